#!/bin/bash
# round 5, call 6: the new tree build under the frames-in-flight gates and in the bench line
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 1500 python -m pytest tests/test_gpu_gof32.py tests/test_gpu_gof_soak.py -m gpu -x -q --durations=5 2>&1 | tail -12) > $O/r05c6_gof.log 2>&1
tail -3 $O/r05c6_gof.log
timeout -k 10 600 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c6_bench.json 2> $O/r05c6_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c6_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['verified'], d.get('per_rank_proxy',{}).get('ms'), d.get('per_rank_proxy',{}).get('predicted_n8_speedup')); print({k:v for k,v in d['stage_ms_per_frame'].items() if 'kd' in k})"
TMC2_KD_FORM=tiers timeout -k 10 600 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c6_bench_tiers.json 2> $O/r05c6_bench_tiers.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c6_bench_tiers.json').read().strip().splitlines()[-1]); print('tiers', d['value'], d['verified'], d.get('per_rank_proxy',{}).get('ms'), d.get('per_rank_proxy',{}).get('predicted_n8_speedup')); print({k:v for k,v in d['stage_ms_per_frame'].items() if 'kd' in k})"
