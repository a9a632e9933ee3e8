#!/bin/bash
# round 5, call 9: the whole GPU tier after the options / reserve / sharded-host changes; first passes with and without the reservation
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25) > $O/r05c9_gpu_tier.log 2>&1
tail -14 $O/r05c9_gpu_tier.log
for r in 1 0; do
timeout -k 10 600 python bench.py --reserve $r --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c9_bench_reserve$r.json 2> $O/r05c9_bench_reserve$r.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c9_bench_reserve$r.json').read().strip().splitlines()[-1]); print('reserve $r', d['value'], d['verified'], 'first', d['first_gof_ms'], d['untimed_pass_ms'], 'steady', d['ms_per_step'], d['pool'], d.get('per_rank_proxy',{}).get('ms'))"
done
