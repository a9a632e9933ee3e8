#!/bin/bash
# round 5, call 12: the rough-shell workload (how often does S3 leave the contracted walk?), the new noisy parity case, the gaps of
# one frame (copies per frame), one frame's kernel stats
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 600 python -m pytest tests/test_gpu_segmenter.py -m gpu -x -q -k "segmenter_compute_matches_oracle" 2>&1 | tail -4) > $O/r05c12_noisy_parity.log 2>&1; tail -1 $O/r05c12_noisy_parity.log
timeout -k 10 900 python bench.py --workload longdress_vox10_noisy --steps 3 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c12_bench_noisy.json 2> $O/r05c12_bench_noisy.err; echo "noisy rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c12_bench_noisy.json').read().strip().splitlines()[-1]); print('noisy', d['value'], d['verified'], d['orientation']); print({k:v for k,v in d['stage_ms_per_frame'].items() if 'orient' in k})"
bash tools/gpu/gaps.sh r05 > /dev/null 2>&1; head -20 $O/r05_gaps_one_frame.txt
bash profiles/collect.sh r05a > /dev/null 2>&1; head -45 $O/r05a_kernel_stats_one_frame.txt | cut -c1-120
