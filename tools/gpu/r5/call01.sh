#!/bin/bash
# round 5, call 1: the new 32-frame / 16-in-flight gates of every configuration, and the starting point of the round
# (default bench line without the side legs; one-frame kernel stats for the tree build)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 1500 python -m pytest tests/test_gpu_gof32.py -m gpu -x -q --durations=10 2>&1 | tail -25) > $O/r05c1_gof32.log 2>&1
timeout -k 10 600 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c1_bench.json 2> $O/r05c1_bench.err; echo "bench rc=$?"
bash tools/gpu/kd_prof.sh 32768 > $O/r05c1_kd_prof.txt 2>&1
tail -5 $O/r05c1_gof32.log; python -c "
import json; d=json.loads(open('$O/r05c1_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['verified'], d.get('per_rank_proxy'))"
cat $O/r05c1_kd_prof.txt
