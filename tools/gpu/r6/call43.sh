#!/bin/bash
# round 6, call 43: one rank's four frames with the second half starting later (FRAME_START_DELAY_US on two of the four contexts): the step's time
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out/r06c43_stagger.txt; : > $O
for round in 1 2; do
for us in 0 1500 3000 4500 6000; do
  v=$( BENCH_PROXY_STAGGER_US=$us timeout 300 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d.get('per_rank_proxy'))" 2>&1 | tail -1 )
  echo "longdress stagger $us us: $v" | tee -a $O
done
done
for us in 0 3000 6000; do
  v=$( BENCH_PROXY_STAGGER_US=$us timeout 300 python bench.py --config loot --steps 4 --warmup 2 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d.get('per_rank_proxy'))" 2>&1 | tail -1 )
  echo "loot stagger $us us: $v" | tee -a $O
done
