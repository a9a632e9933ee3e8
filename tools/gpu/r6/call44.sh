#!/bin/bash
# round 6, call 44: the default bench invocation (20 steps, 5 warm-up) three times on one box, with and without the split searches: how far do runs of the headline scatter
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out/r06c44_headline.txt; : > $O
for i in 1 2 3; do
for v in 1 0; do
  r=$( TMC2_KNN_SPLIT=$v timeout 600 python bench.py --cpu-baseline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['ms_per_step'], d['untimed_pass_ms'], (d.get('per_rank_proxy') or {}).get('ms'), (d.get('decoder') or {}).get('frames_per_s'))" 2>&1 | tail -1 )
  echo "default bench, split=$v: $r" | tee -a $O
done
done
