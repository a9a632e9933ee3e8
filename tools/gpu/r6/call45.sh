#!/bin/bash
# round 6, call 45: the headline line as the driver runs it (20 steps, 5 warm-up), three times, with every timed step's time
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out/r06c45_headline_steps.txt; : > $O
for i in 1 2 3; do
  r=$( timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --decoder 0 --tail 0 --ingest 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['ms_per_step'], 'untimed', d['untimed_pass_ms'], 'steps', d['step_ms'])" 2>&1 | tail -1 )
  echo "bench --steps 20 --warmup 5: $r" | tee -a $O
done
