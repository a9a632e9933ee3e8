#!/bin/bash
# round 6, call 46: does the timed region depend on the side legs being enabled?  the driver's invocation (legs on) against legs off, alternating, same box
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out/r06c46_legs.txt; : > $O
for i in 1 2 3; do
  r=$( timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['ms_per_step'], 'untimed', d['untimed_pass_ms'], 'steps', d['step_ms'])" 2>&1 | tail -1 )
  echo "legs on : $r" | tee -a $O
  r=$( timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --decoder 0 --tail 0 --ingest 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['ms_per_step'], 'untimed', d['untimed_pass_ms'], 'steps', d['step_ms'])" 2>&1 | tail -1 )
  echo "legs off: $r" | tee -a $O
done
