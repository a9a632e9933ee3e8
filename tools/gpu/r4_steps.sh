#!/bin/bash
# round 4: does the number depend on how many steps are timed / warmed up?  (same box, same process set-up)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for sw in "2 1" "10 3" "20 5" "2 1" "10 3"; do
  set -- $sw
  timeout 200 python bench.py --steps $1 --warmup $2 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > gpurun_out/r04n_steps_$1.json 2> gpurun_out/r04n_steps_$1.err; echo "steps $1 warmup $2 rc=$?"
  python - $1 <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/r04n_steps_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", j["steps"], j["warmup"], j["value"], j["ms_per_step"], j["verified"])
except Exception as e:
    print("no line:", e)
PY
done
