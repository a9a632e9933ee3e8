#!/bin/bash
# round 4: one rank's share (4 frames, 4 in flight) through each host
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for h in native python native python; do
  timeout 200 python bench.py --frames 4 --workers 4 --steps 10 --warmup 3 --host $h --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 \
    > gpurun_out/r04n_four_$h.json 2> gpurun_out/r04n_four_$h.err; echo "$h rc=$?"
  python - $h <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/r04n_four_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", j["ms_per_step"], "fps", j["value"])
except Exception as e:
    print("no line:", e)
PY
done
