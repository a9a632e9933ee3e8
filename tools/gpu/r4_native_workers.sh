#!/bin/bash
# round 4: frames in flight under the native GOF host
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for w in 16 12 20 24 32; do
  timeout 240 python bench.py --steps 3 --warmup 1 --workers $w --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 \
    > gpurun_out/r04n_workers_$w.json 2> gpurun_out/r04n_workers_$w.err; echo "workers $w rc=$?"
  python - $w <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/r04n_workers_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("workers", sys.argv[1], j["value"], j["verified"])
except Exception as e:
    print("no line:", e)
PY
done
