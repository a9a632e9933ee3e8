#!/bin/bash
# round 6, call 15: frames in flight with round 6's kernels (the S3 / S7 passes in eighths), same box
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
for w in 16 12 14 18 20 16; do
timeout 600 $B --workers $w --steps 10 --warmup 3 > $O/r06c15_workers_$w.json 2> $O/r06c15_workers_$w.err
python - $w <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06c15_workers_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("workers", sys.argv[1], "value", d["value"], "verified", d["verified"], "first", d["first_gof_ms"])
PY
done
