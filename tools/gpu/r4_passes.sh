#!/bin/bash
# round 4: which pass over the GOF is the slow one?  (one timed step after 0..3 untimed ones, no priming)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for w in 1 2 0 3; do
  timeout 40 python bench.py --prime 0 --warmup $w --steps 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > gpurun_out/r04n_pass_$w.json 2> gpurun_out/r04n_pass_$w.err
  python - $w <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/r04n_pass_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    st = j.get("stage_ms_per_frame") or j.get("stages_ms") or {}
    top = sorted(st.items(), key=lambda kv: -kv[1])[:6] if isinstance(st, dict) else st
    print("pass", int(sys.argv[1]) + 1, "ms", j["ms_per_step"], top)
except Exception as e:
    print("no line:", e)
PY
done
