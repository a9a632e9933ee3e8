#!/bin/bash
# round 4: priming passes -- the number no longer depends on W; then the default line for the record
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 100 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > gpurun_out/r04n_prime_2_1.json 2> gpurun_out/r04n_prime_2_1.err; echo "2/1 rc=$?"
timeout 130 python bench.py > gpurun_out/bench_r04_final_native.json 2> gpurun_out/bench_r04_final_native.err; echo "default rc=$?"
python - <<'PY'
import json
for n in ("r04n_prime_2_1", "bench_r04_final_native"):
    try:
        j = json.loads(open("gpurun_out/%s.json" % n).read().strip().splitlines()[-1])
        print(n, j["steps"], j["warmup"], j["priming_passes"], j["value"], j["verified"], j["roofline"]["traffic"], (j.get("decoder") or {}).get("frames_per_s"))
    except Exception as e:
        print(n, "no line:", e)
PY
