#!/bin/bash
# round 4: the default bench line (native host) for the record, and the other configurations through the native host
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/bench_r04_final_native.json 2> gpurun_out/bench_r04_final_native.err; echo "default rc=$?"
for cfg in loot redandblack soldier basketball; do
  timeout 240 python bench.py --config $cfg --steps 2 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 \
    > gpurun_out/bench_r04_native_$cfg.json 2> gpurun_out/bench_r04_native_$cfg.err; echo "$cfg rc=$?"
done
python - <<'PY'
import json
for n in ["final_native", "native_loot", "native_redandblack", "native_soldier", "native_basketball"]:
    try:
        j = json.loads(open("gpurun_out/bench_r04_%s.json" % n).read().strip().splitlines()[-1])
        print(n, j["value"], j["verified"], j["roofline"].get("traffic"), (j.get("decoder") or {}).get("frames_per_s"))
    except Exception as e:
        print(n, "no line:", e)
PY
