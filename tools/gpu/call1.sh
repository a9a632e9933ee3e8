#!/bin/bash
# round 3, GPU call 1: stale-view experiment, soak test, whole GPU tier, bench with / without the union pre-check
mkdir -p gpurun_out


timeout -k 10 900 python -m pytest tests/test_gpu_gof_soak.py -m gpu -x -q > gpurun_out/r03_gof32.log 2>&1
echo "gof32 rc=$?" >> gpurun_out/r03_gof32.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_gof_soak.py > gpurun_out/r03_gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03_gpu_tests.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r03_bench_pre1.json 2> gpurun_out/r03_bench_pre1.err
TMC2_UF_PRECHECK=0 timeout -k 10 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r03_bench_pre0.json 2> gpurun_out/r03_bench_pre0.err
for f in gpurun_out/r03_gof32.log gpurun_out/r03_gpu_tests.log; do tail -n 4 $f; done
python - <<'PY'
import json
for n in ("pre1","pre0"):
    try:
        d=json.loads(open("gpurun_out/r03_bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["verified"], d["verified_detail"][:200])
    except Exception as e: print(n, "failed", e)
PY
