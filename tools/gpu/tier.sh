#!/bin/bash
# the whole GPU tier (the 16-in-flight soak last), then a bench line
mkdir -p gpurun_out
timeout -k 10 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_gof_soak.py > gpurun_out/tier_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/tier_tests.log
tail -n 5 gpurun_out/tier_tests.log
if [ "$1" = "soak" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_gof_soak.py -m gpu -x -q > gpurun_out/tier_gof32.log 2>&1; echo "gof32 rc=$?" >> gpurun_out/tier_gof32.log
  tail -n 4 gpurun_out/tier_gof32.log
fi
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/tier_bench.json 2> gpurun_out/tier_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/tier_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["verified"], d["roofline"]["kernel"], d["roofline"]["alone_avg_launch_ms"], d["roofline"]["alone_frac"], d.get("per_rank_proxy"), d.get("cpu_baseline"))
print({k: v for k, v in d["stage_ms_per_frame"].items() if v > 0.5 and v < 1000})
PY
