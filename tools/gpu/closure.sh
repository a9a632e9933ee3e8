#!/bin/bash
# S5 closure change check: refine parity tests, per-sweep kernel durations, short bench
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_segmenter.py -m gpu -x -q -k "refine" > gpurun_out/closure_tests.log 2>&1; echo "rc=$?" >> gpurun_out/closure_tests.log
tail -n 12 gpurun_out/closure_tests.log
bash tools/gpu/sweep_prof.sh closure; head -8 gpurun_out/closure_sweep_kernels.txt | cut -c1-220
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/closure_bench.json 2> gpurun_out/closure_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/closure_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["verified"], d["stage_ms_per_frame"].get("refine_sweeps"), d["roofline"]["alone_avg_launch_ms"])
PY
