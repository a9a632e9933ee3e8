#!/bin/bash
# round 6, call 06: dispatch interference microbenchmark; GPU tests of the trees / refinement after the retired forms left; bench
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout -k 10 300 tools/gpu/dispatch_interference > $O/r06c06_dispatch.txt 2>&1; cat $O/r06c06_dispatch.txt
unset GPU_MAX_HW_QUEUES
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_native_gof.py tests/test_gpu_full_size.py -x -q -m gpu > $O/r06c06_tests.log 2>&1; tail -3 $O/r06c06_tests.log
timeout -k 10 900 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r06c06_bench.json 2> $O/r06c06_bench.err; python - <<PY
import json
d=json.loads(open("$O/r06c06_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["first_gof_ms"], d["untimed_pass_ms"], d["verified"], d["per_rank_proxy"]["ms"], d["stage_ms_per_frame"].get("k:ccMutualMask"))
PY
