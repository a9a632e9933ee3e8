#!/bin/bash
# round 6, call 03: (a) what is left of the first pass (call marks; buffers that fit from the start), (b) STREAM_XCDS: every context's
# stream confined to 1 / 2 / 4 of the eight XCDs -- where the workgroups land, and what the GOF does
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_native_gof.py -x -q -m gpu > $O/r06c03_tests.log 2>&1; tail -3 $O/r06c03_tests.log
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config longdress --sets 1 > $O/r06c03_first.json 2> $O/r06c03_first.err
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config longdress --sets 1 --capacity-h 1344 > $O/r06c03_first_fits.json 2> $O/r06c03_first_fits.err
python - <<PY
import json
for n in ("first","first_fits"):
    try:
        s=json.load(open("$O/r06c03_%s.json"%n))["sets"][0]; print(n, s["pass_ms"], s.get("new_buffers_ms"), s.get("call_marks_ms"))
    except Exception as e: print(n,"failed",e)
PY
python - <<PY > $O/r06c03_xcd_histogram.txt 2>&1
import os, sys, ctypes as C
sys.path.insert(0, "mpeg-pcc-tmc2_amd")
import numpy as np
for k in ("", "1", "2", "4"):
    if k: os.environ["TMC2_STREAM_XCDS"] = k
    import tmc2_amd as T
    L = T.load_library()
    for j in range(4 if k else 1):
        ctx = T.Context(0)
        out = np.zeros(8, np.uint32)
        rc = L.tmc2_ctx_xcd_histogram(ctx.h, 8192, out.ctypes.data_as(C.c_void_p))
        print("STREAM_XCDS=%s context %d rc %d workgroups per XCC: %s" % (k or "-", j, rc, out.tolist()))
        ctx.close()
PY
cat $O/r06c03_xcd_histogram.txt
for k in 0 1 2 4; do
if [ $k != 0 ]; then export TMC2_STREAM_XCDS=$k; fi
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r06c03_xcds_$k.json 2> $O/r06c03_xcds_$k.err
python - <<PY
import json
try:
    d=json.loads(open("$O/r06c03_xcds_$k.json").read().strip().splitlines()[-1])
    print("STREAM_XCDS=$k", d["value"], d["ms_per_step"], d["verified"], "rank proxy", d["per_rank_proxy"]["ms"], "sweep alone us", d["roofline"]["alone_avg_launch_ms"]*1e3)
except Exception as e: print("STREAM_XCDS=$k failed", e)
PY
done
