#!/bin/bash
# round 6, call 04: STREAM_XCDS with the other reading of the CU mask (XCC x = bits 32 x .. 32 x + 31), and an API trace of a first pass that is refused and resumed
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
export TMC2_STREAM_XCDS_LAYOUT=block
python - <<PY > $O/r06c04_xcd_histogram.txt 2>&1
import os, sys, ctypes as C
sys.path.insert(0, "mpeg-pcc-tmc2_amd")
import numpy as np
for k in ("1", "2", "4"):
    os.environ["TMC2_STREAM_XCDS"] = k
    import tmc2_amd as T
    L = T.load_library()
    for j in range(3):
        ctx = T.Context(0)
        out = np.zeros(8, np.uint32)
        rc = L.tmc2_ctx_xcd_histogram(ctx.h, 8192, out.ctypes.data_as(C.c_void_p))
        print("layout block STREAM_XCDS=%s context %d rc %d workgroups per XCC: %s" % (k, j, rc, out.tolist()))
        ctx.close()
PY
cat $O/r06c04_xcd_histogram.txt
for k in 1 2 4; do
export TMC2_STREAM_XCDS=$k
timeout -k 10 600 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r06c04_xcds_$k.json 2> $O/r06c04_xcds_$k.err
python - <<PY
import json
try:
    d=json.loads(open("$O/r06c04_xcds_$k.json").read().strip().splitlines()[-1])
    print("block STREAM_XCDS=$k", d["value"], d["ms_per_step"], d["verified"], "rank proxy", d["per_rank_proxy"]["ms"], "sweep alone us", d["roofline"]["alone_avg_launch_ms"]*1e3)
except Exception as e: print("STREAM_XCDS=$k failed", e)
PY
done
unset TMC2_STREAM_XCDS TMC2_STREAM_XCDS_LAYOUT
cd /tmp && timeout -k 10 600 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/r06c04_trace -- python $GRAFT_REPO_ROOT/tools/gpu/r6/first_pass.py --config longdress --sets 1 --passes 2 --gen-procs 1 > $O/r06c04_T.json 2> $O/r06c04_T.err
cd $GRAFT_REPO_ROOT; ls -la $O/r06c04_trace/*/ | head; python - <<PY
import csv, glob, collections, json
d=json.load(open("$O/r06c04_T.json"))["sets"][0]; print(d["pass_ms"], d.get("call_marks_ms"), d.get("new_buffers_ms"))
f=glob.glob("$O/r06c04_trace/**/*hip_api_trace.csv", recursive=True)
rows=[]
for x in f:
    rows+=list(csv.DictReader(open(x)))
print(len(rows), "api rows", rows[0].keys() if rows else None)
# the slowest 25 API calls
rows.sort(key=lambda r:-(int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for r in rows[:25]:
    print(r["Function"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms", "start", int(r["Start_Timestamp"])/1e6 % 100000)
tot=collections.Counter()
for r in rows: tot[r["Function"]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
print(tot.most_common(12))
PY
rm -rf $O/r06c04_trace
