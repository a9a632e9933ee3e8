#!/bin/bash
# round 6, call 01: the first pass of a process -- per context or per process, clock ramp or not, and where the time is (kernels or gaps)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
python __graft_entry__.py smoke > $O/r06c01_smoke.log 2>&1; tail -1 $O/r06c01_smoke.log
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config longdress > $O/r06c01_A.json 2> $O/r06c01_A.err
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config longdress --preheat-ms 600 --sets 1 > $O/r06c01_B.json 2> $O/r06c01_B.err
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config longdress --warm-one 1 --sets 1 > $O/r06c01_C.json 2> $O/r06c01_C.err
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config longdress --warm-all 1 --sets 1 > $O/r06c01_D.json 2> $O/r06c01_D.err
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config redandblack > $O/r06c01_E.json 2> $O/r06c01_E.err
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/r06c01_trace -- python $GRAFT_REPO_ROOT/tools/gpu/r6/first_pass.py --config longdress --sets 1 --passes 4 --marker 1 --gen-procs 1 > $O/r06c01_T.json 2> $O/r06c01_T.err
cd $GRAFT_REPO_ROOT && python tools/gpu/r6/split_trace.py $O/r06c01_trace > $O/r06c01_split.txt 2>&1
rm -rf $O/r06c01_trace
for f in A B C D E T; do python - <<PY
import json
try:
    d=json.load(open("$O/r06c01_$f.json"))
    print("$f", d["flags"], [(s["pass_ms"], s.get("warm_ms"), s.get("preheat_ms")) for s in d["sets"]])
except Exception as e:
    print("$f failed", e)
PY
done
cat $O/r06c01_split.txt
