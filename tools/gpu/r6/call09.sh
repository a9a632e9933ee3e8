#!/bin/bash
# round 6, call 09: all sweeps of S5 in ONE launch (sweepsKernel: phases by ticket) -- parity of the forms, then what it does to a
# frame alone, to one rank's four frames and to the sixteen-in-flight run, by the number of resident workgroups
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 900 python -m pytest tests/test_gpu_segmenter.py -x -q -m gpu -k "refine or segmenter_compute" > $O/r06c09_refine_tests.log 2>&1; tail -3 $O/r06c09_refine_tests.log
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st=d.get("stage_ms_per_frame",{})
    print(sys.argv[1].split("/")[-1], "value", d["value"], "verified", d.get("verified"), "sweeps_ms", st.get("refine_sweeps"), "alone_sweep_ms", d["roofline"].get("alone_avg_launch_ms"), "proxy", d.get("per_rank_proxy",{}).get("ms"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}
# sixteen in flight: two-launch form, then the one-launch form with 32 .. 512 resident workgroups
TMC2_REFINE_PERSISTENT=0 timeout 600 $B --steps 6 --warmup 2 > $O/r06c09_two_launch.json 2> $O/r06c09_two_launch.err; line $O/r06c09_two_launch.json
for G in 32 64 128 256 512; do
TMC2_REFINE_PERSISTENT_BLOCKS=$G timeout 600 $B --steps 6 --warmup 2 > $O/r06c09_resident_$G.json 2> $O/r06c09_resident_$G.err; line $O/r06c09_resident_$G.json
done
timeout 600 $B --steps 6 --warmup 2 > $O/r06c09_default.json 2> $O/r06c09_default.err; line $O/r06c09_default.json
