#!/bin/bash
# round 6, call 22: where a closure launch spends its 29 us (REFINE_TIMING on one longdress frame), few-frames and many-frames grids
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
ENC="python $REPO/tools/gpu/r6/first_pass.py --config longdress --frames 1 --workers 1 --sets 1 --passes 2 --gen-procs 1 --capacity-h 2304"
TMC2_REFINE_TIMING=1 timeout 300 $ENC > $O/r06c22_timing_overlap.log 2>&1
grep -A52 "refine closure" $O/r06c22_timing_overlap.log | tail -53 | cut -c1-160
TMC2_REFINE_TIMING=1 TMC2_REFINE_OVERLAP=0 timeout 300 $ENC > $O/r06c22_timing_many.log 2>&1
grep -A52 "refine closure" $O/r06c22_timing_many.log | tail -53 | head -20 | cut -c1-160
