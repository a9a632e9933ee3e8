#!/bin/bash
# round 6, call 18: the degenerate GOF of seed 6, by pass order
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
for ord in input chunk; do
TMC2_ORIENT_ORDER=$ord TMC2_MUTUAL_ORDER=$ord timeout -k 5 90 python -m pytest "tests/test_gpu_fuzz.py::test_gpu_whole_path_on_degenerate_gofs[6]" -x -q -m gpu > $O/r06c18_fuzz_$ord.log 2>&1; echo "== $ord rc $?"; tail -5 $O/r06c18_fuzz_$ord.log | cut -c1-220
done
TMC2_REFINE_DEBUG=1 timeout -k 5 60 python -m pytest "tests/test_gpu_fuzz.py::test_gpu_whole_path_on_degenerate_gofs[6]" -x -q -m gpu -s > $O/r06c18_fuzz_debug.log 2>&1; echo "== debug rc $?"; tail -30 $O/r06c18_fuzz_debug.log | cut -c1-220
