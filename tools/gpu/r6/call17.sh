#!/bin/bash
# round 6, call 17: the degenerate GOF that failed in call 16, alone; where the tier's time goes
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 600 python -m pytest "tests/test_gpu_fuzz.py::test_gpu_whole_path_on_degenerate_gofs" -x -q -m gpu --durations=5 > $O/r06c17_fuzz.log 2>&1; tail -15 $O/r06c17_fuzz.log | cut -c1-220
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu --durations=8 > $O/r06c17_full.log 2>&1; tail -14 $O/r06c17_full.log | cut -c1-200
