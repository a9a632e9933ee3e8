#!/bin/bash
# round 5, call 26: the C++ front end with the reservation (same bytes over one, two and three device shards; frames/s from the C++ host)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_host_logic.py -x -q -k "native_front_end" 2>&1 | tail -4) > $O/r05c26_front_end.log 2>&1; tail -2 $O/r05c26_front_end.log
bash tools/gpu/native_gof.sh r05 > /dev/null 2>&1; tail -14 $O/r05_native_front_end.txt
