#!/bin/bash
# round 6, call 48: the whole GPU tier, then the end-of-round artefacts (tools/gpu/final.sh) with the final sources
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 2400 python -m pytest tests -x -q -m gpu > $O/r06c48_tests_all.log 2>&1; tail -3 $O/r06c48_tests_all.log
bash tools/gpu/final.sh r06
