#!/bin/bash
# round 6, call 05: the L2 / kernel-boundary microbenchmark, then the whole GPU tier after the round's changes so far
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 300 tools/gpu/l2_boundary > $O/r06c05_l2_boundary.txt 2>&1; cat $O/r06c05_l2_boundary.txt
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/r06c05_gpu_tier.log 2>&1; tail -5 $O/r06c05_gpu_tier.log
