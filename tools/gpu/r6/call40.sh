#!/bin/bash
# round 6, call 40: the hooks fuzzer with this session's options (LDS tiers, kept hits, paired pushes, split searches; options set per context), 80 seeds; then the whole GPU tier
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 1500 python tools/fuzz/fuzz_gpu_hooks.py 0 80 > $O/r06c40_fuzz.log 2>&1; tail -5 $O/r06c40_fuzz.log
timeout -k 10 2400 python -m pytest tests -x -q -m gpu > $O/r06c40_tests_all.log 2>&1; tail -3 $O/r06c40_tests_all.log
