#!/bin/bash
# round 6, call 52: the hooks fuzzer (options per context) on 240 more seeds with the round's final library
# (first run: seed 232 reported a colour-transfer mismatch -- the fuzzer itself handed a colour array shorter than its cloud to both sides
#  (82 points: rgb[50:100] has 32 rows, xyz[:50] has 50): tools/gpu/r6/call53.sh; fixed in the fuzzer, the binding now refuses it)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 2400 python tools/fuzz/fuzz_gpu_hooks.py 80 320 > $O/r06_fuzz_gpu_hooks.log 2>&1; grep -c "^refused" $O/r06_fuzz_gpu_hooks.log; grep "MISMATCH" $O/r06_fuzz_gpu_hooks.log | head -5; tail -1 $O/r06_fuzz_gpu_hooks.log
