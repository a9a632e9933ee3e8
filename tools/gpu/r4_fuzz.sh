#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd tests && timeout -k 5 ${2:-420} python ../tools/fuzz/fuzz_gpu_hooks.py ${1:-1000} ${3:-1400} > ../gpurun_out/r04_fuzz_gpu_hooks.log 2>&1; echo "rc=$?" >> ../gpurun_out/r04_fuzz_gpu_hooks.log
cd ..; grep -c MISMATCH gpurun_out/r04_fuzz_gpu_hooks.log; grep MISMATCH gpurun_out/r04_fuzz_gpu_hooks.log | head -20; tail -n 3 gpurun_out/r04_fuzz_gpu_hooks.log
