#!/bin/bash
# round 5, call 24: where a depth of the piece kernel spends its time (option KD_PIECE_PROFILE: thread 0 of every workgroup, between barriers)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
TMC2_KD_PIECE_PROFILE=1 timeout -k 10 600 python bench.py --steps 1 --warmup 0 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c24.json 2> $O/r05c24.err
grep "pieceKernel:" $O/r05c24.err | head -4
