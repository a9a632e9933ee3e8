#!/bin/bash
# round 5, call 22: how many points move in each of S5's sweeps (does the refinement reach a fixpoint before sweep 50?)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for cfg in longdress basketball; do
TMC2_REFINE_TRACE=1 timeout -k 10 600 python bench.py --config $cfg --steps 1 --warmup 0 --frames 2 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c22_$cfg.json 2> $O/r05c22_$cfg.err
grep "points moved" $O/r05c22_$cfg.err | head -3
done
