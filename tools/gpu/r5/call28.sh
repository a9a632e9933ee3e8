#!/bin/bash
# round 5, call 28: one rank's share of the 8-GPU run (4 frames, 4 in flight), 60 timed steps each: round 5's tree build against round 4's
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for rep in 1 2; do
for form in pieces tiers; do
TMC2_KD_FORM=$form timeout -k 10 600 python bench.py --frames 4 --workers 4 --steps 60 --warmup 10 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c28_$form$rep.json 2> $O/r05c28_$form$rep.err
python -c "
import json; d=json.loads(open('$O/r05c28_$form$rep.json').read().strip().splitlines()[-1]); print('$form run $rep: 4 frames, 4 in flight:', d['ms_per_step'], 'ms per step ->', round(32.0 / (d['ms_per_step'] * 1e-3), 1), 'frames/s bound for N = 8')"
done
done
