#!/bin/bash
# round 5, call 29: one rank's share (4 frames, 4 in flight, 60 steps) by the size of the workgroup-per-segment tier, and with the refinement's geometry not queued ahead
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for hm in 4096 16384 32768; do
TMC2_KD_HUGEMAX=$hm timeout -k 10 600 python bench.py --frames 4 --workers 4 --steps 60 --warmup 10 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c29_$hm.json 2> $O/r05c29_$hm.err
python -c "
import json; d=json.loads(open('$O/r05c29_$hm.json').read().strip().splitlines()[-1]); print('KD_HUGEMAX $hm: 4 frames, 4 in flight:', d['ms_per_step'], 'ms per step')"
done
TMC2_REFINE_OVERLAP=0 timeout -k 10 600 python bench.py --frames 4 --workers 4 --steps 60 --warmup 10 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c29_noov.json 2> $O/r05c29_noov.err
python -c "
import json; d=json.loads(open('$O/r05c29_noov.json').read().strip().splitlines()[-1]); print('REFINE_OVERLAP 0: 4 frames, 4 in flight:', d['ms_per_step'], 'ms per step')"
