#!/bin/bash
# round 5, call 30: one rank's share (4 frames, 4 in flight): REFINE_OVERLAP x KD_HUGEMAX, longdress twice, loot and basketball once
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
run() { # config overlap hugemax steps
TMC2_REFINE_OVERLAP=$2 TMC2_KD_HUGEMAX=$3 timeout -k 10 600 python bench.py --config $1 --frames 4 --workers 4 --steps $4 --warmup 8 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c30.json 2> $O/r05c30.err
python -c "
import json; d=json.loads(open('$O/r05c30.json').read().strip().splitlines()[-1]); print('$1 overlap $2 hugemax $3:', d['ms_per_step'], 'ms per step')"
}
for rep in 1 2; do
run longdress 1 16384 50; run longdress 0 16384 50; run longdress 0 32768 50; run longdress 1 32768 50
done
run loot 1 16384 30; run loot 0 16384 30; run loot 0 32768 30
run basketball 1 16384 20; run basketball 0 16384 20; run basketball 0 32768 20
