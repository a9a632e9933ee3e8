#!/bin/bash
# round 5, call 19: fewer level passes, a longer workgroup-per-segment tier: TMC2_KD_HUGEMAX up to its limit, in flight
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for hm in 16384 32768 65536 131072; do
TMC2_KD_HUGEMAX=$hm timeout -k 10 600 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c19_$hm.json 2> $O/r05c19_$hm.err
python -c "
import json; d=json.loads(open('$O/r05c19_$hm.json').read().strip().splitlines()[-1]); print('hugemax $hm', d['value'], d['verified'], 'proxy', d.get('per_rank_proxy',{}).get('ms'), {k:v for k,v in d['stage_ms_per_frame'].items() if 'kd' in k}, 'alone', d['roofline']['stages']['tree_knn_normals']['alone_ms'])"
done
