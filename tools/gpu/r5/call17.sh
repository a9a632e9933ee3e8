#!/bin/bash
# round 5, call 17: frames in flight with round 5's kernels (the optimum was 16 in rounds 2-4)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for w in 12 16 20 24; do
timeout -k 10 600 python bench.py --workers $w --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c17_w$w.json 2> $O/r05c17_w$w.err
python -c "
import json; d=json.loads(open('$O/r05c17_w$w.json').read().strip().splitlines()[-1]); print('workers $w', d['value'], d['verified'], d['first_gof_ms'], d.get('per_rank_proxy',{}).get('ms'))"
done
