#!/bin/bash
# round 5, call 27: dead tiles in the level passes (a tile none of whose positions still splits is skipped from then on)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 600 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -x -q -k "kdtree or fuzz or full_size" 2>&1 | tail -6) > $O/r05c27_tests.log 2>&1
tail -2 $O/r05c27_tests.log
REPO=$(pwd)
cd /tmp
for cfg in longdress basketball; do
  SOLO="python $REPO/bench.py --config $cfg --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
  rm -rf $O/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/kd_prof.log 2>&1
  DB=$(find $O/prof_solo -name "*_results.db" | head -1)
  echo "$cfg  $(grep -o '"kdtree_build": [0-9.]*' $O/kd_prof.log | head -1)"; python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "pieceKernel|lv[A-Z]|hugeSeg"
done
rm -rf $O/prof_solo
cd $REPO
timeout -k 10 900 python bench.py --config basketball --steps 4 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > $O/r05c27_basketball.json 2> $O/r05c27_basketball.err; echo "rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c27_basketball.json').read().strip().splitlines()[-1]); dec=d.get('decoder',{})
print('basketball', d['value'], d['verified'], 'proxy', d.get('per_rank_proxy',{}).get('ms'), 'decoder', dec.get('frames_per_s'), dec.get('verified'), {k:v for k,v in d['stage_ms_per_frame'].items() if 'kd' in k})"
timeout -k 10 600 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c27_bench.json 2> $O/r05c27_bench.err
python -c "
import json; d=json.loads(open('$O/r05c27_bench.json').read().strip().splitlines()[-1]); print('longdress', d['value'], d['verified'], d.get('per_rank_proxy',{}).get('ms'), {k:v for k,v in d['stage_ms_per_frame'].items() if 'kd' in k})"
