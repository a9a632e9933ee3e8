#!/bin/bash
# round 5, call 8: pieces largest first
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 600 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py -m gpu -x -q -k "kdtree or fuzz" 2>&1 | tail -15) > $O/r05c8_kd_tests.log 2>&1
tail -2 $O/r05c8_kd_tests.log
REPO=$(pwd); SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
for hm in 16384; do
  rm -rf $O/prof_solo; TMC2_KD_HUGEMAX=$hm timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/kd_prof.log 2>&1
  DB=$(find $O/prof_solo -name "*_results.db" | head -1)
  echo "TMC2_KD_HUGEMAX=$hm  $(grep -o '"kdtree_build": [0-9.]*' $O/kd_prof.log | head -1)"; python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "pieceKernel|lv[A-Z]|hugeSeg"
done
rm -rf $O/prof_solo
cd $REPO
timeout -k 10 600 python bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c8_bench.json 2> $O/r05c8_bench.err
python -c "
import json; d=json.loads(open('$O/r05c8_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['verified'], d.get('per_rank_proxy',{}).get('ms'), d.get('per_rank_proxy',{}).get('predicted_n8_speedup')); print({k:v for k,v in d['stage_ms_per_frame'].items() if 'kd' in k})"
