#!/bin/bash
# round 5, call 3: piece kernel v2 (registers instead of LDS re-reads, LDS-only barriers, packed min/max), 4 and 8 positions per thread
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for per in 4 8; do
(TMC2_KD_PIECE_PER=$per timeout -k 10 600 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py -m gpu -x -q -k "kdtree or fuzz" 2>&1 | tail -15) > $O/r05c3_kd_tests_$per.log 2>&1
tail -2 $O/r05c3_kd_tests_$per.log
done
REPO=$(pwd); SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
for per in 4 8; do
  rm -rf $O/prof_solo; TMC2_KD_PIECE_PER=$per timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/kd_prof.log 2>&1
  DB=$(find $O/prof_solo -name "*_results.db" | head -1)
  echo "TMC2_KD_PIECE_PER=$per  $(grep -o '"kdtree_build": [0-9.]*' $O/kd_prof.log | head -1) $(grep -o '"verified": [a-z]*' $O/kd_prof.log | head -1)"; python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "pieceKernel"
done
rm -rf $O/prof_solo
cd $REPO; (timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3) > $O/r05c3_full.log 2>&1; tail -1 $O/r05c3_full.log
