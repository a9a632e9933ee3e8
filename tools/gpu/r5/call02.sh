#!/bin/bash
# round 5, call 2: the piece kernel (one workgroup per <= 4096-point piece, all nodes of a depth at once) against the order tests,
# the full-size fixtures, and its kernel times next to round 4's tiers
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 600 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py -m gpu -x -q -k "kdtree or knn or fuzz" 2>&1 | tail -15) > $O/r05c2_kd_tests.log 2>&1
tail -4 $O/r05c2_kd_tests.log
(timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -8) > $O/r05c2_full.log 2>&1
tail -3 $O/r05c2_full.log
REPO=$(pwd); SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
for form in pieces tiers; do
  rm -rf $O/prof_solo; TMC2_KD_FORM=$form timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/kd_prof.log 2>&1
  DB=$(find $O/prof_solo -name "*_results.db" | head -1)
  echo "TMC2_KD_FORM=$form  $(grep -o '"kdtree_build": [0-9.]*' $O/kd_prof.log | head -1) $(grep -o '"verified": [a-z]*' $O/kd_prof.log | head -1)"; python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "rangeKernel|decide|swapOne|swapTwo|flagTwo|initKernel|splitSegments|finishSubtrees|hugeSegments|pieceKernel"
done
rm -rf $O/prof_solo
