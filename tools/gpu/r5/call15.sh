#!/bin/bash
# round 5, call 15: one vox11 frame alone: the tree build's kernels, new form / round 4's tiers
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
REPO=$(pwd); SOLO="python $REPO/bench.py --config basketball --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
for form in pieces tiers; do
  rm -rf $O/prof_solo; TMC2_KD_FORM=$form timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/kd_prof.log 2>&1
  DB=$(find $O/prof_solo -name "*_results.db" | head -1)
  echo "TMC2_KD_FORM=$form  $(grep -o '"kdtree_build": [0-9.]*' $O/kd_prof.log | head -1)"; python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "pieceKernel|lv[A-Z]|hugeSeg|rangeKernel|decideFlag|swapOne|flagTwo|swapTwo|splitSeg|finishSub|initKernel"
done
rm -rf $O/prof_solo
