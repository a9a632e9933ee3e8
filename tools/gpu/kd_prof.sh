#!/bin/bash
# one frame in flight: kernels of the device tree build, with the workgroup-per-segment tier at several sizes (TMC2_KD_HUGEMAX)
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
for x in ${@:-8192 65536}; do
  rm -rf $OUT/prof_solo; TMC2_KD_HUGEMAX=$x timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/kd_prof.log 2>&1
  DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
  echo "TMC2_KD_HUGEMAX=$x  $(grep -o '"kdtree_build": [0-9.]*' $OUT/kd_prof.log | head -1)"; python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "rangeKernel|decide|swapOne|swapTwo|flagTwo|initKernel|splitSegments|finishSubtrees|hugeSegments"
done
rm -rf $OUT/prof_solo
