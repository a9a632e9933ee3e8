#!/bin/bash
# one frame in flight: k-NN kernel durations with and without the XCD-aware block mapping
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
for x in 1 0; do
  rm -rf $OUT/prof_solo; TMC2_KNN_XCD=$x timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/knn_prof.log 2>&1
  DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
  echo "TMC2_KNN_XCD=$x"; python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "knnKernel|normalsKernel"
done
rm -rf $OUT/prof_solo
