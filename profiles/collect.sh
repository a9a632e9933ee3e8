#!/bin/bash
# Collects the round's rocprofv3 artefacts on the GPU box (run from the repository root through gpurun); the summaries land in
# gpurun_out/ and are copied into profiles/ by hand.   usage: bash profiles/collect.sh <tag>   (e.g. r02)
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
SOLO="python $REPO/bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
FULL="python $REPO/bench.py --steps 4 --warmup 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
db() { find "$1" -name "*_results.db" | head -1; }
# 1. kernel trace, one frame in flight
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/${TAG}_prof_solo.log 2>&1
python $REPO/profiles/summarise_rocpd.py "$(db $OUT/prof_solo)" "$SOLO  (one frame in flight)" > $OUT/${TAG}_kernel_stats_one_frame.txt
python $REPO/profiles/occupancy_rocpd.py "$(db $OUT/prof_solo)" 3 > $OUT/${TAG}_occupancy_one_frame.txt
# 2. kernel trace, the default bench (16 frames in flight)
rm -rf $OUT/prof_full; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_full -- $FULL > $OUT/${TAG}_prof_full.log 2>&1
python $REPO/profiles/summarise_rocpd.py "$(db $OUT/prof_full)" "$FULL  (32-frame GOF, 16 frames in flight)" > $OUT/${TAG}_kernel_stats_default_bench.txt
python $REPO/profiles/concurrency_rocpd.py "$(db $OUT/prof_full)" 0.3 > $OUT/${TAG}_concurrency_default_bench.txt
# 3. HBM traffic counters, separate passes, one frame in flight
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c; timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $SOLO > $OUT/${TAG}_pmc_$c.log 2>&1
done
python $REPO/profiles/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE "longdress_vox10" > $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/prof_solo $OUT/prof_full $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE   # (only the summaries travel back)
