#!/bin/bash
# kernel traces at HEAD: one frame in flight and the default bench (summaries only travel back)
TAG=${1:-r03}
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
FULL="python $REPO/bench.py --steps 4 --warmup 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
db() { find "$1" -name "*_results.db" | head -1; }
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/${TAG}_prof_solo.log 2>&1
python $REPO/profiles/summarise_rocpd.py "$(db $OUT/prof_solo)" "$SOLO  (one frame in flight)" > $OUT/${TAG}_kernel_stats_one_frame.txt
python $REPO/profiles/occupancy_rocpd.py "$(db $OUT/prof_solo)" 3 > $OUT/${TAG}_occupancy_one_frame.txt
python $REPO/profiles/timeline_rocpd.py "$(db $OUT/prof_solo)" > $OUT/${TAG}_timeline_one_frame.txt 2>&1
rm -rf $OUT/prof_full; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_full -- $FULL > $OUT/${TAG}_prof_full.log 2>&1
python $REPO/profiles/summarise_rocpd.py "$(db $OUT/prof_full)" "$FULL  (32-frame GOF, 16 frames in flight)" > $OUT/${TAG}_kernel_stats_default_bench.txt
python $REPO/profiles/concurrency_rocpd.py "$(db $OUT/prof_full)" 0.3 > $OUT/${TAG}_concurrency_default_bench.txt
rm -rf $OUT/prof_solo $OUT/prof_full
