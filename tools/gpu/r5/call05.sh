#!/bin/bash
# round 5, call 5: per-dispatch durations of one tree build (which level's landing pass is slow?)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
REPO=$(pwd); SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
rm -rf $O/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/kd_prof.log 2>&1
DB=$(find $O/prof_solo -name "*_results.db" | head -1)
python $REPO/profiles/timeline_rocpd.py "$DB" 0 100000 | grep -E "lv|piece" | tail -75 > $O/r05c5_tree_timeline.txt
rm -rf $O/prof_solo
cat $O/r05c5_tree_timeline.txt
