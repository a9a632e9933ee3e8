#!/bin/bash
# round 6, call 42: one rank's share of the 8-GPU run (4 frames, 4 in flight) under the kernel trace: how busy is the GPU, where are the holes
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
db() { find "$1" -name "*_results.db" | head -1; }
cd /tmp
RANK="python $REPO/bench.py --frames 4 --workers 4 --steps 12 --warmup 3 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
rm -rf $O/prof_rank; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_rank -- $RANK > $O/r06c42_rank.log 2>&1
python $REPO/profiles/concurrency_rocpd.py "$(db $O/prof_rank)" 0.3 > $O/r06c42_concurrency_rank.txt
python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_rank)" "$RANK  (4 frames, 4 in flight)" > $O/r06c42_kernel_stats_rank.txt
rm -rf $O/prof_rank
head -40 $O/r06c42_concurrency_rank.txt
tail -2 $O/r06c42_rank.log | cut -c1-400
