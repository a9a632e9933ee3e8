#!/bin/bash
# round 4: the kernel trace of the default bench command as it is now (native GOF host in the timed region)
REPO=$(cd "$(dirname "$0")/../.." && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
FULL="python $REPO/bench.py --steps 4 --warmup 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_full; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_full -- $FULL > $OUT/r04n_prof_full.log 2>&1; echo "trace rc=$?"
DB=$(find $OUT/prof_full -name "*_results.db" | head -1)
python $REPO/profiles/summarise_rocpd.py "$DB" "$FULL  (32-frame GOF, 16 frames in flight, native GOF host)" > $OUT/r04_kernel_stats_default_bench_native.txt
python $REPO/profiles/concurrency_rocpd.py "$DB" 0.3 > $OUT/r04_concurrency_default_bench_native.txt
rm -rf $OUT/prof_full
head -12 $OUT/r04_kernel_stats_default_bench_native.txt; tail -2 $OUT/r04n_prof_full.log | cut -c1-300
