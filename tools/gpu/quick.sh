#!/bin/bash
# quick check: a pytest selection ($1), then the durations of the kernels matching $2 in one frame
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$1" --deselect tests/test_gpu_gof_soak.py > gpurun_out/quick_tests.log 2>&1; echo "rc=$?" >> gpurun_out/quick_tests.log
tail -n 3 gpurun_out/quick_tests.log
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/quick_prof.log 2>&1
DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "$2|total kernel"
grep -o '"ms_per_step": [0-9.]*' $OUT/quick_prof.log | head -1
rm -rf $OUT/prof_solo
