#!/bin/bash
# patch-stage change check: parity tests through the patch kernels, then their durations in one frame
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "patch or segmenter or full_size or degenerate or whole_path" --deselect tests/test_gpu_gof_soak.py > gpurun_out/patch_tests.log 2>&1; echo "rc=$?" >> gpurun_out/patch_tests.log
tail -n 4 gpurun_out/patch_tests.log
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/patch_prof.log 2>&1
DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "^patch|^cc|raw|footprint|total kernel"
rm -rf $OUT/prof_solo
