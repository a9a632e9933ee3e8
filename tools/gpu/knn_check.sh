#!/bin/bash
# k-NN change check: parity tests that go through the k-NN kernels, then kernel durations of one frame
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "knn or normals or metrics or transfer or colour or color or full_size" --deselect tests/test_gpu_gof_soak.py > gpurun_out/knn_tests.log 2>&1; echo "rc=$?" >> gpurun_out/knn_tests.log
tail -n 6 gpurun_out/knn_tests.log
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/knn_prof.log 2>&1
DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | grep -E "knnKernel|normalsKernel|total kernel"
rm -rf $OUT/prof_solo
