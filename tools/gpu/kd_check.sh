#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "kdtree or tree or knn or full_size or normals" --deselect tests/test_gpu_gof_soak.py > gpurun_out/kd_tests.log 2>&1; echo "rc=$?" >> gpurun_out/kd_tests.log
tail -n 6 gpurun_out/kd_tests.log
bash tools/gpu/kd_prof.sh 8192 32768 65536 131072
