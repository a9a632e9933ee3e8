cd /root/repo
export TMPDIR=/tmp
(timeout -s ABRT 100 python -X faulthandler -m pytest tests/test_gpu_segmenter.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 60 -k "orient or full_size or segmenter_compute or normals" 2>&1 | tail -5) > gpurun_out/r02_pytest33.log 2>&1
timeout 60 python bench.py --steps 4 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench33.json 2> gpurun_out/r02_bench33.err
