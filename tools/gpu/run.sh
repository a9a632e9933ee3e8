cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 400 python -X faulthandler -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 300 -k "kdtree or knn or fuzz or full_size" 2>&1 | tail -6) > gpurun_out/r02_pytest31.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_solo25.json 2> gpurun_out/r02_solo25.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench31.json 2> gpurun_out/r02_bench31.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench31b.json 2> gpurun_out/r02_bench31b.err
