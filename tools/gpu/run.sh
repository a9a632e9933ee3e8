cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 300 python -X faulthandler -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 200 -k "kdtree or knn or fuzz" 2>&1 | tail -12) > gpurun_out/r02_pytest22.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_solo18.json 2> gpurun_out/r02_solo18.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench23.json 2> gpurun_out/r02_bench23.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench23b.json 2> gpurun_out/r02_bench23b.err
