cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 400 python -X faulthandler -m pytest tests/test_gpu_segmenter.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 300 -k "refine or full_size or segmenter_compute or fuzz" 2>&1 | tail -6) > gpurun_out/r02_pytest29.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_solo23.json 2> gpurun_out/r02_solo23.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench29.json 2> gpurun_out/r02_bench29.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench29b.json 2> gpurun_out/r02_bench29b.err
