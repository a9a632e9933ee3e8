cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 400 python -X faulthandler -m pytest tests/test_gpu_metrics.py tests/test_gpu_images.py tests/test_gpu_fuzz.py -m gpu -q --timeout 150 2>&1 | tail -12) > gpurun_out/r02_pytest17.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 > gpurun_out/r02_solo13.json 2> gpurun_out/r02_solo13.err
