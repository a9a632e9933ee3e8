cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 400 python -X faulthandler -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py tests/test_gpu_metrics.py -m gpu -q --timeout 120 -x 2>&1 | tail -40) > gpurun_out/r02_pytest11.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --kdtree device --cpu-baseline 0 --tail 0 > gpurun_out/r02_solo7.json 2> gpurun_out/r02_solo7.err
timeout 300 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 --tail 0 --kdtree device > gpurun_out/r02_bench6_device.json 2> gpurun_out/r02_bench6_device.err
