cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 300 python -X faulthandler -m pytest tests/test_gpu_metrics.py tests/test_gpu_images.py -m gpu -q -x --timeout 200 -k "metric" 2>&1 | tail -6) > gpurun_out/r02_pytest25.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_metric2.json 2> gpurun_out/r02_metric2.err
