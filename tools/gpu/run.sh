cd /root/repo
export TMPDIR=/tmp
(timeout -s ABRT 85 python -X faulthandler -m pytest tests/test_gpu_fuzz.py tests/test_integration_adaptor.py -m gpu -q --timeout 40 2>&1 | tail -5) > gpurun_out/r02_pytest34.log 2>&1
