cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 240 python -X faulthandler -m pytest tests/test_gpu_segmenter.py -m gpu -q --timeout 100 -x 2>&1 | tail -40) > gpurun_out/r02_pytest9.log 2>&1
(time timeout -s ABRT 600 python -X faulthandler -m pytest tests/test_gpu_full_size.py -m gpu -q --timeout 200 2>&1 | tail -60) > gpurun_out/r02_pytest9b.log 2>&1
