cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 600 python -X faulthandler -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -30) > gpurun_out/r02_pytest10.log 2>&1
(time timeout 900 python bench.py > gpurun_out/r02_bench4.json 2> gpurun_out/r02_bench4.err) > gpurun_out/r02_bench4.time 2>&1
