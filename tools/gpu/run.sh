cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 600 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -8) > gpurun_out/r02_pytest_final.log 2>&1
(time timeout 900 python bench.py > gpurun_out/bench_r02_final.json 2> gpurun_out/bench_r02_final.err) > gpurun_out/bench_r02_final.time 2>&1
bash profiles/collect.sh r02
