cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 600 python -X faulthandler -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -8) > gpurun_out/r02_pytest24.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench25.json 2> gpurun_out/r02_bench25.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench25b.json 2> gpurun_out/r02_bench25b.err
