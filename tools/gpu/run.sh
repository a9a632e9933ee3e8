cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 300 python -X faulthandler -m pytest tests/test_integration_adaptor.py -m gpu -q --timeout 200 2>&1 | tail -25) > gpurun_out/r02_pytest19.log 2>&1
TMC2_REFINE_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_solo15.json 2> gpurun_out/r02_solo15.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench21.json 2> gpurun_out/r02_bench21.err
