cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 600 python -X faulthandler -m pytest tests -m gpu -q -x --timeout 200 2>&1 | tail -15) > gpurun_out/r02_pytest20.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_solo16.json 2> gpurun_out/r02_solo16.err
for i in 1 2 3; do
timeout 300 python -X faulthandler bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench22_$i.json 2> gpurun_out/r02_bench22_$i.err
done
TMC2_REFINE_NO_OVERLAP=1 timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench22_noov.json 2> gpurun_out/r02_bench22_noov.err
