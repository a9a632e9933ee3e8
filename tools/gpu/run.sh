cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 500 python -X faulthandler -m pytest tests -m gpu -q -x --timeout 200 2>&1 | tail -15) > gpurun_out/r02_pytest18.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_solo14.json 2> gpurun_out/r02_solo14.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 > gpurun_out/r02_bench20.json 2> gpurun_out/r02_bench20.err
TMC2_REFINE_CLOSURE=split timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench20_split.json 2> gpurun_out/r02_bench20_split.err
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r02_bench20b.json 2> gpurun_out/r02_bench20b.err
