#!/bin/bash
# round 6, call 02: the native host after the resume / packing-chain / watchdog changes -- its GPU tests, the first pass with the
# resume, the default bench line, basketball (random access) and two ranks on one GPU through bench.py --gpus 2 (gloo: python host)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_native_gof.py tests/test_gpu_gof32.py -x -q -m gpu > $O/r06c02_tests.log 2>&1; tail -3 $O/r06c02_tests.log
for cfg in longdress loot basketball; do
timeout -k 10 400 python tools/gpu/r6/first_pass.py --config $cfg --sets 1 > $O/r06c02_first_$cfg.json 2> $O/r06c02_first_$cfg.err
python - <<PY
import json
try:
    d=json.load(open("$O/r06c02_first_$cfg.json")); s=d["sets"][0]
    print("$cfg", s["pass_ms"], "new buffers", s.get("new_buffers_ms"))
except Exception as e:
    print("$cfg failed", e)
PY
done
timeout -k 10 900 python bench.py > $O/r06c02_bench.json 2> $O/r06c02_bench.err; python - <<PY
import json
d=json.loads(open("$O/r06c02_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["first_gof_ms"], d["untimed_pass_ms"], d.get("passes_resumed_with_larger_buffers"), d["verified"], d["per_rank_proxy"])
PY
timeout -k 10 900 python bench.py --config basketball --cpu-baseline 0 --tail 0 --ingest 0 > $O/r06c02_bench_basketball.json 2> $O/r06c02_bench_basketball.err; python - <<PY
import json
d=json.loads(open("$O/r06c02_bench_basketball.json").read().strip().splitlines()[-1])
print("basketball", d["value"], d["first_gof_ms"], d["untimed_pass_ms"], d["verified"], d["decoder"].get("value"), d["decoder"].get("verified"))
PY
timeout -k 10 900 python bench.py --gpus 2 --dist-backend gloo --steps 4 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r06c02_two.json 2> $O/r06c02_two.err; tail -c 600 $O/r06c02_two.json | head -c 400; echo
