#!/bin/bash
# round 4: the native GOF host (libtmc2gof.so) -- parity with the Python host, then the timed path through each
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_native_gof.py -x -q -m gpu > gpurun_out/r04n_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04n_tests.log
for mode in "native phases" "native one" "python phases"; do
  set -- $mode
  timeout 240 python bench.py --steps 3 --warmup 1 --host $1 --rendezvous $2 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 \
    > gpurun_out/r04n_bench_$1_$2.json 2> gpurun_out/r04n_bench_$1_$2.err; echo "$mode rc=$?"
  python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/r04n_bench_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print(sys.argv[1:], j["value"], j["verified"], j["config"]["host"][:40])
except Exception as e:
    print("no line:", e)
PY
done
