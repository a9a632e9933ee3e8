#!/bin/bash
# round 4, seventh GPU call: 32-frame fixtures of redandblack / soldier in the bench, orientation + context tests
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 5 200 python -m pytest tests/test_gpu_guard.py tests/test_gpu_segmenter.py -m gpu -q -x -k "guard or context or orientation or solid" > $O/r04c7_tests.log 2>&1; echo "rc=$?" >> $O/r04c7_tests.log; tail -n 4 $O/r04c7_tests.log
for c in redandblack soldier; do
  timeout -k 5 300 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 > $O/r04c7_bench_$c.json 2> $O/r04c7_bench_$c.err; echo "$c rc=$?"
done
python - <<'PY'
import json
for c in ("redandblack", "soldier"):
    try:
        d = json.loads(open("gpurun_out/r04c7_bench_%s.json" % c).read().strip().splitlines()[-1])
        dec = d.get("decoder", {})
        print(c, d["value"], "verified", d["verified"], d["verified_detail"][:100], d["config"]["case"], "| proxy", d.get("per_rank_proxy", {}).get("ms"), "| decoder", dec.get("frames_per_s"), dec.get("verified"), str(dec.get("verified_detail"))[:100])
        print("   regrowth:", {k: v for k, v in d["stage_ms_per_frame"].items() if "orient" in k})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
