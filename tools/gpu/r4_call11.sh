#!/bin/bash
# round 4, eleventh GPU call: kNN with the LDS stack for vox11-size trees, orderedSums with a pure adding wave; the whole tier under the guard
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 5 400 python -m pytest tests/test_gpu_metrics.py tests/test_gpu_full_size.py tests/test_gpu_images.py -m gpu -q -x > $O/r04c11_tests.log 2>&1; echo "rc=$?" >> $O/r04c11_tests.log; tail -n 3 $O/r04c11_tests.log
for c in basketball longdress; do
  timeout -k 5 300 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 > $O/r04c11_bench_$c.json 2> $O/r04c11_bench_$c.err; echo "$c rc=$?"
done
python - <<'PY'
import json
for c in ("basketball", "longdress"):
    try:
        d = json.loads(open("gpurun_out/r04c11_bench_%s.json" % c).read().strip().splitlines()[-1])
        dec = d.get("decoder", {})
        print(c, d["value"], "verified", d["verified"], "| proxy", d.get("per_rank_proxy", {}).get("ms"), "| metric ms", d.get("metric_ms_per_frame"), d.get("metric_stage_ms"),
              "| decoder", dec.get("frames_per_s"), dec.get("verified"))
        print("   ", {k: (v["alone_ms"], v["runs_per_frame"]) for k, v in d["roofline"]["stages"].items() if k.startswith("knn")})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
TMC2_GUARD=1 timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/r04_guard_tier.log 2>&1; echo "rc=$?" >> $O/r04_guard_tier.log
tail -n 4 $O/r04_guard_tier.log
