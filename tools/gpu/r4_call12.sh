#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout -k 5 300 python -m pytest tests/test_gpu_segmenter.py -m gpu -q -x -k "refine or compute" > $O/r04c12_tests.log 2>&1; echo "rc=$?" >> $O/r04c12_tests.log; tail -n 3 $O/r04c12_tests.log
for c in longdress loot; do
  timeout -k 5 300 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 > $O/r04c12_bench_$c.json 2> $O/r04c12_bench_$c.err; echo "$c rc=$?"
done
python - <<'PY'
import json
for c in ("longdress", "loot"):
    try:
        d = json.loads(open("gpurun_out/r04c12_bench_%s.json" % c).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(c, d["value"], "verified", d["verified"], "| proxy", d.get("per_rank_proxy", {}).get("ms"), "| dominant", r["kernel"], r["alone_avg_launch_ms"], r["alone_frac"], "in flight", r["avg_launch_ms"], r["frac"],
              {k: (v["alone_ms"], v["runs_per_frame"]) for k, v in r["stages"].items() if k.startswith("refine")})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
