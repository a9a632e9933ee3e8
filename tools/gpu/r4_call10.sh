#!/bin/bash
# round 4, tenth GPU call: the closure ring used once around per sweep -- the refine tests (ring hooks, solid cloud), the voxels-of-2
# soaks, loot three times with 16 frames in flight
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 5 300 python -m pytest tests/test_gpu_segmenter.py -m gpu -q -x -k "refine or solid" > $O/r04c10_refine.log 2>&1; echo "rc=$?" >> $O/r04c10_refine.log; tail -n 3 $O/r04c10_refine.log
timeout -k 5 300 python -m pytest tests/test_gpu_gof_soak.py -m gpu -q -x > $O/r04c10_soak.log 2>&1; echo "rc=$?" >> $O/r04c10_soak.log; tail -n 3 $O/r04c10_soak.log
for r in 1 2 3; do
  timeout -k 5 200 python bench.py --config loot --steps 12 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 > $O/r04c10_bench_loot_$r.json 2> $O/r04c10_bench_loot_$r.err; echo "loot $r rc=$?"
done
timeout -k 5 200 python bench.py --config soldier --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 > $O/r04c10_bench_soldier.json 2> $O/r04c10_bench_soldier.err; echo "soldier rc=$?"
python - <<'PY'
import json
for c in ("loot_1", "loot_2", "loot_3", "soldier"):
    try:
        d = json.loads(open("gpurun_out/r04c10_bench_%s.json" % c).read().strip().splitlines()[-1])
        print(c, d["value"], "verified", d["verified"], "| proxy", d.get("per_rank_proxy", {}).get("ms"),
              {k: (v["alone_ms"], v["runs_per_frame"]) for k, v in d["roofline"]["stages"].items() if k.startswith("refine")},
              {k: v for k, v in d["stage_ms_per_frame"].items() if "orient" in k})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
