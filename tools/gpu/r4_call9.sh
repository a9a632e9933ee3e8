#!/bin/bash
# round 4, ninth GPU call: partial sort of the neighbourhood rows, threshold ladder of the orientation; tests, bench lines, loot trace
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 5 500 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_full_size.py tests/test_gpu_gof_soak.py -m gpu -q -x > $O/r04c9_tests.log 2>&1; echo "rc=$?" >> $O/r04c9_tests.log; tail -n 4 $O/r04c9_tests.log
for c in longdress loot redandblack; do
  timeout -k 5 300 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 > $O/r04c9_bench_$c.json 2> $O/r04c9_bench_$c.err; echo "$c rc=$?"
done
python - <<'PY'
import json
for c in ("longdress", "loot", "redandblack"):
    try:
        d = json.loads(open("gpurun_out/r04c9_bench_%s.json" % c).read().strip().splitlines()[-1])
        print(c, d["value"], "verified", d["verified"], "| proxy", d.get("per_rank_proxy", {}).get("ms"),
              {k: (v["alone_ms"], v["runs_per_frame"]) for k, v in d["roofline"]["stages"].items() if k.startswith("refine")},
              {k: v for k, v in d["stage_ms_per_frame"].items() if "orient" in k})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
cd /tmp
db() { find "$1" -name "*_results.db" | head -1; }
for c in loot; do
  SOLO="python $REPO/bench.py --config $c --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
  rm -rf $O/prof_solo; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/r04_prof_$c.log 2>&1
  python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/r04c9_kernel_stats_one_frame_$c.txt
  rm -rf $O/prof_solo
  head -n 12 $O/r04c9_kernel_stats_one_frame_$c.txt
done
