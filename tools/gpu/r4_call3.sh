#!/bin/bash
# round 4, third GPU call: the row-wise neighbourhood path (tests, full-size fixtures, soaks), gcc-ASan bench, bench lines + loot trace
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_full_size.py tests/test_gpu_gof_soak.py tests/test_gpu_fuzz.py -m gpu -q -x --durations=8 > $O/r04c3_tests.log 2>&1; echo "rc=$?" >> $O/r04c3_tests.log
tail -n 16 $O/r04c3_tests.log
timeout -k 10 900 bash tools/asan_host_gcc.sh run python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --pin 0 > $O/r04_asan_bench.json 2> $O/r04_asan_bench.err; echo "asan rc=$?" | tee -a $O/r04_asan_bench.err
tail -n 4 $O/r04_asan_bench.err
for c in longdress loot basketball; do
  timeout -k 10 600 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 > $O/r04c3_bench_$c.json 2> $O/r04c3_bench_$c.err; echo "$c rc=$?" | tee -a $O/r04c3_bench_$c.err
done
TMC2_REFINE_NEIGHBOURHOOD=cells timeout -k 10 600 python bench.py --config loot --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 > $O/r04c3_bench_loot_cells.json 2> $O/r04c3_bench_loot_cells.err
python - <<'PY'
import json
for c in ("r04_asan_bench", "r04c3_bench_longdress", "r04c3_bench_loot", "r04c3_bench_loot_cells", "r04c3_bench_basketball"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % c).read().strip().splitlines()[-1])
        dec = d.get("decoder", {})
        print(c, d["value"], "verified", d["verified"], "| roofline", d["roofline"]["kernel"], d["roofline"]["alone_avg_launch_ms"], d["roofline"]["alone_frac"],
              "| proxy", d.get("per_rank_proxy", {}).get("ms"), "| decoder", dec.get("frames_per_s"), dec.get("verified"), str(dec.get("error", ""))[:100])
        print("   ", {k: v for k, v in d["stage_ms_per_frame"].items() if k.startswith("refine")}, {k: (v["alone_ms"], v["runs_per_frame"]) for k, v in d["roofline"]["stages"].items() if k.startswith("refine")})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
cd /tmp
db() { find "$1" -name "*_results.db" | head -1; }
for c in loot; do
  SOLO="python $REPO/bench.py --config $c --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
  rm -rf $O/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/r04_prof_$c.log 2>&1
  python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/r04c3_kernel_stats_one_frame_$c.txt
  rm -rf $O/prof_solo
  head -n 14 $O/r04c3_kernel_stats_one_frame_$c.txt
done
