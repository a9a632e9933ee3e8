#!/bin/bash
# round 4, fifth GPU call: where does the solid cloud hang (bounded runs with a trace); rank look-ups + table-free voxelisation
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
bash tools/gpu/solid_debug.sh > /dev/null 2>&1
grep -E "^==|rc=|equal|attempt|sweep . (closure|sweep)" $O/solid_debug.log | cut -c1-170 | tail -n 60
timeout -k 5 240 python -m pytest tests/test_gpu_segmenter.py -m gpu -q -x -k "refine and not solid" > $O/r04c5_refine.log 2>&1; echo "rc=$?" >> $O/r04c5_refine.log; tail -n 4 $O/r04c5_refine.log
timeout -k 5 300 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x > $O/r04c5_full.log 2>&1; echo "rc=$?" >> $O/r04c5_full.log; tail -n 4 $O/r04c5_full.log
for c in longdress loot; do
  for ov in 0 1; do
    if [ $ov = 1 ]; then export TMC2_REFINE_OVERLAP=1; else unset TMC2_REFINE_OVERLAP; fi
    timeout -k 5 300 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 > $O/r04c5_bench_${c}_ov$ov.json 2> $O/r04c5_bench_${c}_ov$ov.err; echo "$c overlap $ov rc=$?"
  done
done
unset TMC2_REFINE_OVERLAP
python - <<'PY'
import json
for c in ("longdress_ov0", "longdress_ov1", "loot_ov0", "loot_ov1"):
    try:
        d = json.loads(open("gpurun_out/r04c5_bench_%s.json" % c).read().strip().splitlines()[-1])
        print(c, d["value"], "verified", d["verified"], "| proxy", d.get("per_rank_proxy", {}).get("ms"),
              {k: (v["alone_ms"], v["runs_per_frame"]) for k, v in d["roofline"]["stages"].items() if k.startswith("refine")})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
timeout -k 5 420 bash tools/asan_host_gcc.sh run python tools/asan_gof.py --config longdress --frames 16 --workers 16 --steps 3 > $O/r04_asan_longdress.log 2>&1; echo "asan longdress rc=$?" >> $O/r04_asan_longdress.log
tail -n 6 $O/r04_asan_longdress.log
