#!/bin/bash
# round 4, sixth GPU call: the whole GPU tier, loot / longdress one-frame kernel traces, the other configs' bench lines
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x --durations=8 > $O/r04_gpu_tests.log 2>&1; echo "rc=$?" >> $O/r04_gpu_tests.log
tail -n 14 $O/r04_gpu_tests.log
cd /tmp
db() { find "$1" -name "*_results.db" | head -1; }
for c in loot longdress; do
  SOLO="python $REPO/bench.py --config $c --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
  rm -rf $O/prof_solo; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/r04_prof_$c.log 2>&1
  python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/r04c6_kernel_stats_one_frame_$c.txt
  rm -rf $O/prof_solo
  head -n 24 $O/r04c6_kernel_stats_one_frame_$c.txt
done
cd $REPO
for c in redandblack soldier basketball; do
  timeout -k 5 300 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 > $O/r04c6_bench_$c.json 2> $O/r04c6_bench_$c.err; echo "$c rc=$?"
done
python - <<'PY'
import json
for c in ("redandblack", "soldier", "basketball"):
    try:
        d = json.loads(open("gpurun_out/r04c6_bench_%s.json" % c).read().strip().splitlines()[-1])
        dec = d.get("decoder", {})
        print(c, d["value"], "verified", d["verified"], d["config"]["case"], "| proxy", d.get("per_rank_proxy", {}).get("ms"), "| decoder", dec.get("frames_per_s"), dec.get("verified"))
    except Exception as e:
        print(c, "no line:", repr(e))
PY
timeout -k 5 300 bash tools/asan_host_gcc.sh run python tools/asan_gof.py --config basketball --frames 4 --workers 4 --steps 2 > $O/r04_asan_basketball.log 2>&1; echo "asan basketball rc=$?" >> $O/r04_asan_basketball.log
tail -n 4 $O/r04_asan_basketball.log
