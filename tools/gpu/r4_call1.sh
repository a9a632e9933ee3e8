#!/bin/bash
# round 4, first GPU call: the whole GPU tier (every BASELINE config with frames in flight, decoder side against the reference),
# the host side under gcc-ASan and under the binding's red-zone guard, one bench line per config
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/r04_gpu_tests.log 2>&1; echo "rc=$?" >> $O/r04_gpu_tests.log
tail -n 25 $O/r04_gpu_tests.log
timeout -k 10 900 bash tools/asan_host_gcc.sh run python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --pin 0 > $O/r04_asan_bench.json 2> $O/r04_asan_bench.err; echo "asan rc=$?" | tee -a $O/r04_asan_bench.err
tail -n 5 $O/r04_asan_bench.err
TMC2_GUARD=1 timeout -k 10 600 python bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $O/r04_guard_bench.json 2> $O/r04_guard_bench.err; echo "guard rc=$?" | tee -a $O/r04_guard_bench.err
for c in longdress loot redandblack soldier basketball; do
  timeout -k 10 600 python bench.py --config $c --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 > $O/r04_bench_$c.json 2> $O/r04_bench_$c.err; echo "$c rc=$?" | tee -a $O/r04_bench_$c.err
done
python - <<'PY'
import json
for c in ("asan_bench", "guard_bench", "bench_longdress", "bench_loot", "bench_redandblack", "bench_soldier", "bench_basketball"):
    try:
        d = json.loads(open("gpurun_out/r04_%s.json" % c).read().strip().splitlines()[-1])
        dec = d.get("decoder", {})
        print(c, d["value"], "verified", d["verified"], d["verified_detail"][:80], "| roofline", d["roofline"]["kernel"], d["roofline"]["alone_frac"],
              "path", d["roofline"]["path"], "| proxy", d.get("per_rank_proxy", {}).get("ms"), "| decoder", dec.get("frames_per_s"), dec.get("verified"),
              str(dec.get("verified_detail", dec.get("error")))[:100], "| tail", d.get("tail", {}).get("gof_frames_per_s", d.get("tail")), "| metric ms", d.get("metric_ms_per_frame"))
    except Exception as e:
        print(c, "no line:", repr(e))
PY
