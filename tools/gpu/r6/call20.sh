#!/bin/bash
# round 6, call 20: S5's ball probes with one record per occupied key (cell | voxel | members) instead of id -> centre: refine tests,
# loot / longdress kernels alone, loot / redandblack / soldier / longdress in flight
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 900 python -m pytest tests/test_gpu_segmenter.py -x -q -m gpu -k "refine or segmenter_compute" > $O/r06c20_tests.log 2>&1; tail -3 $O/r06c20_tests.log
db() { find "$1" -name "*_results.db" | head -1; }
for cfg in loot longdress; do
ENC="python $REPO/tools/gpu/r6/first_pass.py --config $cfg --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
cd /tmp; rm -rf $O/prof_enc; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c20_enc_$cfg.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, S1-S22, 4 passes)" > $O/r06c20_kernel_stats_$cfg.txt
echo "== $cfg"; grep -i "neighbourhoodKernel\|reverseRowsKernel\|rankToVoxel" $O/r06c20_kernel_stats_$cfg.txt
rm -rf $O/prof_enc
done
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
for cfg in loot redandblack soldier longdress; do
timeout 600 $B --config $cfg --steps 10 --warmup 3 > $O/r06c20_bench_$cfg.json 2> $O/r06c20_bench_$cfg.err
python - $cfg <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06c20_bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"]["ms"], {k:v for k,v in d["stage_ms_per_frame"].items() if k.startswith(("refine_s",))})
PY
done
