#!/bin/bash
# round 6, call 21: the one-point-per-lane passes of S2 (normals) and S7-S9 with XCD x on the x-th eighth of the blocks (POINT_CHUNK)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 900 python -m pytest tests/test_gpu_segmenter.py -x -q -m gpu > $O/r06c21_tests.log 2>&1; tail -3 $O/r06c21_tests.log
db() { find "$1" -name "*_results.db" | head -1; }
ENC="python $REPO/tools/gpu/r6/first_pass.py --config longdress --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
for ch in 0 1; do
cd /tmp; rm -rf $O/prof_enc; TMC2_POINT_CHUNK=$ch timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c21_enc_$ch.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, S1-S22, 4 passes, POINT_CHUNK = $ch)" > $O/r06c21_kernel_stats_$ch.txt
echo "== POINT_CHUNK=$ch"; grep -i "normalsKernel\|ccFlattenSeed\|ccLabelCount\|ccSeedFlag\|ccAssign\|patchMinUv\|patchTrimBbox\|patchDepth0\|patchDepth1\|rawDistance" $O/r06c21_kernel_stats_$ch.txt
rm -rf $O/prof_enc
done
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
for ch in 0 1 0 1; do
TMC2_POINT_CHUNK=$ch timeout 600 $B --steps 10 --warmup 3 > $O/r06c21_bench_$ch.json 2> $O/r06c21_bench_$ch.err
python - $ch <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06c21_bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("POINT_CHUNK", sys.argv[1], "value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"]["ms"])
PY
done
