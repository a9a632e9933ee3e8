#!/bin/bash
# round 6, call 12: the passes of S3 / S7 three ways -- blocks as they come (input), XCD x on the x-th eighth (chunk), tree order on
# the same eighths (tree): kernel times of one frame alone, HBM traffic of the default, sixteen in flight for each
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py -x -q -m gpu -k "orientation or patches or segmenter_compute" > $O/r06c12_tests.log 2>&1; tail -3 $O/r06c12_tests.log
db() { find "$1" -name "*_results.db" | head -1; }
ENC="python $REPO/tools/gpu/r6/first_pass.py --config longdress --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
for ord in input chunk tree; do
cd /tmp; rm -rf $O/prof_enc; TMC2_ORIENT_ORDER=$ord TMC2_MUTUAL_ORDER=$ord timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c12_enc_$ord.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, S1-S22, 4 passes, ORIENT_ORDER = MUTUAL_ORDER = $ord)" > $O/r06c12_kernel_stats_$ord.txt
echo "== $ord"; grep -i "initWords\|parityUnion\|flattenKernel\|pairInsert\|pairSelect\|scatterCompact\|ccUnion\|ccRelax\|ccInit\|ccMutual" $O/r06c12_kernel_stats_$ord.txt
rm -rf $O/prof_enc
done
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
for ord in input chunk tree; do
TMC2_ORIENT_ORDER=$ord TMC2_MUTUAL_ORDER=$ord timeout 600 $B --steps 10 --warmup 3 > $O/r06c12_bench_$ord.json 2> $O/r06c12_bench_$ord.err
python - $ord <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06c12_bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"]["ms"], {k:v for k,v in d["stage_ms_per_frame"].items() if k.startswith(("orient","patches","k:"))})
PY
done
