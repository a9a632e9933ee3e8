#!/bin/bash
# round 6, call 11: S3's contraction without the 16 N doubles (classes as bits, cross-edge values recomputed), S3 / S7 passes in tree
# order with XCD x on the x-th eighth: parity, kernel times alone, HBM traffic (PMC), sixteen in flight
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_full_size.py -x -q -m gpu > $O/r06c11_tests.log 2>&1; tail -3 $O/r06c11_tests.log
db() { find "$1" -name "*_results.db" | head -1; }
ENC="python $REPO/tools/gpu/r6/first_pass.py --config longdress --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
cd /tmp; rm -rf $O/prof_enc; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c11_enc.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, the path S1-S22 only, 4 passes)" > $O/r06c11_kernel_stats_encoder_one_frame.txt
python profiles/occupancy_rocpd.py "$(db $O/prof_enc)" 4 > $O/r06c11_occupancy_encoder_one_frame.txt
head -3 $O/r06c11_occupancy_encoder_one_frame.txt | tail -2
grep -i "initWords\|parityUnion\|flattenKernel\|pairInsert\|pairSelect\|scatterCompact\|edgeDot\|ccUnion\|ccRelax\|ccInit\|ccMutual\|clusterFlag" $O/r06c11_kernel_stats_encoder_one_frame.txt
rm -rf $O/prof_enc
SOLO="python $REPO/bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c; timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $SOLO > $O/r06c11_pmc_$c.log 2>&1
done
cd $REPO
python profiles/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "longdress_vox10" > $O/r06c11_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06c11_pmc_traffic.json"))
for k in ("initWordsKernel<16>","parityUnionKernel<16>","flattenKernel","pairInsertKernel<16>","pairSelectKernel<16>","scatterCompactKernel<16>","tmc2::edgeDotKernel","ccMutualMaskKernel<16>","ccUnionKernel<16>","ccInitKernel<16>","ccRelaxKernel<16>"):
    v=d["kernels"].get(k)
    print(k, round(v["hbm_bytes_per_launch"]/1e6,1) if v else None, "MB")
PY
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
timeout 600 $B --steps 10 --warmup 3 > $O/r06c11_bench.json 2> $O/r06c11_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06c11_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"], {k:v for k,v in d["stage_ms_per_frame"].items() if k.startswith(("orient","patches","k:"))})
PY
