#!/bin/bash
# round 6, call 13: S5's set-up and closure with XCD x on the x-th eighth of the voxels (REFINE_CHUNK) against the blocks as they come,
# longdress and loot: kernel times alone, sixteen in flight; HBM traffic of the new defaults; refine / segmenter tests
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py -x -q -m gpu > $O/r06c13_tests.log 2>&1; tail -3 $O/r06c13_tests.log
db() { find "$1" -name "*_results.db" | head -1; }
for cfg in longdress loot; do
ENC="python $REPO/tools/gpu/r6/first_pass.py --config $cfg --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
for ch in 0 1; do
cd /tmp; rm -rf $O/prof_enc; TMC2_REFINE_CHUNK=$ch timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c13_enc_${cfg}_$ch.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, S1-S22, 4 passes, REFINE_CHUNK = $ch)" > $O/r06c13_kernel_stats_${cfg}_$ch.txt
echo "== $cfg REFINE_CHUNK=$ch"; grep -i "neighbourhoodKernel\|reverseRowsKernel\|closureKernel\|sweepKernel" $O/r06c13_kernel_stats_${cfg}_$ch.txt
rm -rf $O/prof_enc
done
done
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
for cfg in longdress loot; do
for ch in 0 1; do
TMC2_REFINE_CHUNK=$ch timeout 600 $B --config $cfg --steps 10 --warmup 3 > $O/r06c13_bench_${cfg}_$ch.json 2> $O/r06c13_bench_${cfg}_$ch.err
python - $cfg $ch <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06c13_bench_%s_%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], "REFINE_CHUNK", sys.argv[2], "value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"]["ms"], {k:v for k,v in d["stage_ms_per_frame"].items() if k.startswith(("refine_s",))})
PY
done
done
SOLO="python $REPO/bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c; timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $SOLO > $O/r06c13_pmc_$c.log 2>&1
done
cd $REPO
python profiles/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "longdress_vox10" > $O/r06c13_pmc_traffic.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06c13_pmc_traffic.json"))
for k in ("initWordsKernel<16>","parityUnionKernel<16>","flattenKernel","pairInsertKernel<16>","pairSelectKernel<16>","scatterCompactKernel<16>","ccMutualMaskKernel<16>","ccUnionKernel<16>","ccInitKernel<16>","ccRelaxKernel<16>","closureKernel","sweepKernel","neighbourhoodKernel<1024, 8>","reverseRowsKernel<1024, 8>"):
    v=d["kernels"].get(k)
    print(k, round(v["hbm_bytes_per_launch"]/1e6,1) if v else None, "MB")
PY
