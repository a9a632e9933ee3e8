#!/bin/bash
# round 6, call 27: pairSelect / scatterCompact with the counts of a workgroup folded per cluster in LDS (one global atomic per workgroup and cluster): parity, times
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/r06c27_tests.log 2>&1; tail -3 $O/r06c27_tests.log
db() { find "$1" -name "*_results.db" | head -1; }
ENC="python $REPO/tools/gpu/r6/first_pass.py --config longdress --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
cd /tmp; rm -rf $O/prof_enc; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c27_enc.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, S1-S22, 4 passes)" > $O/r06c27_kernel_stats.txt
python profiles/occupancy_rocpd.py "$(db $O/prof_enc)" 4 > $O/r06c27_occupancy.txt
head -2 $O/r06c27_occupancy.txt | tail -1
grep -i "pairSelect\|scatterCompact\|pairInsert\|initWords\|parityUnion\|flattenKernel" $O/r06c27_kernel_stats.txt
rm -rf $O/prof_enc
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
for i in 1 2; do
timeout 600 $B --steps 10 --warmup 3 > $O/r06c27_bench_$i.json 2> $O/r06c27_bench_$i.err
python - $i <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06c27_bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"]["ms"], d["stage_ms_per_frame"]["orient_contract"])
PY
done
