#!/bin/bash
# round 6, call 14: where the other configurations stand after the S3 / S7 changes: encoder kernels of one basketball / loot frame alone, bench lines
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
db() { find "$1" -name "*_results.db" | head -1; }
for cfg in basketball loot; do
ENC="python $REPO/tools/gpu/r6/first_pass.py --config $cfg --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
cd /tmp; rm -rf $O/prof_enc; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c14_enc_$cfg.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, the path S1-S22 only, 4 passes)" > $O/r06c14_kernel_stats_$cfg.txt
python profiles/occupancy_rocpd.py "$(db $O/prof_enc)" 4 > $O/r06c14_occupancy_$cfg.txt
head -32 $O/r06c14_occupancy_$cfg.txt | tail -31
rm -rf $O/prof_enc
done
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
for cfg in basketball loot redandblack soldier; do
timeout 900 $B --config $cfg --steps 6 --warmup 2 > $O/r06c14_bench_$cfg.json 2> $O/r06c14_bench_$cfg.err
python - $cfg <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r06c14_bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"]["ms"], d["stage_ms_per_frame"])
PY
done
