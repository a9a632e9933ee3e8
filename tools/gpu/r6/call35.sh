#!/bin/bash
# round 6, call 35: one row reservation per wavefront over 128 region cursors instead of one per workgroup on ONE word (+ the kept hits of call 34): parity, kernel times of one loot frame, the configurations
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/r06c35_tests.log 2>&1; tail -3 $O/r06c35_tests.log
timeout -k 10 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_gof32.py tests/test_gpu_gof_soak.py -x -q -m gpu > $O/r06c35_tests_full.log 2>&1; tail -3 $O/r06c35_tests_full.log
db() { find "$1" -name "*_results.db" | head -1; }
cd /tmp
SOLO="python $REPO/bench.py --config loot --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
rm -rf $O/prof_solo; TMC2_REFINE_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/r06c35_prof_loot.log 2>&1
python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/r06c35_kernel_stats_one_frame_loot.txt
rm -rf $O/prof_solo
cd $REPO
grep "neighbourhood attempt" $O/r06c35_prof_loot.log | sort | uniq -c | head
grep -i "neighbourhoodKernel\|reverseRowsKernel" $O/r06c35_kernel_stats_one_frame_loot.txt
K=$O/r06c35_bench.txt; : > $K
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 400 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), d['stage_ms_per_frame'].get('refine_setup'), d['stage_ms_per_frame'].get('refine_sweeps'), d['stage_ms_per_frame'].get('refine_hits_out_of_room'), d.get('pool'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $K
}
for round in 1 2; do
run kept loot "" X=1
run collect_twice loot "" TMC2_REFINE_HITS=0
done
run kept redandblack "" X=1
run collect_twice redandblack "" TMC2_REFINE_HITS=0
run kept soldier "" X=1
run collect_twice soldier "" TMC2_REFINE_HITS=0
run kept longdress "" X=1
run collect_twice longdress "" TMC2_REFINE_HITS=0
run kept basketball "" X=1
run collect_twice basketball "" TMC2_REFINE_HITS=0
