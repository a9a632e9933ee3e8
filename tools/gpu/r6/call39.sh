#!/bin/bash
# round 6, call 39: the colour transfer's searches in two launches (queries with an identical point in the tree first: one descent, one leaf; then the compacted rest): parity, kernel times, benches
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1200 python -m pytest tests/test_gpu_images.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/r06c39_tests.log 2>&1; tail -3 $O/r06c39_tests.log
timeout -k 10 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_gof32.py tests/test_integration_adaptor.py -x -q -m gpu > $O/r06c39_tests_full.log 2>&1; tail -3 $O/r06c39_tests_full.log
db() { find "$1" -name "*_results.db" | head -1; }
cd /tmp
for c in longdress; do
SOLO="python $REPO/bench.py --config $c --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
rm -rf $O/prof_solo; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/r06c39_prof_$c.log 2>&1
python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/r06c39_kernel_stats_one_frame_$c.txt
rm -rf $O/prof_solo
grep -i "knnKernel\|easyQuery\|gatherHard\|hardFlag" $O/r06c39_kernel_stats_one_frame_$c.txt
done
cd $REPO
K=$O/r06c39_bench.txt; : > $K
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 400 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), d['stage_ms_per_frame'].get('knn8_recon_in_source'), d['stage_ms_per_frame'].get('knn1_source_in_recon'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $K
}
for round in 1 2; do
run split longdress "" X=1
run one_launch longdress "" TMC2_KNN_SPLIT=0
done
run split loot "" X=1
run one_launch loot "" TMC2_KNN_SPLIT=0
run split redandblack "" X=1
run split soldier "" X=1
run split basketball "" X=1
run one_launch basketball "" TMC2_KNN_SPLIT=0
