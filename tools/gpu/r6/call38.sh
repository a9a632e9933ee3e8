#!/bin/bash
# round 6, call 38: the sweep pushes as one signed 64-bit add + one 32-bit add per target instead of three: parity, kernel times, benches
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/r06c38_tests.log 2>&1; tail -3 $O/r06c38_tests.log
timeout -k 10 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_gof32.py -x -q -m gpu > $O/r06c38_tests_full.log 2>&1; tail -3 $O/r06c38_tests_full.log
db() { find "$1" -name "*_results.db" | head -1; }
cd /tmp
for c in loot; do
SOLO="python $REPO/bench.py --config $c --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
rm -rf $O/prof_solo; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/r06c38_prof_$c.log 2>&1
python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/r06c38_kernel_stats_one_frame_$c.txt
rm -rf $O/prof_solo
grep -i "closureKernel\|sweepKernel" $O/r06c38_kernel_stats_one_frame_$c.txt
done
cd $REPO
K=$O/r06c38_bench.txt; : > $K
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 400 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), d['stage_ms_per_frame'].get('refine_setup'), d['stage_ms_per_frame'].get('refine_sweeps'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $K
}
run hop loot "" X=1
run hop loot "" X=1
run hop redandblack "" X=1
run hop soldier "" X=1
run hop longdress "" X=1
run hop longdress "" X=1
run words loot "" TMC2_REFINE_PUSH=words
run words longdress "" TMC2_REFINE_PUSH=words
run paired basketball "" X=1
