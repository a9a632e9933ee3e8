#!/bin/bash
# round 6, call 31: the closure's new default (two workgroups of 256 per CU) against the sweep kernel's grid and the idle groups' naps, three rounds, alternating
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_segmenter.py -x -q -m gpu -k "refine" > $O/r06c31_tests.log 2>&1; tail -2 $O/r06c31_tests.log
K=$O/r06c31_knobs.txt; : > $K
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 400 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), d['stage_ms_per_frame'].get('refine_sweeps'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $K
}
for round in 1 2 3; do
run default longdress "" X=1
run old512x512 longdress "" TMC2_REFINE_CLOSURE_THREADS=512
run sweep512 longdress "" TMC2_REFINE_SWEEP_BLOCKS=512
run sweep256 longdress "" TMC2_REFINE_SWEEP_BLOCKS=256
run naps4 longdress "" TMC2_REFINE_CLOSURE_NAPS=4
run naps16 longdress "" TMC2_REFINE_CLOSURE_NAPS=16
run sweep512naps4 longdress "" TMC2_REFINE_SWEEP_BLOCKS=512 TMC2_REFINE_CLOSURE_NAPS=4
done
run default loot "" X=1
run sweep512 loot "" TMC2_REFINE_SWEEP_BLOCKS=512
run naps4 loot "" TMC2_REFINE_CLOSURE_NAPS=4
run default loot "" X=1
run sweep512 loot "" TMC2_REFINE_SWEEP_BLOCKS=512
run naps4 loot "" TMC2_REFINE_CLOSURE_NAPS=4
