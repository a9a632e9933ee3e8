#!/bin/bash
# round 6, call 49: the closure / sweep grids on a voxels-of-2 configuration (rows of 128, 262 K voxels), two rounds
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out/r06c49_knobs_loot.txt; : > $O
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 400 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), d['stage_ms_per_frame'].get('refine_setup'), d['stage_ms_per_frame'].get('refine_sweeps'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $O
}
for round in 1 2; do
run default loot "" X=1
run closure1024x256 loot "" TMC2_REFINE_CLOSURE_BLOCKS=1024 TMC2_REFINE_CLOSURE_THREADS=256
run closure256x256 loot "" TMC2_REFINE_CLOSURE_BLOCKS=256 TMC2_REFINE_CLOSURE_THREADS=256
run closure512x512 loot "" TMC2_REFINE_CLOSURE_BLOCKS=512 TMC2_REFINE_CLOSURE_THREADS=512
run sweep1024 loot "" TMC2_REFINE_SWEEP_BLOCKS=1024
run sweep256 loot "" TMC2_REFINE_SWEEP_BLOCKS=256
run sweep2048 loot "" TMC2_REFINE_SWEEP_BLOCKS=2048
done
