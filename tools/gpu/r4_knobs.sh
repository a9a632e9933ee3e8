#!/bin/bash
# knob sweep on the voxels-of-2 configuration (and two on longdress): closure grid, sweep grid, frames in flight
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04_knobs.txt; : > $O
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 200 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['per_rank_proxy'].get('ms'), d['roofline']['stages'].get('refine_sweep',{}).get('alone_ms'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $O
}
run default loot "" X=1
run closure1024 loot "" TMC2_REFINE_CLOSURE_BLOCKS=1024
run closure2048 loot "" TMC2_REFINE_CLOSURE_BLOCKS=2048
run closure256 loot "" TMC2_REFINE_CLOSURE_BLOCKS=256
run sweep1024 loot "" TMC2_REFINE_SWEEP_BLOCKS=1024
run sweep2048 loot "" TMC2_REFINE_SWEEP_BLOCKS=2048
run sweep4096 loot "" TMC2_REFINE_SWEEP_BLOCKS=4096
run workers12 loot "--workers 12" X=1
run workers20 loot "--workers 20" GPU_MAX_HW_QUEUES=20
run default longdress "" X=1
run sweep1024 longdress "" TMC2_REFINE_SWEEP_BLOCKS=1024
run closure1024 longdress "" TMC2_REFINE_CLOSURE_BLOCKS=1024
