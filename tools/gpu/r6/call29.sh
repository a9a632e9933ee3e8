#!/bin/bash
# round 6, call 29: knob sweep with this round's kernels (closure / sweep grids, frames in flight) on longdress; the rough shell with 32 frames in flight
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out/r06c29_knobs.txt; : > $O
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 300 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), d['stage_ms_per_frame'].get('refine_sweeps'), d['stage_ms_per_frame'].get('orient_normals_host'), d['stage_ms_per_frame'].get('orient_contract'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $O
}
run default longdress "" X=1
run closure256 longdress "" TMC2_REFINE_CLOSURE_BLOCKS=256
run closure384 longdress "" TMC2_REFINE_CLOSURE_BLOCKS=384
run closure256x1024 longdress "" TMC2_REFINE_CLOSURE_BLOCKS=256 TMC2_REFINE_CLOSURE_THREADS=1024
run closure512x256 longdress "" TMC2_REFINE_CLOSURE_BLOCKS=512 TMC2_REFINE_CLOSURE_THREADS=256
run sweep512 longdress "" TMC2_REFINE_SWEEP_BLOCKS=512
run sweep2048 longdress "" TMC2_REFINE_SWEEP_BLOCKS=2048
run workers12 longdress "--workers 12" X=1
run workers20 longdress "--workers 20 --host-steps 20" X=1
run workers24 longdress "--workers 24 --host-steps 24" X=1
run workers32 longdress "--workers 32 --host-steps 32" X=1
run default longdress "" X=1
run rough16 rough "--steps 4 --warmup 1" X=1
run rough32 rough "--steps 4 --warmup 1 --workers 32 --host-steps 32" X=1
