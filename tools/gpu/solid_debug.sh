#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/solid_debug.log; : > $O
run() { echo "== $*" >> $O; ( timeout -k 3 25 env "$@" ) >> $O 2>&1; echo "rc=$?" >> $O; }
run TMC2_REFINE_DEBUG=1 TMC2_REFINE_NEIGHBOURHOOD=cells python tools/gpu/solid_debug.py 4 24 2
run TMC2_REFINE_DEBUG=1 TMC2_REFINE_NEIGHBOURHOOD=cells python tools/gpu/solid_debug.py 4 60 3
run TMC2_REFINE_DEBUG=1 TMC2_REFINE_CAPTIER=1 python tools/gpu/solid_debug.py 4 60 3
run TMC2_REFINE_DEBUG=1 python tools/gpu/solid_debug.py 4 60 3
run TMC2_REFINE_DEBUG=1 TMC2_REFINE_SWEEPS=full TMC2_REFINE_NEIGHBOURHOOD=cells python tools/gpu/solid_debug.py 4 60 3
run TMC2_REFINE_DEBUG=1 TMC2_REFINE_RING=8192 python tools/gpu/solid_debug.py 4 60 3
run TMC2_REFINE_DEBUG=1 python tools/gpu/solid_debug.py 2 40 3
cat $O | cut -c1-200
