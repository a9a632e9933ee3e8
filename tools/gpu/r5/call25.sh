#!/bin/bash
# round 5, call 25: the host side under gcc's AddressSanitizer on the GPU box (16 frames in flight), and the GPU tier's C-ABI calls
# under the binding's red-zone guard -- the new entries (options, reservation, device staging, sharded host) included
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 5 600 bash tools/asan_host_gcc.sh run python tools/asan_gof.py --config longdress --frames 16 --workers 16 --steps 2 > $O/r05_asan_longdress.log 2>&1; echo "asan longdress rc=$?" >> $O/r05_asan_longdress.log
timeout -k 5 600 bash tools/asan_host_gcc.sh run python tools/asan_gof.py --config loot --frames 8 --workers 8 --steps 2 > $O/r05_asan_loot.log 2>&1; echo "asan loot rc=$?" >> $O/r05_asan_loot.log
tail -n 3 $O/r05_asan_longdress.log; tail -n 3 $O/r05_asan_loot.log
(TMC2_GUARD=1 timeout -k 10 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/r05_guard_tier.log 2>&1; tail -2 $O/r05_guard_tier.log
