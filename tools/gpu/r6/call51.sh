#!/bin/bash
# round 6, call 51: the whole GPU tier in guard mode (4 KiB red zones around every buffer handed to the C-ABI) with the round's final library
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
TMC2_GUARD=1 timeout -k 10 3000 python -m pytest tests -x -q -m gpu > $O/r06_guard_tier.log 2>&1; tail -3 $O/r06_guard_tier.log
