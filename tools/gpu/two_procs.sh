#!/bin/bash
# does the GPU take more from two processes (16 frames, 8 in flight each) than from one (32 frames, 16 in flight)?
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --tail 0 --ingest 0"
$B --frames 32 --workers 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one process, 16 in flight:', d['value'])"
$B --frames 16 --workers 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one process, 8 in flight alone:', d['value'])"
for i in 1 2; do ( $B --frames 16 --workers 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  of two processes, 8 in flight each:', d['value'])" ) & done; wait
for i in 1 2 3 4; do ( $B --frames 8 --workers 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  of four processes, 4 in flight each:', d['value'])" ) & done; wait
