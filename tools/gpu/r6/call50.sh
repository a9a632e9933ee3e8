#!/bin/bash
# round 6, call 50: a graph that barely contracts (the rough shell) walked point by point instead of through the compact graph with every cross edge: parity, the rough shell's bench line both ways
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 1200 python -m pytest tests/test_gpu_segmenter.py -x -q -m gpu -k "orientation or normals or segmenter_matches" > $O/r06c50_tests.log 2>&1; tail -2 $O/r06c50_tests.log
timeout -k 10 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_gof32.py -x -q -m gpu -k "noisy or rough or longdress" > $O/r06c50_tests_full.log 2>&1; tail -2 $O/r06c50_tests_full.log
K=$O/r06c50_rough.txt; : > $K
for v in point_level contracted point_level contracted; do
  if [ $v = contracted ]; then export TMC2_ORIENT_CONTRACT_ALWAYS=1; else unset TMC2_ORIENT_CONTRACT_ALWAYS; fi
  r=$( timeout 600 python bench.py --workload longdress_vox10_noisy --steps 3 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_frame']; print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), 'contract', s.get('orient_contract'), 'walk', s.get('orient_normals_host'), d['orientation'])" 2>&1 | tail -1 )
  echo "rough shell, $v: $r" | tee -a $K
done
unset TMC2_ORIENT_CONTRACT_ALWAYS
r=$( timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['orientation'])" 2>&1 | tail -1 )
echo "longdress: $r" | tee -a $K
