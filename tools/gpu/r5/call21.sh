#!/bin/bash
# round 5, call 21: the default bench line exactly as the driver runs it, twice (is 163 a box or a regression? does the CPU baseline leg survive?)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for k in 1 2; do
timeout -k 10 1200 python bench.py > $O/r05c21_bench$k.json 2> $O/r05c21_bench$k.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c21_bench$k.json').read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['ms_per_step'], d['untimed_pass_ms'], d.get('per_rank_proxy',{}).get('ms'), d['decoder'].get('frames_per_s'), {k: (v if not isinstance(v,str) else v[:200]) for k, v in d['cpu_baseline'].items() if k.endswith('value') or 'error' in k or 'failed' in k})"
done
rocm-smi --showclocks --showtemp --showpower 2>/dev/null | head -30
