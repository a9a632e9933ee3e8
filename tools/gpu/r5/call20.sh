#!/bin/bash
# round 5, call 20: the whole GPU tier and the default bench line at the round's last state
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
(timeout -k 10 1500 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12) > $O/r05c20_gpu_tier.log 2>&1
tail -3 $O/r05c20_gpu_tier.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05c20_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 900 python bench.py > $O/r05c20_bench.json 2> $O/r05c20_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c20_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['verified'], d['first_gof_ms'], d['pool'], d.get('per_rank_proxy',{}).get('ms'), d['decoder'].get('frames_per_s'), d['decoder'].get('verified'), {k: v for k, v in d['cpu_baseline'].items() if k.endswith('value')})"
