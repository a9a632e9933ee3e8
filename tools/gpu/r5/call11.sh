#!/bin/bash
# round 5, call 11: first passes with the staging reserved and the output buffers allocated up front
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for r in 1 0; do
timeout -k 10 600 python bench.py --reserve $r --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r05c11_bench_reserve$r.json 2> $O/r05c11_bench_reserve$r.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/r05c11_bench_reserve$r.json').read().strip().splitlines()[-1]); print("excess", d["first_gof_excess_ms_per_frame"]); print("reserve $r", d["value"], d['verified'], 'first', d['first_gof_ms'], d['untimed_pass_ms'], 'steady', d['ms_per_step'], d['pool'], d.get('per_rank_proxy',{}).get('ms'))"
done
