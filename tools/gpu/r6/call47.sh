#!/bin/bash
# round 6, call 47: boundsKernel as a stride loop with six atomics per workgroup (the de-duplication's bounding box): parity, the metric's stage times
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py tests/test_integration_adaptor.py -x -q -m gpu > $O/r06c47_tests.log 2>&1; tail -2 $O/r06c47_tests.log
timeout -k 10 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_gof32.py -x -q -m gpu -k "decoder or metric" > $O/r06c47_tests_full.log 2>&1; tail -2 $O/r06c47_tests_full.log
for c in longdress basketball; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 2 --cpu-baseline 0 --ingest 0 --tail 0 --gen-procs 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['verified'], 'metric ms', d.get('metric_ms_per_frame'), d.get('metric_stage_ms'), 'decoder', (d.get('decoder') or {}).get('frames_per_s'), (d.get('decoder') or {}).get('verified'))" | tee -a $O/r06c47_metric.txt
done
