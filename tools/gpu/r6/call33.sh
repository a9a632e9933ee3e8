#!/bin/bash
# round 6, call 33: the chain kernel of the ordered sums with its records through v_readlane (unrolled): parity, times; the whole GPU tier with this round's changes
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py -x -q -m gpu > $O/r06c33_tests.log 2>&1; tail -2 $O/r06c33_tests.log
python - > $O/r06c33_sums.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "mpeg-pcc-tmc2_amd"); sys.path.insert(0, "tests")
import tmc2_amd as T
from test_ordered_sum import term_families
ctx = T.Context(0)
rng = np.random.default_rng(1)
for n in (948_000, 3_031_000):
    t = np.zeros((n, 5)); f = term_families(rng, n)
    t[:, 0] = 3; t[:, 1] = f["d2"]; t[:, 2] = f["colour"]; t[:, 3] = f["colour"][::-1]; t[:, 4] = f["uniform"] * 1e-4
    ctx.set_option("METRICS_SUMS_DEBUG", "1"); ctx.metrics_ordered_sums(t, t); ctx.set_option("METRICS_SUMS_DEBUG", None)
    ctx.stage_reset()
    for _ in range(5): out = ctx.metrics_ordered_sums(t, t)
    print(n, "block form metrics_sums ms per call:", ctx.stage_ms().get("metrics_sums", 0) / 5, out[:5])
PY
cat $O/r06c33_sums.txt
timeout -k 10 2400 python -m pytest tests -x -q -m gpu > $O/r06c33_tests_all.log 2>&1; tail -3 $O/r06c33_tests_all.log
