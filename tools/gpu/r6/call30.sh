#!/bin/bash
# round 6, call 30: the block form of S23's ordered sums (parity, times, fallbacks), the closure grid once more around its new optimum, the rough shell with 32 frames in flight
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 900 python -m pytest tests/test_gpu_metrics.py -x -q -m gpu > $O/r06c30_tests.log 2>&1; tail -3 $O/r06c30_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "metric or decoder" > $O/r06c30_tests_full.log 2>&1; tail -3 $O/r06c30_tests_full.log
python - > $O/r06c30_sums.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "mpeg-pcc-tmc2_amd"); sys.path.insert(0, "tests")
import tmc2_amd as T
from test_ordered_sum import term_families
ctx = T.Context(0)
rng = np.random.default_rng(1)
for n in (948_000, 3_031_000):
    t = np.zeros((n, 5)); f = term_families(rng, n)
    t[:, 0] = 3; t[:, 1] = f["d2"]; t[:, 2] = f["colour"]; t[:, 3] = f["colour"][::-1]; t[:, 4] = f["uniform"] * 1e-4
    for form in (None, "sequential"):
        ctx.set_option("METRICS_SUMS", form); ctx.set_option("METRICS_SUMS_DEBUG", "1")
        ctx.metrics_ordered_sums(t, t)
        ctx.set_option("METRICS_SUMS_DEBUG", None)
        ctx.stage_reset()
        for _ in range(5): out = ctx.metrics_ordered_sums(t, t)
        print(n, form or "block form", "metrics_sums ms per call:", ctx.stage_ms().get("metrics_sums", 0) / 5, out[:5])
PY
cat $O/r06c30_sums.txt
K=$O/r06c30_knobs.txt; : > $K
run() { # label, config, extra bench args, env...
  label=$1; cfg=$2; extra=$3; shift 3
  v=$( ( env "$@" timeout -k 5 400 python bench.py --config $cfg --steps 10 --warmup 3 --cpu-baseline 0 --ingest 0 --tail 0 --decoder 0 --gen-procs 8 $extra 2>/dev/null ) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['verified'], (d.get('per_rank_proxy') or {}).get('ms'), d['stage_ms_per_frame'].get('refine_sweeps'), d['stage_ms_per_frame'].get('orient_normals_host'), d['stage_ms_per_frame'].get('orient_contract'))" 2>&1 | tail -1 )
  echo "$cfg $label: $v" | tee -a $K
}
run default longdress "" X=1
for b in 128 256 512 1024; do for t in 128 256; do run closure${b}x$t longdress "" TMC2_REFINE_CLOSURE_BLOCKS=$b TMC2_REFINE_CLOSURE_THREADS=$t; done; done
run closure512x256_sweep512 longdress "" TMC2_REFINE_CLOSURE_BLOCKS=512 TMC2_REFINE_CLOSURE_THREADS=256 TMC2_REFINE_SWEEP_BLOCKS=512
run closure512x256_sweep2048 longdress "" TMC2_REFINE_CLOSURE_BLOCKS=512 TMC2_REFINE_CLOSURE_THREADS=256 TMC2_REFINE_SWEEP_BLOCKS=2048
run workers14 longdress "--workers 14" X=1
run workers18 longdress "--workers 18 --host-steps 18" X=1
run closure512x256 loot "" TMC2_REFINE_CLOSURE_BLOCKS=512 TMC2_REFINE_CLOSURE_THREADS=256
run default loot "" X=1
run closure512x256 basketball "" TMC2_REFINE_CLOSURE_BLOCKS=512 TMC2_REFINE_CLOSURE_THREADS=256
run default basketball "" X=1
run rough16 longdress "--workload longdress_vox10_noisy --steps 3 --warmup 1" X=1
run rough32 longdress "--workload longdress_vox10_noisy --steps 3 --warmup 1 --workers 32 --host-steps 32" X=1
