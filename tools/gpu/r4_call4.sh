#!/bin/bash
# round 4, fourth GPU call (every step under a tight timeout): the hardened closure ring on the solid cloud, the torch-free ASan
# driver, the sharded native front end, the guard tests, bench.py with two ranks on one GPU (gloo control plane)
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 5 240 python -m pytest tests/test_gpu_segmenter.py -m gpu -q -x -k "solid or forms or refine" --durations=5 > $O/r04c4_refine.log 2>&1; echo "rc=$?" >> $O/r04c4_refine.log
tail -n 9 $O/r04c4_refine.log
timeout -k 5 300 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_guard.py -m gpu -q -x > $O/r04c4_ingest_guard.log 2>&1; echo "rc=$?" >> $O/r04c4_ingest_guard.log
tail -n 6 $O/r04c4_ingest_guard.log
for spec in "longdress 16 16 3" "loot 8 8 2" "basketball 4 4 2"; do
  set -- $spec
  timeout -k 5 420 bash tools/asan_host_gcc.sh run python tools/asan_gof.py --config $1 --frames $2 --workers $3 --steps $4 > $O/r04_asan_$1.log 2>&1; echo "asan $1 rc=$?" >> $O/r04_asan_$1.log
  tail -n 6 $O/r04_asan_$1.log
done
timeout -k 5 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --cpu-baseline 0 > $O/r04c4_two_ranks.json 2> $O/r04c4_two_ranks.err; echo "two ranks rc=$?" | tee -a $O/r04c4_two_ranks.err
tail -n 3 $O/r04c4_two_ranks.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04c4_two_ranks.json").read().strip().splitlines()[-1])
    print("two ranks:", d["value"], d["n_gpus"], "verified", d["verified"], d["verified_detail"][:80], "|", d["config"]["canvases"])
except Exception as e:
    print("two ranks: no line", repr(e))
PY
ls /dev/shm | head; df -h /dev/shm | tail -1
