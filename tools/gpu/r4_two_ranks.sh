#!/bin/bash
# round 4: the N > 1 code path of bench.py (two ranks on the one GPU, gloo control plane) after the native-host change
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --cpu-baseline 0 --tail 0 --ingest 0 > gpurun_out/r04n_two_ranks.json 2> gpurun_out/r04n_two_ranks.err; echo "two ranks rc=$?"
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r04n_two_ranks.json").read().strip().splitlines()[-1])
    print(j["value"], j["n_gpus"], j["verified"], j["config"]["host"], "|", j["config"]["canvases"][:80], "| decoder", (j.get("decoder") or {}).get("verified"))
except Exception as e:
    print("no line:", e)
PY
tail -3 gpurun_out/r04n_two_ranks.err
