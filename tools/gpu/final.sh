#!/bin/bash
# end-of-round artefacts: rocprof traces + PMC traffic (profiles/collect.sh), then the bench line that reads the fresh traffic file
TAG=${1:-r03}
mkdir -p gpurun_out
bash profiles/collect.sh $TAG
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # (bench.py reads profiles/; the copy travels back in gpurun_out/)
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_${TAG}_final.json").read().strip().splitlines()[-1])
print(d["value"], d["verified"], d["roofline"]["kernel"], d["roofline"]["alone_avg_launch_ms"], d["roofline"]["alone_frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_source"))
PY
