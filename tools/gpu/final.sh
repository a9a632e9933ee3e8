#!/bin/bash
# end-of-round artefacts (one gpurun call): smoke, rocprof traces + PMC traffic of the headline configuration (profiles/collect.sh),
# one-frame traces of the voxels-of-2 and vox11 configurations, the gaps / copies of one frame, ONE DRIVER-COMPARABLE LINE PER
# BASELINE CONFIGURATION -- default steps, the reference timed on this box's host cores in the same run (north_star: every
# throughput next to the reference's CPU/TBB path; --cpu-baseline 3 = one frame on one thread + 8 frames through its TBB path) --
# the headline line with every side leg, the rough-shell workload, two ranks on one GPU.
TAG=${1:-r06}
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/${TAG}_smoke.log | cut -c1-200
bash profiles/collect.sh $TAG
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # (bench.py reads profiles/; the copy travels back in gpurun_out/)
cd /tmp
db() { find "$1" -name "*_results.db" | head -1; }
for c in loot basketball; do
  SOLO="python $REPO/bench.py --config $c --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
  rm -rf $O/prof_solo; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/${TAG}_prof_$c.log 2>&1
  python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/${TAG}_kernel_stats_one_frame_$c.txt
  rm -rf $O/prof_solo
done
cd $REPO
bash tools/gpu/gaps.sh $TAG > /dev/null 2>&1
timeout -k 10 1200 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_final.json 2> $O/bench_${TAG}_final.err; echo "longdress rc=$?"
for c in loot redandblack soldier basketball; do
  timeout -k 5 600 python bench.py --config $c --cpu-baseline 3 --ingest 0 --tail 0 > $O/bench_${TAG}_$c.json 2> $O/bench_${TAG}_$c.err; echo "$c rc=$?"
done
timeout -k 5 600 python bench.py --workload longdress_vox10_noisy --steps 3 --warmup 1 --cpu-baseline 3 --tail 0 --ingest 0 --decoder 0 > $O/bench_${TAG}_rough_shell.json 2> $O/bench_${TAG}_rough_shell.err; echo "rough shell rc=$?"
timeout -k 5 420 python bench.py --gpus 2 --dist-backend gloo --steps 5 --warmup 2 --cpu-baseline 0 > $O/bench_${TAG}_two_ranks_one_gpu.json 2> $O/bench_${TAG}_two_ranks_one_gpu.err; echo "two ranks (bench.py starts its own launcher) rc=$?"
python - <<PY
import json
for c in ("final", "loot", "redandblack", "soldier", "basketball", "rough_shell", "two_ranks_one_gpu"):
    try:
        d = json.loads(open("gpurun_out/bench_${TAG}_%s.json" % c).read().strip().splitlines()[-1])
        dec = d.get("decoder", {})
        r = d["roofline"]
        print(c, d["value"], "n_gpus", d["n_gpus"], "verified", d["verified"], "first_gof_ms", d.get("first_gof_ms"), "| roofline", r["kernel"], r["alone_avg_launch_ms"], r["alone_frac"], "traffic", r["traffic"],
              "| path", r["path"], "| proxy", d.get("per_rank_proxy", {}).get("ms"), d.get("per_rank_proxy", {}).get("predicted_n8_speedup"),
              "| decoder", dec.get("frames_per_s"), dec.get("verified"), "| cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k.endswith("value")},
              "| S3", {k: v for k, v in d.get("orientation", {}).items() if k.endswith("per_gof") and k != "frames_per_gof"})
    except Exception as e:
        print(c, "no line:", repr(e))
PY
