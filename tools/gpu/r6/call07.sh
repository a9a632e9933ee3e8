#!/bin/bash
# round 6, call 07: the strong-edge threshold of S3's contraction -- lower thresholds contract more (a smaller graph for the host walk) and
# are exact whenever the contraction is consistent; what do 0.95 / 0.9 / 0.8 / 0.6 do on the CTC stand-in and on the rough shell?
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
for tau in 0.98 0.95 0.9 0.8 0.6; do
for wl in longdress noisy; do
if [ $wl = noisy ]; then extra="--workload longdress_vox10_noisy --frames 16 --steps 2 --warmup 1"; else extra="--steps 4 --warmup 1"; fi
TMC2_ORIENT_TAU=$tau timeout -k 10 600 python bench.py $extra --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 > $O/r06c07_${wl}_$tau.json 2> $O/r06c07_${wl}_$tau.err
python - <<PY
import json
try:
    d=json.loads(open("$O/r06c07_${wl}_$tau.json").read().strip().splitlines()[-1])
    s=d["stage_ms_per_frame"]; o=d["orientation"]
    print("tau $tau $wl: %.1f frames/s verified %s | contract %.2f host walk %.2f ms/frame | ladder %.1f fallbacks %.1f overflows %.1f repeats %.1f | rank proxy %.1f ms | compact edges %s clusters %s" % (
        d["value"], d["verified"], s.get("orient_contract",0), s.get("orient_normals_host",0), o["ladder_steps_per_gof"], o["point_level_fallbacks_per_gof"],
        o["pair_table_overflows_per_gof"], o["exact_size_repeats_per_gof"], d["per_rank_proxy"]["ms"], s.get("orient_compact_edges"), s.get("orient_clusters")))
except Exception as e: print("tau $tau $wl failed", e)
PY
done; done
