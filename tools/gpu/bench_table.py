"""The rows of DESIGN.md section 6's table from the bench lines tools/gpu/final.sh left in profiles/ (one JSON line per file).
usage: python tools/gpu/bench_table.py r06"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
names = [("final", "longdress (config 2, the headline)"), ("loot", "loot (config 3, voxels of 2)"), ("redandblack", "redandblack (config 3)"),
         ("soldier", "soldier (config 3)"), ("basketball", "basketball (config 4: 3.03 M points, random access)"),
         ("rough_shell", "rough shell (`--workload longdress_vox10_noisy`: 1.40 M points, not a BASELINE configuration)")]
for key, label in names:
    d = json.loads(open("profiles/bench_%s_%s.json" % (tag, key)).read().strip().splitlines()[-1])
    cpu = d.get("cpu_baseline") or {}
    one, allc, procs = cpu.get("value"), cpu.get("all_cores_value"), cpu.get("frame_processes_value")
    r = d["roofline"]
    px = d.get("per_rank_proxy") or {}
    dec = d.get("decoder") or {}
    steady = d["ms_per_step"]
    print("| %s | **%.1f** | %s / %s%s | %s | %.2f GB -> %.0f GB/s = %.1f %% | %s ms -> %s x | %s | %.0f (%.0f) |" % (
        label, d["value"], "%.4f" % one if one else "--", "%.3f" % allc if allc else "--", " (%.2f as 32 processes)" % procs if procs else "",
        "**%.0f x**" % (d["value"] / allc) if allc else "--", r["path"]["B_alg_GB_per_frame"], r["path"]["achieved"], 100 * r["path"]["frac"],
        px.get("ms"), px.get("predicted_n8_speedup"), "%.1f" % dec["frames_per_s"] if dec.get("frames_per_s") else "--",
        d.get("first_gof_ms") or 0, steady))
d = json.loads(open("profiles/bench_%s_final.json" % tag).read().strip().splitlines()[-1])
print("headline roofline:", json.dumps({k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "alone_avg_launch_ms", "alone_achieved", "alone_frac", "traffic", "avg_launch_ms")}))
print("metric_ms_per_frame", d.get("metric_ms_per_frame"), d.get("metric_stage_ms"))
