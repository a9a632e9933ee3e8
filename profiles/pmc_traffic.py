"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; csv output, one pass each -- the TCC block cannot hold both)
into HBM bytes per launch of the kernels bench.py reports a roofline for.
Corrections (MI355X_MICROARCH.md, "HBM"): both counters come back in KiB-sized units of 1024 B... see `unit` below; on gfx950
FETCH_SIZE tallies 128-byte requests as 64 bytes, so it is doubled.  WRITE_SIZE is used as reported (uncalibrated).
usage: python profiles/pmc_traffic.py <fetch_dir> <write_dir> <workload> > profiles/r02_pmc_traffic.json"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict


def source_stamp(root):
    """sha1 over the kernel sources: bench.py recomputes it and drops `roofline.traffic` when the counters are of other code"""
    h = hashlib.sha1()
    csrc = os.path.join(root, "mpeg-pcc-tmc2_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


STAGE_OF = {"knnKernel<16, true, true>": "knn_self", "knnKernel<8, false, true>": "knn8_recon_in_source",
            "knnKernel<1, false, true>": "knn1_source_in_recon", "normalsKernel<16>": "normals",
            "ccUnionKernel<16>": "k:ccUnion", "ccRelaxKernel<16>": "k:ccRelax", "ccMutualMaskKernel<16>": "k:ccMutualMask",
            "initialSegmentationKernel": "initial_segmentation"}
# stages that are several kernels per run: bytes per run = sum over the kernels of (bytes per launch x launches per run)
COMPOSITE = {"refine_sweep": {"closureKernel": 1, "sweepKernel": 1}}


def per_kernel(directory, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))
                tot[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
unit = 1024.0   # rocprofv3 reports both in KiB
out = {"workload": sys.argv[3], "source_sha1": source_stamp(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), one frame in flight",
       "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 tallies 128-byte requests as 64 bytes); WRITE_SIZE as reported",
       "stages": {}, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fb = 2.0 * unit * fetch.get(k, (0.0, 0))[0]
    wb = unit * write.get(k, (0.0, 0))[0]
    rec = {"fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb),
           "launches": fetch.get(k, write.get(k))[1]}
    out["kernels"][k] = rec
    if k in STAGE_OF:
        out["stages"][STAGE_OF[k]] = rec
for stage, parts in COMPOSITE.items():
    if all(k in out["kernels"] for k in parts):
        rec = {"fetch_bytes_per_launch": 0, "write_bytes_per_launch": 0, "hbm_bytes_per_launch": 0,
               "launches": min(out["kernels"][k]["launches"] for k in parts), "kernels": sorted(parts)}
        for k, per_run in parts.items():
            for f in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch"):
                rec[f] += per_run * out["kernels"][k][f]
        out["stages"][stage] = rec
json.dump(out, sys.stdout, indent=1)
print()
