"""Does S5's state recur?  

The reference's grid refinement (PCCPatchSegmenter.cpp:1386-1561) carries exactly (partition[N], edge[V]) from one iteration to
the next, and an iteration is a deterministic map of that state -- so if the state after iteration t equals the state after
iteration t - p, every later state is known without running it.  This tool runs the ORACLE's restatement (test infrastructure,
CPU) on the synthetic stand-ins of the BASELINE sequences and prints, per iteration, how many points / voxel edge classes differ
from the state 1 and 2 iterations earlier.

    python tools/refine_recurrence.py [workload ...] [--frames 0,1] [--iterations 100]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpeg-pcc-tmc2_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from tmc2_amd import synth  # noqa: E402
from tmc2_amd.configs import FULL_SIZE_CASES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["longdress_vox10_ai_r3"])
    ap.add_argument("--frames", default="0")
    ap.add_argument("--iterations", type=int, default=0, help="0 = the configuration's own count")
    ap.add_argument("--workload", default=None, help="override the case's workload (e.g. small, medium)")
    a = ap.parse_args()
    orc = ob.Oracle()
    ref = ob.Reference() if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libtmc2ref.so")) else orc
    for case in a.cases:
        cfg = FULL_SIZE_CASES[case]
        for f in [int(x) for x in a.frames.split(",")]:
            xyz, rgb = synth.synth_cloud(a.workload or cfg["workload"], f)
            t0 = time.time()
            nrm = ref.normals(xyz, 16, True)
            w = ref.weight_normal(xyz, cfg["bits3d"], 0.6)
            part = ref.initial_segmentation(nrm, w)
            it = a.iterations or cfg["iterations"]
            out, tr = orc.refine_grid_trace(xyz, nrm, part, 1024, 3.0, it, cfg["vox_dim"], 192)
            print("# %s frame %d: %d points, voxels of %d, %d iterations (%.1f s)" % (case, f, len(xyz), cfg["vox_dim"], it,
                                                                                      time.time() - t0))
            print("# iter  points!=t-1  points!=t-2  edges!=t-1  edges!=t-2")
            for i, r in enumerate(tr):
                print("%5d %11d %12d %11d %11d" % (i, r[0], r[1], r[2], r[3]))
            p1 = [i for i, r in enumerate(tr) if r[0] == 0 and r[2] == 0]
            p2 = [i for i, r in enumerate(tr) if i >= 1 and r[1] == 0 and r[3] == 0]
            print("# first iteration whose state equals the previous one: %s; equals the one before that: %s" %
                  (p1[0] if p1 else "never", p2[0] if p2 else "never"))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
