import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, '/root/repo/mpeg-pcc-tmc2_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import tmc2_amd as T, oracle_binding as ob
from tmc2_amd.synth import synth_cloud
o = ob.Oracle(); ctx = T.Context(0)
for name in ("small", "medium"):
    xyz, rgb = synth_cloud(name)
    nrm = o.normals(xyz); p0 = o.initial_segmentation(nrm, o.weight_normal(xyz))
    for iters in (1, 2, 3, 10):
        fr = ctx.frame(xyz, rgb); fr.set_normals(nrm); fr.set_partition(p0)
        fr.segmenter_refine_grid_based(1024, 3.0, iters, 2, 192)
        got = fr.get_partition(); exp = o.refine_grid(xyz, nrm, p0, iterations=iters, vox_dim=2)
        bad = np.flatnonzero(got != exp)
        print(name, len(xyz), "iters", iters, "mismatches", len(bad), bad[:8], flush=True)
        fr.close()
print("done", flush=True)
ctx.close()
print("closed", flush=True)
