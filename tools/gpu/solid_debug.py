import os, sys, time
sys.path.insert(0, "mpeg-pcc-tmc2_amd"); sys.path.insert(0, "tests")
import numpy as np
import tmc2_amd as T
import oracle_binding as ob
vox_dim, side, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
o = ob.Oracle()
g = np.arange(side, dtype=np.int16)
xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + np.int16(200)
rng = np.random.default_rng(5)
nrm = rng.normal(size=(len(xyz), 3)); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
p0 = o.initial_segmentation(nrm, np.ones(3))
exp = o.refine_grid(xyz, nrm, p0, iterations=iters, vox_dim=vox_dim)
ctx = T.Context(0)
fr = ctx.frame(xyz); fr.set_normals(nrm); fr.set_partition(p0)
t = time.time()
fr.segmenter_refine_grid_based(1024, 3.0, iters, vox_dim, 192)
print("vox", vox_dim, "side", side, "iters", iters, "equal", bool(np.array_equal(fr.get_partition(), exp)), "%.2f s" % (time.time() - t), ctx.stage_calls().get("refine_cap_tier_repeat", 0), flush=True)
