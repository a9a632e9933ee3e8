"""GPU: the round-3 code paths behind their test hooks, on degenerate and synthetic clouds, against the oracle / the host
builder.  S5's closure (LDS ring size, grid, workgroup size), the tree build's tiers (TMC2_KD_HUGEMAX), the k-NN bound.
usage: python tools/fuzz/fuzz_gpu_hooks.py <first seed> <last seed>      (needs an MI355X; run through gpurun)"""
import sys, os, collections
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "mpeg-pcc-tmc2_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_binding as ob
import tmc2_amd as T
from test_oracle_golden import degenerate_cloud
from tmc2_amd.synth import synth_cloud

oracle = ob.Oracle()
ctx = T.Context(0)
stats = collections.Counter()
HOOKS = {"TMC2_REFINE_RING": [None, "1", "3", "40"], "TMC2_REFINE_CLOSURE_BLOCKS": [None, "1", "2", "7", "100"],
         "TMC2_REFINE_CLOSURE_THREADS": [None, "64", "256", "1024"], "TMC2_KD_HUGEMAX": [None, "8192", "10000", "16384", "131072"],
         # round 4: S5's neighbourhood forms (row-wise through the occupancy bitmap / cell by cell), the LDS tier the row-wise kernels
         # start in, the sweep kernel's grid, the pair table of the orientation's contraction
         "TMC2_REFINE_NEIGHBOURHOOD": [None, None, "cells"], "TMC2_REFINE_CAPTIER": [None, None, "1", "2", "3", "4"], "TMC2_REFINE_HITS": [None, None, None, "0", "tiny"], "TMC2_REFINE_PUSH": [None, None, "words"], "TMC2_KNN_SPLIT": [None, None, "0"],
         "TMC2_REFINE_SWEEP_BLOCKS": [None, "1", "64", "4096"], "TMC2_ORIENT_PAIRS": [None, None, "6", "10"],
         "TMC2_ORIENT_SPEC": [None, None, "64,16"]}
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(31000 + seed)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        xyz = degenerate_cloud(rng)
    elif kind == 1:
        xyz = synth_cloud(["small", "medium"][int(rng.integers(0, 2))], int(rng.integers(0, 4)))[0]
    elif kind == 2:                                   # dense random blob: many equal coordinates, unbalanced tree pieces
        n = int(rng.integers(9000, 140000)); xyz = np.unique(rng.integers(0, int(rng.integers(20, 200)), (n, 3)).astype(np.int16), axis=0)
        xyz = xyz[rng.permutation(len(xyz))]
    else:                                             # thin shell: long chains of INDIRECT-edge activations
        n = int(rng.integers(20000, 90000)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
        xyz = np.unique(np.round(512 + d * rng.uniform(100, 400)).astype(np.int16), axis=0); xyz = xyz[rng.permutation(len(xyz))]
    if len(xyz) < 64:
        continue
    env = {k: v[int(rng.integers(0, len(v)))] for k, v in HOOKS.items()}
    for k, v in env.items():                          # (options of the context: the environment is read once, when a context is created)
        ctx.set_option(k[len("TMC2_"):], v)
    why = []
    fr = ctx.frame(xyz)
    perm, depth = fr.kdtree_order()
    hperm, _, hdepth = T.host_kdtree_build(xyz)
    if not np.array_equal(perm, hperm) or depth != hdepth:
        why.append("kdtree")
    q = xyz[rng.integers(0, len(xyz), 3000)] + rng.integers(-2, 3, (3000, 3)).astype(np.int16)
    for k in (16, 8, 1):
        if len(xyz) >= k and not np.array_equal(fr.kdtree_search(q, k), oracle.knn(xyz, q, k)):
            why.append("knn%d" % k)
    if len(xyz) <= 60000:                             # the colour transfer's searches (round 6: in two launches; identical and duplicate points)
        rgb = rng.integers(0, 256, (len(xyz), 3)).astype(np.uint8)
        tgt = np.concatenate([xyz[rng.random(len(xyz)) < 0.7], q[:1500].clip(0, 1023).astype(np.int16), xyz[rng.integers(0, len(xyz), 200)]])
        dup = min(50, len(xyz))                       # duplicate positions with other colours in the source
        src, col = np.concatenate([xyz, xyz[:dup]]), np.concatenate([rgb, rng.integers(0, 256, (dup, 3)).astype(np.uint8)])
        if not np.array_equal(ctx.transfer_colors(src, col, tgt), oracle.transfer_colors(src, col, tgt)):
            why.append("transfer_colors")
    if len(xyz) <= 60000:
        nrm = oracle.normals(xyz)
        T.load_library().tmc2_set_refine_overlap(int(rng.integers(0, 2)))     # (few frames in flight: other grids, geometry ahead)
        fr.normals_compute(16, 1)                     # S2 + S3 on the device: contraction (every strong edge inside a cluster checked),
        if not np.array_equal(fr.get_normals().view(np.uint64), nrm.view(np.uint64)):   # threshold ladder, compact walk
            why.append("normals")
        p0 = oracle.initial_segmentation(nrm, oracle.weight_normal(xyz))
        vox = int(rng.choice([4, 2])); it = int(rng.integers(2, 9))
        fr.set_normals(nrm); fr.set_partition(p0)
        try:
            fr.segmenter_refine_grid_based(1024, 3.0, it, vox, 192)
        except T.Tmc2Error as e:                      # (a refusal -- e.g. a grid the dense voxel table does not take -- is not a mismatch)
            stats["refine_refused"] += 1
            print("refused", seed, kind, len(xyz), vox, env, str(e)[:160], flush=True)
            fr.close()
            continue
        if not np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=it, vox_dim=vox)):
            why.append("refine")
    fr.close()
    stats["ok" if not why else "MISMATCH"] += 1
    if why:
        print("MISMATCH", seed, kind, len(xyz), env, why, flush=True)
print(dict(stats))
