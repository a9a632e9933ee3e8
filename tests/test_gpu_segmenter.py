"""GPU tier (-m gpu): the HIP path, called through the C-ABI, against the oracle and the golden vectors."""
import hashlib
import os

import numpy as np
import pytest

import tmc2_amd as T
from tmc2_amd.synth import synth_cloud

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def digest(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_gpu_matches_golden(gpu_ctx, name):
    g = np.load(os.path.join(GOLD, "segmenter_%s.npz" % name))
    xyz, rgb = synth_cloud(name)
    fr = gpu_ctx.frame(xyz, rgb)

    def check(key, value):
        if key in g.files:
            assert np.array_equal(g[key], value), key
        else:
            assert str(g[key + "_md5"]) == digest(value), key

    fr.normals_compute_normals(16)
    check("knn16", fr.get_adjacency(16))
    check("normals_raw", fr.get_normals())
    fr.normals_orient()
    check("normals_oriented", fr.get_normals())
    q = (xyz[::5] + np.array([3, -2, 5], np.int16)).astype(np.int16)
    check("knn8_offcloud", fr.kdtree_search(q, 8))
    check("knn1_offcloud", fr.kdtree_search(q, 1))
    w = fr.weight_normal(11, 0.6)
    assert np.array_equal(bits(w), bits(g["weight_normal"]))
    fr.segmenter_initial_segmentation(w)
    assert np.array_equal(fr.get_partition(), g["partition_initial"])


@pytest.mark.parametrize("name,frame", [("small", 2), ("medium", 0)])
def test_gpu_matches_oracle(gpu_ctx, oracle, name, frame):
    xyz, rgb = synth_cloud(name, frame)
    fr = gpu_ctx.frame(xyz, rgb)
    fr.normals_compute(16, 1)
    o_adj = oracle.knn_self(xyz, 16)
    assert np.array_equal(fr.get_adjacency(16), o_adj)
    o_n = oracle.orient_normals(xyz, o_adj, oracle.compute_normals(xyz, o_adj))
    assert np.array_equal(bits(fr.get_normals()), bits(o_n))
    w = fr.weight_normal(11, 0.6)
    assert np.array_equal(bits(w), bits(oracle.weight_normal(xyz, 11, 0.6)))
    fr.segmenter_initial_segmentation(w)
    assert np.array_equal(fr.get_partition(), oracle.initial_segmentation(o_n, w))
    # distances come back exact
    idx, d = fr.kdtree_search(xyz[:1000], 16, with_dist=True)
    oi, od = oracle.knn(xyz, xyz[:1000], 16, with_dist=True)
    assert np.array_equal(idx, oi) and np.array_equal(d.astype(np.float64), od)


def test_gpu_knn_edge_cases(gpu_ctx, oracle):
    xyz, _ = synth_cloud("tiny")
    small = xyz[:16].copy()                      # k == n, single-leaf..two-leaf tree
    fr = gpu_ctx.frame(small)
    assert np.array_equal(fr.kdtree_search(small, 16), oracle.knn(small, small, 16))
    fr = gpu_ctx.frame(xyz)
    far = np.array([[0, 0, 0], [1023, 1023, 1023], [500, -20, 2000]], np.int16)  # outside the root box
    assert np.array_equal(fr.kdtree_search(far, 16), oracle.knn(xyz, far, 16))
    wide = np.array([[-6000, 10, 10], [20000, 5, 5], [512, 512, -32768]], np.int16)  # beyond the packed-offset range:
    assert np.array_equal(fr.kdtree_search(wide, 16), oracle.knn(xyz, wide, 16))    # takes the scratch-stack kernel
    with pytest.raises(T.Tmc2Error):
        gpu_ctx.frame(xyz[:5]).kdtree_search(xyz[:5], 16)                         # k > n is an error, not UB


@pytest.mark.parametrize("case", ["tiny", "small", "medium", "longdress_vox10", "line", "plane", "duplicates", "eleven",
                                  "root3000", "root8192", "n8193", "dense20000", "root30000", "n40000", "dense60000",
                                  "huge131072:root100000", "huge131072:dense120000", "huge8192:root30000"])
def test_gpu_kdtree_order_matches_reference_build(gpu_ctx, oracle, ctx_options, case):
    """The level-parallel device build must leave exactly the permutation of nanoflann's recursive build (the oracle's
    restatement for the small clouds, the library's host builder -- itself pinned to the oracle on CPU -- for all)."""
    rng = np.random.default_rng(5)
    if case.startswith("huge"):                      # TMC2_KD_HUGEMAX: the largest segment one workgroup splits in global memory
        hook, case = case.split(":")                 # (default 32 768; 8 192 = that tier is off; 131 072 = its limit)
        ctx_options.setenv("TMC2_KD_HUGEMAX", hook[4:])
    if case == "line":                               # every split degenerates to one dimension, long equal runs
        xyz = np.zeros((5000, 3), np.int16); xyz[:, 1] = rng.integers(0, 40, 5000)
    elif case == "plane":
        xyz = rng.integers(0, 64, (20000, 3)).astype(np.int16); xyz[:, 2] = 7
    elif case == "duplicates":                       # count/2 fallback: everything equal to the cut
        xyz = np.repeat(rng.integers(0, 1024, (40, 3)).astype(np.int16), 300, axis=0)
    elif case == "eleven":                           # smallest cloud that splits
        xyz = rng.integers(0, 1024, (11, 3)).astype(np.int16)
    elif case in ("root3000", "root8192"):           # the whole tree is one segment for the in-LDS splitting kernel
        xyz = rng.integers(0, 1024, (int(case[4:]), 3)).astype(np.int16)
    elif case == "n8193":                            # one level pass, then two such segments
        xyz = rng.integers(0, 512, (8193, 3)).astype(np.int16)
    elif case.startswith("dense"):                   # many equal coordinates: unbalanced pieces, tiny children next to big ones
        xyz = rng.integers(0, 12 if case == "dense20000" else 24, (int(case[5:]), 3)).astype(np.int16)
    elif case in ("root30000", "root100000"):        # the whole tree is one segment for the workgroup-per-segment kernel
        xyz = rng.integers(0, 1024, (int(case[4:]), 3)).astype(np.int16)
    elif case == "n40000":                           # one level pass, then two such segments
        xyz = rng.integers(0, 1024, (40000, 3)).astype(np.int16)
    else:
        xyz, _ = synth_cloud(case)
    fr = gpu_ctx.frame(xyz)
    perm, depth = fr.kdtree_order()
    hperm, _, hdepth = T.host_kdtree_build(xyz)
    assert np.array_equal(perm, hperm) and depth == hdepth
    if len(xyz) <= 300000:
        assert np.array_equal(perm, oracle.kdtree_perm(xyz)[0])
    q = xyz[rng.integers(0, len(xyz), 2000)]
    k = 16 if len(xyz) >= 16 else 8
    if len(xyz) <= 300000:
        assert np.array_equal(fr.kdtree_search(q, k), oracle.knn(xyz, q, k))


@pytest.mark.parametrize("option,value,case", [("KD_DECIDE", "global", "n40000"), ("KD_DECIDE", "global", "dense120000"),
                                               ("KD_PIECE_PER", "8", "root30000"), ("KD_HUGEMAX", "4096", "dense120000"),
                                               ("KD_HUGEMAX", "65536", "dense120000"), ("KD_HUGEMAX", "4096", "n40000")])
def test_gpu_kdtree_cross_check_forms(gpu_ctx, ctx_options, option, value, case):
    """The forms of the device builder that are not the default -- the decide pass folding through global memory (what a level of
    more than 2 000 segments takes), eight positions per thread in the piece kernel, no workgroup-per-segment tier / a larger one
    -- leave the host builder's permutation too.  (Round 4's tiers and level passes, options KD_FORM / KD_LEVELS of round 5, left
    the library in round 6: the host builder and the oracle are the cross-checks.)"""
    rng = np.random.default_rng(11)
    n = int("".join(ch for ch in case if ch.isdigit()))
    xyz = rng.integers(0, 24 if case.startswith("dense") else 1024, (n, 3)).astype(np.int16)
    ctx_options.setenv(option, value)
    fr = gpu_ctx.frame(xyz)
    perm, depth = fr.kdtree_order()
    hperm, _, hdepth = T.host_kdtree_build(xyz)
    assert np.array_equal(perm, hperm) and depth == hdepth


def test_gpu_full_size_properties(gpu_ctx):
    """BASELINE-size frame (~0.8 M points): size-independent properties instead of the (slow) oracle."""
    xyz, rgb = synth_cloud("longdress_vox10")
    fr = gpu_ctx.frame(xyz, rgb)
    fr.normals_compute(16, 1)
    adj = fr.get_adjacency(16)
    n = len(xyz)
    assert np.array_equal(adj[:, 0], np.arange(n, dtype=np.uint32))               # self first (distance 0, unique points)
    p = xyz.astype(np.int64)
    d = ((p[adj] - p[:, None, :]) ** 2).sum(-1)
    assert np.all(np.diff(d, axis=1) >= 0)                                        # sorted by distance
    assert np.all(np.sort(adj, 1)[:, 1:] != np.sort(adj, 1)[:, :-1])              # no duplicates
    # exactness of the k-th distance: no point outside the list is strictly closer (checked on a sample)
    rng = np.random.default_rng(0)
    for i in rng.integers(0, n, 64):
        di = ((p - p[i]) ** 2).sum(1)
        assert np.sort(di)[15] == d[i, 15]
    nrm = fr.get_normals()
    assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-9)
    assert (np.einsum("ij,ij->i", nrm, -p.astype(np.float64)) < 0).sum() <= (n + 1) // 2  # majority rule of S3


@pytest.mark.parametrize("name,iters", [("tiny", 10), ("small", 50), ("medium", 20), ("small", 1), ("tiny", 2)])
def test_gpu_refine_matches_oracle(gpu_ctx, oracle, ctx_options, name, iters):
    """The sweep loop of S5 (event-driven: incremental S, re-scoring only where S changed, the closure walked chip-wide without
    levels) against the oracle's in-order restatement.  (The sweep-everything loop of rounds 1-2, option REFINE_SWEEPS=full of
    rounds 3-5, left the library in round 6.)"""
    xyz, rgb = synth_cloud(name)
    nrm = oracle.normals(xyz)
    w = oracle.weight_normal(xyz)
    p0 = oracle.initial_segmentation(nrm, w)
    fr = gpu_ctx.frame(xyz, rgb)
    fr.set_normals(nrm)
    fr.set_partition(p0)
    fr.segmenter_refine_grid_based(1024, 3.0, iters, 4, 192)
    assert np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=iters))


@pytest.mark.parametrize("name,iters", [("tiny", 10), ("small", 10), ("medium", 10)])
def test_gpu_refine_voxels_of_two(gpu_ctx, oracle, name, iters):
    """voxelDimensionRefineSegmentation = 2 (cfg/sequence/{loot,redandblack,soldier}_vox10.cfg): search ball of 3 911 cells,
    rows of ~ 250 voxels, INDIRECT-edge candidates within Chebyshev distance 2 (up to 125, not a prefix of the row)."""
    xyz, rgb = synth_cloud(name)
    nrm = oracle.normals(xyz)
    p0 = oracle.initial_segmentation(nrm, oracle.weight_normal(xyz))
    fr = gpu_ctx.frame(xyz, rgb)
    fr.set_normals(nrm)
    fr.set_partition(p0)
    fr.segmenter_refine_grid_based(1024, 3.0, iters, 2, 192)
    exp = oracle.refine_grid(xyz, nrm, p0, iterations=iters, vox_dim=2)
    assert np.array_equal(fr.get_partition(), exp) and not np.array_equal(exp, p0)


@pytest.mark.parametrize("vox_dim", [4, 2])
def test_gpu_refine_row_capacity_retry(gpu_ctx, oracle, ctx_options, vox_dim):
    """The neighbourhood rows lie back to back in a table sized for twice the expected mean row; a frame that needs more
    repeats the pass with room for whole balls (forced here)."""
    ctx_options.setenv("TMC2_REFINE_ROWCAP", "tiny")
    xyz, rgb = synth_cloud("small")
    nrm = oracle.normals(xyz)
    p0 = oracle.initial_segmentation(nrm, oracle.weight_normal(xyz))
    fr = gpu_ctx.frame(xyz, rgb)
    fr.set_normals(nrm)
    fr.set_partition(p0)
    fr.segmenter_refine_grid_based(1024, 3.0, 10, vox_dim, 192)
    assert np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=10, vox_dim=vox_dim))


@pytest.mark.parametrize("stack,blocks", [("1", None), ("2", "3"), (None, "1")])
@pytest.mark.parametrize("vox_dim", [4, 2])
def test_gpu_refine_closure_spill_ring(gpu_ctx, oracle, ctx_options, vox_dim, stack, blocks):
    """The closure walks depth first with a ring in LDS per workgroup; what does not fit goes through a ring in global
    memory that the spilling group drains before it retires.  A ring with room for one or two voxels beyond the run sends nearly every activated voxel
    through the ring; a grid of 1 / 3 workgroups makes the runs of voxels long.  Every voxel must still be listed -- and
    processed -- exactly once."""
    if stack:
        ctx_options.setenv("TMC2_REFINE_RING", stack)
    if blocks:
        ctx_options.setenv("TMC2_REFINE_CLOSURE_BLOCKS", blocks)
    xyz, rgb = synth_cloud("medium")
    nrm = oracle.normals(xyz)
    p0 = oracle.initial_segmentation(nrm, oracle.weight_normal(xyz))
    fr = gpu_ctx.frame(xyz, rgb)
    fr.set_normals(nrm)
    fr.set_partition(p0)
    fr.segmenter_refine_grid_based(1024, 3.0, 6, vox_dim, 192)
    assert np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=6, vox_dim=vox_dim))


def test_gpu_refine_key_aliasing(gpu_ctx, oracle):
    """Coordinates at the top of the range make the reference's voxel key alias; identity is the key."""
    xyz, _ = synth_cloud("small")
    xyz = (xyz + (1023 - xyz.max(0))).astype(np.int16)
    nrm = oracle.normals(xyz)
    w = oracle.weight_normal(xyz)
    p0 = oracle.initial_segmentation(nrm, w)
    fr = gpu_ctx.frame(xyz)
    fr.set_normals(nrm)
    fr.set_partition(p0)
    fr.segmenter_refine_grid_based(1024, 3.0, 10, 4, 192)
    assert np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=10))


@pytest.mark.parametrize("vox_dim", [4, 2])
@pytest.mark.parametrize("form", ["cells", "rows-tier1", "rows-tier2", "rows-tier3", "rows-tier4"])
def test_gpu_refine_neighbourhood_forms(gpu_ctx, oracle, ctx_options, vox_dim, form):
    """S5's neighbourhood rows two ways -- row-wise through the occupancy bitmap with gathered reverse rows (round 4, default)
    and cell by cell with scattered reverse rows (rounds 1-3, TMC2_REFINE_NEIGHBOURHOOD=cells) -- and in every LDS tier of the
    row-wise kernels: same bits as the reference's refinement.  A cloud with coordinates at the top of the range, so that
    voxel keys alias in both."""
    if form == "cells":
        ctx_options.setenv("TMC2_REFINE_NEIGHBOURHOOD", "cells")
    else:
        ctx_options.setenv("TMC2_REFINE_CAPTIER", form[-1])
    for shift_to_top in (False, True):
        xyz, _ = synth_cloud("small")
        if shift_to_top:
            xyz = (xyz + (1023 - xyz.max(0))).astype(np.int16)
        nrm = oracle.normals(xyz)
        p0 = oracle.initial_segmentation(nrm, oracle.weight_normal(xyz))
        fr = gpu_ctx.frame(xyz)
        fr.set_normals(nrm)
        fr.set_partition(p0)
        fr.segmenter_refine_grid_based(1024, 3.0, 8, vox_dim, 192)
        assert np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=8, vox_dim=vox_dim)), (form, shift_to_top)


@pytest.mark.parametrize("vox_dim", [4, 2])
def test_gpu_refine_push_as_three_words(oracle, ctx_options, gpu_ctx, vox_dim):
    """The sweep's pushes add two words of a target's record in one signed 64-bit add (round 6); TMC2_REFINE_PUSH=words is the
    three 32-bit adds of rounds 2-5: same bits."""
    ctx_options.setenv("TMC2_REFINE_PUSH", "words")
    xyz, _ = synth_cloud("small", 2)
    nrm = oracle.normals(xyz)
    p0 = oracle.initial_segmentation(nrm, oracle.weight_normal(xyz))
    fr = gpu_ctx.frame(xyz)
    fr.set_normals(nrm)
    fr.set_partition(p0)
    fr.segmenter_refine_grid_based(1024, 3.0, 8, vox_dim, 192)
    assert np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=8, vox_dim=vox_dim))


@pytest.mark.parametrize("vox_dim", [4, 2])
@pytest.mark.parametrize("hits", ["0", "tiny"])
def test_gpu_refine_reverse_rows_without_the_kept_hits(oracle, ctx_options, gpu_ctx, vox_dim, hits):
    """Round 6: the forward pass over the balls keeps every ball's hits and the reverse rows pass reads them instead of collecting
    the balls again.  TMC2_REFINE_HITS=0 switches that off (rounds 4-5), =tiny leaves the kept hits no room: a region says so, the
    reverse rows pass collects the balls itself, and the context gives its next frames more room -- same bits either way."""
    ctx_options.setenv("TMC2_REFINE_HITS", hits)
    xyz, _ = synth_cloud("small")
    nrm = oracle.normals(xyz)
    p0 = oracle.initial_segmentation(nrm, oracle.weight_normal(xyz))
    exp = oracle.refine_grid(xyz, nrm, p0, iterations=8, vox_dim=vox_dim)
    for attempt in range(2):
        fr = gpu_ctx.frame(xyz)
        fr.set_normals(nrm)
        fr.set_partition(p0)
        gpu_ctx.stage_reset()
        fr.segmenter_refine_grid_based(1024, 3.0, 8, vox_dim, 192)
        assert np.array_equal(fr.get_partition(), exp), (hits, attempt)
        assert gpu_ctx.stage_calls().get("refine_hits_out_of_room", 0) == (1 if hits == "tiny" else 0), (hits, attempt)


@pytest.mark.parametrize("vox_dim", [4, 2])
def test_gpu_refine_solid_cloud_takes_the_larger_tier(oracle, vox_dim):
    """A SOLID block fills its balls (1 357 / 3 911 occupied cells a voxel): more than the smallest instantiation of the row-wise
    neighbourhood kernels holds in LDS -- the overflow says how much room the fullest ball asks for, the frame is repeated ONCE in
    the tier that holds it (1 536 keys for voxels of 4, 4 096 for voxels of 2), the context remembers, and the result is the
    reference's."""
    ctx = T.Context(0)                                                # (its own context: the tier is remembered per context)
    side = 60 if vox_dim == 4 else 40
    g = np.arange(side, dtype=np.int16)
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + np.int16(200)
    rng = np.random.default_rng(5)
    nrm = rng.normal(size=(len(xyz), 3))
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    p0 = oracle.initial_segmentation(nrm, np.ones(3))
    exp = oracle.refine_grid(xyz, nrm, p0, iterations=3, vox_dim=vox_dim)
    for attempt in range(2):
        fr = ctx.frame(xyz)
        fr.set_normals(nrm)
        fr.set_partition(p0)
        ctx.stage_reset()
        fr.segmenter_refine_grid_based(1024, 3.0, 3, vox_dim, 192)
        assert np.array_equal(fr.get_partition(), exp), attempt
        repeats = ctx.stage_calls().get("refine_cap_tier_repeat", 0)
        assert repeats == (1 if attempt == 0 else 0), (attempt, repeats)
        fr.close()
    ctx.close()


def _to_oracle_params(p):
    import oracle_binding as ob
    return ob.seg_params(p.iterationCountRefineSegmentation, p.geometryBitDepth3D, list(p.weightNormal))


def _assert_patches_equal(got, exp):
    gp, g0, g1, go = got
    assert len(gp) == len(exp["patches"])
    for name in exp["patches"].dtype.names:
        assert np.array_equal(gp[name], exp["patches"][name]), name
    assert np.array_equal(g0, exp["depth0"]) and np.array_equal(g1, exp["depth1"])
    assert np.array_equal(go, exp["occupancy"])


@pytest.mark.parametrize("name,iters,order", [("tiny", 10, None), ("small", 10, None), ("medium", 20, None), ("medium", 20, "input"), ("medium", 20, "tree"), ("medium", 20, "chunk")])
def test_gpu_segment_patches_matches_oracle(gpu_ctx, oracle, ctx_options, name, iters, order):
    """S7-S9 alone: oracle-made adjacency / partition in, patch records + depth maps + occupancy out.  order: how the mutual
    mask and the union / relaxation passes of S7 walk the points (option MUTUAL_ORDER: blocks as they come, XCD eighths, tree order)."""
    if order:
        ctx_options.setenv("TMC2_MUTUAL_ORDER", order)
    xyz, rgb = synth_cloud(name)
    knn = oracle.knn_self(xyz, 16)
    nrm = oracle.orient_normals(xyz, knn, oracle.compute_normals(xyz, knn))
    w = oracle.weight_normal(xyz)
    part = oracle.refine_grid(xyz, nrm, oracle.initial_segmentation(nrm, w), iterations=iters)
    fr = gpu_ctx.frame(xyz, rgb)
    fr.normals_compute_normals(16)     # resident adjacency (checked elsewhere against the oracle)
    fr.set_partition(part)
    p = T.ctc_params(iters, 11, w)
    fr.segmenter_segment_patches(p)
    _assert_patches_equal(fr.get_patches(), oracle.segment_patches(xyz, rgb, knn, part, _to_oracle_params(p)))


@pytest.mark.parametrize("rowcap", [None, "tiny"])
def test_gpu_segmenter_compute_with_refine_geometry_ahead(gpu_ctx, oracle, ctx_options, rowcap):
    """TMC2_REFINE_OVERLAP=1: the refine step's point-only half (voxels, neighbourhood rows) is queued before the orientation's
    host walk and picked up afterwards -- also when the neighbourhood pass has to be repeated with more room."""
    ctx_options.setenv("TMC2_REFINE_OVERLAP", "1")
    if rowcap:
        ctx_options.setenv("TMC2_REFINE_ROWCAP", rowcap)
    xyz, rgb = synth_cloud("small", 2)
    fr = gpu_ctx.frame(xyz, rgb)
    p = T.ctc_params(10, 11, fr.weight_normal(11, 0.6))
    fr.segmenter_compute(p)
    exp = oracle.segment(xyz, rgb, _to_oracle_params(p))
    _assert_patches_equal(fr.get_patches(), exp)
    fr.segmenter_compute(p)                                   # and again on the same frame (nothing stale is picked up)
    _assert_patches_equal(fr.get_patches(), exp)


@pytest.mark.parametrize("name,frame,iters", [("small", 1, 50), ("medium", 2, 10), ("small_noisy", 0, 10)])
def test_gpu_segmenter_compute_matches_oracle(gpu_ctx, oracle, name, frame, iters):
    """PCCPatchSegmenter3::compute end to end (S1..S9) through the C-ABI."""
    xyz, rgb = synth_cloud(name, frame)
    fr = gpu_ctx.frame(xyz, rgb)
    w = fr.weight_normal(11, 0.6)
    p = T.ctc_params(iters, 11, w)
    fr.segmenter_compute(p)
    _assert_patches_equal(fr.get_patches(), oracle.segment(xyz, rgb, _to_oracle_params(p)))


def test_gpu_segmenter_full_size_properties(gpu_ctx):
    """longdress-size frame: structural invariants of the patch set (no oracle at this size)."""
    xyz, rgb = synth_cloud("longdress_vox10")
    fr = gpu_ctx.frame(xyz, rgb)
    w = fr.weight_normal(11, 0.6)
    p = T.ctc_params(50, 11, w)
    fr.segmenter_compute(p)
    patches, d0, d1, occ = fr.get_patches()
    assert 20 < len(patches) < 2000
    assert np.array_equal(patches["index"], np.arange(len(patches)))
    total = 0
    for q in patches:
        a = d0[q["depthOffset"]:q["depthOffset"] + q["sizeU"] * q["sizeV"]].reshape(q["sizeV"], q["sizeU"])
        b = d1[q["depthOffset"]:q["depthOffset"] + q["sizeU"] * q["sizeV"]].reshape(q["sizeV"], q["sizeU"])
        valid = a < 32767
        assert valid.sum() == q["d0Count"] and valid.any()
        assert np.all((b[valid] - a[valid] >= 0) & (b[valid] - a[valid] <= 4))          # surfaceThickness
        assert a[valid].min() >= 0 and b[valid].max() <= 255 and q["sizeD"] in (0, 63, 127, 191)
        assert q["d1"] % 64 == 0 and q["sizeU0"] == (q["sizeU"] - 1) // 16 + 1
        o = occ[q["occOffset"]:q["occOffset"] + q["sizeU0"] * q["sizeV0"]].reshape(q["sizeV0"], q["sizeU0"])
        blk = np.add.reduceat(np.add.reduceat(valid.astype(np.int32), np.arange(0, q["sizeV"], 16), 0),
                              np.arange(0, q["sizeU"], 16), 1) > 0
        assert np.array_equal(o.astype(bool), blk)
        total += int(valid.sum())
    assert total > 0.6 * len(xyz)          # D0 pixels alone cover most of the cloud (the rest are D1 / in-between points)


def test_gpu_refine_many_sweeps(gpu_ctx, oracle):
    """50 sweeps (the longdress setting): the INDIRECT-edge closure chains get their full depth and the per-sweep
    voxel-state update rides in the following sweep -- the partition must still match the in-order reference loop."""
    xyz, rgb = synth_cloud("small")
    nrm = oracle.normals(xyz)
    w = oracle.weight_normal(xyz)
    p0 = oracle.initial_segmentation(nrm, w)
    for iters in (1, 2, 3, 50):
        fr = gpu_ctx.frame(xyz, rgb)
        fr.set_normals(nrm)
        fr.set_partition(p0)
        fr.segmenter_refine_grid_based(1024, 3.0, iters, 4, 192)
        assert np.array_equal(fr.get_partition(), oracle.refine_grid(xyz, nrm, p0, iterations=iters)), iters


@pytest.mark.parametrize("name", ["small", "medium"])
def test_gpu_orientation_contracted_and_point_level(gpu_ctx, oracle, ctx_options, name):
    """S3 both ways: clusters contracted on the device + cluster walk on the host (default), and the point-level walk
    (what a frame falls back to when a cluster's strong edges disagree) -- same bits as the reference's growth."""
    xyz, rgb = synth_cloud(name)
    exp = oracle.normals(xyz)
    fr = gpu_ctx.frame(xyz, rgb)
    gpu_ctx.stage_reset()
    fr.normals_compute(16, 1)
    assert np.array_equal(fr.get_normals().view(np.uint64), exp.view(np.uint64))
    assert gpu_ctx.stage_calls().get("orient_contract", 0) == 1 and gpu_ctx.stage_calls().get("orient_normals_regrowth", 0) == 0
    # the pair table too small for the frame: the walk gets every cross edge instead of one per pair of clusters
    ctx_options.setenv("TMC2_ORIENT_PAIRS", "6")
    fr3 = gpu_ctx.frame(xyz, rgb)
    gpu_ctx.stage_reset()
    fr3.normals_compute(16, 1)
    assert np.array_equal(fr3.get_normals().view(np.uint64), exp.view(np.uint64))
    assert gpu_ctx.stage_calls().get("orient_pair_table_overflow", 0) == 1
    ctx_options.delenv("TMC2_ORIENT_PAIRS")
    # the speculative room of the first attempt too small (what a vox11-size or noisy frame meets: > 64 K clusters or > 384 K
    # kept edges): the scatter is repeated with exact sizes on the SAME selection -- same compact graph (the strong one-way
    # edges included: the first of every implied sign, whatever the scheduling), same bits
    edges = []
    for spec, repeats in (("1000000,1000000", 0), ("64,16", 1), ("1000000,16", 1), ("64,1000000", 1)):
        ctx_options.setenv("TMC2_ORIENT_SPEC", spec)
        fr4 = gpu_ctx.frame(xyz, rgb)
        gpu_ctx.stage_reset()
        fr4.normals_compute(16, 1)
        assert np.array_equal(fr4.get_normals().view(np.uint64), exp.view(np.uint64)), spec
        assert gpu_ctx.stage_calls().get("orient_exact_size_repeat", 0) == repeats, spec
        assert gpu_ctx.stage_calls().get("orient_normals_regrowth", 0) == 0
        edges.append(gpu_ctx.stage_ms()["orient_compact_edges"])
    assert edges[0] > 64 and len(set(edges)) == 1, edges
    ctx_options.delenv("TMC2_ORIENT_SPEC")
    # the passes over the edges with the blocks as they come (rounds 2-5) and in tree order (since round 6 the default is input
    # order with XCD x on the x-th eighth of the blocks): the contraction is a fixed point of the graph, the order is only a
    # matter of speed -- same compact graph, same bits
    ctx_options.setenv("TMC2_ORIENT_SPEC", "1000000,1000000")
    for order in ("input", "tree"):
        ctx_options.setenv("TMC2_ORIENT_ORDER", order)
        fr5 = gpu_ctx.frame(xyz, rgb)
        gpu_ctx.stage_reset()
        fr5.normals_compute(16, 1)
        assert np.array_equal(fr5.get_normals().view(np.uint64), exp.view(np.uint64)), order
        assert gpu_ctx.stage_ms()["orient_compact_edges"] == edges[0], order
    ctx_options.delenv("TMC2_ORIENT_ORDER")
    ctx_options.delenv("TMC2_ORIENT_SPEC")
    ctx_options.setenv("TMC2_ORIENT_NO_CONTRACTION", "1")
    fr2 = gpu_ctx.frame(xyz, rgb)
    fr2.normals_compute(16, 1)
    assert np.array_equal(fr2.get_normals().view(np.uint64), exp.view(np.uint64))
