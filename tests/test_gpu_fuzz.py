"""GPU tier: the HIP path against the oracle on the degenerate clouds and random patch sets the CPU tier fuzzes the
oracle with (planes, lines, lattices, dust, duplicate-heavy blobs).  Part of the default `-m gpu` run."""
import numpy as np
import pytest

import tmc2_amd as T

pytestmark = [pytest.mark.gpu]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("seed", range(24))
def test_gpu_segmenter_on_degenerate_clouds(gpu_ctx, oracle, seed):
    import oracle_binding as ob
    from test_oracle_golden import degenerate_cloud
    rng = np.random.default_rng(9000 + seed)
    xyz = degenerate_cloud(rng)
    if len(xyz) < 64:
        pytest.skip("too few distinct points")
    rgb = rng.integers(0, 256, (len(xyz), 3), dtype=np.uint8)
    fr = gpu_ctx.frame(xyz, rgb)
    assert np.array_equal(fr.kdtree_order()[0], oracle.kdtree_perm(xyz)[0])
    fr.normals_compute(16, 1)
    assert np.array_equal(fr.get_adjacency(16), oracle.knn_self(xyz, 16))
    assert np.array_equal(bits(fr.get_normals()), bits(oracle.normals(xyz, 16, True)))
    w = fr.weight_normal(11, 0.6)
    assert np.array_equal(w, oracle.weight_normal(xyz, 11, 0.6))
    it = int(rng.integers(1, 6))
    fr.segmenter_compute(T.ctc_params(it, 11, w))
    seg = oracle.segment(xyz, rgb, ob.seg_params(it, 11, w))
    patches, d0, d1, occ = fr.get_patches()
    for n in seg["patches"].dtype.names:
        assert n in ("depthOffset", "occOffset") or np.array_equal(patches[n], seg["patches"][n]), n
    assert np.array_equal(d0, seg["depth0"]) and np.array_equal(d1, seg["depth1"]) and np.array_equal(occ, seg["occupancy"])


@pytest.mark.parametrize("seed", range(12))
def test_gpu_whole_path_on_degenerate_gofs(oracle, seed):
    from test_oracle_golden import degenerate_cloud
    rng = np.random.default_rng(13000 + seed)
    frames = []
    for _ in range(int(rng.integers(1, 4))):
        xyz = degenerate_cloud(rng)
        if len(xyz) < 64:
            break
        frames.append((xyz, rng.integers(0, 256, (len(xyz), 3), dtype=np.uint8)))
    if not frames:
        pytest.skip("too few distinct points")
    prec, it = int(rng.choice([4, 2, 1])), int(rng.integers(1, 5))
    mode = int(rng.choice([0, 1, 2])) if len(frames) > 1 else 0
    oa = oracle.phase_a(frames, it, 11, prec, constrained_pack=mode)
    ob_ = oracle.phase_b(frames, oa, prec)
    enc = T.GofEncoder(0, workers=2, iterations=it, occ_precision=prec)
    try:
        frs = enc.upload(frames)
        W, H = enc.phase_a(frs, constrained_pack={0: False, 1: True, 2: 2}[mode])
        enc.phase_b(frs)
        i420 = [np.zeros((2, W * H * 3 // 2), np.uint8) for _ in frs]
        enc.phase_c(frs, i420_out=i420)
        dec = [fr.get_decoded_attribute() for fr in frs]
        oc = oracle.phase_c(oa, ob_, dec, prec)
        for fr, a, b, c, d in zip(frs, oa, ob_, oc, dec):
            assert (W, H) == (a["width"], a["height"])
            img = fr.get_geometry_images()
            for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
                assert np.array_equal(img[k], a[k]), k
            assert np.array_equal(fr.get_attribute_images(), b["attribute"])
            exp_dec = np.stack([oracle.convert_yuv420_to_yuv444(*oracle.convert_rgb444_to_yuv420(b["attribute"][m])) for m in range(2)])
            assert np.array_equal(d, exp_dec)
            post = fr.get_post_reconstruction()
            for k in ("xyz", "colors16", "rgb", "boundary"):
                assert np.array_equal(post[k], c[k]), k
    finally:
        enc.close()


@pytest.mark.parametrize("seed", range(20))
def test_gpu_tail_on_random_clouds(gpu_ctx, oracle, seed):
    """The tail's kernels cannot be fed an arbitrary cloud through the C-ABI (they work on a frame's reconstruction); the
    colour conversion can: white noise and block images of odd sizes."""
    rng = np.random.default_rng(17000 + seed)
    H, W = 2 * int(rng.integers(8, 200)), 2 * int(rng.integers(8, 200))
    rgb = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    ey, eu, ev = oracle.convert_rgb444_to_yuv420(rgb)
    gy, gu, gv = gpu_ctx.color_convert_rgb444_to_yuv420(rgb)
    assert np.array_equal(gy, ey) and np.array_equal(gu, eu) and np.array_equal(gv, ev)
    y2, u2, v2 = (rng.integers(0, 256, a.shape, dtype=np.uint8) for a in (ey, eu, ev))
    assert np.array_equal(gpu_ctx.color_convert_yuv420_to_yuv444(y2, u2, v2), oracle.convert_yuv420_to_yuv444(y2, u2, v2))


@pytest.mark.parametrize("seed", range(40))
def test_gpu_color_conversion_on_random_sizes(gpu_ctx, oracle, seed):
    """C1 / C2 on images from 2x2 upwards (every border clamp of the 15/16-tap filters inside the picture), contents as in
    tools/fuzz/fuzz_color.py (which holds oracle == reference on them)."""
    rng = np.random.default_rng(7000 + seed)
    H, W = 2 * int(rng.integers(1, 80)), 2 * int(rng.integers(1, 80))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        rgb = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    elif kind == 1:
        rgb = rng.choice(np.array([0, 255], np.uint8), (3, H, W))
    else:
        rgb = np.clip(rng.normal(128, 3, (3, H, W)), 0, 255).astype(np.uint8)
    exp = oracle.convert_rgb444_to_yuv420(rgb)
    got = gpu_ctx.color_convert_rgb444_to_yuv420(rgb)
    assert all(np.array_equal(a, b) for a, b in zip(got, exp)), (H, W, kind)
    y, u, v = (rng.integers(0, 256, p.shape, dtype=np.uint8) for p in exp)
    assert np.array_equal(gpu_ctx.color_convert_yuv420_to_yuv444(y, u, v), oracle.convert_yuv420_to_yuv444(y, u, v)), (H, W)


@pytest.mark.parametrize("mode", [1, 2])
def test_gpu_packers_over_patch_records_equal_the_in_process_chain(oracle, mode):
    """The route a multi-rank GOF takes through S10' (patch records -> the rank that runs the chain -> packed lists installed
    with tmc2_frame_set_packing), forced in one process, against the in-process chain on the same frames: canvas, lists,
    matches, geometry / occupancy canvases, and (mode 2) the fixture of the unmodified reference."""
    from test_oracle_golden import RANDOM_ACCESS_CANVAS, _random_access_fixture, check_random_access_against_fixture
    g, frames = _random_access_fixture()
    enc = T.GofEncoder(0, workers=2, iterations=10, min_w=RANDOM_ACCESS_CANVAS[0], min_h=RANDOM_ACCESS_CANVAS[1])
    try:
        res = []
        for records_chain in (False, True):
            frs = enc.upload(frames)
            W, H = enc.phase_a(frs, constrained_pack=mode if mode == 2 else True, records_chain=records_chain)
            a = []
            for fr in frs:
                img = fr.get_geometry_images()
                patches = fr.get_patches()[0][fr.get_patch_order()]
                img.update(patches=patches, width=W, height=H, matches=fr.get_patch_matches())
                a.append(img)
            res.append(a)
        for x, y in zip(*res):
            assert (x["width"], x["height"]) == (y["width"], y["height"])
            for n in ("u0", "v0", "patchOrientation", "sizeU0", "sizeV0", "u1", "v1", "sizeU", "sizeV", "viewId"):
                assert np.array_equal(x["patches"][n], y["patches"][n]), n
            assert np.array_equal(x["matches"], y["matches"])
            for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
                assert np.array_equal(x[k], y[k]), k
        if mode == 2:
            check_random_access_against_fixture(g, res[1])
    finally:
        enc.close()
