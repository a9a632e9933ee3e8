"""GPU tier (-m gpu): packing + occupancy / geometry image generation (S10-S16) through the C-ABI."""
import os
import numpy as np
import pytest

import tmc2_amd as T
from tmc2_amd.synth import synth_cloud

pytestmark = pytest.mark.gpu


def gpu_phase_a(ctx, frames, iterations, occ_precision=4, min_w=1280, min_h=1280):
    """The call sequence a PCCEncoder::encode adaptor issues for S0-S16."""
    frs = [ctx.frame(xyz, rgb) for xyz, rgb in frames]
    w = frs[0].weight_normal(11, 0.6)                    # calculateWeightNormal on frame 0 only
    p = T.ctc_params(iterations, 11, w)
    heights = []
    for fr in frs:
        fr.segmenter_compute(p)
        heights.append(fr.encoder_pack_flexible(min_w, 2, 1.0))
    W, H = T.encoder_canvas_size(heights, min_w, min_w, min_h)
    out = []
    for fr in frs:
        fr.encoder_generate_geometry_images(W, H, occ_precision)
        img = fr.get_geometry_images()
        patches = fr.get_patches()[0][fr.get_patch_order()]
        img.update(patches=patches, width=W, height=H)
        out.append(img)
    return out


def assert_phase_a_equal(got, exp):
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert (g["width"], g["height"]) == (e["width"], e["height"])
        for name in e["patches"].dtype.names:
            if name not in ("depthOffset", "occOffset"):
                assert np.array_equal(g["patches"][name], e["patches"][name]), name
        for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
            assert np.array_equal(g[k], e[k]), k


@pytest.mark.parametrize("name,nframes,iters,prec", [("tiny", 2, 10, 4), ("small", 3, 10, 4), ("small", 2, 10, 2),
                                                    ("medium", 1, 20, 4)])
def test_gpu_phase_a_matches_oracle(gpu_ctx, oracle, name, nframes, iters, prec):
    frames = [synth_cloud(name, f) for f in range(nframes)]
    assert_phase_a_equal(gpu_phase_a(gpu_ctx, frames, iters, prec), oracle.phase_a(frames, iters, 11, prec))


def test_gpu_phase_a_full_size_properties(gpu_ctx):
    xyz, rgb = synth_cloud("longdress_vox10")
    img = gpu_phase_a(gpu_ctx, [(xyz, rgb)], 50)[0]
    W, H = img["width"], img["height"]
    assert W == 1280 and H >= 1280 and H % 64 == 0
    occ, ov = img["occupancy"].astype(bool), img["occ_video"].astype(bool)
    assert occ.sum() == img["patches"]["d0Count"].sum()                        # patches never overlap on valid pixels
    assert np.array_equal(ov, occ.reshape(H // 4, 4, W // 4, 4).any(axis=(1, 3)))    # OM video = 4x4 OR
    blocks = occ.reshape(H // 16, 16, W // 16, 16).any(axis=(1, 3))
    assert np.array_equal(img["block_to_patch"] > 0, blocks)                   # every lit block has exactly one owner
    assert img["block_to_patch"].max() <= len(img["patches"])
    g0, g1 = img["geo0"].astype(np.int32), img["geo1"].astype(np.int32)
    assert np.all((g1[occ] - g0[occ] >= 0) & (g1[occ] - g0[occ] <= 4)) and g0.max() <= 255 and g1.max() <= 255
    unocc_cell = ~np.repeat(np.repeat(ov, 4, 0), 4, 1)
    assert np.array_equal(g0[unocc_cell], g1[unocc_cell])                      # group dilation equalised D0/D1
    # idempotence: a second generation from the same packing gives the same canvases


def gpu_phase_b(ctx_frames):
    out = []
    for fr in ctx_frames:
        fr.encoder_generate_attribute_images()
        xyz, rgb, p2p = fr.get_reconstruction()
        out.append(dict(recon_xyz=xyz, recon_rgb=rgb, point_to_pixel=p2p, attribute=fr.get_attribute_images()))
    return out


@pytest.mark.parametrize("name,nframes,iters,prec", [("tiny", 2, 10, 4), ("small", 2, 10, 4), ("small", 1, 10, 2),
                                                    ("medium", 1, 10, 4)])
def test_gpu_phase_b_matches_oracle(gpu_ctx, oracle, name, nframes, iters, prec):
    frames = [synth_cloud(name, f) for f in range(nframes)]
    frs = [gpu_ctx.frame(xyz, rgb) for xyz, rgb in frames]
    w = frs[0].weight_normal(11, 0.6)
    p = T.ctc_params(iters, 11, w)
    heights = []
    for fr in frs:
        fr.segmenter_compute(p)
        heights.append(fr.encoder_pack_flexible(1280, 2, 1.0))
    W, H = T.encoder_canvas_size(heights, 1280, 1280, 1280)
    for fr in frs:
        fr.encoder_generate_geometry_images(W, H, prec)
    got = gpu_phase_b(frs)
    exp_a = oracle.phase_a(frames, iters, 11, prec)
    exp = oracle.phase_b(frames, exp_a, prec)
    for g, e in zip(got, exp):
        for k in ("recon_xyz", "point_to_pixel", "recon_rgb", "attribute"):
            assert np.array_equal(g[k], e[k]), k


def test_gpu_phase_b_decoded_geometry_roundtrip(gpu_ctx, oracle):
    """Lossy-codec stand-in: perturb the geometry planes on the host, upload them as 'decoded', and check the
    reconstruction + attribute images against the oracle run on the same perturbed planes."""
    xyz, rgb = synth_cloud("small")
    fr = gpu_ctx.frame(xyz, rgb)
    w = fr.weight_normal(11, 0.6)
    p = T.ctc_params(10, 11, w)
    fr.segmenter_compute(p)
    h = fr.encoder_pack_flexible(1280, 2, 1.0)
    W, H = T.encoder_canvas_size([h], 1280, 1280, 1280)
    fr.encoder_generate_geometry_images(W, H, 4)
    img = fr.get_geometry_images()
    rng = np.random.default_rng(3)
    noise = rng.integers(-2, 3, img["geo0"].shape)
    g0 = np.clip(img["geo0"].astype(np.int32) + noise, 0, 255).astype(np.uint16)
    g1 = np.maximum(g0, np.clip(img["geo1"].astype(np.int32) + noise, 0, 255).astype(np.uint16))
    fr.set_decoded_geometry(None, np.stack([g0, g1]))
    got = gpu_phase_b([fr])[0]
    ea = dict(img, geo0=g0, geo1=g1, width=W, height=H, patches=fr.get_patches()[0][fr.get_patch_order()])
    exp = oracle.phase_b([(xyz, rgb)], [ea], 4)[0]
    for k in ("recon_xyz", "point_to_pixel", "recon_rgb", "attribute"):
        assert np.array_equal(got[k], exp[k]), k


def test_gpu_gof_matches_golden_fixture(gpu_ctx):
    """Whole path S0-S23 against the fixture generated from the unmodified reference (no oracle involved)."""
    from test_oracle_golden import _gof_fixture, check_gof_against_fixture
    g, frames = _gof_fixture()
    frs = [gpu_ctx.frame(xyz, rgb) for xyz, rgb in frames]
    w = frs[0].weight_normal(11, 0.6)
    p = T.ctc_params(10, 11, w)
    heights = []
    for fr in frs:
        fr.segmenter_compute(p)
        heights.append(fr.encoder_pack_flexible(1280, 2, 1.0))
    W, H = T.encoder_canvas_size(heights, 1280, 1280, 1280)
    a = []
    for fr in frs:
        fr.encoder_generate_geometry_images(W, H, 4)
        img = fr.get_geometry_images()
        img.update(patches=fr.get_patches()[0][fr.get_patch_order()], width=W, height=H)
        a.append(img)
    b = gpu_phase_b(frs)
    normals = {id(fr._xyz): fr for fr in frs}
    check_gof_against_fixture(g, frames, a, b, gpu_ctx.metrics_compute,
                              lambda xyz: next(fr for fr in frs if fr._xyz is xyz or np.array_equal(fr._xyz, xyz)).get_normals())


def test_gpu_transfer_colors_long_candidate_lists(gpu_ctx, oracle):
    """std::sort's non-stable ordering of > 16 equal-distance candidates is reproduced on the device."""
    from test_oracle_golden import _sparse_target_case
    for seed in (0, 1):
        xyz, rgb, tgt = _sparse_target_case(seed)
        assert np.array_equal(gpu_ctx.transfer_colors(xyz, rgb, tgt), oracle.transfer_colors(xyz, rgb, tgt))


@pytest.mark.parametrize("split", [None, "0"])
def test_gpu_transfer_colors_identical_points_and_duplicates(gpu_ctx, oracle, ctx_options, split):
    """Round 6: both searches of the colour transfer run in two launches -- the queries that have an identical point in the tree
    first (bound 0), then the compacted rest (TMC2_KNN_SPLIT=0: one launch, as before).  The cases that decide whether the first
    pass is exact: targets that ARE source points, targets that are not, DUPLICATE positions with different colours in the source
    and in the target (which of several identical points the reference returns is the traversal's order), targets on split planes
    (coordinates equal to many others along an axis), a target cloud that is all duplicates of one point."""
    if split:
        ctx_options.setenv("TMC2_KNN_SPLIT", split)
    rng = np.random.default_rng(17)
    for case in range(4):
        xyz, rgb = synth_cloud("small", case)
        n = len(xyz)
        if case >= 1:      # duplicate positions in the source, different colours
            dup = rng.integers(0, n, n // 20)
            xyz = np.concatenate([xyz, xyz[dup]])
            rgb = np.concatenate([rgb, rng.integers(0, 256, (len(dup), 3)).astype(np.uint8)])
            order = rng.permutation(len(xyz))
            xyz, rgb = xyz[order], rgb[order]
        keep = rng.random(len(xyz)) < 0.8
        tgt = xyz[keep]                                                                  # identical points
        moved = (xyz[rng.integers(0, len(xyz), len(xyz) // 6)] + rng.integers(-2, 3, (len(xyz) // 6, 3))).astype(np.int16)
        tgt = np.concatenate([tgt, np.clip(moved, 0, 1023).astype(np.int16)])            # points that look around for real
        if case >= 2:      # duplicate positions in the target
            tgt = np.concatenate([tgt, tgt[rng.integers(0, len(tgt), len(tgt) // 10)]])
        if case == 3:      # a plane of equal coordinates: every descent meets divlow == divhigh == the query's coordinate
            plane = np.stack(np.meshgrid(np.arange(300, 340), np.arange(400, 440), indexing="ij"), -1).reshape(-1, 2)
            slab = np.concatenate([plane, np.full((len(plane), 1), 512)], 1).astype(np.int16)
            xyz = np.concatenate([xyz, slab, slab[::3]])
            rgb = np.concatenate([rgb, rng.integers(0, 256, (len(slab) + len(slab[::3]), 3)).astype(np.uint8)])
            tgt = np.concatenate([tgt, slab[::2], slab[::5]])
        tgt = tgt[rng.permutation(len(tgt))]
        assert np.array_equal(gpu_ctx.transfer_colors(xyz, rgb, tgt), oracle.transfer_colors(xyz, rgb, tgt)), (case, split)
    one = np.repeat(xyz[:1], 500, 0)
    assert np.array_equal(gpu_ctx.transfer_colors(xyz, rgb, one), oracle.transfer_colors(xyz, rgb, one)), "all targets one point"


@pytest.mark.parametrize("placement", ["device", "host", "adaptive"])
def test_gpu_gof_encoder_worker_threads(oracle, placement):
    """The GOF orchestration used by bench.py: one pinned worker thread + context per in-flight frame, both k-d tree
    placements.  Five frames over three workers, everything compared with the oracle."""
    frames = [synth_cloud("tiny", f) for f in range(5)]
    T.load_library().tmc2_set_kdtree_placement({"device": 0, "host": 1, "adaptive": 2}[placement])
    T.load_library().tmc2_set_host_parallelism(1 if placement == "adaptive" else 16)   # adaptive: one host slot, the rest spills to the device
    try:
        enc = T.GofEncoder(0, workers=3, iterations=10)
        frs = enc.upload(frames)
        W, H = enc.phase_a(frs)
        enc.phase_b(frs)
        exp_a = oracle.phase_a(frames, 10, 11, 4)
        exp_b = oracle.phase_b(frames, exp_a, 4)
        for fr, ea, eb in zip(frs, exp_a, exp_b):
            img = fr.get_geometry_images()
            assert (W, H) == (ea["width"], ea["height"])
            for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
                assert np.array_equal(img[k], ea[k]), k
            assert np.array_equal(fr.get_attribute_images(), eb["attribute"])
        if placement == "device":      # zero-copy byte views of the canvases (what the multi-GPU gather ships)
            img = frs[0].get_geometry_images()
            geo = T.GofEncoder.canvas_from_bytes(enc.device_tensor(frs[0], "geometry"), "geometry")
            assert np.array_equal(geo[0], img["geo0"]) and np.array_equal(geo[1], img["geo1"])
            b2p = T.GofEncoder.canvas_from_bytes(enc.device_tensor(frs[0], "block_to_patch"), "block_to_patch")
            assert np.array_equal(b2p, img["block_to_patch"])
            att = T.GofEncoder.canvas_from_bytes(enc.device_tensor(frs[0], "attribute"), "attribute")
            assert np.array_equal(att, frs[0].get_attribute_images())
        used = enc.stage_calls()
        host, dev = used.get("kdtree_build_host", 0), used.get("kdtree_build", 0)
        assert (host, dev) == (5, 0) if placement == "host" else (host, dev) == (0, 5) if placement == "device" else host + dev == 5
        enc.close()
    finally:
        T.load_library().tmc2_set_kdtree_placement(0)
        T.load_library().tmc2_set_host_parallelism(16)


@pytest.mark.parametrize("min_w,min_h", [(1280, 1280), (256, 64)])
def test_gpu_all_intra_single_rendezvous(oracle, min_w, min_h):
    """GofEncoder.encode_all_intra: every frame runs S1-S22 on its worker with the canvas its own packed height gives, the common
    size is settled at the end and frames that guessed short rasterise again.  On the 256-wide canvas the frames of this GOF pack
    to different heights (the redo path); 1280 x 1280 is the CTC case (no redo).  Same canvases as the oracle either way."""
    frames = [synth_cloud("tiny", f) for f in range(4)] + [synth_cloud("small", 0)]
    enc = T.GofEncoder(0, workers=3, iterations=4, min_w=min_w, min_h=min_h)
    try:
        frs = enc.upload(frames)
        seen = []
        W, H = enc.encode_all_intra(frs, finish=lambda fr, i, size: seen.append((i, tuple(size))))
        exp_a = oracle.phase_a(frames, 4, 11, 4, min_w, min_h)
        exp_b = oracle.phase_b(frames, exp_a, 4)
        assert (W, H) == (exp_a[0]["width"], exp_a[0]["height"])
        assert {i for i, s in seen if s == (W, H)} == set(range(len(frames)))     # every frame finished on the final canvas
        assert (len(seen) > len(frames)) == (min_w == 256)                        # ... on the narrow one some only at the second attempt
        for fr, ea, eb in zip(frs, exp_a, exp_b):
            img = fr.get_geometry_images()
            for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
                assert np.array_equal(img[k], ea[k]), k
            assert np.array_equal(fr.get_attribute_images(), eb["attribute"])
    finally:
        enc.close()


def test_gpu_vox11_full_path_properties(gpu_ctx):
    """basketball_player_vox11-size frame (3.0 M points, 11-bit geometry, BASELINE configs[1..]): the whole path S0-S22 with
    the CTC settings of that sequence (20 refine iterations, 12-bit 3-D range) -- size-dependent tables (2^33-bit voxel
    bitmap, 2^28-word key table, > 2^21 points: scratch-stack k-NN, deep trees) and size-independent invariants."""
    xyz, rgb = synth_cloud("basketball_player_vox11")
    fr = gpu_ctx.frame(xyz, rgb)
    w = fr.weight_normal(12, 0.6)
    fr.segmenter_compute(T.ctc_params(20, 12, w))
    perm, depth = fr.kdtree_order()
    assert np.array_equal(np.sort(perm), np.arange(len(xyz), dtype=np.uint32)) and 20 < depth < 64
    adj = fr.get_adjacency(16)
    assert np.array_equal(adj[:, 0], np.arange(len(xyz), dtype=np.uint32))
    part = fr.get_partition()
    assert part.max() <= 5 and np.bincount(part, minlength=6).min() > 0.02 * len(xyz)
    h = fr.encoder_pack_flexible(1280, 2, 1.0)
    W, H = T.encoder_canvas_size([h], 1280, 1280, 1280)
    fr.encoder_generate_geometry_images(W, H, 4)
    img = fr.get_geometry_images()
    occ = img["occupancy"].astype(bool)
    assert 0.3 * len(xyz) < occ.sum() <= len(xyz)
    assert np.all(img["geo1"][occ].astype(np.int32) - img["geo0"][occ] >= 0)
    fr.encoder_generate_attribute_images()
    rx, rc, p2p = fr.get_reconstruction()
    assert 0.9 * len(xyz) < len(rx) < 1.2 * len(xyz)
    # every reconstructed point sits within one voxel of a source point along its projection axis: spot check by hashing
    src = set(map(tuple, xyz[::7].tolist()))
    hit = sum(tuple(p) in src for p in rx[::50].tolist())
    assert hit > 0          # lossless D0 positions reappear exactly
    att = fr.get_attribute_images()
    assert att.shape == (2, 3, H, W) and att[0, :, occ].any()


def test_gpu_low_delay_packing_matches_golden_fixture():
    """S0-S16 under the low-delay packing (S10': frames after the first packed against their predecessor) through the GOF
    orchestration, against the fixture generated from the unmodified reference."""
    from test_oracle_golden import _low_delay_fixture, check_low_delay_against_fixture
    g, frames = _low_delay_fixture()
    enc = T.GofEncoder(0, workers=2, iterations=10)
    try:
        frs = enc.upload(frames)
        W, H = enc.phase_a(frs, constrained_pack=True)
        a = []
        for fr in frs:
            img = fr.get_geometry_images()
            img.update(patches=fr.get_patches()[0][fr.get_patch_order()], width=W, height=H, matches=fr.get_patch_matches())
            a.append(img)
        check_low_delay_against_fixture(g, a)
    finally:
        enc.close()


def test_gpu_random_access_packing_matches_golden_fixture(oracle):
    """S0-S16 under the random-access packing (S10': the per-frame chain followed by the global patch allocation over the
    GOF) through the GOF orchestration, against the fixture generated from the unmodified reference; then S17-S22 on
    those canvases against the oracle (tracked patches carry the enlarged block box of their union into the
    reconstruction and the attribute images)."""
    from test_oracle_golden import RANDOM_ACCESS_CANVAS, _random_access_fixture, check_random_access_against_fixture
    g, frames = _random_access_fixture()
    enc = T.GofEncoder(0, workers=2, iterations=10, min_w=RANDOM_ACCESS_CANVAS[0], min_h=RANDOM_ACCESS_CANVAS[1])
    try:
        frs = enc.upload(frames)
        W, H = enc.phase_a(frs, constrained_pack=2)
        a = []
        for fr in frs:
            img = fr.get_geometry_images()
            patches, _, _, occ = fr.get_patches()
            assert np.array_equal(fr.get_patch_order(), np.arange(len(patches)))      # the lists themselves were reordered
            assert len(occ) == int((patches["sizeU0"] * patches["sizeV0"]).sum())    # pools rebuilt for the new boxes
            img.update(patches=patches, width=W, height=H, matches=fr.get_patch_matches())
            a.append(img)
        check_random_access_against_fixture(g, a)
        enc.phase_b(frs)
        exp_b = oracle.phase_b(frames, a, 4)
        for fr, eb in zip(frs, exp_b):
            xyz, rgb, p2p = fr.get_reconstruction()
            assert np.array_equal(xyz, eb["recon_xyz"]) and np.array_equal(rgb, eb["recon_rgb"])
            assert np.array_equal(p2p, eb["point_to_pixel"])
            assert np.array_equal(fr.get_attribute_images(), eb["attribute"])
    finally:
        enc.close()


def test_gpu_post_reconstruction_matches_golden_fixture(oracle):
    """The post-reconstruction tail (boundary points, colorPointCloud, grid geometry smoothing, transferColors16bitBP,
    convertYUV16ToRGB8) through the GOF orchestration, against the fixture generated from the unmodified reference."""
    from test_oracle_golden import _post_fixture, check_post_reconstruction_against_fixture
    g, frames, _, _, dec = _post_fixture(oracle)
    enc = T.GofEncoder(0, workers=2, iterations=10)
    try:
        frs = enc.upload(frames)
        enc.phase_a(frs)
        enc.phase_b(frs)
        for fr, d in zip(frs, dec):   # the canvases the stand-in decoder started from are the product's own
            assert np.array_equal(T.synth_decoded_attribute(fr.get_attribute_images()), d)
        before = []
        for fr in frs:
            fr.codec_identify_boundary_points()
            before.append(fr.get_post_reconstruction(xyz=False, colors16=False, rgb=False)["boundary"])
        enc.phase_c(frs, dec)
        c = [dict(fr.get_post_reconstruction(), boundary_before=bb) for fr, bb in zip(frs, before)]
        check_post_reconstruction_against_fixture(g, c)
    finally:
        enc.close()


@pytest.mark.parametrize("name,prec", [("small", 4), ("small", 2), ("medium", 4)])
def test_gpu_post_reconstruction_matches_oracle(gpu_ctx, oracle, name, prec):
    xyz, rgb = synth_cloud(name, 1)
    fr = gpu_ctx.frame(xyz, rgb)
    fr.segmenter_compute(T.ctc_params(10, 11, fr.weight_normal(11, 0.6)))
    h = fr.encoder_pack_flexible(1280, 2, 1.0)
    W, H = T.encoder_canvas_size([h], 1280, 1280, 1280)
    fr.encoder_generate_geometry_images(W, H, prec)
    b = gpu_phase_b([fr])[0]
    img = fr.get_geometry_images()
    img.update(width=W, height=H)
    dec = T.synth_decoded_attribute(b["attribute"])
    exp = oracle.phase_c([img], [b], [dec], prec)[0]
    fr.codec_identify_boundary_points()
    assert np.array_equal(fr.get_post_reconstruction(xyz=False, colors16=False, rgb=False)["boundary"], exp["boundary_before"])
    fr.codec_post_reconstruct(dec)
    got = fr.get_post_reconstruction()
    assert (got["boundary"] == 3).sum() > 100
    for k in ("boundary", "xyz", "colors16", "rgb"):
        assert np.array_equal(got[k], exp[k]), k
    # idempotence: the tail starts from the reconstruction, not from its own output
    fr.codec_post_reconstruct(dec)
    again = fr.get_post_reconstruction()
    for k in got:
        assert np.array_equal(got[k], again[k]), k


def test_gpu_color_chain_matches_golden_fixture(gpu_ctx):
    """Attribute canvases -> I420 frames (what the video encoder reads) -> 16-bit 4:4:4 frames (what the reconstruction
    reads) -> the post-reconstruction tail, all on the device, against the fixture produced by the reference's own
    PCCInternalColorConverter and codec members (identity video codec in between)."""
    from test_oracle_golden import GOLD, check_color_chain_against_fixture, digest
    g = np.load(os.path.join(GOLD, "gof_tiny2_color.npz"))
    frames = [synth_cloud("tiny", f) for f in range(2)]
    assert str(g["input_md5"]) == "".join(digest(x) + digest(c) for x, c in frames)
    frs = [gpu_ctx.frame(xyz, rgb) for xyz, rgb in frames]
    p = T.ctc_params(10, 11, frs[0].weight_normal(11, 0.6))
    heights = []
    for fr in frs:
        fr.segmenter_compute(p)
        heights.append(fr.encoder_pack_flexible(1280, 2, 1.0))
    W, H = T.encoder_canvas_size(heights, 1280, 1280, 1280)
    for fr in frs:
        fr.encoder_generate_geometry_images(W, H, 4)
        fr.encoder_generate_attribute_images()
    a = W * H

    def split(frame420):
        return frame420[:a].reshape(H, W), frame420[a:a + a // 4].reshape(H // 2, W // 2), frame420[a + a // 4:].reshape(H // 2, W // 2)

    # (1) the context-level converters on host images
    def tail_from_host(dec):
        out = []
        for fr, d in zip(frs, dec):
            fr.codec_post_reconstruct(d)
            out.append(fr.get_post_reconstruction())
        return out
    check_color_chain_against_fixture(g, gpu_ctx.color_convert_rgb444_to_yuv420, gpu_ctx.color_convert_yuv420_to_yuv444,
                                      [fr.get_attribute_images() for fr in frs], tail_from_host)
    # (2) the frame-level path: nothing but the I420 frames crosses the PCIe bus
    for i, fr in enumerate(frs):
        i420 = fr.encoder_attribute_to_yuv420(4)
        for m in range(2):
            assert digest(i420[m]) == str(g["f%d_m%d_yuv420_md5" % (i, m)])
        fr.codec_set_decoded_attribute_yuv420(i420, 0)
        dec = fr.get_decoded_attribute()
        for m in range(2):
            assert digest(dec[m]) == str(g["f%d_m%d_yuv444_md5" % (i, m)])
        fr.codec_post_reconstruct(None)
        pc = fr.get_post_reconstruction()
        for k in ("xyz", "colors16", "rgb", "boundary"):
            assert digest(np.ascontiguousarray(pc[k])) == str(g["f%d_%s_md5" % (i, k)]), (i, k)


@pytest.mark.parametrize("kind,H,W", [("noise", 64, 96), ("noise", 66, 70), ("blocks", 256, 320), ("flat", 32, 48), ("noise", 1280, 1344)])
def test_gpu_color_conversion_matches_oracle(gpu_ctx, oracle, kind, H, W):
    rng = np.random.default_rng(H * 1000 + W)
    if kind == "noise":
        rgb = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    elif kind == "blocks":
        rgb = np.kron(rng.integers(0, 256, (3, H // 16, W // 16), dtype=np.uint8), np.ones((16, 16), np.uint8))
    else:
        rgb = np.full((3, H, W), 255, np.uint8)
    ey, eu, ev = oracle.convert_rgb444_to_yuv420(rgb)
    gy, gu, gv = gpu_ctx.color_convert_rgb444_to_yuv420(rgb)
    assert np.array_equal(gy, ey) and np.array_equal(gu, eu) and np.array_equal(gv, ev)
    y2, u2, v2 = (rng.integers(0, 256, a.shape, dtype=np.uint8) for a in (ey, eu, ev))
    assert np.array_equal(gpu_ctx.color_convert_yuv420_to_yuv444(y2, u2, v2), oracle.convert_yuv420_to_yuv444(y2, u2, v2))
    with pytest.raises(T.Tmc2Error):
        gpu_ctx.color_convert_rgb444_to_yuv420(rgb, downsampling_filter=2)     # only the reference's default filters are built


@pytest.mark.parametrize("name,prec", [("small", 4), ("small", 2)])
def test_gpu_decoder_side_reconstruction(gpu_ctx, oracle, name, prec):
    """The decoder's way to the finished cloud: a frame made of decoded patch records, occupancy video and geometry maps only
    (no source cloud) -> generatePointCloud -> colour conversion of the decoded I420 attribute frames -> the tail; against the
    oracle, and identical to what the encoder-side frame it was cut from produces."""
    xyz, rgb = synth_cloud(name, 2)
    enc = gpu_ctx.frame(xyz, rgb)
    enc.segmenter_compute(T.ctc_params(10, 11, enc.weight_normal(11, 0.6)))
    h = enc.encoder_pack_flexible(1280, 2, 1.0)
    W, H = T.encoder_canvas_size([h], 1280, 1280, 1280)
    enc.encoder_generate_geometry_images(W, H, prec)
    enc.encoder_generate_attribute_images()
    img = enc.get_geometry_images()
    i420 = enc.encoder_attribute_to_yuv420(4)
    patches = enc.get_patches()[0][enc.get_patch_order()]
    sent = np.zeros(len(patches), patches.dtype)                     # only what the bitstream carries
    for k in ("u0", "v0", "sizeU0", "sizeV0", "patchOrientation", "u1", "v1", "d1", "normalAxis", "tangentAxis", "bitangentAxis",
              "projectionMode"):
        sent[k] = patches[k]
    sent["sizeU"], sent["sizeV"] = sent["sizeU0"] * 16, sent["sizeV0"] * 16      # (the decoder knows block sizes only)
    dec = gpu_ctx.decoder_frame(sent, W, H, prec, img["occ_video"], np.stack([img["geo0"], img["geo1"]]))
    assert np.array_equal(dec.get_geometry_images()["block_to_patch"], img["block_to_patch"])
    dec.codec_generate_point_cloud()
    rx, _, rp = dec.get_reconstruction(colors=False)
    ex, _, ep = enc.get_reconstruction()
    assert np.array_equal(rx, ex) and np.array_equal(rp, ep)
    with pytest.raises(T.Tmc2Error):
        dec.get_reconstruction(colors=True)                          # no colour transfer happened on this side
    dec.codec_set_decoded_attribute_yuv420(i420, 0)
    dec.codec_post_reconstruct(None)
    got = dec.get_post_reconstruction()
    enc.codec_set_decoded_attribute_yuv420(i420, 0)
    enc.codec_post_reconstruct(None)
    same = enc.get_post_reconstruction()
    a = dict(img, width=W, height=H)
    b = dict(recon_xyz=ex, point_to_pixel=ep)
    exp = oracle.phase_c([a], [b], [dec.get_decoded_attribute()], prec)[0]
    for k in ("xyz", "colors16", "rgb", "boundary"):
        assert np.array_equal(got[k], exp[k]), k
        assert np.array_equal(got[k], same[k]), k
