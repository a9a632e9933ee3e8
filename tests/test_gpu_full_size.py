"""BASELINE.json configurations at their real sizes and CTC settings: the HIP path against MD5 fixtures generated from the
unmodified reference (tests/golden/full_size.npz, tests/golden/make_golden.py full_size) -- every canvas, the reconstructed
cloud, its colours and pointToPixel, the patch lists, bit for bit.  (The same bytes the reference logs per picture / per cloud
under CONFORMANCE_TRACE: PCCVideoEncoder.cpp:389-396, PCCEncoder.cpp:620-626.)"""
import hashlib
import os

import numpy as np
import pytest

import tmc2_amd as T
from tmc2_amd.synth import synth_cloud

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "full_size.npz")

# one table for the fixture generator (tests/golden/make_golden.py), these tests and bench.py; the fixture's input_md5 pins the
# synthetic input.  The several-frame cases of configs 2-4 run with many frames in flight in tests/test_gpu_gof_soak.py.
from tmc2_amd.configs import FULL_SIZE_CASES as ALL_CASES, constrained_pack

CASES = {k: ALL_CASES[k] for k in ("longdress_vox10_ai_r3", "loot_vox10_ai_r3", "redandblack_vox10_ai_r3", "soldier_vox10_ai_r3",
                                   "basketball_player_vox11_ra_r5", "basketball_player_vox11_ra_r5_gof4", "longdress_vox10_ra_r3_gof3")}


def digest(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture(name):
    g = np.load(FIXTURE)
    return {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "/")}


def check_against_fixture(g, W, H, per_frame):
    """per_frame: [(patch list in list order, geometry images dict, (recon xyz, rgb, pointToPixel), attribute images)]"""
    assert (W, H) == tuple(int(x) for x in g["canvas"])
    for i, (patches, img, (rx, rc, p2p), att) in enumerate(per_frame):
        assert [len(patches), len(rx)] == g["f%d_counts" % i].tolist(), i
        flat = np.stack([patches[n] for n in patches.dtype.names if n not in ("depthOffset", "occOffset")], 1).astype(np.int32)
        assert digest(flat) == str(g["f%d_patches_md5" % i]), "patch list of frame %d" % i
        for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
            assert digest(img[k]) == str(g["f%d_%s_md5" % (i, k)]), "%s of frame %d" % (k, i)
        for k, v in (("recon_xyz", rx), ("recon_rgb", rc), ("point_to_pixel", p2p), ("attribute", att)):
            assert digest(v) == str(g["f%d_%s_md5" % (i, k)]), "%s of frame %d" % (k, i)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_full_size_matches_golden(name):
    c, g = CASES[name], fixture(name)
    if not g:
        pytest.fail("fixture of %s missing from tests/golden/full_size.npz" % name)
    frames = [synth_cloud(c["workload"], f) for f in range(c["frames"])]
    assert "".join(digest(x) + digest(col) for x, col in frames) == str(g["input_md5"]), "synthetic input differs from the fixture's"
    enc = T.GofEncoder(0, workers=min(2, len(frames)), iterations=c["iterations"], bits3d=c["bits3d"], occ_precision=c["precision"],
                       min_w=c["min_w"], min_h=c["min_h"], vox_dim=c["vox_dim"])
    try:
        frs = enc.upload(frames)
        W, H = enc.phase_a(frs, constrained_pack=constrained_pack(c))
        enc.phase_b(frs)
        per = []
        for fr in frs:
            patches = fr.get_patches()[0][fr.get_patch_order()]
            per.append((patches, fr.get_geometry_images(), fr.get_reconstruction(), fr.get_attribute_images()))
        check_against_fixture(g, W, H, per)
        if "f0_metrics" in g:
            # S23 at BASELINE size on the RESIDENT clouds (tmc2_metrics_compute_frame): the 48-bit-key de-duplication, the
            # 16-NN batches, normal copy / scaling and the ordered fp64 sums against PCCMetrics::compute of the unmodified
            # reference, raw doubles.  (The frame's normals are the reference's: checked first.)
            fr, res = frs[0], float((1 << (c["bits3d"] - 1)) - 1)
            assert digest(fr.get_normals()) == str(g["f0_normals_md5"]), "normals of frame 0"
            got, counts = enc.per_frame(frs[:1], lambda fr, i: fr.metrics_compute(0, True, res))[0]
            assert counts.tolist() == g["f0_metric_counts"].tolist()
            assert np.array_equal(got.view(np.uint64), g["f0_metrics"].view(np.uint64)), (got, g["f0_metrics"])
            got, _ = enc.per_frame(frs[:1], lambda fr, i: fr.metrics_compute(0, False, res))[0]
            assert np.array_equal(got.view(np.uint64), g["f0_metrics_no_normals"].view(np.uint64)), (got, g["f0_metrics_no_normals"])
    finally:
        enc.close()


def decoder_side_cut(fr):
    """What the bitstream carries of an encoder-side frame: the patch records a decoder parses (no depth pools, no 3-D boxes
    beyond u1 / v1 / d1), the occupancy video and the two geometry maps (identity video codec), the two I420 attribute frames."""
    patches = fr.get_patches()[0][fr.get_patch_order()]
    sent = np.zeros(len(patches), patches.dtype)
    for k in ("u0", "v0", "sizeU0", "sizeV0", "patchOrientation", "u1", "v1", "d1", "normalAxis", "tangentAxis", "bitangentAxis",
              "projectionMode"):
        sent[k] = patches[k]
    sent["sizeU"], sent["sizeV"] = sent["sizeU0"] * 16, sent["sizeV0"] * 16
    img = fr.get_geometry_images()
    return sent, img["occ_video"], np.stack([img["geo0"], img["geo1"]]), fr.encoder_attribute_to_yuv420(4)


def check_decoder_side(g, i, i420, dec444, post, metrics=None, metrics_no_normals=None, counts=None):
    """Frame i of a case against the reference's digests of the decoder-side chain (make_golden.py full_size_decoder_side)."""
    assert digest(i420) == str(g["f%d_i420_md5" % i]), "I420 attribute frames of frame %d" % i
    if dec444 is not None:
        assert digest(dec444) == str(g["f%d_dec444_md5" % i]), "decoded 4:4:4 attribute frames of frame %d" % i
    for k in ("xyz", "colors16", "rgb", "boundary"):
        assert digest(post[k]) == str(g["f%d_post_%s_md5" % (i, k)]), "post-reconstruction %s of frame %d" % (k, i)
    assert int((post["boundary"] == 3).sum()) == int(g["f%d_post_moved" % i]) > 0     # (the geometry smoothing moved points)
    for got, key in ((metrics, "f%d_post_metrics"), (metrics_no_normals, "f%d_post_metrics_no_normals")):
        if got is not None:
            exp = g[key % i]
            assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), (key % i, got, exp)
    if counts is not None:
        assert np.asarray(counts).tolist() == g["f%d_post_metric_counts" % i].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["longdress_vox10_ai_r3", "loot_vox10_ai_r3", "basketball_player_vox11_ra_r5"])
def test_gpu_full_size_decoder_side(gpu_ctx, name):
    """Config 5 at BASELINE size against the UNMODIFIED REFERENCE: a decoder-side frame (decoded patch records, occupancy video,
    geometry maps: no source cloud) -> generatePointCloud -> decoded attribute frames (I420 -> 16-bit 4:4:4,
    PCCInternalColorConverter.cpp:355-482) -> post-reconstruction tail (smoothPointCloudPostprocess PCCCodec.cpp:54,1067-1106,
    transferColors16bitBP PCCPointSet.cpp:1126) -> D1 / D2 / colour metric against the uncompressed frame (PCCMetrics.cpp:324-375).
    Every intermediate is compared with the reference's digest / doubles (voxels of 4 and of 2; 10- and 11-bit geometry,
    occupancyPrecision 4 and 2); the encoder-side frame it was cut from must give the same bytes."""
    c, g = CASES[name], fixture(name)
    if "f0_post_xyz_md5" not in g:
        pytest.fail("decoder-side digests of %s missing from tests/golden/full_size.npz" % name)
    xyz, rgb = synth_cloud(c["workload"], 0)
    gof = T.GofEncoder(0, workers=1, iterations=c["iterations"], bits3d=c["bits3d"], occ_precision=c["precision"], min_w=c["min_w"],
                       min_h=c["min_h"], vox_dim=c["vox_dim"])
    gpu_ctx = gof.ctxs[0]                                            # (the packing condition of the case: through the GOF encoder)
    enc = gof.upload([(xyz, rgb)])[0]
    W, H = gof.phase_a([enc], constrained_pack=constrained_pack(c))
    assert (W, H) == tuple(int(x) for x in g["canvas"])
    gof.phase_b([enc])
    sent, occ_video, geometry, i420 = decoder_side_cut(enc)
    dec = gpu_ctx.decoder_frame(sent, W, H, c["precision"], occ_video, geometry)
    dec.codec_generate_point_cloud()
    rx, _, rp = dec.get_reconstruction(colors=False)
    assert digest(rx) == str(g["f0_recon_xyz_md5"]) and digest(rp) == str(g["f0_point_to_pixel_md5"])
    nrm, res = enc.get_normals(), float((1 << (c["bits3d"] - 1)) - 1)
    assert digest(nrm) == str(g["f0_src_normals_md5"])
    for fr in (dec, enc):
        fr.codec_set_decoded_attribute_yuv420(i420, 0)
        fr.codec_post_reconstruct(None)
        post = fr.get_post_reconstruction()
        if fr is dec:
            m1, counts = fr.metrics_compute_source(xyz, rgb, nrm, 1, res)
            m0, _ = fr.metrics_compute_source(xyz, rgb, None, 1, res)
        else:
            m1, counts = fr.metrics_compute(1, True, res)
            m0, _ = fr.metrics_compute(1, False, res)
        check_decoder_side(g, 0, i420, fr.get_decoded_attribute(), post, m1, m0, counts)
    via_host, _ = gpu_ctx.metrics_compute(xyz, rgb, post["xyz"], post["rgb"], nrm, res)
    assert np.array_equal(via_host.view(np.uint64), g["f0_post_metrics"].view(np.uint64))
    dec.close(), enc.close()
    gof.close()


def test_oracle_full_size_matches_golden(oracle):
    """CPU tier: the restatement against the same fixture at BASELINE size (one case: the suite stays within minutes)."""
    name = "loot_vox10_ai_r3"
    c, g = CASES[name], fixture(name)
    if not g:
        pytest.skip("fixture of %s not generated yet" % name)
    frames = [synth_cloud(c["workload"], f) for f in range(c["frames"])]
    assert "".join(digest(x) + digest(col) for x, col in frames) == str(g["input_md5"])
    a = oracle.phase_a(frames, c["iterations"], c["bits3d"], c["precision"], c["min_w"], c["min_h"], c["pack"], c["vox_dim"])
    b = oracle.phase_b(frames, a, c["precision"])
    per = [(pa["patches"], pa, (pb["recon_xyz"], pb["recon_rgb"], pb["point_to_pixel"]), pb["attribute"]) for pa, pb in zip(a, b)]
    check_against_fixture(g, a[0]["width"], a[0]["height"], per)
