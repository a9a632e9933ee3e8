"""Memory safety of the C-ABI towards its caller: with the binding's guard mode on (tmc2_amd.lib.set_guard / TMC2_GUARD=1) every
array handed to libtmc2hip.so travels in a copy with a 4 KiB red zone on both sides; a byte written outside a caller's buffer
trips the check after the call.  One frame and one random-access GOF go through every entry of the path, the decoder side, the
metric and the PLY reader under it.  (The whole GPU tier runs under the same mode with TMC2_GUARD=1 in the environment:
tools/gpu/guard.sh; round 4 did that after an unexplained `corrupted double-linked list` in a round-3 benchmark process.)"""
import os

import numpy as np
import pytest

import tmc2_amd as T
from tmc2_amd import lib
from tmc2_amd.synth import synth_cloud


@pytest.fixture
def guard():
    lib.set_guard(True)
    yield
    lib.set_guard(os.environ.get("TMC2_GUARD", "0") == "1")


@pytest.mark.gpu
def test_gpu_guard_trips_on_a_short_buffer(gpu_ctx, guard):
    """The check itself: a getter that is handed a buffer one row short must be caught (and is: the library trusts the sizes
    of the C-ABI contract)."""
    xyz, rgb = synth_cloud("tiny")
    fr = gpu_ctx.frame(xyz, rgb)
    fr.normals_compute(16, 1)
    short = np.zeros((len(xyz) - 1, 3), np.float64)
    with pytest.raises(T.Tmc2Error, match="wrote outside"):
        lib._check(fr.L.tmc2_frame_get_normals(fr.h, lib._ptr(short)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "small"])
def test_gpu_guard_whole_path(gpu_ctx, oracle, guard, name, tmp_path):
    xyz, rgb = synth_cloud(name)
    fr = gpu_ctx.frame(xyz, rgb)
    fr.normals_compute(16, 1)
    assert np.array_equal(fr.get_adjacency(16), oracle.knn_self(xyz, 16))
    fr.kdtree_search(xyz[:100], 8, with_dist=True)
    fr.kdtree_order()
    w = fr.weight_normal(11, 0.6)
    fr.segmenter_compute(T.ctc_params(10, 11, w))
    fr.get_partition(), fr.get_normals(), fr.get_patch_records()
    h = fr.encoder_pack_flexible(1280, 2, 1.0)
    W, H = T.encoder_canvas_size([h], 1280, 1280, 1280)
    fr.encoder_generate_geometry_images(W, H, 4)
    fr.encoder_generate_attribute_images()
    img, att = fr.get_geometry_images(), fr.get_attribute_images()
    o_a = oracle.phase_a([(xyz, rgb)], 10, 11, 4)
    for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"):
        assert np.array_equal(img[k], o_a[0][k]), k
    patches = fr.get_patches()[0][fr.get_patch_order()]
    fr.get_patch_matches(), fr.get_packed_size(), fr.get_reconstruction()
    i420 = fr.encoder_attribute_to_yuv420(4)
    fr.codec_set_decoded_attribute_yuv420(i420, 0)
    fr.get_decoded_attribute()
    fr.codec_post_reconstruct(None)
    post = fr.get_post_reconstruction()
    q, counts = fr.metrics_compute(0, True, 1023.0)
    q2, _ = fr.metrics_compute(1, False, 1023.0)
    gpu_ctx.metrics_compute(xyz, rgb, post["xyz"], post["rgb"], fr.get_normals(), 1023.0)
    # decoder side: a frame from what the bitstream carries
    sent = np.zeros(len(patches), patches.dtype)
    for k in ("u0", "v0", "sizeU0", "sizeV0", "patchOrientation", "u1", "v1", "d1", "normalAxis", "tangentAxis", "bitangentAxis",
              "projectionMode"):
        sent[k] = patches[k]
    sent["sizeU"], sent["sizeV"] = sent["sizeU0"] * 16, sent["sizeV0"] * 16
    dec = gpu_ctx.decoder_frame(sent, W, H, 4, img["occ_video"], np.stack([img["geo0"], img["geo1"]]))
    dec.codec_generate_point_cloud()
    dec.codec_set_decoded_attribute_yuv420(i420, 0)
    dec.codec_post_reconstruct(None)
    d = dec.get_post_reconstruction()
    for k in ("xyz", "colors16", "rgb", "boundary"):
        assert np.array_equal(d[k], post[k]), k
    dec.metrics_compute_source(xyz, rgb, fr.get_normals(), 1, 1023.0)
    # colour conversion and the PLY reader / writer on host buffers
    y, u, v = gpu_ctx.color_convert_rgb444_to_yuv420(att[0], 4)
    gpu_ctx.color_convert_yuv420_to_yuv444(y, u, v, 0)
    gpu_ctx.transfer_colors(xyz, rgb, post["xyz"])
    for ascii_ in (True, False):
        path = str(tmp_path / ("c%d.ply" % ascii_))
        T.ply_write(path, xyz, rgb, None, ascii=ascii_)
        for threads in (1, 4):
            gx, gc, _ = T.ply_read(path, threads=threads)
            assert np.array_equal(gx, xyz) and np.array_equal(gc, rgb)
    T.point_set_checksum(xyz, rgb)


@pytest.mark.gpu
@pytest.mark.parametrize("pack", [1, 2])
def test_gpu_guard_inter_frame_packers(guard, pack):
    """The spatial-consistency chain and the global patch allocation (host code over patch records of several frames)."""
    frames = [synth_cloud("tiny", f) for f in range(4)]
    enc = T.GofEncoder(0, workers=2, iterations=10, bits3d=11, occ_precision=4, min_w=128, min_h=192)
    try:
        frs = enc.upload(frames)
        W, H = enc.phase_a(frs, constrained_pack={1: True, 2: 2}[pack])
        enc.phase_b(frs)
        enc.phase_c(frs)
        for fr in frs:
            fr.get_patches(), fr.get_geometry_images(), fr.get_attribute_images(), fr.get_post_reconstruction()
        # the same chain over patch records (what a rank-0 of a sharded GOF runs)
        for fr in frs:
            fr.reset()
        enc.phase_a(frs, constrained_pack={1: True, 2: 2}[pack], records_chain=True)
    finally:
        enc.close()


@pytest.mark.gpu
def test_gpu_context_outlives_its_frames(oracle):
    """tmc2_ctx_destroy on a context that still has frames must not pull the stream and the memory pool from under them (round 4:
    a test that closed its context first hung in the frame's destructor): the destruction is carried out by the last frame to
    go, and until then the frames keep working."""
    xyz, rgb = synth_cloud("tiny")
    ctx = T.Context(0)
    a, b = ctx.frame(xyz, rgb), ctx.frame(xyz, rgb)
    a.normals_compute(16, 1)
    ctx.close()                                                       # asks for the destruction; two frames are alive
    b.normals_compute(16, 1)                                          # ... and still have their context
    assert np.array_equal(a.get_normals().view(np.uint64), b.get_normals().view(np.uint64))
    assert np.array_equal(b.get_adjacency(16), oracle.knn_self(xyz, 16))
    a.close()
    b.close()                                                         # the last one out destroys the context
    c2 = T.Context(0)                                                 # the device is fine
    fr = c2.frame(xyz, rgb)
    fr.normals_compute(16, 1)
    fr.close()
    c2.close()
