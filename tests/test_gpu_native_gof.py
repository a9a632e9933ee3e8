"""libtmc2gof.so (include/tmc2gof.h): the GOF pass driven by C++ threads in one call must leave the bytes GofEncoder's Python
workers leave -- same C-ABI calls per frame, another host."""
import numpy as np
import pytest

import tmc2_amd as T
from tmc2_amd import native_gof
from tmc2_amd.synth import synth_cloud

MIN_W = MIN_H = 256
P = 4


def buffers(n, W, H):
    return [(dict(occupancy=T.host_array((H, W), np.uint8), occ_video=T.host_array((H // P, W // P), np.uint8),
                  block_to_patch=T.host_array((H // 16, W // 16), np.uint32), geo0=T.host_array((H, W), np.uint16),
                  geo1=T.host_array((H, W), np.uint16)), T.host_array((2, 3, H, W), np.uint8)) for _ in range(n)]


def through_python(enc, frames, packing):
    for fr in frames:
        fr.reset()
    W, H = enc.phase_a(frames, constrained_pack={"all-intra": False, "low-delay": True, "random-access": 2}[packing])
    enc.phase_b(frames)
    out = []
    for fr in frames:
        g = fr.get_geometry_images()
        out.append(([np.array(g[k]) for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1")],
                    np.array(fr.get_attribute_images()), fr.get_patches()[0].tobytes()))
    return (W, H), out


def through_native(frames, workers, packing, guess):
    capacity = [MIN_W, MIN_H]
    refused = 0
    while True:
        bufs = buffers(len(frames), *capacity)
        try:
            size = native_gof.encode(frames, [i % workers for i in range(len(frames))], workers, 3, 4, 11, P, MIN_W, MIN_H, packing,
                                     bufs, capacity, guess_canvas=guess)
            break
        except native_gof.CanvasTooSmall as e:
            refused += 1
            assert refused == 1 and (e.size[0] > capacity[0] or e.size[1] > capacity[1])
            capacity = list(e.size)
    assert list(size) == capacity
    return size, [([np.array(b[0][k]) for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1")], np.array(b[1]),
                   fr.get_patches()[0].tobytes()) for b, fr in zip(bufs, frames)], refused


@pytest.mark.gpu
@pytest.mark.parametrize("packing", ["all-intra", "low-delay", "random-access"])
def test_gpu_native_gof_host_equals_the_python_host(packing):
    workers, n = 3, 7                                          # (uneven: slot 0 holds three frames)
    clouds = [synth_cloud("tiny" if i % 3 else "small", i) for i in range(n)]   # 'small' outgrows the 256 x 256 canvas
    enc = T.GofEncoder(0, workers, 3, 11, P, MIN_W, MIN_H)
    frames = enc.upload(clouds)
    want_size, want = through_python(enc, frames, packing)
    for guess in ((False, True) if packing == "all-intra" else (False,)):
        size, got, refused = through_native(frames, workers, packing, guess)
        assert tuple(size) == tuple(want_size), (size, want_size)
        assert refused == int(tuple(want_size) != (MIN_W, MIN_H))
        for i, (g, w) in enumerate(zip(got, want)):
            for x, y in zip(g[0], w[0]):
                assert x.shape == y.shape and np.array_equal(x, y), (packing, guess, i)
            assert np.array_equal(g[1], w[1]) and g[2] == w[2], (packing, guess, i)
    # and again, a second pass over the same frames (reset inside): same bytes
    size2, again, _ = through_native(frames, workers, packing, False)
    assert tuple(size2) == tuple(want_size) and all(np.array_equal(a[1], w[1]) for a, w in zip(again, want))
    for fr in frames:
        fr.close()
    enc.close()


@pytest.mark.gpu
def test_gpu_native_gof_host_reports_the_failing_call():
    """A frame that cannot be segmented (fewer points than the kNN asks for) fails the pass with the library's status and message."""
    ctx = T.Context(0)
    xyz = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.int16)
    fr = ctx.frame(xyz, np.zeros((3, 3), np.uint8))
    with pytest.raises(T.Tmc2Error, match="tmc2_segmenter_compute.*larger than the cloud"):
        native_gof.encode([fr], [0], 1, 3, 4, 11, P, MIN_W, MIN_H, "all-intra", None, (MIN_W, MIN_H))
    fr.close()
    ctx.close()


@pytest.mark.gpu
def test_gpu_native_gof_sharded_runs_rccl_from_cpp():
    """tmc2_gof_encode_sharded on this box's one GPU: a communicator of ONE rank is still librccl.so's ncclCommInitRank, and the pass
    still runs the 24-byte ncclBroadcast, the ncclAllReduce( max ) of the canvas height and the grouped ncclSend / ncclRecv of the
    packed patch records on the context's stream -- RCCL driven from the C++ host, for real.  Same canvases as the unsharded entry;
    the gathered records are the frames' patch lists in list order.  (Several ranks: tests/test_native_gof_schedule.py, against a
    recorder; the 8-GPU run: bench.py --host native --gpus 8.)"""
    workers, n = 2, 4
    clouds = [synth_cloud("tiny", i) for i in range(n)]
    enc = T.GofEncoder(0, workers, 3, 11, P, MIN_W, MIN_H)
    frames = enc.upload(clouds)
    want_size, want = through_python(enc, frames, "all-intra")
    comm = native_gof.Comm(enc.ctxs[0], rank=0, world=1)
    try:
        for rep in range(2):
            bufs = buffers(n, *want_size)
            W, H, records = native_gof.encode_sharded(comm, frames, [i % workers for i in range(n)], workers, 3, 4, 11, P, MIN_W, MIN_H,
                                                      bufs, want_size)
            assert (W, H) == tuple(want_size)
            for i, (b, w, fr) in enumerate(zip(bufs, want, frames)):
                for k, y in zip(("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"), w[0]):
                    assert np.array_equal(b[0][k], y), (rep, i, k)
                assert np.array_equal(b[1], w[1]), (rep, i)
                assert records[0][i].tobytes() == fr.get_patches()[0][fr.get_patch_order()].tobytes(), (rep, i)
    finally:
        comm.close()
        for fr in frames:
            fr.close()
        enc.close()


@pytest.mark.gpu
def test_gpu_two_contexts_two_sets_of_options():
    """Options are per context (tmc2_ctx_set_option): two contexts of one process build the same tree through different tiers of
    the device builder, and the environment is read once, when a context is created."""
    import os
    xyz, _ = synth_cloud("medium")
    a = T.Context(0)
    os.environ["TMC2_KD_FORM"] = "tiers"
    try:
        b = T.Context(0)
    finally:
        del os.environ["TMC2_KD_FORM"]
    assert a.get_option("KD_FORM") is None and b.get_option("KD_FORM") == "tiers"
    a.set_option("TMC2_KD_HUGEMAX", 4096)                      # (the prefix is accepted)
    assert a.get_option("KD_HUGEMAX") == "4096"
    fa, fb = a.frame(xyz), b.frame(xyz)
    pa, pb = fa.kdtree_order(), fb.kdtree_order()
    assert np.array_equal(pa[0], pb[0]) and pa[1] == pb[1]
    assert np.array_equal(pa[0], T.host_kdtree_build(xyz)[0])
    assert "kdtree_build" in a.stage_ms() and "kdtree_build" in b.stage_ms()
    a.set_option("KD_HUGEMAX", None)
    assert a.get_option("KD_HUGEMAX") is None
    st = a.pool_stats()
    assert st["bytes_held"] > 0 and st["hipmalloc_calls"] > 0 and st["carved_blocks"] == 0
    c = T.Context(0)
    c.reserve(len(xyz), 4, 11, 1280, 1280)
    fc = c.frame(xyz)
    assert np.array_equal(fc.kdtree_order()[0], pa[0])
    st = c.pool_stats()
    assert st["carved_blocks"] > 0 and st["hipmalloc_calls"] == 0, st
    for f in (fa, fb, fc):
        f.close()
