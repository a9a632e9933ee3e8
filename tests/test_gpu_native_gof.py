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
    """A GOF that outgrows the buffers is refused once, with the size it needs, and RESUMED there with larger buffers
    (tmc2_gof_encode_resume: S0-S10 are not repeated; with a guessed canvas there is no second half: the whole pass again)."""
    capacity = [MIN_W, MIN_H]
    refused = 0
    while True:
        bufs = buffers(len(frames), *capacity)
        try:
            size = native_gof.encode(frames, [i % workers for i in range(len(frames))], workers, 3, 4, 11, P, MIN_W, MIN_H, packing,
                                     bufs, capacity, guess_canvas=guess, resume=refused > 0 and not guess)
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
def test_gpu_native_gof_resume_needs_packed_frames():
    """tmc2_gof_encode_resume on frames that were reset: there is no tile to take the canvas from (TMC2_E_STATE, nothing rasterised)."""
    enc = T.GofEncoder(0, 2, 3, 11, P, MIN_W, MIN_H)
    frames = enc.upload([synth_cloud("tiny", i) for i in range(2)])
    native_gof.encode(frames, [0, 1], 2, 3, 4, 11, P, MIN_W, MIN_H, "all-intra", None, (MIN_W, MIN_H))
    for fr in frames:
        fr.reset()
    with pytest.raises(T.Tmc2Error, match="not packed"):
        native_gof.encode(frames, [0, 1], 2, 3, 4, 11, P, MIN_W, MIN_H, "all-intra", None, (MIN_W, MIN_H), resume=True)
    for fr in frames:
        fr.close()
    enc.close()


def _sharded_pass(comm, frames, workers, packing, want_size=None):
    """tmc2_gof_encode_sharded on this rank's frames, the way bench.py drives it: refused once if the GOF outgrows the minimum canvas,
    resumed with the size it needs.  -> (size, buffers, records, times refused)"""
    capacity, refused = [MIN_W, MIN_H], 0
    while True:
        bufs = buffers(len(frames), *capacity)
        try:
            W, H, records = native_gof.encode_sharded(comm, frames, [i % workers for i in range(len(frames))], workers, 3, 4, 11, P, MIN_W,
                                                      MIN_H, bufs, capacity, packing=packing, resume=refused > 0)
            return (W, H), bufs, records, refused
        except native_gof.CanvasTooSmall as e:
            refused += 1
            assert refused == 1
            capacity = list(e.size)


@pytest.mark.gpu
@pytest.mark.parametrize("packing", ["all-intra", "low-delay", "random-access"])
def test_gpu_native_gof_sharded_runs_rccl_from_cpp(packing, monkeypatch):
    """tmc2_gof_encode_sharded on this box's one GPU: a communicator of ONE rank is still librccl.so's ncclCommInitRank, and the pass
    still runs its collectives on the context's stream -- RCCL driven from the C++ host, for real.  all-intra: the 24-byte
    ncclBroadcast, the ncclAllReduce( max ) of the canvas height, the grouped ncclSend / ncclRecv of the packed patch records.
    low-delay / random-access (round 6; TMC2_GOF_RECORDS_CHAIN forces the route several ranks take): records and block-occupancy
    pools through a grouped send / receive to "rank 0", PCCEncoder::placeSegments there over the records (tmc2_host_place_segments),
    the packed lists back through a second one, tmc2_frame_set_packing.  Same canvases as the Python host's frame-based chain;
    the records rank 0 is left with are the frames' patch lists in list order.  The GOF outgrows the minimum canvas: refused once,
    resumed (tmc2_gof_encode_sharded_resume).  (Several ranks: tests/test_native_gof_schedule.py against a recorder, and
    test_gpu_native_gof_sharded_over_the_gpus_of_this_box below wherever the box has them.)"""
    monkeypatch.setenv("TMC2_GOF_RECORDS_CHAIN", "1")
    workers, n = 2, 4
    clouds = [synth_cloud("medium" if i == 2 else "tiny", i) for i in range(n)]       # 'medium' (180 K points) outgrows the 256 x 256 canvas
    enc = T.GofEncoder(0, workers, 3, 11, P, MIN_W, MIN_H)
    frames = enc.upload(clouds)
    want_size, want = through_python(enc, frames, packing)
    assert tuple(want_size) != (MIN_W, MIN_H)
    comm = native_gof.Comm(enc.ctxs[0], rank=0, world=1)
    try:
        for rep in range(2):
            size, bufs, records, refused = _sharded_pass(comm, frames, workers, packing)
            assert size == tuple(want_size) and refused == 1
            for i, (b, w, fr) in enumerate(zip(bufs, want, frames)):
                for k, y in zip(("occupancy", "occ_video", "block_to_patch", "geo0", "geo1"), w[0]):
                    assert np.array_equal(b[0][k], y), (rep, i, k)
                assert np.array_equal(b[1], w[1]), (rep, i)
                mine = fr.get_patches()[0][fr.get_patch_order()]
                assert len(records[0][i]) == len(mine)
                if packing == "all-intra":                         # (the gather moves the frame's own records: every byte)
                    assert records[0][i].tobytes() == mine.tobytes(), (rep, i)
                for field in ("u0", "v0", "patchOrientation", "sizeU0", "sizeV0", "u1", "v1", "d1", "index"):
                    assert np.array_equal(records[0][i][field], mine[field]), (rep, i, field)
    finally:
        comm.close()
        for fr in frames:
            fr.close()
        enc.close()


def _rank_of_a_real_world(args):
    """One rank of a sharded GOF on its OWN GPU (a process of its own): RCCL between real devices, from the C++ host."""
    rank, world, n, rendezvous, out_path = args
    import hashlib
    import pickle
    clouds = [synth_cloud("medium" if f == 2 else "tiny", f) for f in range(rank, n * world, world)]     # frame f on rank f mod world
    enc = T.GofEncoder(rank, 2, 3, 11, P, MIN_W, MIN_H)
    frames = enc.upload(clouds)
    comm = native_gof.Comm(enc.ctxs[0], rank=rank, world=world, rendezvous=rendezvous)
    res = {}
    try:
        for packing in ("all-intra", "low-delay", "random-access"):
            size, bufs, records, refused = _sharded_pass(comm, frames, 2, packing)
            res[packing] = dict(size=size, refused=refused,
                                md5=[[hashlib.md5(np.ascontiguousarray(b[0][k]).tobytes()).hexdigest()
                                      for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1")] +
                                     [hashlib.md5(np.ascontiguousarray(b[1]).tobytes()).hexdigest()] for b in bufs],
                                records=None if records is None else [[(r_["u0"].tolist(), r_["v0"].tolist()) for r_ in per] for per in records])
    finally:
        comm.close()
        for fr in frames:
            fr.close()
        enc.close(join=True)
    with open(out_path, "wb") as f:
        pickle.dump(res, f)
    return 0


@pytest.mark.gpu
def test_gpu_native_gof_sharded_over_the_gpus_of_this_box(tmp_path):
    """ARMS ITSELF wherever the box shows two or more GPUs (skipped on the one-GPU box the builder has -- say so): one process per
    GPU, tmc2_gof_comm_create for real (the id through a file with a nonce, ncclCommInitRank between devices) and
    tmc2_gof_encode_sharded under all three packing conditions -- the weights broadcast, the height all-reduce, the grouped
    send / receive of the records, the records route of the packing chains and the resume of a GOF that outgrows the minimum canvas
    -- against the single-process Python host over the same frames on device 0."""
    import hashlib
    import multiprocessing as mp
    import pickle
    import uuid
    import torch
    have = torch.cuda.device_count()
    if have < 2:
        pytest.skip("this box shows %d GPU: a real world of several RCCL ranks needs one GPU per rank (the test arms itself on a "
                    "multi-GPU box; worlds of 2 and 4 ranks run against a recorder in tests/test_native_gof_schedule.py)" % have)
    world, n = min(have, 4), 2
    rendezvous = "/dev/shm/tmc2_gof_test_id_%s" % uuid.uuid4().hex
    outs = [str(tmp_path / ("rank%d.pkl" % r)) for r in range(world)]
    with mp.get_context("spawn").Pool(world) as pool:
        assert pool.map(_rank_of_a_real_world, [(r, world, n, rendezvous, outs[r]) for r in range(world)]) == [0] * world
    res = []
    for o in outs:
        with open(o, "rb") as f:
            res.append(pickle.load(f))
    clouds = [synth_cloud("medium" if f == 2 else "tiny", f) for f in range(n * world)]
    enc = T.GofEncoder(0, 2, 3, 11, P, MIN_W, MIN_H)
    frames = enc.upload(clouds)
    md5 = lambda x: hashlib.md5(np.ascontiguousarray(x).tobytes()).hexdigest()
    for packing in ("all-intra", "low-delay", "random-access"):
        want_size, want = through_python(enc, frames, packing)
        lists = [fr.get_patches()[0][fr.get_patch_order()] for fr in frames]
        for r in range(world):
            got = res[r][packing]
            assert got["size"] == tuple(want_size) and got["refused"] == 1, (packing, r, got["size"])
            for i in range(n):
                f = r + i * world
                assert got["md5"][i] == [md5(x) for x in want[f][0]] + [md5(want[f][1])], (packing, f)
        for r in range(world):
            for i in range(n):
                f = r + i * world
                assert res[0][packing]["records"][r][i] == (lists[f]["u0"].tolist(), lists[f]["v0"].tolist()), (packing, f)
        assert all(res[r][packing]["records"] is None for r in range(1, world))
    for fr in frames:
        fr.close()
    enc.close()


@pytest.mark.gpu
def test_gpu_two_contexts_two_sets_of_options():
    """Options are per context (tmc2_ctx_set_option): two contexts of one process build the same tree through different forms of
    the piece kernel, and the environment is read once, when a context is created."""
    import os
    xyz, _ = synth_cloud("medium")
    a = T.Context(0)
    os.environ["TMC2_KD_PIECE_PER"] = "8"
    try:
        b = T.Context(0)
    finally:
        del os.environ["TMC2_KD_PIECE_PER"]
    assert a.get_option("KD_PIECE_PER") is None and b.get_option("KD_PIECE_PER") == "8"
    a.set_option("TMC2_KD_HUGEMAX", 4096)                      # (the prefix is accepted)
    assert a.get_option("KD_HUGEMAX") == "4096"
    fa, fb = a.frame(xyz), b.frame(xyz)
    pa, pb = fa.kdtree_order(), fb.kdtree_order()
    assert np.array_equal(pa[0], pb[0]) and pa[1] == pb[1]
    assert np.array_equal(pa[0], T.host_kdtree_build(xyz)[0])
    assert "kdtree_build" in a.stage_ms() and "kdtree_build" in b.stage_ms()
    a.set_option("KD_HUGEMAX", None)
    assert a.get_option("KD_HUGEMAX") is None
    # ... and so is the gate around the host-resident steps (tmc2_host_gate_create / tmc2_ctx_set_host_gate): two contexts behind ONE
    # gate of one slot orient their frames from two threads -- one walk at a time, both finish; a third context keeps the default gate
    import threading
    gate = T.HostGate(1)
    a.set_host_gate(gate)
    b.set_host_gate(gate)
    errs = []

    def orient(fr):
        try:
            fr.normals_compute(16, 1)
        except Exception as e:                                 # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=orient, args=(fr,)) for fr in (fa, fb)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs and np.array_equal(fa.get_normals().view(np.uint64), fb.get_normals().view(np.uint64))
    gate.close()                                               # (the contexts keep it alive)
    fa.reset()
    fa.normals_compute(16, 1)
    a.set_host_gate(None)
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
