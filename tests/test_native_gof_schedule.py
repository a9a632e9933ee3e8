"""The schedule libtmc2gof.so drives (mpeg-pcc-tmc2_amd/host/gof_runner.cpp), without a device: the runner is compiled against a
recorder of the C-ABI entries it calls (tests/mock/mock_tmc2hip.cpp) and the log is checked -- which calls, on which frames, in
which order on each slot, one rendezvous, the packing chains in frame order, the guessed canvas and its repeat, a canvas larger
than the buffers, a failing call.  The bytes are the GPU tier's business (tests/test_gpu_native_gof.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESET, WEIGHT, SEGMENT, PACK_FLEXIBLE, PACK_CHAIN, GPA, PACKED_SIZE, GEOMETRY, ATTRIBUTE, GET_GEOMETRY, GET_ATTRIBUTE = range(1, 12)
MIN_W, MIN_H = 128, 128


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("iterations", "voxel", "bits3d", "precision", "min_w", "min_h", "packing", "guess")]


@pytest.fixture(scope="module")
def runner(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("mock_gof"))
    inc = os.path.join(ROOT, "include")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + inc,
                    os.path.join(ROOT, "tests", "mock", "mock_tmc2hip.cpp"), "-o", os.path.join(d, "libtmc2hipmock.so"), "-pthread"], check=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + inc,
                    os.path.join(ROOT, "mpeg-pcc-tmc2_amd", "host", "gof_runner.cpp"), "-o", os.path.join(d, "libtmc2gofmock.so"),
                    "-L" + d, "-ltmc2hipmock", "-Wl,-rpath," + d, "-pthread"], check=True)
    M = C.CDLL(os.path.join(d, "libtmc2hipmock.so"), mode=C.RTLD_GLOBAL)
    G = C.CDLL(os.path.join(d, "libtmc2gofmock.so"))
    M.mock_frame.restype = C.c_void_p
    M.mock_frame_free.argtypes = [C.c_void_p]
    G.tmc2_gof_last_error.restype = C.c_char_p
    return M, G


def run(runner, heights, slots, packing, guess=0, fail=None, widths=None, capacity=(MIN_W, MIN_H), buffers=True):
    """-> (status, (W, H), log [(call, frame, a, b, thread)], occupancy buffers, attribute buffers)"""
    M, G = runner
    n = len(heights)
    frames = [M.mock_frame(i, int(heights[i]), int((widths or [MIN_W] * n)[i]), int(fail[1]) if fail and fail[0] == i else 0) for i in range(n)]
    handles = (C.c_void_p * n)(*frames)
    slot_of = (C.c_int32 * n)(*[i % slots for i in range(n)])
    cfg = Config(7, 4, 11, 4, MIN_W, MIN_H, packing, guess)
    occ = [np.zeros(capacity[0] * capacity[1] + 64, np.uint8) for _ in range(n)]        # (64 bytes of red zone behind)
    att = [np.zeros(16, np.uint8) for _ in range(n)]
    ptrs = lambda arrs: (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    W, H = C.c_int32(0), C.c_int32(0)
    M.mock_log_clear()
    rc = G.tmc2_gof_encode(handles, slot_of, n, slots, C.byref(cfg), ptrs(occ) if buffers else None, None, None, None, None,
                           ptrs(att) if buffers else None, capacity[0], capacity[1], C.byref(W), C.byref(H))
    log = []
    for i in range(M.mock_log_size()):
        v = [C.c_int32() for _ in range(4)] + [C.c_uint64()]
        M.mock_log_get(i, *[C.byref(x) for x in v])
        log.append(tuple(x.value for x in v))
    for f in frames:
        M.mock_frame_free(f)
    return rc, (W.value, H.value), log, occ, att


def per_frame(log, frame):
    return [e[0] for e in log if e[1] == frame and e[0] != WEIGHT]


def test_all_intra_meets_once_and_keeps_the_order_of_a_slot(runner):
    heights = [100, 90, 128, 70, 60, 127, 50]
    rc, size, log, occ, att = run(runner, heights, 3, 0)
    assert rc == 0 and size == (MIN_W, MIN_H)
    # S0 on frame 0 only, after every frame has been reset and before any frame is segmented; its weights reach every frame
    calls = [e[0] for e in log]
    assert calls[:7] == [RESET] * 7 and calls[7] == WEIGHT and log[7][1:3] == (0, 11) and calls.count(WEIGHT) == 1
    assert {e[2:4] for e in log if e[0] == SEGMENT} == {(7, 116)}                 # (iterations, weight[2] = 11 + 0.6)
    for f in range(7):
        assert per_frame(log, f) == [RESET, SEGMENT, PACK_FLEXIBLE, GEOMETRY, ATTRIBUTE, GET_GEOMETRY, GET_ATTRIBUTE]
    # one rendezvous: no frame rasterises before the last frame is packed; every frame on the common canvas
    assert max(i for i, e in enumerate(log) if e[0] == PACK_FLEXIBLE) < min(i for i, e in enumerate(log) if e[0] == GEOMETRY)
    assert {e[2:4] for e in log if e[0] == GEOMETRY} == {(MIN_W, MIN_H)}
    # a slot is one thread per phase and takes its frames in index order
    for phase in ((SEGMENT, PACK_FLEXIBLE), (GEOMETRY, ATTRIBUTE, GET_GEOMETRY, GET_ATTRIBUTE)):
        for s in range(3):
            mine = [e for e in log if e[0] in phase and e[1] % 3 == s]
            assert len({e[4] for e in mine}) == 1 and [e[1] for e in mine] == sorted(e[1] for e in mine)
        assert len({e[4] for e in log if e[0] in phase}) == 3
    assert all(o[:MIN_W * MIN_H].min() == 1 + i and o[MIN_W * MIN_H:].max() == 0 for i, o in enumerate(occ))
    assert [int(a[0]) for a in att] == [100 + i for i in range(7)]


def test_a_gof_that_outgrows_the_buffers_is_refused_before_anything_is_written(runner):
    rc, size, log, occ, _ = run(runner, [100, 200, 130], 2, 0)
    assert rc == -3 and size == (MIN_W, 256)                                      # TMC2_E_INVALID, and what it needs
    assert b"needs a 128 x 256 canvas" in runner[1].tmc2_gof_last_error()
    assert not [e for e in log if e[0] >= GEOMETRY] and all(o.max() == 0 for o in occ)
    rc, size, log, occ, _ = run(runner, [100, 200, 130], 2, 0, capacity=size)    # the caller grows the buffers and calls again
    assert rc == 0 and size == (MIN_W, 256) and {e[2:4] for e in log if e[0] == GET_GEOMETRY} == {(MIN_W, 256)}
    rc, size, log, _, _ = run(runner, [100, 200, 130], 2, 0, buffers=False)      # nothing leaves the device: no capacity to respect
    assert rc == 0 and size == (MIN_W, 256) and not [e for e in log if e[0] in (GET_GEOMETRY, GET_ATTRIBUTE)]
    assert sum(e[0] == ATTRIBUTE for e in log) == 3


def test_guessed_canvas_repeats_only_the_frames_that_guessed_short(runner):
    heights = [100, 200, 90, 129]                                                 # own canvases: 128, 256, 128, 192 rows -> the GOF's: 256
    guesses = [128, 256, 128, 192]
    rc, size, log, occ, _ = run(runner, heights, 2, 0, guess=1, capacity=(MIN_W, 256))
    assert rc == 0 and size == (MIN_W, 256)
    whole = [GEOMETRY, ATTRIBUTE, GET_GEOMETRY, GET_ATTRIBUTE]
    for f, g in enumerate(guesses):
        assert per_frame(log, f) == [RESET, SEGMENT, PACK_FLEXIBLE] + whole + (whole if g != 256 else [])
        sizes = [e[2:4] for e in log if e[0] == GEOMETRY and e[1] == f]
        assert sizes == ([(MIN_W, g), (MIN_W, 256)] if g != 256 else [(MIN_W, 256)])
    # nobody waits in the first pass: a frame rasterises on its guess before the last frame of the GOF is packed
    assert min(i for i, e in enumerate(log) if e[0] == GEOMETRY) < max(i for i, e in enumerate(log) if e[0] == PACK_FLEXIBLE)
    assert all(o[:MIN_W * 256].min() == 1 + i for i, o in enumerate(occ))
    # a guess larger than the buffers is not rasterised at all; the call is refused at the rendezvous
    rc, size, log, occ, _ = run(runner, heights, 2, 0, guess=1)
    assert rc == -3 and size == (MIN_W, 256) and all(o[MIN_W * MIN_H:].max() == 0 for o in occ)
    assert {e[1] for e in log if e[0] == GEOMETRY} == {0, 2}                    # (the two whose own canvas fits the buffers)


@pytest.mark.parametrize("packing", [1, 2])
def test_chained_packers_run_in_frame_order_at_the_rendezvous(runner, packing):
    heights, widths = [100, 140, 90, 60, 120], [128, 128, 192, 128, 128]
    rc, size, log, _, _ = run(runner, heights, 2, packing, widths=widths, capacity=(192, 256))
    assert rc == 0
    seg_end = max(i for i, e in enumerate(log) if e[0] == SEGMENT)
    chain = [e for e in log if e[0] in (PACK_FLEXIBLE, PACK_CHAIN, GPA, PACKED_SIZE)]
    assert log.index(chain[0]) > seg_end and len({e[4] for e in chain}) == 1     # after every segmentation, on one thread
    assert [(e[0], e[1], e[2]) for e in chain[:5]] == [(PACK_FLEXIBLE, 0, 0)] + [(PACK_CHAIN, f, f - 1) for f in range(1, 5)]
    if packing == 2:
        assert [(e[0], e[2]) for e in chain[5:]] == [(GPA, 5)]
        assert size == (192, 192)                                                 # widths / heights as the allocation left them (140 + 16 -> 192)
    else:
        assert [e[0] for e in chain[5:]] == [PACKED_SIZE] * 5
        assert size == (192, 192)                                                 # (140 -> 192 rows; the widest tile)
    assert log.index(chain[-1]) < min(i for i, e in enumerate(log) if e[0] == GEOMETRY)
    assert {e[2:4] for e in log if e[0] == GEOMETRY} == {size}


@pytest.mark.parametrize("call,packing", [(SEGMENT, 0), (PACK_FLEXIBLE, 0), (ATTRIBUTE, 0), (GET_ATTRIBUTE, 0), (PACK_CHAIN, 1), (GPA, 2), (RESET, 0)])
def test_a_failing_call_ends_the_pass_with_its_status_and_message(runner, call, packing):
    victim = 0 if call == GPA else 3
    rc, _, log, _, _ = run(runner, [100] * 8, 4, packing, fail=(victim, call))
    assert rc == -2                                                               # the failing call's own status (TMC2_E_HIP)
    msg = runner[1].tmc2_gof_last_error().decode()
    assert ("mock failure of call %d on frame %d" % (call, victim)) in msg and msg.startswith("tmc2_")
    if call == RESET:
        assert not [e for e in log if e[0] > RESET]
    # the slot of the failing frame stops there; nobody starts a later phase
    after = [e for e in log[max(i for i, e in enumerate(log) if e[1] == victim and e[0] == call) + 1:] if e[1] == victim]
    assert not after
    if call in (SEGMENT, PACK_FLEXIBLE, PACK_CHAIN, GPA):
        assert not [e for e in log if e[0] >= GEOMETRY]


# ---- the sharded mode: one process per rank, RCCL replaced by a recorder that moves the bytes through files -------------------------
class _Patch(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("index", "viewId", "normalAxis", "tangentAxis", "bitangentAxis", "projectionMode", "u1", "v1", "d1",
                                         "sizeU", "sizeV", "sizeD", "sizeDPixel", "sizeU0", "sizeV0", "size2DXInPixel", "size2DYInPixel",
                                         "d0Count", "eomAndD1Count", "u0", "v0", "patchOrientation")] + [("depthOffset", C.c_int64), ("occOffset", C.c_int64)]


def _sharded_rank(args):
    """One rank of a sharded GOF pass against the recorders (runs in a process of its own)."""
    d, rank, world, frames_per_rank, heights, packing, fail = args
    os.environ["TMC2_RCCL_LIBRARY"] = os.path.join(d, "libmockrccl.so")
    os.environ["MOCK_RCCL_DIR"] = d
    M = C.CDLL(os.path.join(d, "libtmc2hipmock.so"), mode=C.RTLD_GLOBAL)
    G = C.CDLL(os.path.join(d, "libtmc2gofmock.so"))
    M.mock_frame.restype = C.c_void_p
    M.mock_ctx.restype = C.c_void_p
    G.tmc2_gof_last_error.restype = C.c_char_p
    ctx = C.c_void_p(M.mock_ctx(rank))
    comm = C.c_void_p()
    rc = G.tmc2_gof_comm_create(rank, world, ctx, os.path.join(d, "id").encode(), C.byref(comm))
    if rc != 0:
        return {"rc": rc, "err": G.tmc2_gof_last_error().decode()}
    n = frames_per_rank
    ids = [rank + i * world for i in range(n)]                       # frame f of the GOF on rank f mod world
    frames = [M.mock_frame(f, int(heights[f]), MIN_W, int(fail[1]) if fail and fail[0] == f else 0) for f in ids]
    handles = (C.c_void_p * n)(*frames)
    slot_of = (C.c_int32 * n)(*[i % 2 for i in range(n)])
    cfg = Config(7, 4, 11, 4, MIN_W, MIN_H, packing, 0)
    W, H = C.c_int32(0), C.c_int32(0)
    slots_rec = 16
    gathered = (_Patch * (world * n * slots_rec))()
    counts = (C.c_int64 * (world * n))()
    M.mock_log_clear()
    rc = G.tmc2_gof_encode_sharded(comm, handles, slot_of, n, 2, C.byref(cfg), None, None, None, None, None, None, 1 << 20, 1 << 20,
                                   C.byref(W), C.byref(H), slots_rec, gathered, counts)
    err = G.tmc2_gof_last_error().decode()
    log = []
    for i in range(M.mock_log_size()):
        call, frame, a, b, th = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
        M.mock_log_get(i, C.byref(call), C.byref(frame), C.byref(a), C.byref(b), C.byref(th))
        log.append((call.value, frame.value, a.value, b.value))
    G.tmc2_gof_comm_destroy(comm)
    recs = [[(gathered[(s * slots_rec) + k].u0, gathered[(s * slots_rec) + k].v0) for k in range(int(counts[s]))] for s in range(world * n)]
    with open(os.path.join(d, "log_%d" % rank)) as f:
        rccl = f.read().split("\n")
    return {"rc": rc, "err": err, "size": (W.value, H.value), "log": log, "counts": list(counts), "records": recs, "rccl": [x for x in rccl if x]}


@pytest.fixture(scope="module")
def sharded_libs(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("mock_gof_sharded"))
    inc = os.path.join(ROOT, "include")
    flags = ["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + inc]
    subprocess.run(flags + [os.path.join(ROOT, "tests", "mock", "mock_tmc2hip.cpp"), "-o", os.path.join(d, "libtmc2hipmock.so"), "-pthread"], check=True)
    subprocess.run(flags + [os.path.join(ROOT, "tests", "mock", "mock_rccl.cpp"), "-o", os.path.join(d, "libmockrccl.so")], check=True)
    subprocess.run(flags + [os.path.join(ROOT, "mpeg-pcc-tmc2_amd", "host", "gof_runner.cpp"), "-o", os.path.join(d, "libtmc2gofmock.so"),
                            "-L" + d, "-ltmc2hipmock", "-Wl,-rpath," + d, "-pthread", "-ldl"], check=True)
    return d


def _run_world(d, world, frames_per_rank, heights, packing=0, fail=None):
    import multiprocessing as mp
    for name in os.listdir(d):                                         # (files of an earlier world)
        if name.startswith(("log_", "p2p_", "ar_", "bcast_", "id")):
            os.unlink(os.path.join(d, name))
    with mp.get_context("spawn").Pool(world) as pool:
        return pool.map(_sharded_rank, [(d, r, world, frames_per_rank, heights, packing, fail) for r in range(world)])


@pytest.mark.parametrize("world", [1, 2, 4])
def test_sharded_gof_crosses_the_node_three_times(sharded_libs, world):
    """tmc2_gof_encode_sharded (RCCL from C++: include/tmc2gof.h), one process per rank against recorders: S0 runs on rank 0 only
    and its weights reach every rank's frames (24-byte broadcast); every rank rasterises on the canvas the TALLEST frame of the
    GOF needs (all-reduce of one int32, max); the packed records of all frames end on rank 0, in list order (one grouped send /
    receive) -- and nothing else crosses: per pass one broadcast, one all-reduce, one group.  World 1 runs the same collectives."""
    n = 3
    heights = [64 + 8 * f for f in range(world * n)]
    heights[world * n - 1] = 400                                        # the tallest frame lives on the LAST rank
    res = _run_world(sharded_libs, world, n, heights)
    for r, x in enumerate(res):
        assert x["rc"] == 0, (r, x["err"])
        assert x["size"] == (MIN_W, 448), x["size"]                      # (the mock rounds the height up to a multiple of 64)
        weights = [e for e in x["log"] if e[0] == WEIGHT]
        assert (len(weights) == 1 and weights[0][1] == 0) if r == 0 else not weights
        seg = [e for e in x["log"] if e[0] == SEGMENT]
        assert sorted(e[1] for e in seg) == [r + i * world for i in range(n)] and all(e[2] == 7 and e[3] == 116 for e in seg), seg
        assert all(e[2:] == (MIN_W, 448) for e in x["log"] if e[0] == GEOMETRY)
        calls = [c.split()[0] for c in x["rccl"]]
        per_pass = calls[calls.index("allreduce") + 1:-1]                # (after create's pre-flight all-reduce, before destroy)
        assert per_pass.count("broadcast") == 1 and per_pass.count("allreduce") == 1 and per_pass.count("group") == 1, x["rccl"]
        assert per_pass.count("send") == 1 and per_pass.count("recv") == (world if r == 0 else 0), x["rccl"]
        assert "broadcast 24 root 0" in x["rccl"] and "allreduce 4 op 2" in x["rccl"]
    got = res[0]
    for r in range(world):
        for i in range(n):
            f = r + i * world
            assert got["counts"][r * n + i] == 3 + f % 5
            assert got["records"][r * n + i] == [(f, k) for k in range(3 + f % 5)], (f, got["records"][r * n + i])


def test_sharded_gof_refuses_the_packing_chains(sharded_libs):
    """The low-delay / random-access chains run over all frames of the GOF in order: with the frames on several ranks that is the
    caller's job, and the sharded entry says so on every rank before anything is queued (nobody is left waiting in a collective)."""
    res = _run_world(sharded_libs, 2, 2, [64, 64, 64, 64], packing=2)
    for x in res:
        assert x["rc"] != 0 and "packing chains" in x["err"], x
        assert not [e for e in x["log"] if e[0] in (SEGMENT, WEIGHT)]


@pytest.mark.parametrize("frame,call", [(0, WEIGHT), (3, SEGMENT), (1, GEOMETRY), (2, ATTRIBUTE)])
def test_sharded_gof_a_failing_rank_does_not_leave_the_others_waiting(sharded_libs, frame, call):
    """A call that fails on ONE rank (S0 on rank 0; a frame's segmentation before the rendezvous; a frame's images after it): every
    rank still goes through the same collectives -- the failed rank with a value that says so (negative weights, a height no canvas
    has, a negative record count) -- and every rank returns with an error instead of waiting in a receive for a rank that has left.
    The rank that failed reports the library's message, the others that another rank failed."""
    world, n = 2, 2
    res = _run_world(sharded_libs, world, n, [64] * (world * n), fail=(frame, call))      # (returns at all: nobody hangs)
    owner = frame % world
    assert res[owner]["rc"] != 0 and "mock failure of call %d on frame %d" % (call, frame) in res[owner]["err"], res[owner]
    assert res[0]["rc"] != 0                                           # the rank that gathers always knows
    other = res[1 - owner]
    if call in (WEIGHT, SEGMENT) or owner != 0:                        # before the rendezvous everybody learns it; after it, rank 0 does
        assert other["rc"] != 0 and "rank" in other["err"] and "its own call says why" in other["err"], other
    else:                                                              # (rank 0 failed after the rendezvous: rank 1 delivered its records)
        assert other["rc"] == 0, other
