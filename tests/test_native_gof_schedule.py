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
(RESET, WEIGHT, SEGMENT, PACK_FLEXIBLE, PACK_CHAIN, GPA, PACKED_SIZE, GEOMETRY, ATTRIBUTE, GET_GEOMETRY, GET_ATTRIBUTE, PLACE,
 SET_PACKING) = range(1, 14)
MIN_W, MIN_H = 128, 128
# TMC2_TEST_SANITIZE=thread | address,undefined: the recorders and the runner are built with that sanitizer (run the module with
# the matching runtime preloaded: LD_PRELOAD=$(g++ -print-file-name=libtsan.so); tools/sanitize_gof_runner.sh does both)
SANITIZE = ["-fsanitize=" + os.environ["TMC2_TEST_SANITIZE"], "-g", "-fno-omit-frame-pointer"] if os.environ.get("TMC2_TEST_SANITIZE") else []


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("iterations", "voxel", "bits3d", "precision", "min_w", "min_h", "packing", "guess")]


@pytest.fixture(scope="module")
def runner(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("mock_gof"))
    inc = os.path.join(ROOT, "include")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + inc] + SANITIZE +
                   [os.path.join(ROOT, "tests", "mock", "mock_tmc2hip.cpp"), "-o", os.path.join(d, "libtmc2hipmock.so"), "-pthread"], check=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + inc] + SANITIZE +
                   [os.path.join(ROOT, "mpeg-pcc-tmc2_amd", "host", "gof_runner.cpp"), "-o", os.path.join(d, "libtmc2gofmock.so"),
                    "-L" + d, "-ltmc2hipmock", "-Wl,-rpath," + d, "-pthread", "-ldl"], check=True)
    M = C.CDLL(os.path.join(d, "libtmc2hipmock.so"), mode=C.RTLD_GLOBAL)
    G = C.CDLL(os.path.join(d, "libtmc2gofmock.so"))
    M.mock_frame.restype = C.c_void_p
    M.mock_frame_free.argtypes = [C.c_void_p]
    G.tmc2_gof_last_error.restype = C.c_char_p
    return M, G


def run(runner, heights, slots, packing, guess=0, fail=None, widths=None, capacity=(MIN_W, MIN_H), buffers=True, resume_with=None):
    """-> (status, (W, H), log [(call, frame, a, b, thread)], occupancy buffers, attribute buffers); resume_with = (W, H): after the
    pass, tmc2_gof_encode_resume on the same frames with buffers of that size -- the returned values are the resumed call's"""
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
    if resume_with is not None:
        first = (rc, (W.value, H.value), M.mock_log_size())
        occ = [np.zeros(resume_with[0] * resume_with[1] + 64, np.uint8) for _ in range(n)]
        M.mock_log_clear()
        rc = G.tmc2_gof_encode_resume(handles, slot_of, n, slots, C.byref(cfg), ptrs(occ), None, None, None, None, ptrs(att),
                                      resume_with[0], resume_with[1], C.byref(W), C.byref(H))
        run.first = first
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


@pytest.mark.parametrize("packing", [0, 1, 2])
def test_resume_starts_at_the_canvas_and_repeats_nothing(runner, packing):
    """A GOF that outgrows the buffers is refused at the rendezvous with the size it needs; tmc2_gof_encode_resume with buffers of that
    size does the second half of THAT pass: no reset, no S0, no segmentation, no packing, no chain -- the canvas comes from the tiles
    the packers left in the frames (round 6: bench.py used to call the whole pass again, which is what made the first pass of a
    process over the longdress GOF -- 1280 x 1344 on a 1280 x 1280 minimum canvas -- twice as long as the later ones)."""
    heights = [100, 200, 130, 90]
    need = (MIN_W, 256)                                                           # 200 rows (+ 16 under random access) -> 256
    rc, size, log, occ, att = run(runner, heights, 2, packing, resume_with=need)
    assert run.first[0] == -3 and run.first[1] == need                            # the pass itself: TMC2_E_INVALID + what the GOF needs
    assert rc == 0 and size == need
    calls = {e[0] for e in log}
    assert not calls & {RESET, WEIGHT, SEGMENT, PACK_FLEXIBLE, PACK_CHAIN, GPA, PLACE, SET_PACKING}, sorted(calls)
    for f in range(4):
        assert [e[0] for e in log if e[1] == f] == [PACKED_SIZE, GEOMETRY, ATTRIBUTE, GET_GEOMETRY, GET_ATTRIBUTE]
    assert {e[2:4] for e in log if e[0] == GEOMETRY} == {need}
    assert all(o[:need[0] * need[1]].min() == 1 + i and o[need[0] * need[1]:].max() == 0 for i, o in enumerate(occ))


def test_resume_refuses_frames_that_are_not_packed(runner):
    """Frames that were reset (or never packed: a mock frame made with a tile width of 0) have no tile to take the canvas from."""
    rc, _, log, _, _ = run(runner, [100, 100], 2, 0, widths=[MIN_W, 0], capacity=(MIN_W, 64), resume_with=(MIN_W, MIN_H))
    assert rc == -5 and b"not packed" in runner[1].tmc2_gof_last_error()          # TMC2_E_STATE
    assert not [e for e in log if e[0] >= GEOMETRY]


def test_the_slot_threads_are_kept_between_passes(runner):
    """A pass leases its slot threads from the library's set and hands them back: the threads of the second pass are those of the first
    (rounds 4-5: 2 x slots thread births per pass), and a slot's frames still share one thread per phase."""
    _, _, log1, _, _ = run(runner, [100] * 6, 3, 0)
    _, _, log2, _, _ = run(runner, [100] * 6, 3, 0)
    t1, t2 = {e[4] for e in log1 if e[0] == SEGMENT}, {e[4] for e in log2 if e[0] == SEGMENT}
    assert len(t1) == 3 and t1 == t2
    assert {e[4] for e in log1 if e[0] == GEOMETRY} == t1


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
    d, rank, world, frames_per_rank, heights, packing, fail, opt = args
    os.environ["TMC2_RCCL_LIBRARY"] = os.path.join(d, "libmockrccl.so")
    os.environ["MOCK_RCCL_DIR"] = d
    os.environ.update(opt.get("env", {}))
    if rank == 0:
        os.environ.update(opt.get("env0", {}))
    M = C.CDLL(os.path.join(d, "libtmc2hipmock.so"), mode=C.RTLD_GLOBAL)
    G = C.CDLL(os.path.join(d, "libtmc2gofmock.so"))
    M.mock_frame.restype = C.c_void_p
    M.mock_ctx.restype = C.c_void_p
    G.tmc2_gof_last_error.restype = C.c_char_p
    ctx = C.c_void_p(M.mock_ctx(rank))
    comm = C.c_void_p()
    if rank == 0 and opt.get("rank0_late"):
        import time
        time.sleep(opt["rank0_late"])
    rc = G.tmc2_gof_comm_create(rank, world, ctx, os.path.join(d, "id").encode(), C.byref(comm))
    if rc != 0:
        return {"rc": rc, "err": G.tmc2_gof_last_error().decode()}
    if rank in opt.get("absent", ()):                                 # a rank that dies after the communicator is up
        return {"rc": None}
    n = frames_per_rank
    ids = [rank + i * world for i in range(n)]                       # frame f of the GOF on rank f mod world
    widths = opt.get("widths") or [MIN_W] * (world * n)
    frames = [M.mock_frame(f, int(heights[f]), int(widths[f]), int(fail[1]) if fail and fail[0] == f else 0) for f in ids]
    handles = (C.c_void_p * n)(*frames)
    slot_of = (C.c_int32 * n)(*[i % 2 for i in range(n)])
    cfg = Config(7, 4, 11, 4, MIN_W, MIN_H, packing, 0)
    W, H = C.c_int32(0), C.c_int32(0)
    slots_rec = 16
    gathered = (_Patch * (world * n * slots_rec))()
    counts = (C.c_int64 * (world * n))()
    cap = opt.get("capacity", (1 << 20, 1 << 20))
    # (buffers only where the test expects the pass to be refused for their size: the mock's getter writes W x H bytes)
    occ = [np.zeros(16, np.uint8) for _ in range(n)] if "capacity" in opt else None
    M.mock_log_clear()
    call = lambda fn, cw, ch, bufs: fn(comm, handles, slot_of, n, 2, C.byref(cfg), bufs, None, None, None, None, None, cw, ch,
                                       C.byref(W), C.byref(H), slots_rec, gathered, counts)
    rc = call(G.tmc2_gof_encode_sharded, cap[0], cap[1], None if occ is None else (C.c_void_p * n)(*[a.ctypes.data for a in occ]))
    first = None
    if opt.get("resume"):
        first = {"rc": rc, "size": (W.value, H.value), "err": G.tmc2_gof_last_error().decode(), "calls": M.mock_log_size()}
        occ = [np.zeros(W.value * H.value + 64, np.uint8) for _ in range(n)]
        M.mock_log_clear()
        rc = call(G.tmc2_gof_encode_sharded_resume, W.value, H.value, (C.c_void_p * n)(*[a.ctypes.data for a in occ]))
    err = G.tmc2_gof_last_error().decode()
    again = None
    if opt.get("again"):                                              # a second pass on the same communicator
        again = call(G.tmc2_gof_encode_sharded, cap[0], cap[1], None), G.tmc2_gof_last_error().decode()
    log = []
    for i in range(M.mock_log_size()):
        c_, frame, a, b, th = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
        M.mock_log_get(i, C.byref(c_), C.byref(frame), C.byref(a), C.byref(b), C.byref(th))
        log.append((c_.value, frame.value, a.value, b.value))
    G.tmc2_gof_comm_destroy(comm)
    recs = [[(gathered[(s * slots_rec) + k].u0, gathered[(s * slots_rec) + k].v0) for k in range(int(counts[s]))] for s in range(world * n)]
    with open(os.path.join(d, "log_%d" % rank)) as f:
        rccl = f.read().split("\n")
    return {"rc": rc, "err": err, "size": (W.value, H.value), "log": log, "counts": list(counts), "records": recs, "rccl": [x for x in rccl if x],
            "first": first, "again": again, "occ": None if occ is None else [int(o[:max(1, W.value * H.value)].min()) for o in occ]}


@pytest.fixture(scope="module")
def sharded_libs(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("mock_gof_sharded"))
    inc = os.path.join(ROOT, "include")
    flags = ["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-fPIC", "-shared", "-I" + inc] + SANITIZE
    subprocess.run(flags + [os.path.join(ROOT, "tests", "mock", "mock_tmc2hip.cpp"), "-o", os.path.join(d, "libtmc2hipmock.so"), "-pthread"], check=True)
    subprocess.run(flags + [os.path.join(ROOT, "tests", "mock", "mock_rccl.cpp"), "-o", os.path.join(d, "libmockrccl.so")], check=True)
    subprocess.run(flags + [os.path.join(ROOT, "mpeg-pcc-tmc2_amd", "host", "gof_runner.cpp"), "-o", os.path.join(d, "libtmc2gofmock.so"),
                            "-L" + d, "-ltmc2hipmock", "-Wl,-rpath," + d, "-pthread", "-ldl"], check=True)
    return d


def _run_world(d, world, frames_per_rank, heights, packing=0, fail=None, plant=None, **opt):
    import multiprocessing as mp
    for name in os.listdir(d):                                         # (files of an earlier world)
        if name.startswith(("log_", "p2p_", "ar_", "bcast_", "id")):
            os.unlink(os.path.join(d, name))
    if plant:
        plant(os.path.join(d, "id"))
    with mp.get_context("spawn").Pool(world) as pool:
        return pool.map(_sharded_rank, [(d, r, world, frames_per_rank, heights, packing, fail, opt) for r in range(world)])


def _per_pass(x):
    calls = [c.split()[0] for c in x["rccl"]]
    return calls[calls.index("allreduce") + 1:-1]                       # (after create's pre-flight all-reduce, before destroy)


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
        per_pass = _per_pass(x)
        assert per_pass.count("broadcast") == 1 and per_pass.count("allreduce") == 1 and per_pass.count("group") == 1, x["rccl"]
        assert per_pass.count("send") == 1 and per_pass.count("recv") == (world if r == 0 else 0), x["rccl"]
        assert "broadcast 24 root 0" in x["rccl"] and "allreduce 4 op 2" in x["rccl"]
    got = res[0]
    for r in range(world):
        for i in range(n):
            f = r + i * world
            assert got["counts"][r * n + i] == 3 + f % 5
            assert got["records"][r * n + i] == [(f, k) for k in range(3 + f % 5)], (f, got["records"][r * n + i])


@pytest.mark.parametrize("world,packing", [(1, 2), (2, 1), (2, 2), (4, 2)])
def test_sharded_gof_runs_the_packing_chains_on_rank_0_over_the_records(sharded_libs, world, packing):
    """The low-delay / random-access conditions (BASELINE config 4) through tmc2_gof_encode_sharded: the chains run over ALL frames of
    the GOF in frame order, so every rank sends the records and block-occupancy pools of its frames to rank 0 (an all-reduce sizes
    the blocks, one grouped send / receive moves them), rank 0 runs PCCEncoder::placeSegments ONCE over the GOF's records in frame
    order, the canvas goes out in a 32-byte broadcast and every rank gets the packed lists of ITS frames back (a second group) and
    installs them before it rasterises; rank 0 keeps every frame's records in list order, so the pass ends with an all-reduce of its
    status instead of a gather.  The mock's placeSegments refuses frames out of order or with another frame's pool, its
    tmc2_frame_set_packing a list that is not the frame's own.  (World 1: the same route, forced by TMC2_GOF_RECORDS_CHAIN.)"""
    n = 3
    heights = [64 + 8 * f for f in range(world * n)]
    heights[world * n - 2] = 300                                        # the tallest tile: not on rank 0 (unless the world is one rank)
    widths = [MIN_W] * (world * n)
    widths[1] = 192                                                     # and one frame needs a wider tile
    res = _run_world(sharded_libs, world, n, heights, packing=packing, widths=widths, env={"TMC2_GOF_RECORDS_CHAIN": "1"})
    size = (192, 320)                                                   # 300 rows (+ 16 under random access) -> 320
    for r, x in enumerate(res):
        assert x["rc"] == 0, (r, x["err"])
        assert x["size"] == size, x["size"]
        place = [e for e in x["log"] if e[0] == PLACE]
        assert place == ([(PLACE, -1, world * n, packing)] if r == 0 else []), place
        assert not [e for e in x["log"] if e[0] in (PACK_FLEXIBLE, PACK_CHAIN, GPA)]            # (the frame-based packers are not the route)
        for i in range(n):
            f = r + i * world
            mine = [e for e in x["log"] if e[1] == f and e[0] != WEIGHT]
            assert [e[0] for e in mine] == [RESET, SEGMENT, SET_PACKING, GEOMETRY, ATTRIBUTE], (f, mine)
            assert mine[2][2:] == (3 + f % 5, heights[f] + (16 if packing == 2 else 0)) and mine[3][2:] == size
        per_pass = _per_pass(x)
        assert per_pass.count("broadcast") == 2 and per_pass.count("allreduce") == 2 and per_pass.count("group") == 2, x["rccl"]
        assert per_pass.count("send") == (1 + world if r == 0 else 1) and per_pass.count("recv") == (1 + world if r == 0 else 1), x["rccl"]
        assert "broadcast 24 root 0" in x["rccl"] and "broadcast 32 root 0" in x["rccl"] and "allreduce 12 op 2" in x["rccl"]
    got = res[0]
    for r in range(world):
        for i in range(n):
            f = r + i * world
            assert got["counts"][r * n + i] == 3 + f % 5
            assert got["records"][r * n + i] == [(f, k) for k in range(3 + f % 5)], (f, got["records"][r * n + i])


@pytest.mark.parametrize("frame,call", [(0, WEIGHT), (3, SEGMENT), (1, GEOMETRY), (2, ATTRIBUTE), (1, RESET), (0, RESET)])
def test_sharded_gof_a_failing_rank_does_not_leave_the_others_waiting(sharded_libs, frame, call):
    """A call that fails on ONE rank (a reset; S0 on rank 0; a frame's segmentation before the rendezvous; a frame's images after
    it): every rank still goes through the same collectives -- the failed rank with a value that says so (negative weights, a height
    no canvas has, a negative record count) -- and every rank returns with an error instead of waiting in a receive for a rank that
    has left.  The rank that failed reports the library's message, the others that another rank failed."""
    world, n = 2, 2
    res = _run_world(sharded_libs, world, n, [64] * (world * n), fail=(frame, call))      # (returns at all: nobody hangs)
    owner = frame % world
    assert res[owner]["rc"] != 0 and "mock failure of call %d on frame %d" % (call, frame) in res[owner]["err"], res[owner]
    assert res[0]["rc"] != 0                                           # the rank that gathers always knows
    other = res[1 - owner]
    if call in (WEIGHT, SEGMENT, RESET) or owner != 0:                 # before the rendezvous everybody learns it; after it, rank 0 does
        assert other["rc"] != 0 and "rank" in other["err"] and "its own call says why" in other["err"], other
    else:                                                              # (rank 0 failed after the rendezvous: rank 1 delivered its records)
        assert other["rc"] == 0, other


@pytest.mark.parametrize("frame,call", [(3, SEGMENT), (1, SET_PACKING), (2, GEOMETRY), (None, PLACE)])
def test_sharded_chained_gof_a_failing_rank_does_not_leave_the_others_waiting(sharded_libs, frame, call):
    """The same under the random-access condition: a frame that fails before the chain (every rank learns it from the all-reduce that
    sizes the blocks), the chain itself failing on rank 0 (from the header broadcast), a rank that cannot install or rasterise its
    packed lists (from the status all-reduce that ends the pass).  Everybody returns, everybody with an error."""
    world, n = 2, 2
    if call == PLACE:
        res = _run_world(sharded_libs, world, n, [64] * (world * n), packing=2, env0={"MOCK_PLACE_FAIL": "1"})
        assert res[0]["rc"] != 0 and "mock failure of the packing chain" in res[0]["err"], res[0]
        assert res[1]["rc"] != 0 and "the packing chain failed on rank 0" in res[1]["err"], res[1]
        assert not [e for x in res for e in x["log"] if e[0] >= GEOMETRY and e[0] != PLACE]
        return
    res = _run_world(sharded_libs, world, n, [64] * (world * n), packing=2, fail=(frame, call))
    owner = frame % world
    assert res[owner]["rc"] != 0 and "mock failure of call %d on frame %d" % (call, frame) in res[owner]["err"], res[owner]
    other = res[1 - owner]
    assert other["rc"] != 0 and "another rank" in other["err"] and "its own call says why" in other["err"], other
    if call == SEGMENT:
        assert not [e for x in res for e in x["log"] if e[0] in (PLACE, SET_PACKING, GEOMETRY)]


@pytest.mark.parametrize("packing", [0, 2])
def test_sharded_gof_resumes_a_pass_that_outgrew_the_buffers(sharded_libs, packing):
    """Every rank is refused at the rendezvous with the size the GOF needs (the same everywhere) and resumes with buffers of that
    size: nothing before the canvas is repeated, the tiles meet in one all-reduce, the records are gathered to rank 0."""
    world, n = 2, 2
    heights = [64, 200, 90, 64]
    res = _run_world(sharded_libs, world, n, heights, packing=packing, capacity=(MIN_W, MIN_H), resume=True)
    for r, x in enumerate(res):
        assert x["first"]["rc"] == -3 and x["first"]["size"] == (MIN_W, 256) and "needs a 128 x 256 canvas" in x["first"]["err"], x["first"]
        assert x["rc"] == 0 and x["size"] == (MIN_W, 256), x
        assert not {e[0] for e in x["log"]} & {RESET, WEIGHT, SEGMENT, PACK_FLEXIBLE, PACK_CHAIN, GPA, PLACE, SET_PACKING}, x["log"]
        assert sorted(e[1] for e in x["log"] if e[0] == GEOMETRY) == [r + i * world for i in range(n)]
        assert x["occ"] == [1 + r + i * world for i in range(n)]
    got = res[0]
    assert got["records"] == [[(r + i * world, k) for k in range(3 + (r + i * world) % 5)] for r in range(world) for i in range(n)]


def test_a_rank_that_never_arrives_ends_the_wait_after_the_timeout(sharded_libs):
    """Rank 1 dies after the communicator is up.  Rank 0's pass waits in the height all-reduce -- for TMC2_GOF_COLLECTIVE_TIMEOUT
    seconds: then the watchdog aborts the communicator (ncclCommAbort from its own thread), the wait ends, the call fails with
    TMC2_E_STATE and a message that says what happened, and the communicator refuses every later call."""
    import time
    t0 = time.time()
    res = _run_world(sharded_libs, 2, 2, [64] * 4, absent=(1,), env={"TMC2_GOF_COLLECTIVE_TIMEOUT": "1.5"}, again=True)
    assert time.time() - t0 < 30
    x = res[0]
    assert x["rc"] == -5 and "no answer from the other ranks" in x["err"] and "aborted" in x["err"], x
    assert "abort" in x["rccl"]
    assert x["again"][0] == -5 and "make a new one" in x["again"][1], x["again"]


def _plant_stale(path):
    with open(path, "wb") as f:                                         # what a crashed run of the same port left: a complete file, another id
        f.write(b"tmc2id01" + b"stale-id".ljust(128, b"\0"))
    os.utime(path, (1_000_000_000, 1_000_000_000))


def test_a_stale_id_file_is_not_taken_for_the_new_one(sharded_libs):
    """A complete id file from an earlier run sits under the rendezvous name and rank 0 is late: rank 1 must not join the old world
    (the mock's ncclCommInitRank refuses any id but its own) -- the file is older than rank 1's process, so it waits for the one
    rank 0 writes after removing the old."""
    res = _run_world(sharded_libs, 2, 1, [64, 64], plant=_plant_stale, rank0_late=1.0)
    assert [x["rc"] for x in res] == [0, 0], res


def test_the_id_file_is_created_exclusively_and_follows_no_link(sharded_libs, tmp_path):
    """The rendezvous name is a symbolic link to somebody's file: rank 0 removes the LINK and creates its own file; the target is
    untouched (rounds 5's std::ofstream would have written the id through the link)."""
    target = tmp_path / "precious"
    target.write_bytes(b"do not touch")

    def plant(path):
        os.symlink(str(target), path)
        os.symlink(str(target), path + ".tmp.0")
    res = _run_world(sharded_libs, 2, 1, [64, 64], plant=plant)
    assert [x["rc"] for x in res] == [0, 0], res
    assert target.read_bytes() == b"do not touch"


def test_several_ranks_without_a_rendezvous_name_or_a_port_are_refused(runner, monkeypatch):
    """No file named and no MASTER_PORT: /dev/shm/tmc2_gof_id_0 would be every such job's file.  Refused before anything waits."""
    M, G = runner
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setenv("TMC2_RCCL_LIBRARY", "/nonexistent/librccl.so")
    M.mock_ctx.restype = C.c_void_p
    comm = C.c_void_p()
    rc = G.tmc2_gof_comm_create(1, 2, C.c_void_p(M.mock_ctx(0)), None, C.byref(comm))
    msg = G.tmc2_gof_last_error().decode()
    assert rc != 0 and not comm.value
    assert "MASTER_PORT" in msg or "RCCL not available" in msg
