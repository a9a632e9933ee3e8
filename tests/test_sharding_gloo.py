"""CPU tier: the multi-GPU sharding logic (frame -> rank, axis-weight broadcast, canvas-height all-reduce, gather of the
finished canvases to rank 0) under torch.distributed with the gloo backend, world_size 2.  The per-frame compute is
faked (this tier has no GPU); what is covered is exactly the code bench.py runs between the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tmc2_amd.gof import Sharder


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, frame_count, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = Sharder(rank, world, dist, "cpu")
    mine = sh.frames_of(frame_count)
    w = sh.broadcast_weight([0.75, 0.6, 1.0] if rank == 0 else [0.0, 0.0, 0.0])
    gof_h = sh.max_height([1280 + 16 * f for f in mine])               # frame f "packs" to height 1280 + 16 f
    canv = torch.stack([torch.full((2, 4, 4), f, dtype=torch.int16) for f in mine])   # fake canvases, value = frame id
    got = sh.gather(canv)
    sh.barrier()
    if rank == 0:
        order = [f for r in range(world) for f in sh.frames_of(frame_count, r)]
        q.put((mine, w.tolist(), gof_h, [int(t[i, 0, 0, 0]) for t in got for i in range(t.shape[0])], order))
    else:
        q.put((mine, w.tolist(), gof_h, None, None))
    dist.destroy_process_group()


def test_sharder_world2_gloo():
    world, frames = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = sorted(f for r in res for f in r[0])
    assert covered == list(range(frames))                               # every frame on exactly one rank
    for mine, w, h, gathered, order in res:
        assert w == [0.75, 0.6, 1.0]                                    # rank 0's weights everywhere
        assert h == 1280 + 16 * (frames - 1)                            # max over ALL frames, not just the local ones
        if gathered is not None:
            assert gathered == order and sorted(gathered) == list(range(frames))


def test_sharder_single_process_is_identity():
    sh = Sharder()
    assert sh.frames_of(5) == [0, 1, 2, 3, 4]
    assert sh.max_height([3, 9, 4]) == 9
    assert np.array_equal(sh.broadcast_weight([1, 2, 3]), np.array([1.0, 2.0, 3.0]))
    t = torch.zeros(3)
    assert sh.gather(t)[0] is t


def _gof_records(seed, frames):
    from test_host_logic import _random_patch_gof
    rng = np.random.default_rng(seed)
    return _random_patch_gof(rng, frames, 25, drift=6, churn=0.1)


def _pack_worker(rank, world, port, frame_count, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = Sharder(rank, world, dist, "cpu")
    gof = _gof_records(seed, frame_count)                              # the same GOF on every rank; each keeps its own frames
    local = [gof[f] for f in sh.frames_of(frame_count)]
    out = {}
    for mode in (1, 2):
        mine, tiles = sh.pack_gof_records(local, frame_count, mode, 512, 512)
        out[mode] = (sh.frames_of(frame_count), mine, tiles)
    # a chain rank 0 refuses (here: a canvas of width 0) raises on EVERY rank: nobody is left waiting in the collective
    try:
        sh.pack_gof_records(local, frame_count, 1, 0, 512)
        out["error"] = None
    except Exception as e:
        out["error"] = type(e).__name__
    sh.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_inter_frame_packers_over_sharded_frames_world2_gloo():
    """S10' with the frames of a GOF on two ranks: patch records gathered to rank 0, the chain / the global patch
    allocation run there, the packed lists scattered back -- against the same chain in one process."""
    import tmc2_amd as T
    world, frames, seed = 2, 6, 77
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pack_worker, args=(r, world, port, frames, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gof = _gof_records(seed, frames)
    for mode in (1, 2):
        exp = T.host_pack_gof_records(gof, mode, 512, 512)
        seen = []
        for rank in range(world):
            ids, mine, tiles = res[rank][mode]
            assert tiles == [(e[3], e[4]) for e in exp]                 # every rank knows the tile sizes of the whole GOF
            for f, got in zip(ids, mine):
                seen.append(f)
                for a, b in zip(got, exp[f]):
                    assert np.array_equal(a, b), (mode, f)
        assert sorted(seen) == list(range(frames))
        if mode == 2:
            assert any((e[2] >= 0).any() for e in exp)            # the GOF has tracked patches: the allocation did something
    assert res[0]["error"] == res[1]["error"] == "Tmc2Error"


def test_phase_a_records_route_orchestration_with_fake_frames():
    """GofEncoder's records route through S10' (segment -> records -> chain on plain records -> install -> rasterise) with
    the device calls faked: what is installed in every frame and the GOF canvas are those of the chain on plain records."""
    import tmc2_amd as T
    from tmc2_amd.gof import GofEncoder

    gof = _gof_records(5, 4)

    class FakeFrame:
        def __init__(self, rec, occ):
            self.rec, self.occ, self.installed, self.canvas = rec, occ, None, None

        def weight_normal(self, bits, thr):
            return np.array([1.0, 1.0, 1.0])

        def segmenter_compute(self, params):
            pass

        def get_patch_records(self):
            return self.rec, self.occ

        def set_packing(self, patch_list, matches, occupancy, w, h):
            self.installed = (patch_list, occupancy, matches, w, h)

        def encoder_generate_geometry_images(self, W, H, prec):
            self.canvas = (W, H, prec)

    class FakeEncoder:
        iterations, bits3d, min_w, min_h, occ_precision, vox_dim = 2, 10, 512, 512, 4, 4

        def _per_worker(self, frames, fn):
            return [fn(fr) for fr in frames]

        def per_frame(self, frames, fn):
            return [fn(fr, i) for i, fr in enumerate(frames)]

    for mode in (1, 2):
        frames = [FakeFrame(r, o) for r, o in gof]
        W, H = GofEncoder._phase_a_sharded_chain(FakeEncoder(), frames, Sharder(), None, mode, None)
        exp = T.host_pack_gof_records(gof, mode, 512, 512)
        for fr, e in zip(frames, exp):
            assert fr.canvas == (W, H, 4)
            for a, b in zip(fr.installed, e):
                assert np.array_equal(a, b)
        tile_h = max(e[4] for e in exp)
        assert (W, H) == T.encoder_canvas_size([max(tile_h, 512) if mode == 2 else tile_h], max([512] + [e[3] for e in exp]), 512, 512)


# ---- bench.py's own N > 1 code path, before the driver's 8-GPU run: gloo, faked frames, world 2 and 8 ----------------------
def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


class _FakeFrame:
    def __init__(self, gof_index):
        self.gof_index = gof_index


class _FakeEncoder:
    """What bench.gather_canvases needs of a GofEncoder: byte views of a frame's finished canvases (here: CPU tensors whose
    bytes name the frame and the canvas kind)."""
    SHAPES = {"geometry": (2, 8, 6, 2), "occ_video": (2, 3, 1), "attribute": (2, 3, 8, 6, 1)}

    def device_tensor(self, frame, name):
        kind = list(self.SHAPES).index(name)
        return torch.full(self.SHAPES[name], (frame.gof_index * 3 + kind) % 251, dtype=torch.uint8)


def _bench_gather_worker(rank, world, port, frame_count, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench = _load_bench()
    sh = Sharder(rank, world, dist, "cpu")
    frames = [_FakeFrame(f) for f in sh.frames_of(frame_count)]
    cache = {}
    for _ in range(2):                                                   # two steps: the second one reuses rank 0's buffers
        bench.gather_canvases(_FakeEncoder(), frames, sh, cache, pin=False)
    sh.barrier()
    if rank == 0:
        q.put({k: v.numpy().copy() for k, v in cache.items()})            # (plain arrays: nothing shared outlives the process)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_gather_branch_world_2_and_8_gloo(world):
    """The exact tail of bench.py's step() for N > 1 (bench.gather_canvases: one gather per canvas kind and frame slot to rank
    0, copies into its host buffers) with 32 frames over 2 and over 8 ranks: rank 0 ends up with every frame's canvases in
    the slot its (rank, index) names."""
    frames = 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_gather_worker, args=(r, world, port, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    per_rank = frames // world
    assert sorted(got) == sorted((i, name) for i in range(per_rank) for name in _FakeEncoder.SHAPES)
    for (slot, name), buf in got.items():
        assert tuple(buf.shape) == (world,) + _FakeEncoder.SHAPES[name]
        kind = list(_FakeEncoder.SHAPES).index(name)
        for r in range(world):
            gof_index = r + slot * world                                 # frame f lives on rank f % world, slot f // world
            assert int(buf[r].min()) == int(buf[r].max()) == (gof_index * 3 + kind) % 251, (slot, name, r)


def test_bench_host_slot_budget_per_rank():
    """--host-steps is a node budget; a rank never gets fewer slots than its in-flight frames need to progress side by side."""
    bench = _load_bench()
    assert bench.host_slots(16, 1, 16) == 16
    assert bench.host_slots(16, 2, 16) == 8
    assert bench.host_slots(16, 8, 4) == 4          # 16 // 8 = 2 would serialise four frames behind two slots
    assert bench.host_slots(16, 8, 1) == 2
    assert bench.host_slots(0, 8, 4) == 4


class _FakeRecordFrame:
    """What bench.gather_records needs of a frame: its packed patch list."""

    def __init__(self, gof_index):
        from tmc2_amd.lib import PATCH_DTYPE
        self.gof_index = gof_index
        self.records = np.zeros(3 + gof_index % 5, PATCH_DTYPE)
        self.records["u0"] = gof_index
        self.records["v0"] = np.arange(len(self.records))

    def get_patches(self):
        return (self.records, None, None, None)

    def get_patch_order(self):
        return np.arange(len(self.records))


class _FakeRecordEncoder:
    def per_frame(self, frames, fn):
        return [fn(fr, i) for i, fr in enumerate(frames)]


def _bench_records_worker(rank, world, port, frame_count, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench = _load_bench()
    sh = Sharder(rank, world, dist, "cpu")
    frames = [_FakeRecordFrame(f) for f in sh.frames_of(frame_count)]
    # every rank lands its "canvases" in its own shared host segment (no GPU here: not page-locked), rank 0 maps them all
    from tmc2_amd.lib import SharedHostArray
    seg = SharedHostArray("tmc2_test_%d_r%d" % (port, rank), 64 * len(frames), create=True, register=False)
    for i, fr in enumerate(frames):
        seg.array[64 * i:64 * (i + 1)] = fr.gof_index
    cache = {}
    for _ in range(2):
        bench.gather_records(_FakeRecordEncoder(), frames, sh, cache)
    sh.barrier()
    if rank == 0:
        seen = []
        for r in range(world):
            other = SharedHostArray("tmc2_test_%d_r%d" % (port, r), 64 * len(frames), create=False, register=False)
            seen.append(other.array.reshape(len(frames), 64)[:, 0].copy())
            other.close()
        q.put(([x.copy() for x in cache["records"]], seen))
    sh.barrier()
    seg.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_records_gather_and_shared_host_canvases_world_2_and_8_gloo(world):
    """bench.py --gather host (the default for N > 1): every rank copies its frames' canvases into its own shared host segment
    (tmc2_amd.lib.SharedHostArray; page-locked on a GPU box) that rank 0 maps, and the only collective of the tail is ONE gather
    of the per-frame patch records (bench.gather_records).  32 frames over 2 and 8 ranks: rank 0 sees every frame's records in
    the slot its (rank, index) names, and every rank's segment."""
    from tmc2_amd.lib import PATCH_DTYPE
    frames = 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_records_worker, args=(r, world, port, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    records, seen = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bench = _load_bench()
    size = PATCH_DTYPE.itemsize
    for r in range(world):
        assert seen[r].tolist() == [r + slot * world for slot in range(frames // world)]
        for slot in range(frames // world):
            f = r + slot * world
            row = records[r][slot]
            count = int(np.frombuffer(row[:8].tobytes(), np.int64)[0])
            assert count == 3 + f % 5
            recs = np.frombuffer(row[8:8 + count * size].tobytes(), PATCH_DTYPE)
            assert (recs["u0"] == f).all() and recs["v0"].tolist() == list(range(count))
    assert not [n for n in os.listdir("/dev/shm") if n.startswith("tmc2_test_%d_" % port)]
    assert bench.RECORD_SLOTS >= 512


def test_bench_gpus_n_starts_its_own_launcher():
    """`python bench.py --gpus 2` with no launcher around it (the way the N = 1 line is invoked) must not run one rank and print
    n_gpus 1: it becomes `torch.distributed.run --nproc-per-node 2` itself, the ranks that come up are the ranks asked for, and the
    collectives of the sharded GOF (24-byte broadcast, height all-reduce, record gather) run once as a pre-flight before any
    set-up.  --preflight-only stops there, so the route is covered on a box without a GPU (gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--preflight-only", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-600:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"preflight": "ok", "n_gpus": 2, "backend": "gloo"}
    assert "pre-flight ok: 2 ranks over gloo" in r.stderr


def test_bench_refuses_a_world_that_is_not_the_gpus_asked_for():
    """A launcher that started another number of ranks than --gpus says: refused before anything is set up (the line's n_gpus
    would not be what was asked for); and RCCL ranks without a GPU each are refused by the parent, before the launcher starts."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29573")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 2 rank(s)" in r.stderr, r.stderr[-400:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "--gpus 8 but 0 GPU(s) visible" in r.stderr, r.stderr[-400:]
