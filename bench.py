#!/usr/bin/env python
"""bench.py -- encoder patch-generation + image-generation throughput (frames/s) on a 32-frame GOF.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: under python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...,
     or bare: without WORLD_SIZE in the environment bench.py starts that launcher itself; either way the ranks that came up must
     be the N asked for, and the collectives of the sharded GOF run once as a pre-flight before the set-up)

One "step" = one pass of the hot path over one GOF (default, --config longdress: 32 synthetic longdress_vox10-like frames,
~0.84 M points each, CTC all-intra r3 flags: 50 refine iterations, occupancyPrecision 4, 1280x1280 minimum canvas; --config
loot | redandblack | soldier | basketball: the other BASELINE configurations with their own CTC parameters, tmc2_amd/configs.py).
Frames are sharded frame f -> rank f % N (strong scaling: the GOF is fixed); the only collectives are the
24-byte axis-weight broadcast, the canvas-height all-reduce(max) and one gather of the packed patch records to rank 0 (RCCL over
xGMI); every rank copies its frames' finished canvases into page-locked shared host memory over its own PCIe link (--gather
rccl: gathered to rank 0 first, round 3's route).  The point arrays are resident in HBM before the timed region starts;
everything from the k-d tree build to the finished, host-resident canvases is inside it.  "verified": every frame of the last
timed step against the unmodified reference's digests (tests/golden/full_size.npz); "decoder": the decoder side of the same
GOF (reconstruct + post-reconstruction tail + D1/D2 metric per frame), timed and verified separately.

Printed JSON (one line, rank 0): metric/value/unit as BASELINE.json, plus
  roofline     -- the dominant GPU step of the timed region, chosen over ALL timed stages by GPU time with the GPU to
                  itself (SURVEY.md 8d's S1 + S2 + S6 row -- the source tree build, the 16-NN lists and the normals, 94 N
                  bytes -- is one of the candidates, "tree_knn_normals"): algorithmic bytes per launch (SURVEY.md 8d
                  formulas, with the V and L of this very run) / mean launch duration (HIP events on the launching
                  stream, inside the timed region) against 8 TB/s HBM;
                  "path" = the contract figure of the whole path, B_alg per frame x frames/s / peak; "stages" = achieved
                  GB/s of every stage whose algorithmic bytes are defined
  first_gof_ms / untimed_pass_ms / pool / orientation -- the first pass of this process over the GOF as it is (no priming
                  passes; the contexts reserve their worst case up front: pool.hipmalloc_calls), S3's ladder steps and
                  fallbacks per GOF
  cpu_baseline -- the unmodified reference (oracle/_ref, kind "reference") or our restatement (kind "port") on the host
                  cores of this box: "value" = one frame, one thread (what MPEG's anchors use); "all_cores_value" = a GOF
                  through the reference's own TBB path (ENABLE_TBB build with its vendored TBB: frames in parallel, points
                  and voxels in parallel inside a frame), --nbThread = the physical cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.environ.get("TMC2_PACKAGE_DIR") or os.path.join(ROOT, "mpeg-pcc-tmc2_amd"))  # (TMC2_PACKAGE_DIR: tools/asan_gpu.sh)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prime", type=int, default=0, help="untimed passes of set-up before the W warmup steps (rounds 3-4 needed three: "
                    "the contexts' pools sized their buffers by first use, and a first-use hipMalloc synchronises the device; round "
                    "5 reserves the sequence's worst case at context creation, tmc2_ctx_reserve, and reports the first passes as "
                    "first_gof_ms)")
    ap.add_argument("--reserve", type=int, default=1, help="0: no tmc2_ctx_reserve (the pools grow by hipMalloc on first use)")
    ap.add_argument("--config", default="longdress", help="BASELINE configuration: a short name (%s) or a case of "
                    "tmc2_amd/configs.py FULL_SIZE_CASES -- workload, frames, refine iterations / voxel size, bit depth, occupancy "
                    "precision, minimum canvas and packing condition come from the CTC table there, and the timed step is checked "
                    "against that case's digests of the unmodified reference" % ", ".join(sorted(_bench_configs())))
    ap.add_argument("--frames", type=int, default=None, help="frames per GOF (default: the configuration's)")
    ap.add_argument("--workload", default=None, help="synthetic sequence (default: the configuration's)")
    ap.add_argument("--workers", type=int, default=0, help="concurrent frames per GPU (0 = auto)")
    ap.add_argument("--host-steps", type=int, default=16, help="max concurrent host-resident steps (tree build, orientation)")
    ap.add_argument("--kdtree", default="auto", choices=["auto", "device", "host", "adaptive"],
                    help="where the k-d trees are built (auto = device)")
    ap.add_argument("--iterations", type=int, default=None, help="iterationCountRefineSegmentation (default: the configuration's)")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 disables the CPU baseline leg; 2 runs it inside this process "
                    "(the round-3 form: two builds of the reference loaded next to the product library); 3: the bounded form -- one "
                    "frame on one thread and 8 frames through the reference's TBB path, no CLI run, no process-per-frame bound (what "
                    "the per-configuration lines of tools/gpu/final.sh use)")
    ap.add_argument("--ingest", type=int, default=1, help="0 skips the (untimed) PLY ingest measurement")
    ap.add_argument("--tail", type=int, default=1, help="0 skips the (untimed) post-reconstruction tail measurement")
    ap.add_argument("--decoder", type=int, default=1, help="0 skips the decoder-side GOF leg (BASELINE config 5: reconstruct + "
                    "tail + D1/D2 metric for every frame, outside the headline metric)")
    ap.add_argument("--gen-procs", type=int, default=0, help="processes for synthetic data generation (1 = in-process; "
                    "use 1 under rocprofv3, whose signal handler deadlocks multiprocessing pools)")
    ap.add_argument("--rendezvous", default="phases", choices=["one", "phases", "four"],
                    help="all-intra: phases = the frames of the GOF meet once, after segmentation + packing (the common canvas "
                         "size), and go on independently from there (default); four = they also meet after phase A, after phase "
                         "B and after the copies (round 2); one = every frame runs its whole chain on a guessed canvas size, "
                         "the GOF meets at the end (GofEncoder.encode_all_intra: frames in different phases share the chip "
                         "worse -- measured 132 against 152 frames/s)")
    ap.add_argument("--host", default="auto", choices=["auto", "native", "python"],
                    help="who drives the C-ABI inside the timed region: native = one tmc2_gof_encode call per GOF (libtmc2gof.so: C++ "
                         "threads, include/tmc2gof.h; with several ranks tmc2_gof_encode_sharded: the all-intra GOF's collectives are "
                         "RCCL calls from C++), python = GofEncoder's worker threads.  auto: native wherever libtmc2gof.so loads -- one GPU or several ranks under RCCL, every packing condition (the "
                         "low-delay / random-access chains run on rank 0 over the gathered records); python only under --dist-backend "
                         "gloo (ranks sharing a GPU cannot form an RCCL world)")
    ap.add_argument("--pin", type=int, default=1, help="0: plain host buffers instead of page-locked ones for the canvases (for runs "
                    "under a sanitizer runtime, where torch's pinned allocator does not come up; slower copies)")
    ap.add_argument("--gather", default="host", choices=["host", "rccl"],
                    help="N > 1, where the finished canvases of a rank's frames go: host = straight into page-locked host memory of "
                         "the node from the GPU that made them (a shared-memory segment per rank, mapped by rank 0: 8 PCIe links "
                         "side by side, nothing crosses xGMI but the per-frame patch records, gathered to rank 0 over RCCL); rccl = "
                         "gathered to rank 0's HBM first and copied out from there (round 3: ~ 550 MB per GOF through one GPU and one link)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="N > 1: nccl = RCCL, one GPU per rank; gloo = control plane on the CPU and rank r on GPU r mod the visible "
                         "ones (several ranks share a GPU: for trying the N > 1 path on a one-GPU box)")
    ap.add_argument("--cpu-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--preflight-only", type=int, default=0, help="1: bring the process group of --gpus N ranks up, run the pre-flight "
                    "collectives (broadcast, all-reduce, gather) and print {\"preflight\": \"ok\", \"n_gpus\": N}: no GPU work (with "
                    "--dist-backend gloo it runs on a box without a GPU)")
    ap.add_argument("--packing", default=None, choices=["all-intra", "low-delay", "random-access"],
                    help="S10 condition (default: the configuration's): every frame on its own, the spatial-consistency chain, "
                         "or the chain + global patch allocation (with several ranks the chain runs on rank 0 over the patch records)")
    a = ap.parse_args()
    from tmc2_amd import configs
    a.case_name = configs.BENCH_CONFIGS.get(a.config, a.config)
    if a.case_name not in configs.FULL_SIZE_CASES:
        ap.error("unknown --config %s (known: %s)" % (a.config, ", ".join(sorted(configs.BENCH_CONFIGS) + sorted(configs.FULL_SIZE_CASES))))
    a.case = dict(configs.FULL_SIZE_CASES[a.case_name], name=a.case_name)
    # explicit flags override the table (such a run has no reference fixture: "verified" is null)
    a.workload = a.workload or a.case["workload"]
    a.frames = a.frames or a.case["frames"]
    a.iterations = a.iterations or a.case["iterations"]
    a.packing = a.packing or configs.PACKING_NAME[a.case["pack"]]
    a.is_case = (a.workload, a.frames, a.iterations, a.packing) == (a.case["workload"], a.case["frames"], a.case["iterations"],
                                                                    configs.PACKING_NAME[a.case["pack"]])
    return a


def _bench_configs():
    from tmc2_amd import configs
    return configs.BENCH_CONFIGS


def _gen(arg):
    from tmc2_amd.synth import synth_cloud
    return synth_cloud(arg[0], arg[1])


def stream_copy_ceiling(torch, device, nbytes=1 << 30, repeats=10):
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    src.fill_(1)
    for _ in range(2):
        dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(repeats):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / repeats
    del src, dst
    return {"GB/s": round(2.0 * nbytes / (ms * 1e-3) / 1e9, 1), "what": "device-to-device copy of %d MiB, read + written bytes" % (nbytes >> 20)}


def under_profiler():
    """rocprofv3 injects its tool library into every child; its signal handler deadlocks multiprocessing pools and would
    trace the CPU baseline's subprocesses: side legs that start processes are left out under it."""
    return any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))


def make_frames(workload, indices, gen_procs=0):
    return start_frames(workload, indices, gen_procs)()


def start_frames(workload, indices, gen_procs=0):
    """Starts the synthetic frames' generation in forked processes (BEFORE any GPU context exists: fork-safe) and returns the
    callable that waits for them -- the rendezvous pre-flight of a sharded run goes in between."""
    import multiprocessing as mp
    if under_profiler():
        gen_procs = 1
    procs = gen_procs or max(1, min(len(indices), (os.cpu_count() or 8) // max(1, int(os.environ.get("WORLD_SIZE", "1"))), 16))
    if procs == 1 or len(indices) < 2:
        return lambda: [_gen((workload, i)) for i in indices]
    pool = mp.get_context("fork").Pool(procs)
    pending = pool.map_async(_gen, [(workload, i) for i in indices])

    def finish():
        try:
            return pending.get()
        finally:
            pool.close()
            pool.join()
    return finish


def relaunch_under_launcher(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>`
    -- one rank per GPU, the frame loop of PCCEncoder.cpp:4729-4750 sharded f -> rank f mod N.  Refuses, before anything is set up,
    when the node has fewer GPUs than ranks (unless --dist-backend gloo, where ranks share the visible GPUs)."""
    import socket
    if a.dist_backend == "nccl" and not a.preflight_only:
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible (RCCL needs one GPU per rank; --dist-backend gloo shares them)" % (a.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: %s\n" % (a.gpus, " ".join(cmd[1:9])))
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def preflight(a, torch, dist, sharder, rank, world):
    """Before the set-up (tens of seconds of data generation, uploads and priming passes): the process group is up, and every
    collective the sharded GOF uses -- the 24-byte weight broadcast, the height all-reduce(max), one gather of a dummy patch
    record block to rank 0 -- has run once over the backend the timed region will use.  A RCCL / rendezvous problem surfaces
    here, with the rank that saw it named."""
    import numpy as np
    t0 = time.time()
    try:
        w = sharder.broadcast_weight(np.array([1.0, 2.0, 3.0]) if rank == 0 else np.zeros(3))
        assert w.tolist() == [1.0, 2.0, 3.0], w
        assert sharder.max_height([100 + rank]) == 100 + world - 1
        rec = torch.full((2, 8 + 64), rank, dtype=torch.uint8).to(sharder.device)
        got = sharder.gather(rec)
        if rank == 0:
            assert [int(g[0, 0].item()) for g in got] == list(range(world))
        sharder.barrier()
    except Exception as e:
        raise SystemExit("bench.py: pre-flight of the %s process group failed on rank %d of %d: %r" % (a.dist_backend, rank, world, e))
    if rank == 0:
        sys.stderr.write("bench.py: pre-flight ok: %d ranks over %s (%s), broadcast + all-reduce + gather in %.2f s\n" %
                         (world, "RCCL" if a.dist_backend == "nccl" else "gloo", sharder.device, time.time() - t0))
        sys.stderr.flush()


# Algorithmic HBM bytes per frame of each timed stage (N points, M reconstructed points, V refinement
# voxels, L mean neighbourhood row, A canvas pixels, p occupancy precision).  A stage that runs several times per frame
# (the refine sweeps: I times) is quoted per run.
def algorithmic_bytes(stage, N, M=0, V=0, L=0.0, A=0, p=4):
    table = {
        "knn_self": (8 + 64) * N,                       # point in, 16 x 4 B neighbour ids out (S1 + S2 + S6 share the lists)
        "normals": (64 + 24) * N,                       # neighbour ids in, fp64 normal out
        "initial_segmentation": (24 + 1) * N,
        "refine_setup": 14 * N + 4 * V * L,             # voxel keys + radius adjacency
        "refine_sweep": 4 * V * L + 24 * V + 26 * N,    # one sweep: adjacency, histograms r/w, normals + partition
        "k:ccMutualMask": (64 + 2) * N,
        "k:ccUnion": (64 + 2 + 1 + 1 + 4) * N,
        "k:ccRelax": (64 + 2 + 1 + 1 + 4) * N,
        "knn8_recon_in_source": (8 + 64) * M,           # per reconstructed point: point in, 8 ids + 8 distances out
        "knn1_source_in_recon": (8 + 8) * N,
        "geometry_images": A * (1 + 1.0 / (p * p)) + 12 * A,
        "reconstruct": 5 * A + 16 * M,
        "attribute_images": 6 * A + 16 * A,             # scatter + push-pull + group dilation
    }
    return float(table.get(stage, 0))


def path_bytes(N, M, V, L, I, A, p):
    """B_alg of the whole path S0-S22 per frame, with this run's V, L, canvas."""
    return (94 * N + 25 * N + 14 * N + 4 * V * L + I * (4 * V * L + 24 * V + 26 * N) + 102 * N
            + A * (1 + 1.0 / (p * p)) + 12 * A + 5 * A + 16 * M + 13 * N + 45 * M + 6 * A + 16 * A)


def physical_cores():
    try:
        seen = set()
        for cpu in os.sched_getaffinity(0):
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
                seen.add(f.read().strip())
        return max(1, len(seen))
    except (OSError, AttributeError):
        return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(workload, iterations, gof, case, bounded=False):
    """The same workload through the CPU checker (test infrastructure, used here only as the reported baseline): the
    unmodified reference if oracle/_ref travelled with the repo, else our restatement.  One frame on one thread; then a GOF
    through the reference's own TBB path on all physical cores (its ENABLE_TBB build with the vendored TBB)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    frames = gof[:1]
    if os.path.exists(ob.REF_PATH):
        eng, kind = ob.Reference(), "reference"
    else:
        eng, kind = ob.Oracle(), "port"
    args = (iterations, case["bits3d"], case["precision"], case["min_w"], case["min_h"], case["pack"], case["vox_dim"])
    t = time.time()
    a = eng.phase_a(frames, *args)
    eng.phase_b(frames, a, case["precision"])
    dt = time.time() - t
    res = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": 1, "kind": kind,
           "sample": "1 frame of %s (%d points), stages S0-S22 (patch generation + occupancy/geometry/attribute images), "
                     "1 thread, %.1f s" % (workload, len(frames[0][0]), dt)}
    if under_profiler() or not os.path.exists(ob.REF_TBB_PATH):
        return res
    try:                                                       # a side figure: never lose the line over it
        cores = physical_cores()
        avail_gb = 8.0
        with open("/proc/meminfo") as f:
            avail_gb = [int(l.split()[1]) for l in f if l.startswith("MemAvailable")][0] / 1048576.0
        scale = max(1.0, len(gof[0][0]) / 0.84e6)                              # (memory per frame follows the point count)
        nfr = int(max(1, min(len(gof), 8 if bounded else 16, avail_gb * 0.5 / (1.2 * scale))))   # bounded sample; ~1.2 GB per frame inside the reference's containers
        gof = gof[:nfr]                                                      # (the frames of the timed GOF, generated already)
        eng = ob.Reference(tbb=True, nb_thread=cores)
        t = time.time()
        a = eng.phase_a(gof, *args)
        eng.phase_b(gof, a, case["precision"])
        wall = time.time() - t
        res["all_cores_value"] = round(nfr / wall, 5)
        res["all_cores"] = cores
        res["all_cores_sample"] = ("%d frames of the GOF through the reference's own TBB path (ENABLE_TBB build, vendored TBB, "
                                   "--nbThread=%d = physical cores of this host: frames in parallel, points / voxels in parallel "
                                   "inside a frame), same stages, %.1f s wall" % (nfr, cores, wall))
    except Exception as e:
        res["all_cores_error"] = repr(e)
    if bounded:                                                # (--cpu-baseline 3: one thread + the TBB path, nothing else)
        return res
    try:                                                       # the CLI's wall time next to the stage sum
        res["cli"] = cpu_baseline_cli(frames[0], iterations, dt)
    except Exception as e:
        res["cli"] = {"error": repr(e)}
    try:
        fp = cpu_baseline_frame_processes(workload, iterations, dt, case)
        res["frame_processes_value"], res["frame_processes"], res["frame_processes_sample"] = fp["value"], fp["cores"], fp["sample"]
    except Exception as e:
        res["frame_processes_error"] = repr(e)
    return res


CLI_STUB = """#!/bin/sh
# identity "video codec" behind the reference's HMAPP wrapper (PCCHMAppVideoEncoder.cpp:59-90): reconstruction = input
for a in "$@"; do case $a in --InputFile=*) IN=${a#*=};; --ReconFile=*) REC=${a#*=};; --BitstreamFile=*) BIN=${a#*=};; esac; done
cp "$IN" "$REC"; printf '\\000\\000\\000\\001\\100\\001\\014\\001' > "$BIN"
"""


def cpu_baseline_cli(frame, iterations, stage_seconds):
    """The reference's own command-line encoder (oracle/_ref/PccAppEncoder, built from the unmodified sources) on ONE frame of
    the workload, whole encode() with an identity "video codec" behind its HMAPP wrapper, one thread: wall time end to end
    (PLY read, the path, colour conversion, the post-reconstruction tail, bitstream) next to the time of the path's stages
    alone.  CTC parameters from tests/golden/cli_ctc_args.json (the cfg files of the reference flattened into options: the
    reference tree does not exist on the GPU box)."""
    import subprocess
    import tempfile
    import tmc2_amd as T
    app = os.path.join(ROOT, "oracle", "_ref", "PccAppEncoder")
    if not os.path.exists(app):
        return {"error": "oracle/_ref/PccAppEncoder not built"}
    with open(os.path.join(ROOT, "tests", "golden", "cli_ctc_args.json")) as f:
        args = json.load(f)["args"]
    with tempfile.TemporaryDirectory() as d:
        stub = os.path.join(d, "stub.sh")
        with open(stub, "w") as f:
            f.write(CLI_STUB)
        os.chmod(stub, 0o755)
        T.ply_write(os.path.join(d, "in_0000.ply"), frame[0], frame[1], None, ascii=True)     # (the 8i content ships as ASCII)
        cmd = [app] + args + ["--uncompressedDataPath=" + os.path.join(d, "in_%04d.ply"), "--startFrameNumber=0", "--frameCount=1",
                              "--groupOfFramesSize=1", "--iterationCountRefineSegmentation=%d" % iterations,
                              "--videoEncoderOccupancyCodecId=HMAPP", "--videoEncoderGeometryCodecId=HMAPP",
                              "--videoEncoderAttributeCodecId=HMAPP", "--videoEncoderOccupancyPath=" + stub,
                              "--videoEncoderGeometryPath=" + stub, "--videoEncoderAttributePath=" + stub, "--nbThread=1",
                              "--computeMetrics=0", "--computeChecksum=0", "--compressedStreamPath=" + os.path.join(d, "S.bin")]
        t = time.time()
        r = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        wall = time.time() - t
        if r.returncode != 0 or not os.path.exists(os.path.join(d, "S.bin")):
            return {"error": "PccAppEncoder exited with %d" % r.returncode, "tail": r.stdout.decode(errors="replace")[-300:]}
    return {"wall_s": round(wall, 2), "path_stages_s": round(stage_seconds, 2), "path_share": round(stage_seconds / wall, 3),
            "what": "PccAppEncoder (unmodified reference, ENABLE_TBB build, --nbThread=1), 1 frame, whole encode() with an identity "
                    "video codec: end-to-end wall time; path_stages_s = the same frame through the path's stages alone"}


def cpu_baseline_frame_processes(workload, iterations, one_frame_seconds, case):
    """An upper bound on what frame-level parallelism alone can give the reference on this host (its own TBB path also runs
    the frames of a GOF side by side, PCCEncoder.cpp:4729-4750, but serialises parts of the path): the serial build, one
    PROCESS per frame, P different frames at once, frames/s = P / wall time from a common start."""
    import subprocess
    import tempfile
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    avail_gb = 8.0
    try:
        with open("/proc/meminfo") as f:
            avail_gb = [int(l.split()[1]) for l in f if l.startswith("MemAvailable")][0] / 1048576.0
    except Exception:
        pass
    procs = int(max(1, min(32, cores, avail_gb * 0.25 / (0.6 * (4.0 if case["bits3d"] > 11 else 1.0)))))      # a child peaks at ~0.45 GB on the longdress-like frame
    if procs < 2:
        return {"value": round(1.0 / one_frame_seconds, 5), "cores": 1, "sample": "single core host"}
    with tempfile.TemporaryDirectory() as d:
        kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-child", "%d,%s" % (i, d), "--workload", workload,
                                  "--iterations", str(iterations), "--config", case["name"]], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(procs)]
        limit = time.time() + 120 + 20 * one_frame_seconds
        try:
            while sum(os.path.exists(os.path.join(d, "ready%d" % i)) for i in range(procs)) < procs:
                if time.time() > limit or any(k.poll() not in (None, 0) for k in kids):
                    raise RuntimeError("a CPU baseline child failed before the start")
                time.sleep(0.05)
            t0 = time.time()
            open(os.path.join(d, "go"), "w").close()
            for k in kids:
                k.wait(timeout=max(1.0, limit - time.time()))
            wall = time.time() - t0
            if any(k.returncode != 0 for k in kids):
                raise RuntimeError("a CPU baseline child failed")
        finally:
            for k in kids:
                if k.poll() is None:
                    k.kill()
    return {"value": round(procs / wall, 5), "unit": "frames/s", "cores": procs,
            "sample": "%d frames of the GOF at once, one process of the serial build per frame, same stages, "
                      "%.1f s wall on %d usable hardware threads" % (procs, wall, cores)}


def cpu_child(spec, workload, iterations, case):
    index, d = spec.split(",", 1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from tmc2_amd.synth import synth_cloud
    frames = [synth_cloud(workload, int(index))]
    eng = ob.Reference() if os.path.exists(ob.REF_PATH) else ob.Oracle()
    open(os.path.join(d, "ready" + index), "w").close()
    while not os.path.exists(os.path.join(d, "go")):
        time.sleep(0.01)
    a = eng.phase_a(frames, iterations, case["bits3d"], case["precision"], case["min_w"], case["min_h"], case["pack"], case["vox_dim"])
    eng.phase_b(frames, a, case["precision"])


def verify_frames(a, frames, indices, W, H, host_bufs):
    """After the timed loop, outside the timing: every frame this rank holds, exactly as the LAST timed step left it, against
    the per-frame MD5 fixture of the unmodified reference (patch list, occupancy map / video, blockToPatch, both geometry
    layers, reconstructed cloud, its colours, pointToPixel, both attribute layers).  At N = 1 the canvases digested are the
    ones in the page-locked host buffers -- the bytes the video encoder would read.  Returns (verdict, detail): True / False,
    or None when the run is not the fixture's configuration (other workload / frame count / iterations / packing)."""
    import hashlib
    import numpy as np
    if not a.is_case:
        return None, "no reference fixture for this configuration (flags override the case %s)" % a.case_name
    g = case_fixture(a.case_name)
    if not g:
        return None, "fixture %s missing" % a.case_name

    def md5(x):
        return hashlib.md5(np.ascontiguousarray(x).tobytes()).hexdigest()
    if (W, H) != tuple(int(x) for x in g["canvas"]):
        return False, "canvas %dx%d, reference %s" % (W, H, g["canvas"].tolist())
    bad = []
    for slot, (fr, i) in enumerate(zip(frames, indices)):
        patches = fr.get_patches()[0][fr.get_patch_order()]
        rx, rc, p2p = fr.get_reconstruction()
        if host_bufs is not None:
            img, att = host_bufs[slot]
        else:
            img, att = fr.get_geometry_images(), fr.get_attribute_images()
        if [len(patches), len(rx)] != g["f%d_counts" % i].tolist():
            bad.append("frame %d: %d patches / %d points, reference %s" % (i, len(patches), len(rx), g["f%d_counts" % i].tolist()))
            continue
        flat = np.stack([patches[n] for n in patches.dtype.names if n not in ("depthOffset", "occOffset")], 1).astype(np.int32)
        got = {"patches": flat, "recon_xyz": rx, "recon_rgb": rc, "point_to_pixel": p2p, "attribute": att}
        got.update({k: img[k] for k in ("occupancy", "occ_video", "block_to_patch", "geo0", "geo1")})
        bad += ["frame %d: %s" % (i, k) for k, v in got.items() if md5(v) != str(g["f%d_%s_md5" % (i, k)])]
    return (not bad), ("%d frames x 10 digests equal the reference's" % len(frames) if not bad else "; ".join(bad[:8]))


def case_fixture(name):
    """tests/golden/full_size.npz (make_golden.py full_size): digests of the unmodified reference for one case, {} if absent."""
    import numpy as np
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "full_size.npz"))
        return {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "/")}
    except OSError:
        return {}


def host_slots(host_steps, world, frames_per_rank):
    """Concurrent host-resident steps (the orientation walk; host k-d tree builds if selected) of ONE rank: the node's budget
    split over the ranks, but never fewer than the frames a rank has in flight need to make progress side by side
    (at most 4): 8 ranks x 4 frames must not queue behind 2 slots each."""
    return int(max(min(max(frames_per_rank, 1), 4), host_steps // max(world, 1)))


def ingest_leg(T, torch, ctx, cloud):
    """PLY file -> page-locked host arrays (tmc2_ply_read) -> frame resident in HBM (tmc2_frame_create), per format."""
    import tempfile
    import numpy as np
    xyz, rgb = cloud
    n = len(xyz)
    res = {"points": int(n), "what": "PCCPointSet3::read replacement: file (page cache) -> page-locked buffers -> frame in HBM; "
                                     "ms per frame, one frame at a time; threads = parser threads"}
    hx, hc = T.host_array((n, 3), np.int16), T.host_array((n, 3), np.uint8)
    with tempfile.TemporaryDirectory() as d:
        for name, ascii_ in (("ascii", True), ("binary", False)):
            path = os.path.join(d, name + ".ply")
            T.ply_write(path, xyz, rgb, None, ascii=ascii_)
            size = os.path.getsize(path)
            for threads in (1, 16):
                T.ply_read(path, threads=threads, out=(hx, hc))          # warm (page cache, allocations)
                t0 = time.time()
                T.ply_read(path, threads=threads, out=(hx, hc))
                t1 = time.time()
                fr = ctx.frame(hx, hc)
                torch.cuda.synchronize()
                t2 = time.time()
                del fr
                res["%s_threads%d" % (name, threads)] = {"file_MB": round(size / 1e6, 1), "parse_ms": round(1e3 * (t1 - t0), 2),
                                                         "upload_ms": round(1e3 * (t2 - t1), 2),
                                                         "parse_MB_per_s": round(size / 1e6 / (t1 - t0), 1)}
        assert np.array_equal(hx, xyz) and np.array_equal(hc, rgb)
    return res


def decoder_leg(a, T, torch, enc, frames, clouds, indices, W, H, reps=3):
    """BASELINE config 5 as a GOF: every frame as the DECODER sees it -- the patch records a bitstream carries, the decoded
    occupancy video, geometry maps and I420 attribute frames (identity video codec) in page-locked host memory, no source cloud
    on the device -- through generatePointCloud, the inverse colour conversion, the post-reconstruction tail (boundary points,
    colorPointCloud, grid smoothing, transferColors16bitBP, YUV -> RGB) and the D1 / D2 / colour metric against the uncompressed
    source frame (host arrays, uploaded inside the timing: PccAppDecoder / PccAppMetrics read them from files).  All frames of the
    GOF, as many in flight as the encoder run.  Outside the headline metric; checked frame by frame against the reference's
    digests and PCCMetrics doubles (tests/golden/full_size.npz, make_golden.py full_size_decoder_side)."""
    import hashlib
    import numpy as np
    c, P = a.case, a.case["precision"]
    res = float((1 << (c["bits3d"] - 1)) - 1)

    def pinned(x):
        out = T.host_array(x.shape, x.dtype) if a.pin else np.empty(x.shape, x.dtype)
        out[...] = x
        return out

    def cut(fr, i):
        patches = fr.get_patches()[0][fr.get_patch_order()]
        sent = np.zeros(len(patches), patches.dtype)
        for k in ("u0", "v0", "sizeU0", "sizeV0", "patchOrientation", "u1", "v1", "d1", "normalAxis", "tangentAxis", "bitangentAxis",
                  "projectionMode"):
            sent[k] = patches[k]
        sent["sizeU"], sent["sizeV"] = sent["sizeU0"] * 16, sent["sizeV0"] * 16
        img = fr.get_geometry_images()
        return (sent, pinned(img["occ_video"]), pinned(np.stack([img["geo0"], img["geo1"]])), pinned(fr.encoder_attribute_to_yuv420(4)),
                fr.get_normals())
    cuts = enc.per_frame(frames, cut)
    dec = enc.per_frame(frames, lambda fr, i: fr.ctx.decoder_frame(cuts[i][0], W, H, P, cuts[i][1], cuts[i][2]))

    def chain(fr, i):
        fr.set_decoded_geometry(cuts[i][1], cuts[i][2])            # the decoded video of this frame arrives (H2D)
        fr.codec_generate_point_cloud()
        fr.codec_set_decoded_attribute_yuv420(cuts[i][3], 0)
        fr.codec_post_reconstruct(None)
        return fr.metrics_compute_source(clouds[i][0], clouds[i][1], cuts[i][4], 1, res)
    enc.per_frame(dec, chain)                                      # warm-up (allocations, trees' level counts)
    enc.stage_reset()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        got = enc.per_frame(dec, chain)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    ms, calls = enc.stage_ms(), enc.stage_calls()
    out = {"what": "decoder side of the GOF: decoded occupancy / geometry / I420 attribute frames (page-locked host memory) -> "
                   "generatePointCloud -> YUV420ToYUV444_8_0 -> boundary points, colorPointCloud, grid smoothing, "
                   "transferColors16bitBP, YUV16 -> RGB8 -> PCCMetrics::compute (D1, D2, colour; both directions) against the "
                   "uncompressed source frame (host arrays); %d frames, %d in flight, %d repetitions" % (len(dec), enc.workers, reps),
           "frames_per_s": round(len(dec) / dt, 2), "ms_per_gof": round(1e3 * dt, 2),
           "frame0": {"d1_psnr": round(float(got[0][0][2, 1]), 4), "d2_psnr": round(float(got[0][0][2, 3]), 4),
                      "y_psnr": round(float(got[0][0][2, 7]), 4)},
           "stage_ms_per_frame": {k: round(v / (reps * len(dec)), 3) for k, v in sorted(ms.items()) if v > 0 and calls.get(k)}}
    # parity of what was just timed, against the unmodified reference
    g = case_fixture(a.case_name) if a.is_case else {}
    if "f0_post_xyz_md5" not in g:
        out["verified"], out["verified_detail"] = None, "no decoder-side reference digests for this configuration"
    else:
        md5 = lambda x: hashlib.md5(np.ascontiguousarray(x).tobytes()).hexdigest()
        posts = enc.per_frame(dec, lambda fr, i: fr.get_post_reconstruction())
        bad = []
        for slot, i in enumerate(indices):
            if md5(cuts[slot][3]) != str(g["f%d_i420_md5" % i]):
                bad.append("frame %d: i420" % i)
            if md5(cuts[slot][4]) != str(g["f%d_src_normals_md5" % i]):
                bad.append("frame %d: source normals" % i)
            bad += ["frame %d: post %s" % (i, k) for k in ("xyz", "colors16", "rgb", "boundary")
                    if md5(posts[slot][k]) != str(g["f%d_post_%s_md5" % (i, k)])]
            if not np.array_equal(got[slot][0].view(np.uint64), g["f%d_post_metrics" % i].view(np.uint64)):
                bad.append("frame %d: metric doubles" % i)
        out["verified"] = not bad
        out["verified_detail"] = ("%d frames x (I420 frames, source normals, 4 post-reconstruction digests, 24 metric doubles) equal the "
                                  "reference's" % len(dec)) if not bad else "; ".join(bad[:8])
    for fr in dec:
        fr.close()
    return out


RECORD_SLOTS = 1024      # patch records per frame the side-information gather has room for (a CTC frame has 100 - 200)


def gather_records(enc, frames, sharder, cache):
    """The N > 1 tail of a step when every rank lands its own canvases in host memory (--gather host): what is left to
    gather is the per-frame side information of the bitstream -- the packed patch records (atlas data: ~ 100 bytes a patch) --
    one RCCL gather per step to rank 0.  Returns rank 0's list of [frames per rank][RECORD_SLOTS] record arrays per rank."""
    import numpy as np
    import torch
    import tmc2_amd as T
    recs = enc.per_frame(frames, lambda fr, i: fr.get_patches()[0][fr.get_patch_order()])
    size = recs[0].dtype.itemsize if recs else np.dtype(T.lib.PATCH_DTYPE).itemsize
    buf = np.zeros((max(1, len(frames)), 8 + RECORD_SLOTS * size), np.uint8)
    for i, r in enumerate(recs):
        if len(r) > RECORD_SLOTS:
            raise RuntimeError("frame with %d patches: more than the side-information gather holds" % len(r))
        buf[i, :8] = np.frombuffer(np.int64(len(r)).tobytes(), np.uint8)
        buf[i, 8:8 + len(r) * size] = np.frombuffer(np.ascontiguousarray(r).tobytes(), np.uint8)
    got = sharder.gather(torch.from_numpy(buf).to(sharder.device))
    if sharder.rank == 0:
        cache["records"] = [g.cpu().numpy() for g in got]
    return cache


def gather_canvases(enc, frames, sharder, cache, pin=True):
    """The N > 1 tail of a step: the finished canvases of every frame slot go to rank 0 (one gather per canvas kind and
    slot: every rank holds the same number of frames), which moves what arrives into (page-locked) host memory with
    asynchronous copies -- completed by the synchronize after the timed steps.  Returns rank 0's buffers
    {(slot, kind): uint8 tensor [world, ...]} (the frame of slot i on rank r is GOF frame r + i * world)."""
    import torch
    for i, fr in enumerate(frames):
        for name in ("geometry", "occ_video", "attribute"):
            src = enc.device_tensor(fr, name)
            got = sharder.gather(src)
            if sharder.rank == 0:
                key = (i, name)
                if key not in cache or cache[key].shape[1:] != src.shape:
                    cache[key] = torch.empty((sharder.world,) + tuple(src.shape), dtype=torch.uint8, pin_memory=pin)
                for r, x in enumerate(got):
                    cache[key][r].copy_(x, non_blocking=True)
    return cache


def main():
    a = parse()
    if a.cpu_child in ("baseline", "baseline-bounded"):        # the whole CPU baseline leg, in a process of its own
        from tmc2_amd.synth import synth_cloud
        bounded = a.cpu_child == "baseline-bounded"
        print(json.dumps(cpu_baseline(a.workload, a.iterations, [synth_cloud(a.workload, i) for i in range(min(a.frames, 8 if bounded else 16))],
                                      a.case, bounded=bounded)))
        return
    if a.cpu_child:
        return cpu_child(a.cpu_child, a.workload, a.iterations, a.case)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_launcher(a)                              # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): the line's n_gpus would not be what "
                         "was asked for" % (a.gpus, world))
    import numpy as np
    import tmc2_amd as T
    if a.frames % world:
        raise SystemExit("bench.py: --frames %d is not a multiple of the %d ranks (the canvas gather runs once per frame slot)" % (a.frames, world))
    my_indices = list(range(rank, a.frames, world))
    # generation starts in forked processes before any GPU context exists (fork-safe); the rendezvous comes up meanwhile
    clouds_ready = (lambda: []) if a.preflight_only else start_frames(a.workload, my_indices, a.gen_procs)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.dist_backend == "gloo":
            if not a.preflight_only:
                local = local % max(1, torch.cuda.device_count())
                torch.cuda.set_device(local)
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() <= local:
                raise SystemExit("bench.py: rank %d has no GPU %d (%d visible): RCCL needs one GPU per rank" % (rank, local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    sharder = T.Sharder(rank, world, dist, "cpu" if (world > 1 and a.dist_backend == "gloo") else "cuda:%d" % local)
    if world > 1:
        preflight(a, torch, dist, sharder, rank, world)
    if a.preflight_only:
        if rank == 0:
            print(json.dumps({"preflight": "ok", "n_gpus": world, "backend": a.dist_backend}))
            sys.stdout.flush()
        if world > 1:
            dist.destroy_process_group()
        return
    clouds = clouds_ready()
    to_host_here = world == 1 or a.gather == "host"     # every rank lands its own frames' canvases in host memory
    # one hardware queue per in-flight frame (GPU_MAX_HW_QUEUES = 16, tmc2_amd/lib.py): streams that share a queue serialise
    # behind each other, and 16 frames in flight keep the chip busy (measured: 12 -> 62, 16 -> 73, 20 -> 68, 24 -> 55, 32 -> 64 frames/s)
    workers = a.workers or max(1, min(len(clouds), 16, (os.cpu_count() or 8) // world))
    slots = host_slots(a.host_steps, world, min(len(clouds), workers))
    # many frames in flight and idle host cores: the (exact) host k-d tree builder leaves the GPU to the other stages
    # (round 1 preferred host builds above 8 frames in flight; with the subtree-finishing device build the device wins:
    #  16 frames in flight, measured: device 85, host 75 frames/s -- and the host cores stay free)
    kd_mode = {"device": 0, "host": 1, "adaptive": 2}.get(a.kdtree, 0)
    c = a.case
    P = c["precision"]
    enc = T.GofEncoder(local, workers, a.iterations, c["bits3d"], P, c["min_w"], c["min_h"], timing=True, first_domain=rank * workers,
                       vox_dim=c["vox_dim"])
    enc.set_option("KDTREE_HOST", kd_mode)                # (options of this encoder's contexts: nothing process-wide)
    enc.set_host_slots(slots)                             # (this encoder's own gate around the host-resident steps)
    reserve_note = None
    if a.reserve:                                          # the sequence's largest frame, before the first one arrives
        try:
            enc.reserve(max(len(c_[0]) for c_ in clouds), c["min_w"], max(c["min_h"], c["min_w"]))
        except T.Tmc2Error as e:                           # (not enough device memory for every context's worst case: grow on demand)
            reserve_note = "tmc2_ctx_reserve failed (%s): the pools grow by hipMalloc" % (e,)
    frames = enc.upload(clouds)                          # inputs resident in HBM
    n_points = sum(len(c[0]) for c in clouds)

    import threading
    host_cache, host_lock = {}, threading.Lock()
    gather_cache = {}

    shared = {}                                             # (W, H) -> this rank's shared-memory segment (N > 1)

    def host_out(W, H):
        """Host-side destination of the finished canvases (what the video encoder reads), allocated once: page-locked
        host memory, so that the copies are plain DMA instead of being staged through the runtime's bounce buffers.
        N > 1: a shared-memory segment per rank (page-locked by the rank that writes it, mapped by rank 0 afterwards): one
        node, one address space -- the process that runs the video encoder reads every rank's canvases where they landed."""
        per_frame = W * H * (1 + 2 + 2 + 6) + (W // P) * (H // P) + (W // 16) * (H // 16) * 4 + 64
        seg = [None]

        def pinned(shape, dtype):
            if seg[0] is not None:                              # carve the next 64-byte aligned piece out of the segment
                n = int(np.prod(shape)) * np.dtype(dtype).itemsize
                at = seg[1]
                seg[1] = (at + n + 63) & ~63
                return seg[0].array[at:at + n].view(dtype).reshape(shape)
            return T.host_array(shape, dtype) if a.pin else np.empty(shape, dtype)   # (tmc2_host_alloc: page-locked, portable)
        with host_lock:                                         # (called from the worker threads)
          if (W, H) not in host_cache:
            if world > 1 and a.pin:
                try:
                    name = "tmc2_bench_%s_%dx%d_r%d" % (os.environ.get("MASTER_PORT", "0"), W, H, rank)
                    shared[(W, H)] = T.SharedHostArray(name, (per_frame + 64 * 8) * len(frames), create=True)
                    seg[:] = [shared[(W, H)], 0]
                except (OSError, T.Tmc2Error) as e:             # /dev/shm too small or not there: private page-locked memory
                    shared[(W, H)] = "unavailable (%r): private page-locked buffers per rank" % (e,)
                    seg[:] = [None]
            host_cache[(W, H)] = [(dict(occupancy=pinned((H, W), np.uint8), occ_video=pinned((H // P, W // P), np.uint8),
                                        block_to_patch=pinned((H // 16, W // 16), np.uint32),
                                        geo0=pinned((H, W), np.uint16), geo1=pinned((H, W), np.uint16)),
                                   pinned((2, 3, H, W), np.uint8)) for _ in frames]
        return host_cache[(W, H)]

    # auto: the C++ host (north_star: "host code stays C++ ... RCCL only for the final gather") whenever its libraries load -- one GPU
    # or several; with several ranks under RCCL that needs librccl.so (tmc2_gof_comm_create says so), under --dist-backend gloo (ranks
    # sharing a GPU: no RCCL world to make) the ranks meet over torch.distributed as before
    native = a.host == "native" or (a.host == "auto" and (world == 1 or a.dist_backend == "nccl"))
    comm = None
    native_fallback = [None]
    if native:
        from tmc2_amd import native_gof
        native_gof.load_library()                           # (fails here, loudly, if it was not built)
        if world > 1:                                       # the ranks of the sharded GOF meet in C++: RCCL on this rank's stream
            # the communicator's id travels through a file only this job knows: a name with a nonce from rank 0, agreed over the
            # launcher's process group (include/tmc2gof.h: the default name is per user and port, a caller with a launcher does better)
            import uuid
            box = ["/dev/shm/tmc2_gof_id_%d_%s" % (os.getuid(), uuid.uuid4().hex)]
            dist.broadcast_object_list(box, src=0)
            why = None
            try:
                comm = native_gof.Comm(enc.ctxs[0], rank, world, rendezvous=box[0])
            except Exception as e:                          # (no librccl.so to load, a communicator that does not come up, ...)
                why = repr(e)
            # every rank or none: a world in which one rank could not make its communicator meets over torch.distributed instead
            # (the Python host of rounds 3-5) -- and the line says so (config.host)
            ok = [None] * world
            dist.all_gather_object(ok, why)
            if any(w is not None for w in ok):
                if a.host == "native":                      # asked for by name: no quiet substitute
                    raise RuntimeError("bench: --host native: %s" % next(w for w in ok if w is not None))
                if comm is not None:
                    comm.close()
                comm, native = None, False
                native_fallback[0] = next(w for w in ok if w is not None)
                if rank == 0:
                    print("bench: the C++ / RCCL host is not available (%s): the ranks meet over torch.distributed" % native_fallback[0], file=sys.stderr)
    capacity = [c["min_w"], c["min_h"]]
    resumed_passes = [0]

    def native_step():
        # reset, S0, S1-S9 + packing, the rendezvous, S12-S22 and the copies into page-locked host memory: one call into C++.
        # A GOF that outgrows the buffers (longdress: 1280 x 1344 on a 1280 x 1280 minimum canvas) is refused at the rendezvous with
        # the size it needs; the pass is RESUMED there with larger buffers (tmc2_gof_encode_resume) -- rounds 4-5 called the whole
        # pass again, which made the first pass of the process twice as long as the others (profiles/r06_first_pass.txt)
        resume = False
        while True:
            try:
                slot_of = [i % workers for i in range(len(frames))]
                if comm is not None:                        # this rank's frames; weights, canvas height and records cross over RCCL
                    W_, H_, recs = native_gof.encode_sharded(comm, frames, slot_of, workers, a.iterations, c["vox_dim"], c["bits3d"], P,
                                                             c["min_w"], c["min_h"], host_out(*capacity), capacity,
                                                             record_slots=RECORD_SLOTS, packing=a.packing, resume=resume)
                    if recs is not None:
                        gather_cache["records"] = recs
                    return W_, H_
                return native_gof.encode(frames, slot_of, workers, a.iterations, c["vox_dim"], c["bits3d"], P, c["min_w"], c["min_h"],
                                         a.packing, host_out(*capacity), capacity, guess_canvas=a.rendezvous == "one", resume=resume)
            except native_gof.CanvasTooSmall as e:          # (the same on every rank: the size is the GOF's)
                capacity[:] = [max(capacity[0], e.size[0]), max(capacity[1], e.size[1])]
                resume = a.rendezvous != "one"              # (a guessed canvas has no second half to resume: the whole pass again)
                resumed_passes[0] += 1

    def step():
        if native:
            return native_step()
        for fr in frames:
            fr.reset()
        if a.packing == "all-intra" and a.rendezvous == "one":
            # every frame runs its whole chain on its worker; one rendezvous per GOF (the common canvas size is verified at
            # the end: GofEncoder.encode_all_intra).  Finished canvases -> host memory (N = 1) / rank 0 (N > 1).
            def to_host(fr, i, size):
                bufs = host_out(size[0], size[1])
                fr.get_geometry_images(bufs[i][0])
                fr.get_attribute_images(bufs[i][1])
            W, H = enc.encode_all_intra(frames, sharder, finish=to_host if to_host_here else None)
            if world > 1:
                (gather_records if a.gather == "host" else gather_canvases)(enc, frames, sharder, gather_cache)
            return W, H
        # identity video codec between the phases (HM/VTM on the host is outside the metric): phase B runs on the
        # resident canvases.  Finished canvases -> rank 0 -> host memory, where the video encoder reads them.
        # Once the GOF's canvas size is settled the frames are independent: a frame's worker goes on with its phase B and the
        # copy of its canvases in the same pass (one rendezvous per GOF instead of four: each one waits for its slowest frame).
        def rest(fr, i, W_, H_):
            fr.encoder_generate_attribute_images()
            if to_host_here:
                bufs = host_out(W_, H_)
                fr.get_geometry_images(bufs[i][0])
                fr.get_attribute_images(bufs[i][1])
        W, H = enc.phase_a(frames, sharder, constrained_pack={"all-intra": False, "low-delay": True, "random-access": 2}[a.packing],
                           frame_count=a.frames, then=None if a.rendezvous == "four" else rest)
        if a.rendezvous == "four":                              # (the round-2 schedule: a rendezvous after every phase)
            enc.phase_b(frames)
            if to_host_here:
                bufs = host_out(W, H)
                enc.per_frame(frames, lambda fr, i: (fr.get_geometry_images(bufs[i][0]), fr.get_attribute_images(bufs[i][1])))
        if world > 1:
            (gather_records if a.gather == "host" else gather_canvases)(enc, frames, sharder, gather_cache)
        return W, H

    def sync():
        torch.cuda.synchronize()
        sharder.barrier()

    if to_host_here:
        host_out(*capacity)                                 # the encoder's own output buffers (page-locked): set-up, not a pass
    pass_ms = []                                            # the first passes of this process over the GOF, one by one
    first_stage_ms = {}
    for k_pass in range(max(0, a.prime) + a.warmup):
        if k_pass == 0:
            enc.stage_reset()
        t_pass = time.time()
        step()
        pass_ms.append(round(1e3 * (time.time() - t_pass), 2))
        if k_pass == 0:
            first_stage_ms = enc.stage_ms()                   # (event-bracketed: in flight from the first to the last launch of a stage)
    enc.stage_reset()
    sync()
    t0 = time.time()
    step_ms, t_last = [], t0                                # (every timed step by the host's clock: the scatter inside the region)
    for _ in range(a.steps):
        W, H = step()
        t_now = time.time()
        step_ms.append(round(1e3 * (t_now - t_last), 2))
        t_last = t_now
    sync()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=sharder.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms, calls = enc.stage_ms(), enc.stage_calls()
    # parity of what was just timed (outside the timing): every frame of every rank against the reference's MD5s
    try:
        verdict, detail = verify_frames(a, frames, my_indices, W, H, host_out(W, H) if to_host_here else None)
    except Exception as e:
        verdict, detail = False, "verification failed to run: %r" % (e,)
    if world > 1:
        t = torch.tensor([-1 if verdict is None else int(bool(verdict))], dtype=torch.int32, device=sharder.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if verdict is not None or int(t.item()) >= 0:
            verdict = bool(int(t.item()) > 0) and verdict is not False
    def drop_shared():
        for v in shared.values():
            if hasattr(v, "close"):
                v.close()
    if rank != 0:
        if world > 1:
            dist.barrier()                                  # (rank 0 may still be reading this rank's segment)
            dist.destroy_process_group()
        drop_shared()
        if comm is not None:
            comm.close()
        return
    # The timed region runs 32 frames at once: their launches share the chip and queue behind each other, so a kernel's
    # event-bracketed time in the region says how long it was in flight, not how much of the GPU it needs.  The kernel the
    # roofline is reported for is therefore chosen by its time with the GPU to itself (one frame, outside the timing);
    # both durations are reported.
    enc.set_option("REFINE_OVERLAP", 1)                     # one frame in flight: the library's few-frames-in-flight regime (what a
    for warm in (True, False):                              # GofEncoder of <= 4 workers sets; scheduling only, never a result);
        enc.stage_reset()                                   # once untimed: the regime holds its buffers in another order, and the
        frames[0].reset()                                   # context's pool has to have seen that (first-use hipMallocs otherwise)
        enc.phase_a(frames[:1], sharder=T.Sharder())
        enc.phase_b(frames[:1])
    enc.set_option("REFINE_OVERLAP", 1 if workers <= 4 else 0)
    solo_ms, solo_calls = enc.stage_ms(), enc.stage_calls()
    n_frames = max(1, len(frames))
    N = n_points / n_frames
    M = float(frames[0].recon_count())
    V = solo_ms.get("refine_voxels", 0.0) / max(1, solo_calls.get("refine_voxels", 1))
    L = (solo_ms.get("refine_row_entries", 0.0) / max(1, solo_calls.get("refine_row_entries", 1))) / V if V else 0.0
    A, I = float(W * H), a.iterations

    def stage_view(name):
        """(in-flight ms per launch, alone ms per launch, launches in the timed region, algorithmic bytes per launch)"""
        if name == "refine_sweep":                         # one of the I sweeps of the refine_sweeps stage
            fl, al, n_l = ms.get("refine_sweeps", 0.0), solo_ms.get("refine_sweeps", 0.0), calls.get("refine_sweeps", 0) * I
            return fl / max(1, n_l), al / max(1, solo_calls.get("refine_sweeps", 1) * I), n_l, algorithmic_bytes(name, N, M, V, L, A, P)
        if name == "tree_knn_normals":                     # SURVEY 8d's S1 + S2 + S6 row: source k-d tree build + 16-NN lists + normals = 94 N
            parts = ("kdtree_build", "knn_self", "normals")
            n_l = calls.get("knn_self", 0)
            fl = sum(ms.get(k, 0.0) / max(1, calls.get(k, 1)) for k in parts)
            al = sum(solo_ms.get(k, 0.0) / max(1, solo_calls.get(k, 1)) for k in parts)
            return fl, al, n_l, 94.0 * N
        if name == "patches":                              # S7-S9: connected components + per-patch build, all rounds
            fl = ms.get("patches_cc", 0.0) + ms.get("patches_build", 0.0)
            al = solo_ms.get("patches_cc", 0.0) + solo_ms.get("patches_build", 0.0)
            n_l = calls.get("patches_build", 0)
            return fl / max(1, n_l), al / max(1, solo_calls.get("patches_build", 1)), n_l, 102.0 * N
        n_l = calls.get(name, 0)
        return (ms.get(name, 0.0) / max(1, n_l), solo_ms.get(name, 0.0) / max(1, solo_calls.get(name, 1)), n_l,
                algorithmic_bytes(name, N, M, V, L, A, P))

    names = ["tree_knn_normals", "knn_self", "normals", "initial_segmentation", "refine_setup", "refine_sweep", "patches", "k:ccMutualMask",
             "k:ccUnion", "k:ccRelax", "geometry_images", "reconstruct", "knn8_recon_in_source", "knn1_source_in_recon",
             "attribute_images"]
    per_stage, alone_total = {}, {}
    for nm in names:
        fl, al, n_l, by = stage_view(nm)
        if n_l == 0 or al <= 0 or by <= 0:
            continue
        runs_per_frame = n_l / float(a.steps * n_frames)
        alone_total[nm] = al * runs_per_frame             # GPU time per frame with the GPU to itself
        per_stage[nm] = {"alone_ms": round(al, 4), "in_flight_ms": round(fl, 4), "runs_per_frame": round(runs_per_frame, 2),
                         "MB": round(by / 1e6, 2), "alone_GB/s": round(by / (al * 1e-3) / 1e9, 1),
                         "in_flight_GB/s": round(by / (fl * 1e-3) / 1e9, 1) if fl > 0 else None}
    dom = max(alone_total, key=alone_total.get)
    avg_ms, s_avg, launches, dom_bytes = stage_view(dom)
    achieved = dom_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    s_ach = dom_bytes / (s_avg * 1e-3) / 1e9 if s_avg > 0 else 0.0
    fps = a.frames * a.steps / dt
    b_alg = path_bytes(N, M, V, L, I, A, P)
    # HBM bytes per launch from the PMC passes of profiles/collect.sh (FETCH_SIZE / WRITE_SIZE need their own runs under
    # rocprofv3, so they cannot be taken here): used only if they were collected from THESE kernel sources -- the file carries
    # a sha1 over mpeg-pcc-tmc2_amd/csrc, recomputed here; counters of other code are dropped, not quoted
    traffic, traffic_note = None, "no PMC file"
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import glob
        latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        with open(latest) as f:
            pmc = json.load(f)
        stamp = None
        if pmc.get("source_sha1"):
            import hashlib
            h = hashlib.sha1()
            csrc = os.path.join(ROOT, "mpeg-pcc-tmc2_amd", "csrc")
            for name in sorted(os.listdir(csrc)):
                if name.endswith((".hip", ".cpp", ".h")):
                    with open(os.path.join(csrc, name), "rb") as f:
                        h.update(name.encode() + b"\0" + f.read())
            stamp = h.hexdigest()
        if stamp is None or stamp != pmc.get("source_sha1"):
            traffic_note = "%s is of other kernel sources (sha1 %s, here %s): dropped" % (os.path.basename(latest), str(pmc.get("source_sha1"))[:12], str(stamp)[:12])
        elif dom in pmc.get("stages", {}) and a.workload == pmc.get("workload"):
            traffic, traffic_note = pmc["stages"][dom]["hbm_bytes_per_launch"], "%s (same kernel sources, sha1 %s)" % (os.path.basename(latest), stamp[:12])
        else:
            traffic_note = "%s has no entry for %s" % (os.path.basename(latest), dom)
    except (OSError, ValueError, KeyError, IndexError):
        pass
    out = {
        "metric": "encoder patch+image-gen frames/sec, %s %d-frame GOF" % (a.workload, a.frames),
        "value": round(a.frames * a.steps / dt, 4), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "priming_passes": max(0, a.prime), "ms_per_step": round(1000.0 * dt / a.steps, 2), "step_ms": step_ms, "higher_is_better": True,
        "first_gof_ms": pass_ms[0] if pass_ms else None, "untimed_pass_ms": pass_ms, "passes_resumed_with_larger_buffers": resumed_passes[0], "pool": dict(enc.pool_stats(), reserved=bool(a.reserve) and reserve_note is None, note=reserve_note),
        "first_gof_excess_ms_per_frame": dict(sorted(((k, round((first_stage_ms.get(k, 0.0) - v / a.steps) / max(1, len(frames)), 3))
                                                      for k, v in ms.items() if first_stage_ms.get(k, 0.0) - v / a.steps > 0.2 * len(frames)
                                                      and not k.startswith(("refine_row_entries", "refine_voxels", "refine_sweeps_executed"))),   # (counters, not times)
                                                     key=lambda kv: -kv[1])[:8]),
        "scaling": "strong", "vs_baseline": None, "dtype": "int32/f64", "data": "synthetic",
        "verified": verdict, "verified_detail": detail,
        "config": {"workload": "%s-like synthetic, %d frames, %d points/frame avg, ctc-common + %s "
                               "(refine iterations %d on voxels of %d, %d-bit geometry, occupancyPrecision %d), canvas %dx%d" %
                               (a.workload, a.frames, n_points // max(1, len(frames)), a.packing, a.iterations, c["vox_dim"],
                                c["bits3d"] - 1, P, W, H),
                   "case": a.case_name if a.is_case else None,
                   "stages": "S0-S22: k-d tree, kNN, normals, orientation, segmentation, refinement, patches, packing, "
                             "occupancy + geometry images, dilation, reconstruction, colour transfer, attribute images, "
                             "push-pull padding (identity video codec between the phases); the D1/D2 metric (S23) is "
                             "reported separately (metric_ms_per_frame)",
                   "canvases": ("each rank -> page-locked shared host memory (%s); patch records -> rank 0 over %s" %
                                ("; ".join(sorted(set("a /dev/shm segment per rank" if hasattr(v, "close") else str(v) for v in shared.values()))) or "private buffers",
                                 "RCCL" if a.dist_backend == "nccl" else "gloo")) if (world > 1 and a.gather == "host")
                               else ("gathered to rank 0 over RCCL, copied out from there" if world > 1 else "page-locked host memory"),
                   "host": (("native: one tmc2_gof_encode_sharded call per GOF and rank (libtmc2gof.so: C++ threads over the C-ABI; weights, "
                             "canvas height and patch records cross the node as RCCL collectives issued from C++)" if comm is not None else
                             "native: one tmc2_gof_encode call per GOF (libtmc2gof.so, C++ threads over the C-ABI)") if native
                            else "python: GofEncoder's worker threads over the C-ABI" + ("" if native_fallback[0] is None else " (the C++ / RCCL host was asked for and not available: %s)" % native_fallback[0])),
                   "frames_per_gpu": len(frames), "host_workers_per_gpu": workers, "host_step_slots_per_gpu": slots, "parallelism": "frames f%%%d" % world},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_note, "avg_launch_ms": round(avg_ms, 4),
                     "launches": launches, "alone_avg_launch_ms": round(s_avg, 4), "alone_achieved": round(s_ach, 2),
                     "alone_frac": round(s_ach / 8000.0, 5),
                     "algorithmic_MB_per_launch": round(dom_bytes / 1e6, 2),
                     "what": "the dominant GPU step by time with the GPU to itself (one frame in flight: the library's "
                             "few-frames-in-flight regime, option REFINE_OVERLAP), over every timed stage with a contract byte "
                             "count (SURVEY.md 8d; tree_knn_normals = its S1 + S2 + S6 row, the source tree build included: 94 N); "
                             "refine_sweep = one of the %d sweeps of S5 = 4VL + 24V + 26N bytes" % I,
                     "N": int(N), "M": int(M), "V": int(V), "L": round(L, 1),
                     "path": {"B_alg_GB_per_frame": round(b_alg / 1e9, 3), "achieved": round(b_alg * fps / 1e9, 1),
                              "frac": round(b_alg * fps / 1e9 / 8000.0, 5)},
                     "stages": per_stage},
        "stage_ms_per_frame": {k: round(v / (a.steps * len(frames)), 3) for k, v in sorted(ms.items())},
        # S3's cliff: a frame whose contracted graph is inconsistent at 0.98 walks the threshold ladder (a contraction + a cluster
        # walk per step: milliseconds) and, past its last step, falls back to the growth point by point (hundreds of milliseconds,
        # and the GOF's rendezvous waits for it).  Per GOF of the timed region:
        "orientation": {"frames_per_gof": a.frames,
                        "contracted_walks_per_gof": round(calls.get("orient_contract", 0) / a.steps, 2),
                        "ladder_steps_per_gof": round(calls.get("orient_tau_retry", 0) / a.steps, 2),
                        "point_level_fallbacks_per_gof": round(calls.get("orient_normals_regrowth", 0) / a.steps, 2),
                        "pair_table_overflows_per_gof": round(calls.get("orient_pair_table_overflow", 0) / a.steps, 2),
                        "exact_size_repeats_per_gof": round(calls.get("orient_exact_size_repeat", 0) / a.steps, 2),
                        "what": "this rank's frames; SURVEY 8a S3 (PCCNormalsGenerator.cpp:198-242): the exact directed growth runs on the "
                                "contracted graph; ladder = thresholds 0.99 / 0.995 / 0.998 tried after 0.98"},
    }
    # the measured ceiling next to the nominal 8 TB/s: a device-to-device copy of 1 GiB, bytes read +
    # bytes written per second, with the GPU to itself (outside the timed region)
    try:
        out["roofline"]["stream_copy"] = stream_copy_ceiling(torch, "cuda:%d" % local)
    except Exception as e:
        out["roofline"]["stream_copy"] = {"error": repr(e)}
    # S23 is reported separately: one frame, D1 + D2 + colour, both directions
    # (tmc2_metrics_compute_frame: source, normals and reconstruction are the frame's resident arrays -- nothing is uploaded)
    resolution = float((1 << (c["bits3d"] - 1)) - 1)
    enc.per_frame(frames[:1], lambda fr, i: fr.metrics_compute(0, True, resolution))             # warm (allocations)
    enc.ctxs[0].stage_reset()
    t0 = time.time()
    q, qc = enc.per_frame(frames[:1], lambda fr, i: fr.metrics_compute(0, True, resolution))[0]
    out["metric_ms_per_frame"] = round(1000.0 * (time.time() - t0), 1)
    out["metric_stage_ms"] = {k: round(v, 3) for k, v in sorted(enc.ctxs[0].stage_ms().items()) if v > 0}
    out["metric_frame0"] = {"d1_psnr": round(float(q[2, 1]), 4), "d2_psnr": round(float(q[2, 3]), 4), "y_psnr": round(float(q[2, 7]), 4),
                            "points": [int(qc[0]), int(qc[1])]}
    try:                                                       # ... and against the reference's doubles where the fixture has them
        # (the single-frame case of the same sequence holds PCCMetrics::compute's doubles for frame 0; an all-intra frame's
        #  reconstruction does not depend on the other frames of its GOF)
        import re
        g1 = case_fixture(re.sub(r"_gof\d+$", "", a.case_name))
        if a.is_case and c["pack"] == 0 and "f0_metrics" in g1:
            out["metric_frame0"]["equals_reference"] = bool(np.array_equal(q.view(np.uint64), g1["f0_metrics"].view(np.uint64)))
    except OSError:
        pass
    # What ONE rank of the 8-GPU run does (the driver's SCALE run is the only real measurement; this is its single-GPU proxy):
    # 32 / 8 = 4 frames, 4 in flight, the same stages -- the wall time of that step bounds the 8-GPU step from below (the
    # gather of the finished canvases to rank 0 comes on top), so 32 frames / that time is what the node can reach at most.
    if world == 1 and len(frames) >= 4:
        try:
            sub = frames[:4]

            def rank_step():
                for fr in sub:
                    fr.reset()
                def rest4(fr, i, w_, h_):
                    fr.encoder_generate_attribute_images()
                    b4 = host_out(w_, h_)
                    fr.get_geometry_images(b4[i][0])
                    fr.get_attribute_images(b4[i][1])
                enc.phase_a(sub, sharder=T.Sharder(), then=rest4)
            enc.set_option("REFINE_OVERLAP", 1)                 # (what a GofEncoder of <= 4 workers -- a rank of the 8-GPU run -- sets)
            # (... and delays the start of the second half of its frames: tmc2_amd/gof.py; BENCH_PROXY_STAGGER_US: another delay)
            for late in enc.ctxs[2:4]:
                late.set_option("FRAME_START_DELAY_US", os.environ.get("BENCH_PROXY_STAGGER_US", str(T.gof.FEW_FRAMES_START_DELAY_US)))
            rank_step()
            torch.cuda.synchronize()
            t0 = time.time()
            reps = 5
            for _ in range(reps):
                rank_step()
            torch.cuda.synchronize()
            ms4 = 1000.0 * (time.time() - t0) / reps
            enc.set_option("REFINE_OVERLAP", 1 if workers <= 4 else 0)
            enc.set_option("FRAME_START_DELAY_US", None)
            out["per_rank_proxy"] = {"frames": 4, "workers": 4, "ms": round(ms4, 2),
                                     "predicted_n8_frames_per_s": round(a.frames / (ms4 * 1e-3), 1),
                                     "predicted_n8_speedup": round(a.frames / (ms4 * 1e-3) / out["value"], 2),
                                     "what": "one rank's share of the 8-GPU run on this GPU alone: 4 frames, 4 in flight, full path, "
                                             "canvases to host; upper bound for N = 8 = frames / this time (no gather, no skew)"}
            step()                                             # (leave the frames as a full step leaves them)
        except Exception as e:
            out["per_rank_proxy"] = {"error": repr(e)}
    # the post-reconstruction tail is outside the metric as well: one frame with the GPU to
    # itself (stage times), then the whole GOF through the worker threads
    if world == 1 and a.tail:
        try:
            enc.stage_reset()
            enc.phase_c(frames[:1])                           # warm-up (allocations)
            enc.stage_reset()
            t0 = time.time()
            enc.phase_c(frames[:1])
            solo = 1000.0 * (time.time() - t0)
            tail_ms = enc.stage_ms()
            post = frames[0].get_post_reconstruction(xyz=False, colors16=False, rgb=False)
            W, H = step()                                     # (the one-frame run above left frame 0 on its own canvas)
            i420 = [(T.host_array if a.pin else np.empty)((2, W * H * 3 // 2), np.uint8) for _ in frames]
            enc.phase_c(frames, i420_out=i420)
            torch.cuda.synchronize()
            t0 = time.time()
            enc.phase_c(frames, i420_out=i420)
            torch.cuda.synchronize()
            gof = time.time() - t0
            out["tail"] = {"what": "attribute canvases -> I420 (RGB444ToYUV420_8_4) -> host -> device -> 16-bit 4:4:4 "
                                   "(YUV420ToYUV444_8_0) -> colorPointCloud, grid smoothing, transferColors16bitBP, YUV16 -> RGB8",
                           "ms_per_frame_alone": round(solo, 2), "gof_frames_per_s": round(len(frames) / gof, 2),
                           "reconstructed_points": int(len(post["boundary"])), "boundary_points": int((post["boundary"] != 0).sum()),
                           "moved_points": int((post["boundary"] == 3).sum()),
                           "stage_ms_alone": {k: round(v, 3) for k, v in sorted(tail_ms.items()) if v > 0}}
        except Exception as e:                                 # never lose the metric line over the side measurement
            out["tail"] = {"error": repr(e)}
    if world == 1 and a.decoder:
        try:
            W, H = step()                                     # (every frame as a full step leaves it)
            out["decoder"] = decoder_leg(a, T, torch, enc, frames, clouds, my_indices, W, H)
        except Exception as e:
            out["decoder"] = {"error": repr(e)}
    # PLY ingest, outside the metric too: one frame written as the reference writes it (ASCII,
    # as the 8i / Owlii content ships, and binary), read straight into page-locked buffers, uploaded and bound to a frame
    if world == 1 and a.ingest and a.pin:
        try:
            out["ingest"] = ingest_leg(T, torch, enc.ctxs[0], clouds[0])
        except Exception as e:
            out["ingest"] = {"error": repr(e)}
    if a.cpu_baseline == 2 and world == 1:                     # the round-3 form, kept for reproducing its heap corruption: in-process
        out["cpu_baseline"] = cpu_baseline(a.workload, a.iterations, clouds[:min(a.frames, 16)], a.case)
    elif a.cpu_baseline and world == 1:                        # rank 0 at N = 1 only (the contract of the bench line)
        # In a process of its own: the leg loads the reference's libraries (two builds of the same C++ code, one with TBB and
        # 128 threads) -- foreign code that must not be able to take the metric line with it (a full run died once with glibc's
        # "corrupted double-linked list" seconds into this leg; the path itself ran clean under MALLOC_CHECK_=3).
        import subprocess

        def baseline_child(kind):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", kind, "--workload", a.workload,
                                "--iterations", str(a.iterations), "--frames", str(a.frames), "--config", a.case_name],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
            lines = r.stdout.decode(errors="replace").strip().splitlines()
            if r.returncode != 0 or not lines:
                raise RuntimeError("exit code %d, stderr tail: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:].replace("\n", " | ")))
            return json.loads(lines[-1])
        try:
            out["cpu_baseline"] = baseline_child("baseline-bounded" if a.cpu_baseline == 3 else "baseline")
        except Exception as e:                                  # foreign code (the reference's libraries, 128 TBB threads): once more, bounded
            first_error = repr(e)
            try:
                out["cpu_baseline"] = dict(baseline_child("baseline-bounded"), first_attempt_failed=first_error)
            except Exception as e2:
                out["cpu_baseline"] = {"error": "the CPU baseline process failed twice: %s; %r" % (first_error, e2)}
        for key in ("all_cores_value", "frame_processes_value"):
            if out["cpu_baseline"].get(key):
                out["cpu_baseline"]["gpu_over_" + key[:-6]] = round(out["value"] / out["cpu_baseline"][key], 2)
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    drop_shared()
    # Orderly teardown (round 3 left through os._exit after an unexplained heap corruption in a bench process; DESIGN.md section 9
    # has what round 4 found): frames before their contexts, each worker thread ended and joined, contexts closed, then the
    # interpreter's own exit.
    if comm is not None:
        comm.close()
    for fr in frames:
        fr.close()
    enc.close(join=True)


if __name__ == "__main__":
    main()
