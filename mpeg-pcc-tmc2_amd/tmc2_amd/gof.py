"""Group-of-frames orchestration: the call sequence a PCCEncoder::encode adaptor issues,
frames of a GOF processed concurrently by host worker threads (one tmc2_ctx + HIP stream each -- the
reference does the same with a tbb::parallel_for over frames, PCCEncoder.cpp:4729-4750), and, for
multi-GPU runs, frames sharded over ranks: frame f -> rank f % world.  The only cross-frame coupling of
the all-intra path is (a) the axis weights of frame 0 (24 bytes, broadcast) and (b) the common canvas
size (max over frames, all-reduce); finished canvases are gathered to rank 0 (RCCL over xGMI under the
"nccl" backend, gloo in the CPU tests).  This module holds no algorithmic code."""
import os
import queue
import threading

import numpy as np

from . import lib


class Sharder:
    """Frame -> rank assignment and the three tiny collectives of a GOF.  `dist` is torch.distributed (or None
    for a single process); tensors live on `device` ('cpu' under gloo, 'cuda:N' under nccl)."""

    def __init__(self, rank=0, world=1, dist=None, device="cpu"):
        self.rank, self.world, self.dist, self.device = rank, world, dist, device

    def frames_of(self, frame_count, rank=None):
        r = self.rank if rank is None else rank
        return list(range(r, frame_count, self.world))

    def broadcast_weight(self, w):
        if self.world == 1:
            return np.asarray(w, np.float64)
        import torch
        t = torch.zeros(3, dtype=torch.float64, device=self.device)
        if self.rank == 0:
            t.copy_(torch.as_tensor(np.asarray(w, np.float64)))
        self.dist.broadcast(t, src=0)
        return t.cpu().numpy()

    def max_height(self, local_heights):
        h = int(max(local_heights)) if len(local_heights) else 0
        if self.world == 1:
            return h
        import torch
        t = torch.tensor([h], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def sum_count(self, local_count):
        """Frames of the whole GOF = sum over the ranks of the frames each holds."""
        if self.world == 1:
            return int(local_count)
        import torch
        t = torch.tensor([int(local_count)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather(self, tensor):
        """Gather equally-shaped tensors to rank 0; returns the list on rank 0, None elsewhere."""
        if self.world == 1:
            return [tensor]
        import torch
        raw = tensor.contiguous().view(torch.uint8)          # transport as bytes: collectives lack 16-bit integer types
        out = [torch.empty_like(raw) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(raw, out, dst=0)
        return [o.view(tensor.dtype).view(tensor.shape) for o in out] if out is not None else None

    def pack_gof_records(self, local_records, frame_count, mode, min_w, min_h, tiles_hor=2, ratio=1.0):
        """The inter-frame packers over a GOF whose frames live on several ranks ("gather to rank 0 of the
        per-frame patch table before packing").  local_records: (patch records, block-occupancy pool) of this rank's
        frames, in the order of frames_of(frame_count) -- a few KB per frame.  Rank 0 runs the chain over the GOF in
        frame order (lib.host_pack_gof_records) and every rank gets back the packed lists of its own frames:
        [(patch list, pool, matches, tile width, tile height)], plus the tile sizes of ALL frames (for the GOF canvas)."""
        if self.world == 1:
            res = lib.host_pack_gof_records(local_records, mode, min_w, min_h, tiles_hor, ratio)
            return res, [(r[3], r[4]) for r in res]
        gathered = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(local_records, gathered, dst=0)
        box = [None]
        if self.rank == 0:
            try:                                             # whatever goes wrong here, every rank must leave the
                records = [None] * frame_count               # broadcast below, and with the same error
                for r, recs in enumerate(gathered):
                    mine = self.frames_of(frame_count, r)
                    if len(recs) != len(mine):
                        raise ValueError("rank %d sent %d frames, the GOF of %d frames gives it %d" % (r, len(recs), frame_count, len(mine)))
                    for k, f in enumerate(mine):
                        records[f] = recs[k]
                box[0] = lib.host_pack_gof_records(records, mode, min_w, min_h, tiles_hor, ratio)
            except Exception as e:
                box[0] = e
        self.dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], Exception):
            raise box[0]
        return [box[0][f] for f in self.frames_of(frame_count)], [(r[3], r[4]) for r in box[0]]

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()


class _DevArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


def l3_domains():
    """CPU ids grouped by shared last-level cache (one entry per CCD on EPYC), SMT siblings dropped; [] if unknown."""
    try:
        allowed = os.sched_getaffinity(0)
        groups, seen_core = {}, set()
        for cpu in sorted(allowed):
            base = "/sys/devices/system/cpu/cpu%d/" % cpu
            with open(base + "topology/thread_siblings_list") as f:
                first = int(f.read().replace("-", ",").split(",")[0])
            if first in seen_core and first != cpu:
                continue                                     # second hardware thread of a core already listed
            seen_core.add(first)
            with open(base + "cache/index3/shared_cpu_list") as f:
                key = f.read().strip()
            groups.setdefault(key, []).append(cpu)
        return [g for _, g in sorted(groups.items(), key=lambda kv: kv[1][0])]
    except (OSError, ValueError, AttributeError):
        return []


FEW_FRAMES_START_DELAY_US = 2000   # (what the late half of <= 4 frames in flight waits before it starts: GofEncoder.__init__)


class _Worker(threading.Thread):
    """One host thread per in-flight frame slot.  It owns its tmc2_ctx (created after the thread has been pinned, so
    the context's page-locked staging and scratch land on the thread's NUMA node) and runs every call on it."""

    def __init__(self, index, device, cpu, timing):
        super().__init__(daemon=True)
        self.index, self.device, self.cpu, self.timing = index, device, cpu, timing
        self.jobs, self.ctx, self.ready, self.failed = queue.Queue(), None, threading.Event(), None
        self.start()
        self.ready.wait()
        if self.failed is not None:                           # (no device, no memory ..: the caller hears it, nobody waits for ever)
            raise self.failed

    def run(self):
        if self.cpu is not None:
            try:
                os.sched_setaffinity(0, {self.cpu})          # applies to the calling thread
            except OSError:
                pass
        try:
            self.ctx = lib.Context(self.device)
            self.ctx.set_timing(self.timing)
        except BaseException as e:
            self.failed = e
            return
        finally:
            self.ready.set()
        while True:
            job = self.jobs.get()
            if job is None:
                return
            fn, done = job
            try:
                done.append(("ok", fn()))
            except BaseException as e:                        # handed back to the caller
                done.append(("err", e))
            done.event.set()


class _Done(list):
    def __init__(self):
        super().__init__()
        self.event = threading.Event()


class GofEncoder:
    """Phase A (S0-S16) and phase B (S17-S22) of a GOF on one GPU with `workers` concurrent frames.
    Worker w is pinned to a core of L3 domain (first_domain + w) % domains: the host-resident steps (orientation, the
    host k-d tree builds) are cache- and memory-latency bound, so they are spread over all last-level caches and both
    sockets instead of wherever the scheduler happens to wake them."""

    def __init__(self, device=0, workers=4, iterations=50, bits3d=11, occ_precision=4, min_w=1280, min_h=1280,
                 timing=True, pin=True, first_domain=0, vox_dim=4):
        self.device, self.workers = device, workers
        self.iterations, self.bits3d, self.occ_precision, self.vox_dim = iterations, bits3d, occ_precision, vox_dim
        self.min_w, self.min_h = min_w, min_h
        doms = l3_domains() if pin and workers > 1 and os.environ.get("TMC2_PIN", "1") != "0" else []
        cpus = [None] * workers
        if doms:
            for w in range(workers):
                d = doms[(first_domain + w) % len(doms)]
                cpus[w] = d[((first_domain + w) // len(doms)) % len(d)]
        # few frames in flight (one rank's share of a many-GPU run): the refinement's geometry goes ahead of the orientation's host
        # walk -- it shortens a frame's chain; with the chip full it would only compete (include/tmc2hip.h)
        self.closed = False
        self.threads = [_Worker(w, device, cpus[w], timing) for w in range(workers)]
        self.ctxs = [t.ctx for t in self.threads]
        if os.environ.get("TMC2_REFINE_OVERLAP") is None:      # (an option of THIS encoder's contexts: nothing process-wide)
            self.set_option("REFINE_OVERLAP", 1 if workers <= 4 else 0)
        # ... and the second half of the frames start 2 ms after the first: frames that start together reach S3's host walk (~ 3 ms)
        # together and leave the GPU with nothing to do meanwhile (round 6: profiles/r06_rank_concurrency.txt, r06_rank_stagger.txt)
        if os.environ.get("TMC2_FRAME_START_DELAY_US") is None and 2 <= workers <= 4:
            for c in self.ctxs[(workers + 1) // 2:]:
                c.set_option("FRAME_START_DELAY_US", FEW_FRAMES_START_DELAY_US)
        self.gate = None

    def set_host_slots(self, slots):
        """At most `slots` host-resident steps (orientation walks, host tree builds) of THIS encoder's frames at a time (0: no
        limit): a gate of its own on its contexts (tmc2_host_gate_create / tmc2_ctx_set_host_gate), nothing process-wide."""
        old, self.gate = self.gate, lib.HostGate(slots)
        for c in self.ctxs:
            c.set_host_gate(self.gate)
        if old is not None:
            old.close()

    def set_option(self, key, value):
        """tmc2_ctx_set_option on every context of this encoder (value None: unset)."""
        for c in self.ctxs:
            c.set_option(key, value)

    def reserve(self, max_points, max_w=None, max_h=None):
        """tmc2_ctx_reserve on every context (on its own worker thread): the sequence's largest frame, with this encoder's
        refinement voxels / bit depth; canvas: the minimum one unless the caller knows better."""
        w, h = max_w or self.min_w, max_h or max(self.min_h, self.min_w)
        self._dispatch([(i, (lambda c=c: c.reserve(max_points, self.vox_dim, self.bits3d, w, h))) for i, c in enumerate(self.ctxs)])

    def pool_stats(self):
        tot = {}
        for c in self.ctxs:
            for k, v in c.pool_stats().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def close(self, join=False):
        """Ends the worker threads; join=True also waits for them and closes their contexts (every frame of this encoder must
        have been closed before: a frame's device buffers go back to its context's pool)."""
        self.closed = True
        for t in self.threads:
            t.jobs.put(None)
        if join:
            for t in self.threads:
                t.join()
            for c in self.ctxs:
                c.close()
        if self.gate is not None:
            self.gate.close()
            self.gate = None

    def upload(self, clouds):
        """Untimed: copy the GOF's point arrays to HBM (frame i lives on worker i % workers)."""
        frames = [None] * len(clouds)

        def make(i):
            frames[i] = self.ctxs[i % self.workers].frame(*clouds[i])
        self._dispatch([(i % self.workers, (lambda i=i: make(i))) for i in range(len(clouds))])
        return frames

    def _dispatch(self, jobs):
        """jobs: [(worker, callable)]; calls on one worker run in order; returns the results in job order."""
        per = [[] for _ in range(self.workers)]
        for j, (w, fn) in enumerate(jobs):
            per[w].append((j, fn))
        out, waits = [None] * len(jobs), []
        for w, lst in enumerate(per):
            if not lst:
                continue

            def run(lst=lst):
                for j, fn in lst:
                    out[j] = fn()
            d = _Done()
            if self.closed or not self.threads[w].is_alive():
                raise lib.Tmc2Error("GofEncoder: worker %d has ended (close() was called): nobody would answer this call" % w)
            self.threads[w].jobs.put((run, d))
            waits.append(d)
        for d in waits:
            d.event.wait()
            if d[0][0] == "err":
                raise d[0][1]
        return out

    def _per_worker(self, frames, fn):
        return self._dispatch([(i % self.workers, (lambda fr=fr: fn(fr))) for i, fr in enumerate(frames)])

    def per_frame(self, frames, fn):
        """fn(frame, index) on the frame's own worker thread (e.g. the copies of finished canvases to host memory)."""
        return self._dispatch([(i % self.workers, (lambda fr=fr, i=i: fn(fr, i))) for i, fr in enumerate(frames)])

    def phase_a(self, frames, sharder=None, weight=None, constrained_pack=False, frame_count=None, records_chain=False,
                then=None):
        """constrained_pack: True = the low-delay condition -- frames after the first are packed against their predecessor
        (S10', a sequential chain over the GOF, microseconds per frame on the host); 2 = the random-access condition -- the same chain followed by the global patch allocation over the
        GOF (tracked patches share one place in all frames of a sub-context).  With several ranks (frame f on rank
        f mod world; frame_count = frames of the whole GOF) the chain runs on rank 0 over the gathered patch records and the
        packed lists come back (records_chain=True forces that route in a single process).
        then(frame, index, W, H): what a frame goes on with once the canvas size is settled and its geometry images exist (phase B,
        the copy of its canvases to the host, ...), on the frame's worker in the same pass -- the frames are independent from
        there on, and every rendezvous of the GOF costs the wait for its slowest frame."""
        sharder = sharder or Sharder()
        if constrained_pack and (sharder.world > 1 or records_chain):
            size = self._phase_a_sharded_chain(frames, sharder, weight, int(constrained_pack), frame_count)
            if then is not None:
                self.per_frame(frames, lambda fr, i: then(fr, i, size[0], size[1]))
            return size
        if weight is None:
            w = frames[0].weight_normal(self.bits3d, 0.6) if sharder.rank == 0 else np.zeros(3)
            weight = sharder.broadcast_weight(w)
        params = lib.ctc_params(self.iterations, self.bits3d, weight, self.vox_dim)

        def segment_and_pack(fr):
            fr.segmenter_compute(params)
            return fr.encoder_pack_flexible(self.min_w, 2, 1.0)
        if constrained_pack:
            self._per_worker(frames, lambda fr: fr.segmenter_compute(params))
            # the chain is sequential over the GOF; each call still runs on the worker thread that owns the frame's context
            heights = self._dispatch([(0, lambda: frames[0].encoder_pack_flexible(self.min_w, 2, 1.0))])
            for i in range(1, len(frames)):
                heights += self._dispatch([(i % self.workers, lambda i=i: frames[i].encoder_pack_spatial_consistency(
                    frames[i - 1], self.min_w, 2, 1.0))])
            if constrained_pack == 2:
                widths, heights = lib.encoder_global_patch_allocation(frames, self.min_w, self.min_h)
                W, H = lib.encoder_canvas_size([max(int(max(heights)), self.min_h)], max(int(max(widths)), self.min_w),
                                               self.min_w, self.min_h)
                self.per_frame(frames, lambda fr, i: (fr.encoder_generate_geometry_images(W, H, self.occ_precision),
                                                      then(fr, i, W, H) if then is not None else None))
                return W, H
        else:
            heights = self._per_worker(frames, segment_and_pack)
        gof_h = sharder.max_height(heights)
        # the chained packer writes the width of its canvas back into the tile (a patch wider than the preset width widens it)
        tile_w = max([self.min_w] + [fr.get_packed_size()[0] for fr in frames]) if constrained_pack else self.min_w
        W, H = lib.encoder_canvas_size([gof_h], tile_w, self.min_w, self.min_h)
        self.per_frame(frames, lambda fr, i: (fr.encoder_generate_geometry_images(W, H, self.occ_precision),
                                              then(fr, i, W, H) if then is not None else None))
        return W, H

    def encode_all_intra(self, frames, sharder=None, weight=None, finish=None):
        """Phase A + phase B of a GOF under the all-intra condition with ONE rendezvous.  The frames are independent but for
        the common canvas: its height is the maximum over the GOF of the packed heights (resizeGeometryVideo,
        PCCEncoder.cpp:5546-5591), never below the minimum canvas -- which is what every frame of the CTC sequences packs
        into.  So every frame runs its whole chain (segment, pack, rasterise, reconstruct, colour, attribute images, then
        `finish(frame, index)`, e.g. the copy of its canvases to the host) on its own worker with the canvas size ITS OWN
        height gives; when all are done the real size is settled (max over the ranks) and a frame whose guess was short
        rasterises again on the right canvas.  Same bytes as phase_a() + phase_b(); no worker waits for another mid-GOF."""
        sharder = sharder or Sharder()
        if weight is None:
            w = frames[0].weight_normal(self.bits3d, 0.6) if sharder.rank == 0 else np.zeros(3)
            weight = sharder.broadcast_weight(w)
        params = lib.ctc_params(self.iterations, self.bits3d, weight, self.vox_dim)

        def images(fr, i, size):
            fr.encoder_generate_geometry_images(size[0], size[1], self.occ_precision)
            fr.encoder_generate_attribute_images()
            if finish is not None:
                finish(fr, i, size)

        def chain(fr, i):
            fr.segmenter_compute(params)
            h = fr.encoder_pack_flexible(self.min_w, 2, 1.0)
            size = lib.encoder_canvas_size([h], self.min_w, self.min_w, self.min_h)
            images(fr, i, size)
            return h, size
        done = self.per_frame(frames, chain)
        gof_h = sharder.max_height([h for h, _ in done])
        size = lib.encoder_canvas_size([gof_h], self.min_w, self.min_w, self.min_h)
        redo = [(i, fr) for i, (fr, (_, guess)) in enumerate(zip(frames, done)) if tuple(guess) != tuple(size)]
        if redo:
            self._dispatch([(i % self.workers, (lambda fr=fr, i=i: images(fr, i, size))) for i, fr in redo])
        return size

    def _phase_a_sharded_chain(self, frames, sharder, weight, mode, frame_count):
        if frame_count is None:                              # (uneven shards: the ranks need not hold equally many frames)
            frame_count = sharder.sum_count(len(frames))
        if weight is None:
            w = frames[0].weight_normal(self.bits3d, 0.6) if sharder.rank == 0 else np.zeros(3)
            weight = sharder.broadcast_weight(w)
        params = lib.ctc_params(self.iterations, self.bits3d, weight, self.vox_dim)
        self._per_worker(frames, lambda fr: fr.segmenter_compute(params))
        local = self._per_worker(frames, lambda fr: fr.get_patch_records())
        mine, tiles = sharder.pack_gof_records(local, frame_count, mode, self.min_w, self.min_h)
        self.per_frame(frames, lambda fr, i: fr.set_packing(mine[i][0], mine[i][2], mine[i][1], mine[i][3], mine[i][4]))
        tile_w = max([self.min_w] + [t[0] for t in tiles])
        tile_h = max([t[1] for t in tiles]) if tiles else 0
        if mode == 2:
            tile_h = max(tile_h, self.min_h)
        W, H = lib.encoder_canvas_size([tile_h], tile_w, self.min_w, self.min_h)
        self._per_worker(frames, lambda fr: fr.encoder_generate_geometry_images(W, H, self.occ_precision))
        return W, H

    def phase_b(self, frames):
        """S17-S22 on the resident (decoded == generated) occupancy / geometry canvases."""
        self._per_worker(frames, lambda fr: fr.encoder_generate_attribute_images())

    def phase_c(self, frames, decoded_attribute=None, grid_size=8, threshold=64.0, i420_out=None):
        """What follows the attribute images, per frame: the colour-space conversion to the I420 frames the video encoder
        reads, and the post-reconstruction tail (boundary points, 16-bit colours from the decoded attribute frames, grid
        geometry smoothing, colour transfer onto the moved points, YUV -> RGB).
        decoded_attribute[i]: uint16 [2][3][H][W], the decoded attribute frames after the inverse conversion; None: identity
        video codec -- the I420 frames produced here are converted back on the device (i420_out[i], uint8 [2][H*W*3/2],
        receives them if given)."""
        if decoded_attribute is not None:
            self.per_frame(frames, lambda fr, i: fr.codec_post_reconstruct(decoded_attribute[i], grid_size, threshold))
            return

        def chain(fr, i):
            i420 = fr.encoder_attribute_to_yuv420(4, None if i420_out is None else i420_out[i])
            fr.codec_set_decoded_attribute_yuv420(i420, 0)
            fr.codec_post_reconstruct(None, grid_size, threshold)
        self.per_frame(frames, chain)

    def stage_ms(self):
        tot = {}
        for c in self.ctxs:
            for k, v in c.stage_ms().items():
                tot[k] = tot.get(k, 0.0) + v
        return tot

    def stage_calls(self):
        tot = {}
        for c in self.ctxs:
            for k, v in c.stage_calls().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def stage_reset(self):
        for c in self.ctxs:
            c.stage_reset()

    def device_tensor(self, frame, name):
        """torch view (zero copy) of a frame's canvas as BYTES, shape (*canvas shape, itemsize), for collectives
        (they lack 16/32-bit unsigned types; the receiver reinterprets: see canvas_from_bytes)."""
        import torch
        ptr, shape, typestr = frame.device_images()[name]
        itemsize = int(typestr[2:])
        return torch.as_tensor(_DevArray(ptr, tuple(shape) + (itemsize,), "|u1"), device="cuda:%d" % self.device)

    @staticmethod
    def canvas_from_bytes(tensor, name):
        """numpy canvas from the byte tensor device_tensor() / a gather of it delivers."""
        dtype = {"occupancy": np.uint8, "occ_video": np.uint8, "block_to_patch": np.uint32, "geometry": np.uint16,
                 "attribute": np.uint8}[name]
        a = tensor.cpu().numpy()
        return np.ascontiguousarray(a).view(dtype).reshape(a.shape[:-1])
