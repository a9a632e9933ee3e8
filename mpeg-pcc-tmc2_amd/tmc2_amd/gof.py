"""Group-of-frames orchestration: the call sequence a PCCEncoder::encode adaptor issues,
frames of a GOF processed concurrently by host worker threads (one tmc2_ctx + HIP stream each -- the
reference does the same with a tbb::parallel_for over frames, PCCEncoder.cpp:4729-4750), and, for
multi-GPU runs, frames sharded over ranks: frame f -> rank f % world.  The only cross-frame coupling of
the all-intra path is (a) the axis weights of frame 0 (24 bytes, broadcast) and (b) the common canvas
size (max over frames, all-reduce); finished canvases are gathered to rank 0 (RCCL over xGMI under the
"nccl" backend, gloo in the CPU tests).  This module holds no algorithmic code."""
import threading
from concurrent.futures import ThreadPoolExecutor

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

    def gather(self, tensor):
        """Gather equally-shaped tensors to rank 0; returns the list on rank 0, None elsewhere."""
        if self.world == 1:
            return [tensor]
        import torch
        raw = tensor.contiguous().view(torch.uint8)          # transport as bytes: collectives lack 16-bit integer types
        out = [torch.empty_like(raw) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(raw, out, dst=0)
        return [o.view(tensor.dtype).view(tensor.shape) for o in out] if out is not None else None

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()


class _DevArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


class GofEncoder:
    """Phase A (S0-S16) and phase B (S17-S22) of a GOF on one GPU with `workers` concurrent frames."""

    def __init__(self, device=0, workers=4, iterations=50, bits3d=11, occ_precision=4, min_w=1280, min_h=1280,
                 timing=True):
        self.device, self.workers = device, workers
        self.iterations, self.bits3d, self.occ_precision = iterations, bits3d, occ_precision
        self.min_w, self.min_h = min_w, min_h
        self.ctxs = [lib.Context(device) for _ in range(workers)]
        for c in self.ctxs:
            c.set_timing(timing)
        self.pool = ThreadPoolExecutor(max_workers=workers)

    def upload(self, clouds):
        """Untimed: copy the GOF's point arrays to HBM (frame i lives on worker i % workers)."""
        return [self.ctxs[i % self.workers].frame(xyz, rgb) for i, (xyz, rgb) in enumerate(clouds)]

    def _per_worker(self, frames, fn):
        buckets = [[] for _ in range(self.workers)]
        for i, fr in enumerate(frames):
            buckets[i % self.workers].append((i, fr))
        out = [None] * len(frames)

        def run(bucket):
            for i, fr in bucket:
                out[i] = fn(fr)
        list(self.pool.map(run, [b for b in buckets if b]))
        return out

    def phase_a(self, frames, sharder=None, weight=None):
        sharder = sharder or Sharder()
        if weight is None:
            w = frames[0].weight_normal(self.bits3d, 0.6) if sharder.rank == 0 else np.zeros(3)
            weight = sharder.broadcast_weight(w)
        params = lib.ctc_params(self.iterations, self.bits3d, weight)

        def segment_and_pack(fr):
            fr.segmenter_compute(params)
            return fr.encoder_pack_flexible(self.min_w, 2, 1.0)
        heights = self._per_worker(frames, segment_and_pack)
        gof_h = sharder.max_height(heights)
        W, H = lib.encoder_canvas_size([gof_h], self.min_w, self.min_w, self.min_h)
        self._per_worker(frames, lambda fr: fr.encoder_generate_geometry_images(W, H, self.occ_precision))
        return W, H

    def phase_b(self, frames):
        """S17-S22 on the resident (decoded == generated) occupancy / geometry canvases."""
        self._per_worker(frames, lambda fr: fr.encoder_generate_attribute_images())

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
        """torch view (zero copy) of a frame's canvas, for collectives."""
        import torch
        ptr, shape, typestr = frame.device_images()[name]
        return torch.as_tensor(_DevArray(ptr, shape, typestr), device="cuda:%d" % self.device)
