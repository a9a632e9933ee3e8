"""The native host of a GOF pass: libtmc2gof.so (include/tmc2gof.h, mpeg-pcc-tmc2_amd/host/gof_runner.cpp) behind ctypes.

GofEncoder (gof.py) drives the C-ABI from Python worker threads; this hands the same pass -- reset, S0, S1-S9 + packing per
frame, one rendezvous for the canvas, S12-S22, the copies of the finished canvases -- to C++ threads in one call, which is what
the reference's own host is (PCCEncoder::encode, PccLibEncoder/source/PCCEncoder.cpp:85-172, with its tbb::parallel_for over the
frames at :4729-4750).  Same C-ABI calls in the same order per frame, so the same bytes; no interpreter between the launches.
"""
import ctypes as C
import os

import numpy as np

from . import lib as _lib

PACKING = {"all-intra": 0, "low-delay": 1, "random-access": 2}
_GOF = None


class GofConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("iterationCountRefineSegmentation", "voxelDimensionRefineSegmentation", "geometryBitDepth3D",
                                         "occupancyPrecision", "minimumImageWidth", "minimumImageHeight", "packing", "guessCanvas")]


def library_path():
    return os.path.join(os.path.dirname(_lib.library_path()), "libtmc2gof.so")


def load_library():
    """Load libtmc2gof.so (after libtmc2hip.so, which it links against); raises if it has not been built."""
    global _GOF
    if _GOF is None:
        _lib.load_library()
        path = library_path()
        if not os.path.exists(path):
            raise _lib.Tmc2Error("libtmc2gof.so not built: run `python __graft_entry__.py build` (make -C mpeg-pcc-tmc2_amd/host)")
        G = C.CDLL(path)
        G.tmc2_gof_last_error.restype = C.c_char_p
        G.tmc2_gof_comm_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_void_p]
        G.tmc2_gof_comm_destroy.argtypes = [C.c_void_p]
        G.tmc2_gof_comm_destroy.restype = None
        _GOF = G
    return _GOF


class Comm:
    """The ranks of a sharded GOF (tmc2_gof_comm: RCCL from C++, include/tmc2gof.h).  ctx: a lib.Context on this rank's device.
    rank / world default to the launcher's RANK / WORLD_SIZE; the communicator's id travels through a file rank 0 writes
    (rendezvous, default /dev/shm/tmc2_gof_id_$MASTER_PORT).  Creation ends with a checked all-reduce."""

    def __init__(self, ctx, rank=None, world=None, rendezvous=None):
        G = load_library()
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.h = C.c_void_p()
        rc = G.tmc2_gof_comm_create(self.rank, self.world, ctx.h, None if rendezvous is None else rendezvous.encode(), C.byref(self.h))
        if rc != 0:
            raise _lib.Tmc2Error("tmc2_gof_comm_create failed (%d): %s" % (rc, G.tmc2_gof_last_error().decode()))

    def close(self):
        if self.h:
            load_library().tmc2_gof_comm_destroy(self.h)
            self.h = C.c_void_p()


def _pointers(arrays, dtype):
    if arrays is None:
        return None
    out = (C.c_void_p * len(arrays))()
    for i, x in enumerate(arrays):
        if x is not None:
            assert x.dtype == dtype and x.flags["C_CONTIGUOUS"]
            out[i] = x.ctypes.data
    return out


class CanvasTooSmall(_lib.Tmc2Error):
    def __init__(self, W, H):
        super().__init__("the GOF needs a %d x %d canvas" % (W, H))
        self.size = (W, H)


def encode(frames, slot_of, slots, iterations, vox_dim, bits3d, precision, min_w, min_h, packing, out, capacity, guess_canvas=False,
           resume=False):
    """One pass over the GOF (tmc2_gof_encode).  frames: lib.Frame; slot_of[i]: the host thread of frame i (the frames of one
    context on one slot); out: per frame (dict(occupancy, occ_video, block_to_patch, geo0, geo1), attribute) numpy buffers sized
    for capacity = (W, H), or None (nothing leaves the device).  Returns (W, H); raises CanvasTooSmall with the size the GOF needs --
    the caller then calls again with resume=True and buffers of that size (tmc2_gof_encode_resume: the second half of the same
    pass; S0-S10 are not repeated)."""
    G = load_library()
    n = len(frames)
    cfg = GofConfig(iterations, vox_dim, bits3d, precision, min_w, min_h, PACKING[packing] if isinstance(packing, str) else int(packing),
                    int(bool(guess_canvas)))
    handles = (C.c_void_p * n)(*[fr.h.value for fr in frames])
    slot = (C.c_int32 * n)(*[int(s) for s in slot_of])
    col = (lambda k, dt: _pointers([o[0][k] for o in out], dt)) if out is not None else (lambda k, dt: None)
    W, H = C.c_int32(0), C.c_int32(0)
    fn = G.tmc2_gof_encode_resume if resume else G.tmc2_gof_encode
    rc = fn(handles, slot, n, int(slots), C.byref(cfg), col("occupancy", np.uint8), col("occ_video", np.uint8),
            col("block_to_patch", np.uint32), col("geo0", np.uint16), col("geo1", np.uint16),
            _pointers([o[1] for o in out], np.uint8) if out is not None else None,
            int(capacity[0]), int(capacity[1]), C.byref(W), C.byref(H))
    if rc != 0:
        if W.value > capacity[0] or H.value > capacity[1]:
            raise CanvasTooSmall(W.value, H.value)
        raise _lib.Tmc2Error("tmc2gof error %d: %s" % (rc, G.tmc2_gof_last_error().decode()))
    for fr in frames:                                         # (what Frame.encoder_generate_geometry_images notes for its getters)
        fr._canvas = (W.value, H.value, int(precision))
    return W.value, H.value


def encode_sharded(comm, frames, slot_of, slots, iterations, vox_dim, bits3d, precision, min_w, min_h, out, capacity, record_slots=1024,
                   packing="all-intra", resume=False):
    """One pass over THIS rank's frames of a sharded GOF (tmc2_gof_encode_sharded): weights from rank 0, canvases into `out` (this
    rank's page-locked / shared buffers).  all-intra: canvas height all-reduced, the packed patch records of every frame gathered
    to rank 0; low-delay / random-access: records and pools to rank 0, the chain there (tmc2_host_place_segments), packed lists back.
    resume: after CanvasTooSmall, the second half of the same pass (tmc2_gof_encode_sharded_resume) -- on every rank.
    Returns (W, H, records): records = on rank 0 a list [rank][frame] of PATCH_DTYPE arrays in list order, else None."""
    G = load_library()
    n = len(frames)
    cfg = GofConfig(iterations, vox_dim, bits3d, precision, min_w, min_h, PACKING[packing] if isinstance(packing, str) else int(packing), 0)
    handles = (C.c_void_p * n)(*[fr.h.value for fr in frames])
    slot = (C.c_int32 * n)(*[int(s) for s in slot_of])
    col = (lambda k, dt: _pointers([o[0][k] for o in out], dt)) if out is not None else (lambda k, dt: None)
    W, H = C.c_int32(0), C.c_int32(0)
    gathered = counts = None
    if comm.rank == 0:
        gathered = np.zeros((comm.world, n, record_slots), _lib.PATCH_DTYPE)
        counts = np.zeros((comm.world, n), np.int64)
    fn = G.tmc2_gof_encode_sharded_resume if resume else G.tmc2_gof_encode_sharded
    rc = fn(comm.h, handles, slot, n, int(slots), C.byref(cfg), col("occupancy", np.uint8), col("occ_video", np.uint8),
            col("block_to_patch", np.uint32), col("geo0", np.uint16), col("geo1", np.uint16),
            _pointers([o[1] for o in out], np.uint8) if out is not None else None,
            int(capacity[0]), int(capacity[1]), C.byref(W), C.byref(H), int(record_slots),
            None if gathered is None else C.c_void_p(gathered.ctypes.data),
            None if counts is None else C.c_void_p(counts.ctypes.data))
    if rc != 0:
        if W.value > capacity[0] or H.value > capacity[1]:
            raise CanvasTooSmall(W.value, H.value)
        raise _lib.Tmc2Error("tmc2gof error %d: %s" % (rc, G.tmc2_gof_last_error().decode()))
    for fr in frames:
        fr._canvas = (W.value, H.value, int(precision))
    records = None
    if gathered is not None:
        records = [[gathered[r, i, :int(counts[r, i])].copy() for i in range(n)] for r in range(comm.world)]
    return W.value, H.value, records
