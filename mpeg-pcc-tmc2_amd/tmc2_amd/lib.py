"""ctypes binding of libtmc2hip.so (include/tmc2hip.h).  Class/method names mirror the reference's
seams: Frame.normals_compute <-> PCCNormalsGenerator3::compute, Frame.segmenter_* <-> PCCPatchSegmenter3::*,
Frame.kdtree_search <-> PCCKdTree::search."""
import ctypes as C
import os as _os

# One HIP stream per in-flight frame (32 per GPU in a GOF run): the runtime multiplexes streams onto 4 hardware queues by
# default, and a queue is held by whatever kernel is at its head (e.g. a 140 us single-workgroup closure tail).  Must be
# set before the HIP runtime initialises; an explicit setting of the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import os
import threading
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def library_path():
    return os.path.normpath(os.path.join(_HERE, "..", "libtmc2hip.so"))


class Tmc2Error(RuntimeError):
    pass


class SegmenterParams(C.Structure):
    """Mirror of tmc2_segmenter_params (PCCPatchSegmenter3Parameters, PCCPatchSegmenter.h:48-100)."""
    _fields_ = [(n, C.c_int32) for n in (
        "nnNormalEstimation", "normalOrientation", "gridBasedRefineSegmentation", "maxNNCountRefineSegmentation",
        "iterationCountRefineSegmentation", "voxelDimensionRefineSegmentation", "searchRadiusRefineSegmentation",
        "occupancyResolution", "enablePatchSplitting", "maxPatchSize", "quantizerSizeX", "quantizerSizeY",
        "minPointCountPerCCPatchSegmentation", "maxNNCountPatchSegmentation", "surfaceThickness", "mapCountMinus1",
        "minLevel", "maxAllowedDepth", "geometryBitDepth2D", "geometryBitDepth3D")] + [
        ("maxAllowedDist2RawPointsDetection", C.c_double), ("maxAllowedDist2RawPointsSelection", C.c_double),
        ("lambdaRefineSegmentation", C.c_double), ("weightNormal", C.c_double * 3)]


class Patch(C.Structure):
    """Mirror of tmc2_patch (PCCPatch fields produced by the hot path)."""
    _fields_ = [(n, C.c_int32) for n in (
        "index", "viewId", "normalAxis", "tangentAxis", "bitangentAxis", "projectionMode", "u1", "v1", "d1", "sizeU",
        "sizeV", "sizeD", "sizeDPixel", "sizeU0", "sizeV0", "size2DXInPixel", "size2DYInPixel", "d0Count",
        "eomAndD1Count", "u0", "v0", "patchOrientation")] + [("depthOffset", C.c_int64), ("occOffset", C.c_int64)]


PATCH_DTYPE = np.dtype([(n, t) for n, t in [(f[0], np.int32) for f in Patch._fields_[:22]] +
                        [("depthOffset", np.int64), ("occOffset", np.int64)]])


def ctc_params(iterations=10, bits3d=11, weight=(1.0, 1.0, 1.0), vox_dim=4):
    """CTC lossy settings (cfg/common/ctc-common.cfg + sequence cfg).  Per sequence: iterations 50
    (longdress), 20 (basketball_player), 10 (the others); vox_dim = voxelDimensionRefineSegmentation: 4, but 2 for loot,
    redandblack and soldier."""
    p = SegmenterParams()
    p.nnNormalEstimation = 16
    p.normalOrientation = 1
    p.gridBasedRefineSegmentation = 1
    p.maxNNCountRefineSegmentation = 1024
    p.iterationCountRefineSegmentation = iterations
    p.voxelDimensionRefineSegmentation = vox_dim
    p.searchRadiusRefineSegmentation = 192
    p.occupancyResolution = 16
    p.enablePatchSplitting = 1
    p.maxPatchSize = 1024
    p.quantizerSizeX = 16
    p.quantizerSizeY = 16
    p.minPointCountPerCCPatchSegmentation = 16
    p.maxNNCountPatchSegmentation = 16
    p.surfaceThickness = 4
    p.mapCountMinus1 = 1
    p.minLevel = 64
    p.maxAllowedDepth = 255
    p.geometryBitDepth2D = 8
    p.geometryBitDepth3D = bits3d
    p.maxAllowedDist2RawPointsDetection = 9.0
    p.maxAllowedDist2RawPointsSelection = 1.0
    p.lambdaRefineSegmentation = 3.0
    p.weightNormal[0], p.weightNormal[1], p.weightNormal[2] = weight
    return p


# TMC2_GUARD=1 (tests, bench.py --guard): every array handed to the C-ABI travels in a copy with a red zone of _GUARD bytes on
# both sides, filled with a pattern; after the call (_check) the zones are verified -- the library wrote outside a buffer of the
# caller's iff they changed -- and the payload is copied back.  Calls are synchronous on host buffers (the getters end with a
# stream synchronisation), so the copy-back sees the final bytes.  Costs a copy per argument: a debugging aid, not a mode to time.
_GUARD = 4096
_guard_on = os.environ.get("TMC2_GUARD", "0") == "1"
_guard_tls = threading.local()


def set_guard(on):
    global _guard_on
    _guard_on = bool(on)


def _ptr(a):
    if not _guard_on or a.nbytes == 0:
        return a.ctypes.data_as(C.c_void_p)
    if not a.flags["C_CONTIGUOUS"]:
        raise Tmc2Error("guard mode: array handed to the C-ABI is not contiguous")
    g = np.full(a.nbytes + 2 * _GUARD, 0xA5, np.uint8)
    g[_GUARD:_GUARD + a.nbytes] = a.reshape(-1).view(np.uint8)
    if not hasattr(_guard_tls, "pending"):
        _guard_tls.pending = []
    _guard_tls.pending.append((a, g))
    return C.c_void_p(g.ctypes.data + _GUARD)


def _guard_finish():
    pending, _guard_tls.pending = getattr(_guard_tls, "pending", []), []
    for a, g in pending:
        n = a.nbytes
        lo, hi = g[:_GUARD], g[_GUARD + n:]
        if (lo != 0xA5).any() or (hi != 0xA5).any():
            before, after = int((lo != 0xA5).sum()), int((hi != 0xA5).sum())
            first = int(np.argmax(hi != 0xA5)) if after else -1
            raise Tmc2Error("guard mode: the library wrote outside a caller's buffer (%s %s, %d bytes): %d bytes changed before it, "
                            "%d after it (first at +%d)" % (a.dtype, a.shape, n, before, after, first))
        if a.flags["WRITEABLE"]:
            a.reshape(-1).view(np.uint8)[:] = g[_GUARD:_GUARD + n]


def load_library():
    """Load libtmc2hip.so; raises if it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise Tmc2Error("libtmc2hip.so not built: run `python __graft_entry__.py build` (hipcc, gfx950)")
    L = C.CDLL(path)
    L.tmc2_last_error.restype = C.c_char_p
    L.tmc2_set_kdtree_placement.restype = None
    L.tmc2_set_refine_overlap.restype = None
    L.tmc2_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.tmc2_ctx_reserve.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.tmc2_ctx_device_alloc.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.tmc2_ctx_device_free.argtypes = [C.c_void_p, C.c_void_p]
    L.tmc2_ctx_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.tmc2_ctx_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.tmc2_ctx_stream.argtypes = [C.c_void_p]
    L.tmc2_ctx_device.argtypes = [C.c_void_p]
    L.tmc2_ctx_make_current.argtypes = [C.c_void_p]
    L.tmc2_ctx_pool_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tmc2_ctx_get_option.argtypes = [C.c_void_p, C.c_char_p]
    L.tmc2_ctx_get_option.restype = C.c_char_p
    L.tmc2_set_host_parallelism.restype = None
    L.tmc2_host_gate_create.argtypes = [C.c_int, C.c_void_p]
    L.tmc2_host_gate_destroy.argtypes = [C.c_void_p]
    L.tmc2_host_gate_destroy.restype = None
    L.tmc2_ctx_set_host_gate.argtypes = [C.c_void_p, C.c_void_p]
    L.tmc2_ctx_stage_name.restype = C.c_char_p
    L.tmc2_ctx_stage_ms.restype = C.c_double
    L.tmc2_frame_point_count.restype = C.c_uint64
    _LIB = L
    return L


def _check(rc):
    if _guard_on:
        _guard_finish()
    if rc != 0:
        raise Tmc2Error("tmc2hip error %d: %s" % (rc, load_library().tmc2_last_error().decode()))


class HostGate:
    """One encoder's budget of concurrently running host-resident steps (tmc2_host_gate_create): shared by the contexts it is set
    on (Context.set_host_gate) and by nobody else -- two encoders of one process do not draw on one process-wide count."""

    def __init__(self, limit):
        self.L = load_library()
        self.h = C.c_void_p()
        _check(self.L.tmc2_host_gate_create(int(limit), C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.tmc2_host_gate_destroy(self.h)       # (contexts that still hold the gate keep it alive)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    def __init__(self, device=0):
        self.L = load_library()
        self.h = C.c_void_p()
        _check(self.L.tmc2_ctx_create(int(device), C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.tmc2_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self.L.tmc2_ctx_synchronize(self.h))

    def set_option(self, key, value):
        """Per-context option (include/tmc2hip.h: tmc2_ctx_set_option); value None unsets it.  A context starts with the TMC2_*
        variables of the environment it was created in."""
        _check(self.L.tmc2_ctx_set_option(self.h, key.encode(), None if value is None else str(value).encode()))

    def get_option(self, key):
        v = self.L.tmc2_ctx_get_option(self.h, key.encode())
        return None if v is None else v.decode()

    def set_host_gate(self, gate):
        """This context's host-resident steps wait at `gate` (a HostGate; None: the process' default gate again)."""
        _check(self.L.tmc2_ctx_set_host_gate(self.h, None if gate is None else gate.h))

    def reserve(self, max_points, vox_dim, bits3d, max_w, max_h):
        """tmc2_ctx_reserve: the device memory of the sequence's largest frame, allocated now (no first-use hipMalloc later)."""
        _check(self.L.tmc2_ctx_reserve(self.h, C.c_uint64(int(max_points)), int(vox_dim), int(bits3d), int(max_w), int(max_h)))

    # device staging (what libtmc2gof.so's RCCL mode uses between its collectives): raw device memory of this context's device,
    # copies ordered on its stream
    def device_alloc(self, nbytes):
        p = C.c_void_p()
        _check(self.L.tmc2_ctx_device_alloc(self.h, C.c_size_t(int(nbytes)), C.byref(p)))
        return p

    def device_free(self, p):
        _check(self.L.tmc2_ctx_device_free(self.h, p))

    def upload(self, device_ptr, array):
        a = np.ascontiguousarray(array)
        _check(self.L.tmc2_ctx_upload(self.h, device_ptr, _ptr(a), C.c_size_t(a.nbytes)))

    def download(self, array, device_ptr):
        _check(self.L.tmc2_ctx_download(self.h, _ptr(array), device_ptr, C.c_size_t(array.nbytes)))
        return array

    def stream(self):
        self.L.tmc2_ctx_stream.restype = C.c_void_p
        return self.L.tmc2_ctx_stream(self.h)

    def device(self):
        return int(self.L.tmc2_ctx_device(self.h))

    def make_current(self):
        _check(self.L.tmc2_ctx_make_current(self.h))

    def pool_stats(self):
        held, calls, carved, ms = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_double()
        _check(self.L.tmc2_ctx_pool_stats(self.h, C.byref(held), C.byref(calls), C.byref(ms), C.byref(carved)))
        return {"bytes_held": held.value, "hipmalloc_calls": calls.value, "hipmalloc_ms": ms.value, "carved_blocks": carved.value}

    def stage_ms(self):
        return {self.L.tmc2_ctx_stage_name(self.h, i).decode(): self.L.tmc2_ctx_stage_ms(self.h, i)
                for i in range(self.L.tmc2_ctx_stage_count(self.h))}

    def stage_calls(self):
        self.L.tmc2_ctx_stage_calls.restype = C.c_long
        return {self.L.tmc2_ctx_stage_name(self.h, i).decode(): self.L.tmc2_ctx_stage_calls(self.h, i)
                for i in range(self.L.tmc2_ctx_stage_count(self.h))}

    def stage_reset(self):
        self.L.tmc2_ctx_stage_reset(self.h)

    def set_timing(self, enabled):
        self.L.tmc2_ctx_set_timing(self.h, 1 if enabled else 0)

    def frame(self, xyz, rgb=None):
        return Frame(self, xyz, rgb)

    # PCCPointSet3::transferColors
    def transfer_colors(self, src_xyz, src_rgb, tgt_xyz):
        a = np.ascontiguousarray(src_xyz, np.int16)
        b = np.ascontiguousarray(src_rgb, np.uint8)
        c = np.ascontiguousarray(tgt_xyz, np.int16)
        if a.shape != b.shape:                                    # (the C entry takes ONE count for both: a short colour array is read past its end)
            raise Tmc2Error("transfer_colors: %d source points, %d source colours" % (len(a), len(b)))
        out = np.zeros((len(c), 3), np.uint8)
        _check(self.L.tmc2_transfer_colors(self.h, _ptr(a), _ptr(b), C.c_uint64(len(a)), _ptr(c), C.c_uint64(len(c)), _ptr(out)))
        return out

    # PCCMetrics::compute (one frame)
    def decoder_frame(self, patches, width, height, occ_precision, occ_video, geometry):
        """A frame without a source cloud: decoded patch records (list order), occupancy video, geometry maps [2][H][W]."""
        pt = np.ascontiguousarray(patches, dtype=PATCH_DTYPE)
        ov = np.ascontiguousarray(occ_video, dtype=np.uint8)
        geo = np.ascontiguousarray(geometry, dtype=np.uint16)
        if ov.shape != (height // occ_precision, width // occ_precision) or geo.shape != (2, height, width):
            raise ValueError("decoded canvases do not match %dx%d at precision %d" % (width, height, occ_precision))
        fr = Frame.__new__(Frame)
        fr.L, fr.ctx, fr.h = self.L, self, C.c_void_p()
        fr._xyz, fr._rgb, fr.n = None, None, 0
        _check(self.L.tmc2_decoder_frame_create(self.h, _ptr(pt), len(pt), int(width), int(height), int(occ_precision), _ptr(ov),
                                               _ptr(geo), C.byref(fr.h)))
        fr._canvas = (int(width), int(height), int(occ_precision))
        return fr

    # PCCInternalColorConverter (the attribute video's colour-space conversion)
    def color_convert_rgb444_to_yuv420(self, rgb, downsampling_filter=4):
        """rgb uint8 [3][H][W] -> (y [H][W], u, v [H/2][W/2])"""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        _, H, W = rgb.shape
        out = np.zeros(W * H * 3 // 2, np.uint8)
        _check(self.L.tmc2_color_convert_rgb444_to_yuv420(self.h, _ptr(rgb), int(W), int(H), int(downsampling_filter), _ptr(out)))
        a = W * H
        return out[:a].reshape(H, W), out[a:a + a // 4].reshape(H // 2, W // 2), out[a + a // 4:].reshape(H // 2, W // 2)

    def color_convert_yuv420_to_yuv444(self, y, u, v, upsampling_filter=0):
        """-> uint16 [3][H][W]"""
        H, W = y.shape
        src = np.concatenate([np.ascontiguousarray(a, dtype=np.uint8).reshape(-1) for a in (y, u, v)])
        out = np.zeros((3, H, W), np.uint16)
        _check(self.L.tmc2_color_convert_yuv420_to_yuv444(self.h, _ptr(src), int(W), int(H), int(upsampling_filter), _ptr(out)))
        return out

    def metrics_compute(self, src_xyz, src_rgb, rec_xyz, rec_rgb, normals=None, resolution=1023.0):
        a = np.ascontiguousarray(src_xyz, np.int16)
        b = np.ascontiguousarray(src_rgb, np.uint8)
        c = np.ascontiguousarray(rec_xyz, np.int16)
        d = np.ascontiguousarray(rec_rgb, np.uint8)
        nm = None if normals is None else np.ascontiguousarray(normals, np.float64)
        out = np.zeros((3, 8), np.float64)
        counts = np.zeros(2, np.int64)
        _check(self.L.tmc2_metrics_compute(self.h, _ptr(a), _ptr(b), C.c_uint64(len(a)), _ptr(c), _ptr(d), C.c_uint64(len(c)),
                                           None if nm is None else _ptr(nm), C.c_double(resolution), _ptr(out), _ptr(counts)))
        return out, counts

    def metrics_ordered_sums(self, terms_a, terms_b):
        """The ordered sums of per-point terms [n][5] (tmc2_metrics_ordered_sums): out[10], five per direction."""
        a = np.ascontiguousarray(terms_a, np.float64).reshape(-1, 5)
        b = np.ascontiguousarray(terms_b, np.float64).reshape(-1, 5)
        out = np.zeros(10, np.float64)
        _check(self.L.tmc2_metrics_ordered_sums(self.h, _ptr(a) if len(a) else None, C.c_uint64(len(a)),
                                                _ptr(b) if len(b) else None, C.c_uint64(len(b)), _ptr(out)))
        return out


class Frame:
    """Device-resident state of one point-cloud frame (tmc2_frame)."""

    def __init__(self, ctx, xyz, rgb=None):
        self.ctx, self.L = ctx, ctx.L
        xyz = np.ascontiguousarray(xyz, dtype=np.int16)
        assert xyz.ndim == 2 and xyz.shape[1] == 3
        self.n = len(xyz)
        self._xyz = xyz
        self._rgb = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
        self.h = C.c_void_p()
        _check(self.L.tmc2_frame_create(ctx.h, _ptr(xyz), None if rgb is None else _ptr(self._rgb),
                                        C.c_uint64(self.n), C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.tmc2_frame_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _check(self.L.tmc2_frame_reset(self.h))

    def metrics_compute(self, which=0, use_normals=True, resolution=1023.0):
        """S23 on the frame's resident clouds (nothing is uploaded): source = the frame, reconstruction = which 0: the cloud of
        the attribute-image step, 1: the finished cloud of the post-reconstruction tail.  Returns (q[3][8], counts[2])."""
        out = np.zeros((3, 8), np.float64)
        counts = np.zeros(2, np.int64)
        _check(self.L.tmc2_metrics_compute_frame(self.h, C.c_int(which), C.c_int(1 if use_normals else 0), C.c_double(resolution),
                                                 _ptr(out), _ptr(counts)))
        return out, counts

    def metrics_compute_source(self, src_xyz, src_rgb, normals=None, which=1, resolution=1023.0):
        """S23 of a source cloud held on the host against this frame's resident reconstruction (the decoder side)."""
        a = np.ascontiguousarray(src_xyz, np.int16)
        b = np.ascontiguousarray(src_rgb, np.uint8)
        nm = None if normals is None else np.ascontiguousarray(normals, np.float64)
        out = np.zeros((3, 8), np.float64)
        counts = np.zeros(2, np.int64)
        _check(self.L.tmc2_metrics_compute_frame_source(self.h, C.c_int(which), _ptr(a), _ptr(b), C.c_uint64(len(a)),
                                                        None if nm is None else _ptr(nm), C.c_double(resolution), _ptr(out), _ptr(counts)))
        return out, counts

    def kdtree_build(self):
        _check(self.L.tmc2_kdtree_build(self.h))

    def kdtree_order(self):
        """(perm, depth): the permutation the tree build leaves behind (tree order -> point index) and the levels."""
        perm = np.empty(self.n, np.uint32)
        depth = C.c_int32()
        _check(self.L.tmc2_frame_get_kdtree_order(self.h, _ptr(perm), C.byref(depth)))
        return perm, depth.value

    def device_images(self):
        """Device addresses + shapes of the canvases (for RCCL hand-off): dict name -> (ptr, shape, typestr)."""
        W, H, p = self._canvas
        ptrs = [C.c_void_p() for _ in range(4)]
        _check(self.L.tmc2_frame_device_images(self.h, *[C.byref(x) for x in ptrs]))
        out = dict(occupancy=(ptrs[0].value, (H, W), "|u1"), occ_video=(ptrs[1].value, (H // p, W // p), "|u1"),
                   block_to_patch=(ptrs[2].value, (H // 16, W // 16), "<u4"), geometry=(ptrs[3].value, (2, H, W), "<u2"))
        a = C.c_void_p()
        if self.L.tmc2_frame_device_attribute(self.h, C.byref(a)) == 0:
            out["attribute"] = (a.value, (2, 3, H, W), "|u1")
        return out

    # PCCKdTree::search
    def kdtree_search(self, queries, k, with_dist=False):
        q = np.ascontiguousarray(queries, dtype=np.int16)
        idx = np.empty((len(q), k), np.uint32)
        d = np.empty((len(q), k), np.uint32) if with_dist else None
        _check(self.L.tmc2_kdtree_search(self.h, _ptr(q), C.c_uint64(len(q)), int(k), _ptr(idx),
                                         None if d is None else _ptr(d)))
        return (idx, d) if with_dist else idx

    # PCCNormalsGenerator3
    def normals_compute_normals(self, k=16):
        _check(self.L.tmc2_normals_compute_normals(self.h, int(k)))

    def normals_orient(self):
        _check(self.L.tmc2_normals_orient(self.h))

    def normals_compute(self, k=16, orientation=1):
        _check(self.L.tmc2_normals_compute(self.h, int(k), int(orientation)))

    def get_normals(self):
        out = np.empty((self.n, 3), np.float64)
        _check(self.L.tmc2_frame_get_normals(self.h, _ptr(out)))
        return out

    def set_normals(self, normals):
        a = np.ascontiguousarray(normals, dtype=np.float64)
        _check(self.L.tmc2_frame_set_normals(self.h, _ptr(a)))

    def get_adjacency(self, k=16):
        out = np.empty((self.n, k), np.uint32)
        _check(self.L.tmc2_frame_get_adjacency(self.h, _ptr(out)))
        return out

    # PCCEncoder::calculateWeightNormal
    def weight_normal(self, bits3d=11, min_weight_epp=0.6):
        w = (C.c_double * 3)()
        _check(self.L.tmc2_weight_normal(self.h, int(bits3d), C.c_double(min_weight_epp), w))
        return np.array([w[0], w[1], w[2]])

    # PCCPatchSegmenter3
    def segmenter_initial_segmentation(self, weight):
        w = (C.c_double * 3)(*[float(x) for x in weight])
        _check(self.L.tmc2_segmenter_initial_segmentation(self.h, w))

    def segmenter_refine_grid_based(self, max_nn=1024, lam=3.0, iterations=10, vox_dim=4, radius=192):
        _check(self.L.tmc2_segmenter_refine_grid_based(self.h, int(max_nn), C.c_double(lam), int(iterations),
                                                       int(vox_dim), int(radius)))

    def get_partition(self):
        out = np.empty(self.n, np.uint32)
        _check(self.L.tmc2_frame_get_partition(self.h, _ptr(out)))
        return out

    def set_partition(self, partition):
        a = np.ascontiguousarray(partition, dtype=np.uint32)
        _check(self.L.tmc2_frame_set_partition(self.h, _ptr(a)))

    def segmenter_segment_patches(self, params):
        _check(self.L.tmc2_segmenter_segment_patches(self.h, C.byref(params)))

    def segmenter_compute(self, params):
        _check(self.L.tmc2_segmenter_compute(self.h, C.byref(params)))

    def get_patches(self):
        cnt = self.L.tmc2_frame_patch_count(self.h)
        dc, oc = C.c_int64(), C.c_int64()
        _check(self.L.tmc2_frame_patch_pool_sizes(self.h, C.byref(dc), C.byref(oc)))
        patches = np.zeros(cnt, PATCH_DTYPE)
        d0 = np.zeros(dc.value, np.int16)
        d1 = np.zeros(dc.value, np.int16)
        occ = np.zeros(oc.value, np.uint8)
        _check(self.L.tmc2_frame_get_patches(self.h, _ptr(patches), _ptr(d0), _ptr(d1), _ptr(occ)))
        return patches, d0, d1, occ


    def get_patch_records(self):
        """(patch records, block-occupancy pool) without the depth pools: what the inter-frame packers work on."""
        cnt = self.L.tmc2_frame_patch_count(self.h)
        dc, oc = C.c_int64(), C.c_int64()
        _check(self.L.tmc2_frame_patch_pool_sizes(self.h, C.byref(dc), C.byref(oc)))
        patches, occ = np.zeros(cnt, PATCH_DTYPE), np.zeros(oc.value, np.uint8)
        _check(self.L.tmc2_frame_get_patches(self.h, _ptr(patches), None, None, _ptr(occ)))
        return patches, occ

    def set_packing(self, patch_list, matches, occupancy, packed_width, packed_height):
        """Install a packed patch list (list order) computed elsewhere -- on the rank that runs the inter-frame chain."""
        p = np.ascontiguousarray(patch_list, dtype=PATCH_DTYPE)
        occ = np.ascontiguousarray(occupancy, dtype=np.uint8)
        m = None if matches is None else np.ascontiguousarray(matches, dtype=np.int32)
        _check(self.L.tmc2_frame_set_packing(self.h, _ptr(p), len(p), None if m is None else _ptr(m), _ptr(occ), C.c_int64(len(occ)),
                                             int(packed_width), int(packed_height)))

    # PCCEncoder image generation, phase A
    def encoder_pack_flexible(self, preset_width=1280, tiles_hor=2, ratio=1.0):
        h = C.c_int32()
        _check(self.L.tmc2_encoder_pack_flexible(self.h, int(preset_width), int(tiles_hor), C.c_double(ratio), C.byref(h)))
        return h.value

    def encoder_pack_spatial_consistency(self, previous, preset_width=1280, tiles_hor=2, ratio=1.0):
        """S10' (constrainedPack): pack against the previous frame of the GOF (which must be packed already)."""
        h = C.c_int32()
        _check(self.L.tmc2_encoder_pack_spatial_consistency(self.h, previous.h, int(preset_width), int(tiles_hor),
                                                            C.c_double(ratio), C.byref(h)))
        return h.value

    def get_packed_size(self):
        """(width, height) of the tile as the packers left it."""
        w, h = C.c_int32(), C.c_int32()
        _check(self.L.tmc2_frame_get_packed_size(self.h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def get_patch_matches(self):
        m = np.zeros(self.L.tmc2_frame_patch_count(self.h), np.int32)
        _check(self.L.tmc2_frame_get_patch_matches(self.h, _ptr(m)))
        return m

    def get_patch_order(self):
        order = np.zeros(self.L.tmc2_frame_patch_count(self.h), np.int32)
        _check(self.L.tmc2_frame_get_patch_order(self.h, _ptr(order)))
        return order

    def encoder_generate_geometry_images(self, width, height, occ_precision=4):
        _check(self.L.tmc2_encoder_generate_geometry_images(self.h, int(width), int(height), int(occ_precision)))
        self._canvas = (int(width), int(height), int(occ_precision))

    def get_geometry_images(self, out=None):
        W, H, p = self._canvas
        if out is None:
            out = dict(occupancy=np.zeros((H, W), np.uint8), occ_video=np.zeros((H // p, W // p), np.uint8),
                       block_to_patch=np.zeros((H // 16, W // 16), np.uint32), geo0=np.zeros((H, W), np.uint16),
                       geo1=np.zeros((H, W), np.uint16))
        _check(self.L.tmc2_frame_get_geometry_images(self.h, _ptr(out["occupancy"]), _ptr(out["occ_video"]),
                                                     _ptr(out["block_to_patch"]), _ptr(out["geo0"]), _ptr(out["geo1"])))
        return out


    # PCCEncoder image generation, phase B
    def encoder_generate_attribute_images(self):
        _check(self.L.tmc2_encoder_generate_attribute_images(self.h))

    def recon_count(self):
        """Points of the reconstructed cloud (0 before the reconstruction)."""
        self.L.tmc2_frame_recon_count.restype = C.c_int64
        return int(self.L.tmc2_frame_recon_count(self.h))

    def get_reconstruction(self, colors=True):
        self.L.tmc2_frame_recon_count.restype = C.c_int64
        M = self.L.tmc2_frame_recon_count(self.h)
        xyz, rgb, p2p = np.zeros((M, 3), np.int16), (np.zeros((M, 3), np.uint8) if colors else None), np.zeros((M, 3), np.uint32)
        _check(self.L.tmc2_frame_get_reconstruction(self.h, _ptr(xyz), None if rgb is None else _ptr(rgb), _ptr(p2p)))
        return xyz, rgb, p2p

    def codec_generate_point_cloud(self):
        """PCCCodec::generatePointCloud alone (no colour transfer, no attribute images)."""
        _check(self.L.tmc2_codec_generate_point_cloud(self.h))

    # post-reconstruction tail (PCCEncoder::encode :571-719 / PCCDecoder::decode :330-470)
    def codec_identify_boundary_points(self):
        _check(self.L.tmc2_codec_identify_boundary_points(self.h))

    def encoder_attribute_to_yuv420(self, downsampling_filter=4, out=None):
        """The two attribute canvases as the I420 frames the video encoder reads: uint8 [2][H*W*3/2]."""
        W, H, _ = self._canvas
        if out is None:
            out = np.zeros((2, W * H * 3 // 2), np.uint8)
        _check(self.L.tmc2_encoder_attribute_to_yuv420(self.h, int(downsampling_filter), _ptr(out)))
        return out

    def codec_set_decoded_attribute_yuv420(self, yuv420, upsampling_filter=0):
        """The two decoded I420 attribute frames (uint8 [2][H*W*3/2]) -> 16-bit 4:4:4 planes kept on the device."""
        W, H, _ = self._canvas
        src = np.ascontiguousarray(yuv420, dtype=np.uint8)
        if src.size != 2 * (W * H * 3 // 2):
            raise ValueError("decoded I420 frames must hold 2 x %d bytes" % (W * H * 3 // 2))
        _check(self.L.tmc2_codec_set_decoded_attribute_yuv420(self.h, _ptr(src), int(upsampling_filter)))

    def get_decoded_attribute(self):
        W, H, _ = self._canvas
        out = np.zeros((2, 3, H, W), np.uint16)
        _check(self.L.tmc2_frame_get_decoded_attribute(self.h, _ptr(out)))
        return out

    def codec_color_point_cloud(self, attribute16=None):
        """attribute16: the decoded attribute frames of this frame, uint16 [2][3][H][W]; None: those already on the device
        (codec_set_decoded_attribute_yuv420)."""
        if attribute16 is None:
            _check(self.L.tmc2_codec_color_point_cloud(self.h, None))
            return
        W, H, _ = self._canvas
        att = np.ascontiguousarray(attribute16, dtype=np.uint16)
        if att.shape != (2, 3, H, W):
            raise ValueError("decoded attribute frames must be uint16 [2][3][%d][%d]" % (H, W))
        _check(self.L.tmc2_codec_color_point_cloud(self.h, _ptr(att)))

    def codec_smooth_point_cloud_postprocess(self, grid_size=8, threshold=64.0):
        _check(self.L.tmc2_codec_smooth_point_cloud_postprocess(self.h, int(grid_size), C.c_double(threshold)))

    def codec_transfer_colors_16bit_bp(self):
        _check(self.L.tmc2_codec_transfer_colors_16bit_bp(self.h))

    def codec_convert_yuv16_to_rgb8(self):
        _check(self.L.tmc2_codec_convert_yuv16_to_rgb8(self.h))

    def codec_post_reconstruct(self, attribute16=None, grid_size=8, threshold=64.0):
        """The whole tail in the reference's order: boundary points, 16-bit colours from the decoded attribute frames, grid
        geometry smoothing, colour transfer onto the moved points, YUV -> RGB."""
        self.codec_identify_boundary_points()
        self.codec_color_point_cloud(attribute16)
        self.codec_smooth_point_cloud_postprocess(grid_size, threshold)
        self.codec_transfer_colors_16bit_bp()
        self.codec_convert_yuv16_to_rgb8()

    def get_post_reconstruction(self, xyz=True, colors16=True, rgb=True, boundary=True):
        self.L.tmc2_frame_recon_count.restype = C.c_int64
        M = self.L.tmc2_frame_recon_count(self.h)
        out = dict(xyz=np.zeros((M, 3), np.int16) if xyz else None, colors16=np.zeros((M, 3), np.uint16) if colors16 else None,
                   rgb=np.zeros((M, 3), np.uint8) if rgb else None, boundary=np.zeros(M, np.uint16) if boundary else None)
        _check(self.L.tmc2_frame_get_post_reconstruction(self.h, *[None if out[k] is None else _ptr(out[k])
                                                                    for k in ("xyz", "colors16", "rgb", "boundary")]))
        return {k: v for k, v in out.items() if v is not None}

    def get_attribute_images(self, out=None):
        W, H, _ = self._canvas
        if out is None:
            out = np.zeros((2, 3, H, W), np.uint8)
        _check(self.L.tmc2_frame_get_attribute_images(self.h, _ptr(out)))
        return out

    def set_decoded_geometry(self, occ_video=None, geometry=None):
        ov = None if occ_video is None else np.ascontiguousarray(occ_video, dtype=np.uint8)
        g = None if geometry is None else np.ascontiguousarray(geometry, dtype=np.uint16)
        _check(self.L.tmc2_frame_set_decoded_geometry(self.h, None if ov is None else _ptr(ov), None if g is None else _ptr(g)))


class _Pinned:
    """tmc2_host_alloc / tmc2_host_free: page-locked host memory every device of the node can DMA into."""

    def __init__(self, nbytes):
        self.L, self.p, self.nbytes = load_library(), C.c_void_p(), int(nbytes)
        _check(self.L.tmc2_host_alloc(C.c_size_t(self.nbytes), C.byref(self.p)))
        self.__array_interface__ = dict(shape=(self.nbytes,), typestr="|u1", data=(self.p.value, False), version=3)

    def __del__(self):
        try:
            if self.p:
                self.L.tmc2_host_free(self.p)
                self.p = C.c_void_p()
        except Exception:
            pass


class SharedHostArray:
    """A named shared-memory segment (/dev/shm) mapped by several processes of the node, page-locked in the process that writes
    it from a GPU (tmc2_host_register).  create=True makes (and on close removes) the segment; every opener gets a numpy view."""

    def __init__(self, name, nbytes, create, register=True):
        import mmap
        self.path, self.nbytes, self.created, self.registered = os.path.join("/dev/shm", name), int(nbytes), bool(create), False
        fd = os.open(self.path, (os.O_CREAT | os.O_RDWR) if create else os.O_RDWR, 0o600)
        try:
            if create:
                os.ftruncate(fd, self.nbytes)
            self.map = mmap.mmap(fd, self.nbytes)
        finally:
            os.close(fd)
        self.array = np.frombuffer(self.map, dtype=np.uint8)
        if register:
            # (the mapping's own address, never a red-zone copy of the guard mode: the call page-locks the memory, it moves no payload)
            _check(load_library().tmc2_host_register(C.c_void_p(self.array.ctypes.data), C.c_size_t(self.nbytes)))
            self.registered = True

    def close(self):
        if self.registered:
            rc = load_library().tmc2_host_unregister(C.c_void_p(self.array.ctypes.data))
            self.registered = False
            if rc != 0:
                raise Tmc2Error("tmc2_host_unregister(%s) failed: %d" % (self.path, rc))
        if self.created:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.created = False


def host_array(shape, dtype=np.uint8):
    """A numpy array in page-locked host memory (the array keeps the allocation alive): destination of the canvas getters,
    so that the copies out of HBM are plain DMA."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    return np.asarray(_Pinned(max(n, 1)))[:n].view(dtype).reshape(shape)


def encoder_canvas_size(heights, tile_width=1280, min_w=1280, min_h=1280):
    """resizeTileGeometryVideo + resizeGeometryVideo: common canvas of a GOF."""
    L = load_library()
    hs = np.ascontiguousarray(heights, dtype=np.int32)
    W, H = C.c_int32(), C.c_int32()
    _check(L.tmc2_encoder_canvas_size(_ptr(hs), len(hs), int(tile_width), int(min_w), int(min_h), C.byref(W), C.byref(H)))
    return W.value, H.value


# ---- host-only pieces (no device needed) -----------------------------------------------------------
def host_kdtree_build(xyz):
    L = load_library()
    xyz = np.ascontiguousarray(xyz, dtype=np.int16)
    perm = np.empty(len(xyz), np.uint32)
    nodes, depth = C.c_uint64(), C.c_int32()
    _check(L.tmc2_host_kdtree_build(_ptr(xyz), C.c_uint64(len(xyz)), _ptr(perm), C.byref(nodes), C.byref(depth)))
    return perm, nodes.value, depth.value


def host_pack_flexible(patches, occupancy, preset_width=1280, tiles_hor=2, ratio=1.0):
    """(placed patches by index, order, height) -- the S10 placement on plain records."""
    L = load_library()
    p = np.array(patches, dtype=PATCH_DTYPE, order="C", copy=True)
    occ = np.ascontiguousarray(occupancy, dtype=np.uint8)
    order, h = np.zeros(len(p), np.int32), C.c_int32()
    _check(L.tmc2_host_pack_flexible(_ptr(p), len(p), _ptr(occ), int(preset_width), int(tiles_hor), C.c_double(ratio), _ptr(order),
                                     C.byref(h)))
    return p, order, h.value


def host_pack_spatial_consistency(patches, occupancy, previous_list, preset_width=1280, tiles_hor=2, ratio=1.0):
    """(placed patches by index, order, matches per list position, height) -- the S10' placement on plain records."""
    L = load_library()
    p = np.array(patches, dtype=PATCH_DTYPE, order="C", copy=True)
    prev = np.ascontiguousarray(previous_list, dtype=PATCH_DTYPE)
    occ = np.ascontiguousarray(occupancy, dtype=np.uint8)
    order, match, h = np.zeros(len(p), np.int32), np.zeros(len(p), np.int32), C.c_int32()
    _check(L.tmc2_host_pack_spatial_consistency(_ptr(p), len(p), _ptr(occ), _ptr(prev), len(prev), int(preset_width),
                                                int(tiles_hor), C.c_double(ratio), _ptr(order), _ptr(match), C.byref(h)))
    return p, order, match, h.value


def encoder_global_patch_allocation(frames, min_w=1280, min_h=1280):
    """S10' second half (random-access condition) over the packed frames of a GOF, in order; returns the tile
    (widths, heights) per frame.  Afterwards every frame's patch records are in list order."""
    L = load_library()
    n = len(frames)
    handles = (C.c_void_p * n)(*[fr.h for fr in frames])
    w, h = np.zeros(n, np.int32), np.zeros(n, np.int32)
    _check(L.tmc2_encoder_global_patch_allocation(handles, n, int(min_w), int(min_h), _ptr(w), _ptr(h)))
    return w, h


def host_global_patch_allocation(lists, pools, matches, tile_w, tile_h, min_w=1280, min_h=1280):
    """The allocation on plain records: per frame the patches in list order, their block-occupancy pool and matches.
    Returns per frame (list, pool, matches, tile width, tile height)."""
    L = load_library()
    n = len(lists)
    counts = np.array([len(x) for x in lists], np.int32)
    patches = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=PATCH_DTYPE) for x in lists]), dtype=PATCH_DTYPE)
    m = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.int32) for x in matches]), dtype=np.int32)
    pools = [np.ascontiguousarray(x, dtype=np.uint8) for x in pools]
    base = np.zeros(n, np.int64)
    base[1:] = np.cumsum([len(x) for x in pools])[:-1]
    occ = np.ascontiguousarray(np.concatenate(pools)) if n else np.zeros(0, np.uint8)
    # a patch can grow to the box of its track's union: never beyond the largest box of the GOF in either direction
    cap = int(len(patches) * max(1, int(patches["sizeU0"].max(initial=1))) * max(1, int(patches["sizeV0"].max(initial=1))))
    out, out_base = np.zeros(max(cap, 1), np.uint8), np.zeros(n + 1, np.int64)
    w, h = np.zeros(n, np.int32), np.zeros(n, np.int32)
    _check(L.tmc2_host_global_patch_allocation(n, _ptr(counts), _ptr(patches), _ptr(occ), _ptr(base), _ptr(m), int(tile_w),
                                               int(tile_h), int(min_w), int(min_h), _ptr(out), C.c_int64(cap), _ptr(out_base),
                                               _ptr(w), _ptr(h)))
    res, at = [], 0
    for f in range(n):
        c = int(counts[f])
        res.append((patches[at:at + c].copy(), out[out_base[f]:out_base[f + 1] if f + 1 < n else out_base[n]].copy(),
                    m[at:at + c].copy(), int(w[f]), int(h[f])))
        at += c
    return res


def host_pack_gof_records(records, mode, min_w=1280, min_h=1280, tiles_hor=2, ratio=1.0):
    """PCCEncoder::placeSegments over the patch RECORDS of a GOF (tmc2_host_place_segments; no device): records[f] =
    (patches by index, occupancy pool) of frame f.  mode 0: every frame on its own (packFlexible); 1: the low-delay chain
    (frame f against frame f-1); 2: the chain followed by the global patch allocation (random access).
    Returns per frame (patch list in list order, pool the list's occOffsets point into, matches, tile width, tile height)."""
    L = load_library()
    n = len(records)
    if n == 0:
        return []
    counts = np.array([len(r) for r, _ in records], np.int32)
    patches = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=PATCH_DTYPE) for r, _ in records]), dtype=PATCH_DTYPE)
    pools = [np.ascontiguousarray(o, dtype=np.uint8) for _, o in records]
    base = np.zeros(n, np.int64)
    base[1:] = np.cumsum([len(x) for x in pools])[:-1]
    occ = np.ascontiguousarray(np.concatenate(pools))
    # (random access) a patch can grow to the box of its track's union: never beyond the largest box of the GOF in either direction
    cap = int(len(patches) * max(1, int(patches["sizeU0"].max(initial=1))) * max(1, int(patches["sizeV0"].max(initial=1)))) if mode == 2 else len(occ)
    out, out_base = np.zeros(max(cap, 1), np.uint8), np.zeros(n + 1, np.int64)
    m = np.zeros(max(len(patches), 1), np.int32)
    w, h = np.zeros(n, np.int32), np.zeros(n, np.int32)
    _check(L.tmc2_host_place_segments(n, _ptr(counts), _ptr(patches), _ptr(occ), _ptr(base), int(mode), int(min_w), int(min_h),
                                      int(tiles_hor), C.c_double(ratio), _ptr(m), _ptr(out), C.c_int64(cap), _ptr(out_base), _ptr(w), _ptr(h)))
    res, at = [], 0
    for f in range(n):
        c = int(counts[f])
        res.append((patches[at:at + c].copy(), out[out_base[f]:out_base[f + 1]].copy(), m[at:at + c].copy(), int(w[f]), int(h[f])))
        at += c
    return res


def segmenter_params_check(params):
    """Raises Tmc2Error for a parameter set outside what the path implements (the CTC lossy conditions are inside)."""
    _check(load_library().tmc2_segmenter_params_check(C.byref(params)))


def metrics_display(out, source_points, reconstruct_points, counts, resolution=1023, with_c2p=True, precision=9):
    """The text PCCMetrics::display() prints for one frame (what the CTC log parsers read)."""
    L = load_library()
    q = np.ascontiguousarray(out, dtype=np.float64)
    c = np.ascontiguousarray(counts, dtype=np.int64)
    need = C.c_uint64()
    _check(L.tmc2_metrics_display(_ptr(q), C.c_uint64(int(source_points)), C.c_uint64(int(reconstruct_points)), _ptr(c),
                                  C.c_uint64(int(resolution)), int(bool(with_c2p)), int(precision), None, C.c_uint64(0), C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _check(L.tmc2_metrics_display(_ptr(q), C.c_uint64(int(source_points)), C.c_uint64(int(reconstruct_points)), _ptr(c),
                                  C.c_uint64(int(resolution)), int(bool(with_c2p)), int(precision), buf, C.c_uint64(need.value), None))
    return buf.value.decode()


def checksum_file_write(path, digests):
    """PCCChecksum::write: digests = list of 16-byte MD5s, one per frame."""
    L = load_library()
    d = np.frombuffer(b"".join(digests), np.uint8).copy() if len(digests) else np.zeros(0, np.uint8)
    _check(L.tmc2_checksum_file_write(str(path).encode(), _ptr(d) if len(d) else None, C.c_uint64(len(digests))))


def checksum_file_read(path):
    L = load_library()
    n = C.c_uint64()
    _check(L.tmc2_checksum_file_read(str(path).encode(), None, C.c_uint64(0), C.byref(n)))
    d = np.zeros(16 * max(1, n.value), np.uint8)
    _check(L.tmc2_checksum_file_read(str(path).encode(), _ptr(d), C.c_uint64(n.value), C.byref(n)))
    return [d[16 * f:16 * f + 16].tobytes() for f in range(n.value)]


def ply_info(path, read_normals=False):
    """(point count, has colours, has float normals) from the header of a PLY file."""
    L = load_library()
    n, c, m = C.c_uint64(), C.c_int(), C.c_int()
    _check(L.tmc2_ply_info(str(path).encode(), int(bool(read_normals)), C.byref(n), C.byref(c), C.byref(m)))
    return n.value, bool(c.value), bool(m.value)


def ply_read(path, read_normals=False, threads=0, out=None):
    """PCCPointSet3::read: (xyz int16[n][3], rgb uint8[n][3] or None, normals float64[n][3] or None).
    out = (xyz, rgb) preallocated arrays (e.g. views of page-locked memory) to land the cloud in."""
    L = load_library()
    n, has_rgb, has_nrm = ply_info(path, read_normals)
    if out is not None:
        xyz, rgb = out
        if len(xyz) < n or (rgb is not None and len(rgb) < n):
            raise ValueError("buffers too small for %d points" % n)
    else:
        xyz, rgb = np.zeros((n, 3), np.int16), (np.zeros((n, 3), np.uint8) if has_rgb else None)
    nrm = np.zeros((n, 3), np.float64) if has_nrm else None
    cnt = C.c_uint64()
    _check(L.tmc2_ply_read(str(path).encode(), _ptr(xyz), None if rgb is None or not has_rgb else _ptr(rgb),
                           None if nrm is None else _ptr(nrm), C.c_uint64(len(xyz)), int(threads), C.byref(cnt)))
    return xyz[:n], (rgb[:n] if rgb is not None and has_rgb else None), nrm


def ply_write(path, xyz, rgb=None, normals=None, ascii=True):
    """PCCPointSet3::write, byte for byte."""
    L = load_library()
    xyz = np.ascontiguousarray(xyz, dtype=np.int16)
    rgb = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
    nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
    _check(L.tmc2_ply_write(str(path).encode(), _ptr(xyz), None if rgb is None else _ptr(rgb), None if nrm is None else _ptr(nrm),
                            C.c_uint64(len(xyz)), int(bool(ascii))))


def point_set_checksum(xyz, rgb=None, reorder=False):
    """PCCPointSet3::computeChecksum: the 16 MD5 bytes."""
    L = load_library()
    xyz = np.ascontiguousarray(xyz, dtype=np.int16)
    rgb = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
    d = np.zeros(16, np.uint8)
    _check(L.tmc2_point_set_checksum(_ptr(xyz), None if rgb is None else _ptr(rgb), C.c_uint64(len(xyz)), int(bool(reorder)), _ptr(d)))
    return d.tobytes()


def host_orient_normals(xyz, knn, normals):
    L = load_library()
    xyz = np.ascontiguousarray(xyz, dtype=np.int16)
    knn = np.ascontiguousarray(knn, dtype=np.uint32)
    out = np.array(normals, dtype=np.float64, order="C", copy=True)
    _check(L.tmc2_host_orient_normals(_ptr(xyz), C.c_uint64(len(xyz)), _ptr(knn), int(knn.shape[1]), _ptr(out)))
    return out
