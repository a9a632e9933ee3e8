"""ctypes bindings of the two CPU checkers (TEST INFRASTRUCTURE; imported only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg):
    Oracle    -> oracle/liboracle.so        our CPU restatement (oracle/port_*.cpp)
    Reference -> oracle/_ref/libtmc2ref.so  the unmodified reference behind oracle/ref_harness.cpp
Both expose the same method names so a test can be parametrised over them."""
import ctypes as C
import os
import subprocess

import numpy as np


class SegParams(C.Structure):
    """orc_seg_params (oracle/oracle.h)"""
    _fields_ = [(n, C.c_int32) for n in (
        "nnNormalEstimation", "normalOrientation", "gridBasedRefineSegmentation", "maxNNCountRefineSegmentation",
        "iterationCountRefineSegmentation", "voxelDimensionRefineSegmentation", "searchRadiusRefineSegmentation",
        "occupancyResolution", "enablePatchSplitting", "maxPatchSize", "quantizerSizeX", "quantizerSizeY",
        "minPointCountPerCC", "maxNNCountPatchSegmentation", "surfaceThickness", "mapCountMinus1", "minLevel",
        "maxAllowedDepth", "geometryBitDepth2D", "geometryBitDepth3D")] + [
        ("maxAllowedDist2RawPointsDetection", C.c_double), ("maxAllowedDist2RawPointsSelection", C.c_double),
        ("lambdaRefineSegmentation", C.c_double), ("weightNormal", C.c_double * 3)]


PATCH_DTYPE = np.dtype([(n, np.int32) for n in (
    "index", "viewId", "normalAxis", "tangentAxis", "bitangentAxis", "projectionMode", "u1", "v1", "d1", "sizeU",
    "sizeV", "sizeD", "sizeDPixel", "sizeU0", "sizeV0", "size2DXInPixel", "size2DYInPixel", "d0Count",
    "eomAndD1Count", "u0", "v0", "patchOrientation")] + [("depthOffset", np.int64), ("occOffset", np.int64)])


def seg_params(iterations=10, bits3d=11, weight=(1.0, 1.0, 1.0), vox_dim=4):
    p = SegParams(16, 1, 1, 1024, iterations, vox_dim, 192, 16, 1, 1024, 16, 16, 16, 16, 4, 1, 64, 255, 8, bits3d, 9.0, 1.0,
                  3.0)
    p.weightNormal[0], p.weightNormal[1], p.weightNormal[2] = [float(x) for x in weight]
    return p


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libtmc2ref.so")
REF_TBB_PATH = os.path.join(ROOT, "oracle", "_ref", "libtmc2ref_tbb.so")   # the same sources with ENABLE_TBB + the vendored TBB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_PATH):
            subprocess.check_call(["make", "port"], cwd=os.path.join(ROOT, "oracle"))
        self.L = C.CDLL(ORACLE_PATH)
        self.L.orc_kdtree_build.restype = C.c_void_p
        self.L.orc_kdtree_perm.restype = C.POINTER(C.c_uint32)
        self.L.orc_kdtree_node_count.restype = C.c_size_t

    # S1
    def kdtree_perm(self, xyz):
        xyz = _i16(xyz)
        t = C.c_void_p(self.L.orc_kdtree_build(_p(xyz), C.c_size_t(len(xyz))))
        perm = np.ctypeslib.as_array(self.L.orc_kdtree_perm(t), (len(xyz),)).copy()
        nodes = self.L.orc_kdtree_node_count(t)
        self.L.orc_kdtree_free(t)
        return perm, nodes

    def knn(self, xyz, queries, k, with_dist=False):
        xyz, q = _i16(xyz), _i16(queries)
        t = C.c_void_p(self.L.orc_kdtree_build(_p(xyz), C.c_size_t(len(xyz))))
        idx = np.empty((len(q), k), np.uint32)
        d = np.empty((len(q), k), np.float64) if with_dist else None
        rc = self.L.orc_knn(t, _p(q), C.c_size_t(len(q)), int(k), _p(idx), None if d is None else _p(d))
        self.L.orc_kdtree_free(t)
        assert rc == 0
        return (idx, d) if with_dist else idx

    def knn_self(self, xyz, k):
        return self.knn(xyz, xyz, k)

    def radius(self, xyz, queries, r2, cap):
        xyz, q = _i16(xyz), _i16(queries)
        t = C.c_void_p(self.L.orc_kdtree_build(_p(xyz), C.c_size_t(len(xyz))))
        cnt = np.zeros(len(q), np.int32)
        idx = np.zeros((len(q), cap), np.uint32)
        self.L.orc_radius(t, _p(q), C.c_size_t(len(q)), C.c_double(r2), int(cap), _p(cnt), _p(idx))
        self.L.orc_kdtree_free(t)
        return cnt, idx

    # S2 / S3
    def compute_normals(self, xyz, knn):
        xyz = _i16(xyz)
        knn = np.ascontiguousarray(knn, dtype=np.uint32)
        out = np.empty((len(xyz), 3), np.float64)
        self.L.orc_compute_normals(_p(xyz), C.c_size_t(len(xyz)), _p(knn), int(knn.shape[1]), _p(out))
        return out

    def orient_normals(self, xyz, knn, normals):
        xyz = _i16(xyz)
        knn = np.ascontiguousarray(knn, dtype=np.uint32)
        out = np.array(normals, dtype=np.float64, order="C", copy=True)
        self.L.orc_orient_normals(_p(xyz), C.c_size_t(len(xyz)), _p(knn), int(knn.shape[1]), _p(out))
        return out

    def normals(self, xyz, k=16, oriented=True):
        knn = self.knn_self(xyz, k)
        n = self.compute_normals(xyz, knn)
        return self.orient_normals(xyz, knn, n) if oriented else n

    # S0 / S4
    def weight_normal(self, xyz, bits3d=11, min_weight=0.6):
        xyz = _i16(xyz)
        w = np.zeros(3)
        self.L.orc_weight_normal(_p(xyz), C.c_size_t(len(xyz)), int(bits3d), C.c_double(min_weight), _p(w))
        return w

    def initial_segmentation(self, normals, weight):
        nm = np.ascontiguousarray(normals, dtype=np.float64)
        w = np.ascontiguousarray(weight, dtype=np.float64)
        out = np.empty(len(nm), np.uint32)
        self.L.orc_initial_segmentation(_p(nm), C.c_size_t(len(nm)), _p(w), _p(out))
        return out


    # S5
    def refine_grid(self, xyz, normals, partition, max_nn=1024, lam=3.0, iterations=10, vox_dim=4, radius=192):
        xyz = _i16(xyz)
        nm = np.ascontiguousarray(normals, dtype=np.float64)
        part = np.array(partition, dtype=np.uint32, order="C", copy=True)
        self.L.orc_refine_grid(_p(xyz), _p(nm), C.c_size_t(len(xyz)), _p(part), int(max_nn), C.c_double(lam),
                               int(iterations), int(vox_dim), int(radius))
        return part

    def refine_grid_trace(self, xyz, normals, partition, max_nn=1024, lam=3.0, iterations=10, vox_dim=4, radius=192):
        """refine_grid + per iteration [points that differ from 1 / 2 iterations earlier, voxel edge classes that do] -- the state
        the reference carries across iterations is (partition, edge): a zero pair in a column means the state recurs."""
        xyz = _i16(xyz)
        nm = np.ascontiguousarray(normals, dtype=np.float64)
        part = np.array(partition, dtype=np.uint32, order="C", copy=True)
        trace = np.zeros((int(iterations), 4), np.uint32)
        self.L.orc_refine_grid_trace(_p(xyz), _p(nm), C.c_size_t(len(xyz)), _p(part), int(max_nn), C.c_double(lam),
                                     int(iterations), int(vox_dim), int(radius), _p(trace))
        return part, trace


    # S7-S9 and the whole segmenter
    def _collect(self, r):
        np_, dc, oc, rc, rounds = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        self.L.orc_seg_result_sizes(r, C.byref(np_), C.byref(dc), C.byref(oc), C.byref(rc), C.byref(rounds))
        patches = np.zeros(np_.value, PATCH_DTYPE)
        d0 = np.zeros(dc.value, np.int16)
        d1 = np.zeros(dc.value, np.int16)
        occ = np.zeros(oc.value, np.uint8)
        res = np.zeros((rc.value, 3), np.int16)
        rr = np.zeros(rounds.value, np.int32)
        self.L.orc_seg_result_copy(r, _p(patches), _p(d0), _p(d1), _p(occ), _p(res), _p(rr))
        self.L.orc_seg_result_free(r)
        return dict(patches=patches, depth0=d0, depth1=d1, occupancy=occ, resampled=res, round_raw=rr)

    def segment_patches(self, xyz, rgb, knn, partition, params):
        xyz = _i16(xyz)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        knn = np.ascontiguousarray(knn, dtype=np.uint32)
        part = np.ascontiguousarray(partition, dtype=np.uint32)
        self.L.orc_segment_patches.restype = C.c_void_p
        r = C.c_void_p(self.L.orc_segment_patches(_p(xyz), _p(rgb), C.c_size_t(len(xyz)), _p(knn), int(knn.shape[1]),
                                                  _p(part), C.byref(params)))
        return self._collect(r)

    def segment(self, xyz, rgb, params):
        xyz = _i16(xyz)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        self.L.orc_segment.restype = C.c_void_p
        r = C.c_void_p(self.L.orc_segment(_p(xyz), _p(rgb), C.c_size_t(len(xyz)), C.byref(params)))
        return self._collect(r)


    # S10-S16
    def pack_flexible(self, patches, occupancy, preset_width=1280, occ_res=16, tiles_hor=2, ratio=1.0):
        p = np.array(patches, dtype=PATCH_DTYPE, copy=True)
        occ = np.ascontiguousarray(occupancy, dtype=np.uint8)
        order = np.zeros(len(p), np.int32)
        h = C.c_int32()
        self.L.orc_pack_flexible(_p(p), len(p), _p(occ), int(preset_width), int(occ_res), int(tiles_hor),
                                 C.c_double(ratio), _p(order), C.byref(h))
        return p, order, h.value

    def gof_canvas_size(self, heights, tile_width=1280, min_w=1280, min_h=1280):
        hs = np.ascontiguousarray(heights, dtype=np.int32)
        W, H = C.c_int32(), C.c_int32()
        self.L.orc_gof_canvas_size(_p(hs), len(hs), int(tile_width), int(min_w), int(min_h), C.byref(W), C.byref(H))
        return W.value, H.value

    def geometry_images(self, patches, order, depth0, depth1, W, H, occ_res=16, occ_precision=4):
        p = np.ascontiguousarray(patches, dtype=PATCH_DTYPE)
        order = np.ascontiguousarray(order, dtype=np.int32)
        out = dict(occupancy=np.zeros((H, W), np.uint8), occ_video=np.zeros((H // occ_precision, W // occ_precision), np.uint8),
                   block_to_patch=np.zeros((H // occ_res, W // occ_res), np.uint32),
                   geo0=np.zeros((H, W), np.uint16), geo1=np.zeros((H, W), np.uint16))
        rc = self.L.orc_generate_geometry_images(_p(p), _p(order), len(p), _p(np.ascontiguousarray(depth0, np.int16)),
                                                 _p(np.ascontiguousarray(depth1, np.int16)), int(W), int(H), int(occ_res),
                                                 int(occ_precision), _p(out["occupancy"]), _p(out["occ_video"]),
                                                 _p(out["block_to_patch"]), _p(out["geo0"]), _p(out["geo1"]))
        assert rc == 0
        return out

    def pack_spatial_consistency(self, patches, occupancy, prev_list, preset_width=1280, occ_res=16, tiles_hor=2, ratio=1.0):
        """S10' for one frame against the previous frame's patch list (in list order).
        Returns (placed patches by index, order, matches per list position, height)."""
        p = np.array(patches, dtype=PATCH_DTYPE, order="C", copy=True)
        prev = np.ascontiguousarray(prev_list, dtype=PATCH_DTYPE)
        occ = np.ascontiguousarray(occupancy, dtype=np.uint8)
        order = np.zeros(len(p), np.int32)
        match = np.zeros(len(p), np.int32)
        h = C.c_int32()
        rc = self.L.orc_pack_spatial_consistency(_p(p), len(p), _p(occ), _p(prev), len(prev), int(preset_width), int(occ_res),
                                                 int(tiles_hor), C.c_double(ratio), _p(order), _p(match), C.byref(h))
        if rc == -2:
            return None       # a patch fits at no canvas height: the reference never returns
        return p, order, match, h.value

    @staticmethod
    def tile_size(per, min_w, min_h, chained=True):
        """resizeTileGeometryVideo after the per-frame packing: the common tile of the GOF.  packFlexible works on a copy of
        the tile width (PCCEncoder.cpp:2312), so a frame it packed keeps the preset width whatever its patches needed;
        spatialConsistencyPackFlexible writes the width of its canvas back (:1190, :1308)."""
        widths = [min_w]
        for f, (_, placed, _, _) in enumerate(per):
            if chained and f > 0 and len(placed):
                widths.append(max(min_w // 16, int((placed["sizeU0"] + 1).max())) * 16)
        return max(widths), max([h for _, _, _, h in per] + [min_h])

    def global_patch_allocation(self, per, min_w, min_h):
        """S10' second half (random-access condition): GPA over the frames packed by the per-frame chain.
        per: [(seg, placed, order, height)] -> [(list patches, occupancy pool, matches, tile width, tile height)]."""
        L = self.L
        L.orc_gpa_begin.restype = C.c_void_p
        L.orc_gpa_occ_bytes.restype = C.c_int64
        tw, th = self.tile_size(per, min_w, min_h)
        h = C.c_void_p(L.orc_gpa_begin(len(per), int(min_w), int(min_h), 16))
        for f, (seg, placed, order, _) in enumerate(per):
            lst = np.ascontiguousarray(placed[order], dtype=PATCH_DTYPE)
            occ = np.ascontiguousarray(seg["occupancy"], dtype=np.uint8)
            m = np.ascontiguousarray(seg["matches"], dtype=np.int32)
            L.orc_gpa_set_frame(h, f, _p(lst), len(lst), _p(occ), _p(m), int(tw), int(th))
        if L.orc_gpa_run(h):
            L.orc_gpa_free(h)
            return None       # undefined in the reference (a tracked patch without a place in the realigned lists), or it never returns
        out = []
        for f, (seg, placed, order, _) in enumerate(per):
            n = len(placed)
            lst = np.zeros(n, PATCH_DTYPE)
            occ = np.zeros(max(1, L.orc_gpa_occ_bytes(h, f)), np.uint8)
            m = np.zeros(n, np.int32)
            wh = np.zeros(2, np.int32)
            L.orc_gpa_get_frame(h, f, _p(lst), _p(occ), _p(m), _p(wh))
            out.append((lst, occ, m, int(wh[0]), int(wh[1])))
        L.orc_gpa_free(h)
        return out

    def phase_a(self, frames, iterations=10, bits3d=11, occ_precision=4, min_w=1280, min_h=1280, constrained_pack=False, vox_dim=4):
        """S0..S16 for a GOF given as [(xyz, rgb), ...]; mirrors Reference.phase_a.  constrained_pack: True = the low-delay
        condition (frames after the first packed against their predecessor, S10'); 2 = the random-access condition
        (the same chain followed by the global patch allocation)."""
        w = self.weight_normal(frames[0][0], bits3d, 0.6)
        sp = seg_params(iterations, bits3d, w, vox_dim)
        per = []
        for xyz, rgb in frames:
            seg = self.segment(xyz, rgb, sp)
            if constrained_pack and per:
                pseg, pplaced, porder, _ = per[-1][:4]
                placed, order, match, h = self.pack_spatial_consistency(seg["patches"], seg["occupancy"], pplaced[porder], min_w)
                seg["matches"] = match
            else:
                placed, order, h = self.pack_flexible(seg["patches"], seg["occupancy"], min_w)
                seg["matches"] = np.full(len(order), -1, np.int32)
            per.append((seg, placed, order, h))
        if constrained_pack == 2:
            gpa = self.global_patch_allocation(per, min_w, min_h)
            tw, th = max([g[3] for g in gpa] + [min_w]), max([g[4] for g in gpa] + [min_h])    # resizeTileGeometryVideo
            W, H = self.gof_canvas_size([th], tw, min_w, min_h)
            out = []
            for (seg, _, _, _), (lst, occ, m, _, _) in zip(per, gpa):
                order = np.arange(len(lst), dtype=np.int32)
                img = self.geometry_images(lst, order, seg["depth0"], seg["depth1"], W, H, 16, occ_precision)
                img.update(patches=lst, width=W, height=H, matches=m)
                out.append(img)
            return out
        W, H = self.gof_canvas_size([x[3] for x in per], self.tile_size(per, min_w, min_h, bool(constrained_pack))[0], min_w, min_h)
        out = []
        for seg, placed, order, h in per:
            img = self.geometry_images(placed, order, seg["depth0"], seg["depth1"], W, H, 16, occ_precision)
            img.update(patches=placed[order], width=W, height=H, matches=seg["matches"])
            out.append(img)
        return out


    _metrics_fn = "orc_metrics"

    def metrics(self, src_xyz, src_rgb, rec_xyz, rec_rgb, normals=None, resolution=1023.0):
        """S23: returns (q[3][8], counts[2]); rows = A->B, B->A, symmetric; columns = c2cMse, c2cPsnr, c2pMse, c2pPsnr,
        colorMse Y/U/V, colorPsnr Y."""
        src_xyz, rec_xyz = _i16(src_xyz), _i16(rec_xyz)
        src_rgb = np.ascontiguousarray(src_rgb, dtype=np.uint8)
        rec_rgb = np.ascontiguousarray(rec_rgb, dtype=np.uint8)
        nm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
        out = np.zeros((3, 8), np.float64)
        counts = np.zeros(2, np.int64)
        rc = getattr(self.L, self._metrics_fn)(_p(src_xyz), _p(src_rgb), C.c_size_t(len(src_xyz)), _p(rec_xyz), _p(rec_rgb),
                                               C.c_size_t(len(rec_xyz)), None if nm is None else _p(nm),
                                               C.c_double(resolution), _p(out), _p(counts))
        assert rc == 0, rc
        return out, counts

    # S17-S22
    def generate_point_cloud(self, img):
        W, H = img["width"], img["height"]
        p = np.ascontiguousarray(img["patches"], dtype=PATCH_DTYPE)   # already in packing order
        order = np.arange(len(p), dtype=np.int32)
        xyz = np.zeros((2 * W * H, 3), np.int16)
        p2p = np.zeros((2 * W * H, 3), np.uint32)
        prec = W // img["occ_video"].shape[1]
        self.L.orc_generate_point_cloud.restype = C.c_int64
        M = self.L.orc_generate_point_cloud(_p(p), _p(order), len(p), _p(img["occ_video"]), _p(img["block_to_patch"]),
                                            _p(img["geo0"]), _p(img["geo1"]), int(W), int(H), 16, int(prec), _p(xyz), _p(p2p))
        return xyz[:M].copy(), p2p[:M].copy()

    def transfer_colors(self, src_xyz, src_rgb, tgt_xyz):
        src_xyz, tgt_xyz = _i16(src_xyz), _i16(tgt_xyz)
        src_rgb = np.ascontiguousarray(src_rgb, dtype=np.uint8)
        out = np.zeros((len(tgt_xyz), 3), np.uint8)
        rc = self.L.orc_transfer_colors(_p(src_xyz), _p(src_rgb), C.c_size_t(len(src_xyz)), _p(tgt_xyz),
                                        C.c_size_t(len(tgt_xyz)), _p(out))
        assert rc == 0, rc
        return out

    def attribute_images(self, rgb, p2p, occ_video, W, H, occ_precision=4):
        out = np.zeros((2, 3, H, W), np.uint8)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        p2p = np.ascontiguousarray(p2p, dtype=np.uint32)
        self.L.orc_attribute_images(_p(rgb), _p(p2p), C.c_int64(len(rgb)), _p(np.ascontiguousarray(occ_video)), int(W),
                                    int(H), int(occ_precision), _p(out))
        return out

    def phase_b(self, frames, phase_a_out, occ_precision=4):
        """S17..S22 on top of phase_a() output (identity video codec)."""
        out = []
        for (xyz, rgb), img in zip(frames, phase_a_out):
            rec, p2p = self.generate_point_cloud(img)
            col = self.transfer_colors(xyz, rgb, rec)
            att = self.attribute_images(col, p2p, img["occ_video"], img["width"], img["height"], occ_precision)
            out.append(dict(recon_xyz=rec, recon_rgb=col, point_to_pixel=p2p, attribute=att))
        return out



    # ---- colour-space conversion around the attribute video codec ----
    _conv_prefix = "orc_"

    def convert_rgb444_to_yuv420(self, rgb, filter=4):
        """rgb u8 [3][H][W] -> (y u8 [H][W], u, v u8 [H/2][W/2])"""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        _, H, W = rgb.shape
        y, u, v = np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)
        rc = getattr(self.L, self._conv_prefix + "convert_rgb444_to_yuv420")(_p(rgb), int(W), int(H), int(filter), _p(y), _p(u), _p(v))
        assert rc == 0, rc
        return y, u, v

    def convert_yuv420_to_yuv444(self, y, u, v, filter=0):
        """-> u16 [3][H][W]"""
        y, u, v = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v))
        H, W = y.shape
        out = np.zeros((3, H, W), np.uint16)
        rc = getattr(self.L, self._conv_prefix + "convert_yuv420_to_yuv444")(_p(y), _p(u), _p(v), int(W), int(H), int(filter), _p(out))
        assert rc == 0, rc
        return out

    # ---- post-reconstruction tail ----
    def identify_boundary_points(self, p2p, occ_video, W, H, occ_precision):
        p2p = np.ascontiguousarray(p2p, dtype=np.uint32)
        ov = np.ascontiguousarray(occ_video, dtype=np.uint8)
        bt = np.zeros(len(p2p), np.uint16)
        self.L.orc_identify_boundary_points(_p(p2p), C.c_int64(len(p2p)), _p(ov), int(W), int(H), int(occ_precision), _p(bt))
        return bt

    def color_point_cloud(self, p2p, attribute16):
        p2p = np.ascontiguousarray(p2p, dtype=np.uint32)
        att = np.ascontiguousarray(attribute16, dtype=np.uint16)
        out = np.zeros((len(p2p), 3), np.uint16)
        self.L.orc_color_point_cloud(_p(p2p), C.c_int64(len(p2p)), _p(att), int(att.shape[-1]), int(att.shape[-2]), _p(out))
        return out

    def smooth_point_cloud_grid(self, xyz, btype, partition, grid_size=8, threshold=64.0):
        xyz = np.array(_i16(xyz), copy=True)
        bt = np.array(btype, dtype=np.uint16, copy=True)
        part = np.ascontiguousarray(partition, dtype=np.uint32)
        self.L.orc_smooth_point_cloud_grid(_p(xyz), _p(bt), _p(part), C.c_int64(len(xyz)), int(grid_size), C.c_double(threshold))
        return xyz, bt

    def transfer_colors16_bp(self, src_xyz, src_c16, tgt_xyz, tgt_btype):
        src_xyz, tgt_xyz = _i16(src_xyz), _i16(tgt_xyz)
        src = np.ascontiguousarray(src_c16, dtype=np.uint16)
        bt = np.ascontiguousarray(tgt_btype, dtype=np.uint16)
        out = np.array(src, copy=True)
        self.L.orc_transfer_colors16_bp(_p(src_xyz), _p(src), _p(tgt_xyz), _p(bt), C.c_int64(len(tgt_xyz)), _p(out))
        return out

    def convert_yuv16_to_rgb8(self, c16):
        c16 = np.ascontiguousarray(c16, dtype=np.uint16)
        out = np.zeros((len(c16), 3), np.uint8)
        self.L.orc_convert_yuv16_to_rgb8(_p(c16), C.c_int64(len(c16)), _p(out))
        return out

    def phase_c(self, phase_a_out, phase_b_out, decoded_attribute, occ_precision=4):
        """Post-reconstruction tail on top of phase_a() / phase_b() output; mirrors Reference.phase_c."""
        out = []
        for a, b, att in zip(phase_a_out, phase_b_out, decoded_attribute):
            p2p = b["point_to_pixel"]
            bt0 = self.identify_boundary_points(p2p, a["occ_video"], a["width"], a["height"], occ_precision)
            part = (a["block_to_patch"][p2p[:, 1] // 16, p2p[:, 0] // 16] - 1).astype(np.uint32)
            c16 = self.color_point_cloud(p2p, att)
            xyz, bt = self.smooth_point_cloud_grid(b["recon_xyz"], bt0, part)
            c16 = self.transfer_colors16_bp(b["recon_xyz"], c16, xyz, bt)
            out.append(dict(boundary_before=bt0, partition=part, xyz=xyz, colors16=c16, rgb=self.convert_yuv16_to_rgb8(c16), boundary=bt))
        return out


class Reference:
    """The unmodified reference compiled in place (oracle/Makefile).  tbb=True: the ENABLE_TBB build with the vendored TBB;
    nb_thread = the reference's --nbThread (width of its TBB arenas) for the GOF entry points."""

    def __init__(self, tbb=False, nb_thread=1):
        self.L = C.CDLL(REF_TBB_PATH if tbb else REF_PATH)
        self.nb_thread = int(nb_thread)
        assert bool(self.L.ref_built_with_tbb()) == bool(tbb)

    def transfer_colors(self, src_xyz, src_rgb, tgt_xyz):
        src_xyz, tgt_xyz = _i16(src_xyz), _i16(tgt_xyz)
        src_rgb = np.ascontiguousarray(src_rgb, dtype=np.uint8)
        out = np.zeros((len(tgt_xyz), 3), np.uint8)
        self.L.ref_transfer_colors(_p(src_xyz), _p(src_rgb), C.c_size_t(len(src_xyz)), _p(tgt_xyz), C.c_size_t(len(tgt_xyz)), _p(out))
        return out

    _metrics_fn = "ref_metrics"

    def metrics(self, src_xyz, src_rgb, rec_xyz, rec_rgb, normals=None, resolution=1023.0):
        """S23: returns (q[3][8], counts[2]); rows = A->B, B->A, symmetric; columns = c2cMse, c2cPsnr, c2pMse, c2pPsnr,
        colorMse Y/U/V, colorPsnr Y."""
        src_xyz, rec_xyz = _i16(src_xyz), _i16(rec_xyz)
        src_rgb = np.ascontiguousarray(src_rgb, dtype=np.uint8)
        rec_rgb = np.ascontiguousarray(rec_rgb, dtype=np.uint8)
        nm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
        out = np.zeros((3, 8), np.float64)
        counts = np.zeros(2, np.int64)
        rc = getattr(self.L, self._metrics_fn)(_p(src_xyz), _p(src_rgb), C.c_size_t(len(src_xyz)), _p(rec_xyz), _p(rec_rgb),
                                               C.c_size_t(len(rec_xyz)), None if nm is None else _p(nm),
                                               C.c_double(resolution), _p(out), _p(counts))
        assert rc == 0, rc
        return out, counts

    def phase_b(self, frames, phase_a_out, occ_precision=4):
        """Must follow phase_a() on the same GOF (state lives in the harness)."""
        L = self.L
        L.ref_gof_phase_b()
        L.ref_gof_recon_count.restype = C.c_int64
        out = []
        for i, img in enumerate(phase_a_out):
            M = L.ref_gof_recon_count(i)
            rec = np.zeros((M, 3), np.int16)
            col = np.zeros((M, 3), np.uint8)
            p2p = np.zeros((M, 3), np.uint32)
            L.ref_gof_get_recon(i, _p(rec), _p(col), _p(p2p))
            att = np.zeros((2, 3, img["height"], img["width"]), np.uint8)
            assert L.ref_gof_get_attribute_images(i, _p(att)) == 0
            out.append(dict(recon_xyz=rec, recon_rgb=col, point_to_pixel=p2p, attribute=att))
        return out


    # ---- colour-space conversion around the attribute video codec ----
    _conv_prefix = "ref_"

    def convert_rgb444_to_yuv420(self, rgb, filter=4):
        """rgb u8 [3][H][W] -> (y u8 [H][W], u, v u8 [H/2][W/2])"""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        _, H, W = rgb.shape
        y, u, v = np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)
        rc = getattr(self.L, self._conv_prefix + "convert_rgb444_to_yuv420")(_p(rgb), int(W), int(H), int(filter), _p(y), _p(u), _p(v))
        assert rc == 0, rc
        return y, u, v

    def convert_yuv420_to_yuv444(self, y, u, v, filter=0):
        """-> u16 [3][H][W]"""
        y, u, v = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v))
        H, W = y.shape
        out = np.zeros((3, H, W), np.uint16)
        rc = getattr(self.L, self._conv_prefix + "convert_yuv420_to_yuv444")(_p(y), _p(u), _p(v), int(W), int(H), int(filter), _p(out))
        assert rc == 0, rc
        return out

    def place_records(self, gof, min_w, min_h, constrained_pack):
        """PCCEncoder::placeSegments on patch records: gof = [(records by index, occupancy pool)] per frame.
        -> (per frame (list patches, occupancy pool in list order, matches), (canvas width, canvas height))."""
        L = self.L
        counts = np.array([len(r) for r, _ in gof], np.int32)
        recs = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=PATCH_DTYPE) for r, _ in gof]), dtype=PATCH_DTYPE)
        pools = [np.ascontiguousarray(o, dtype=np.uint8) for _, o in gof]
        base = np.zeros(len(gof), np.int64)
        base[1:] = np.cumsum([len(o) for o in pools])[:-1]
        occ = np.ascontiguousarray(np.concatenate(pools))
        L.ref_place_records(len(gof), _p(counts), _p(recs), _p(occ), _p(base), int(min_w), int(min_h), int(constrained_pack))
        L.ref_gof_get_patch_occupancy.restype = C.c_int64
        out = []
        for f in range(len(gof)):
            n = L.ref_gof_patch_count(f)
            pt = np.zeros(n, PATCH_DTYPE)
            L.ref_gof_get_patches(f, _p(pt))
            m = np.zeros(n, np.int32)
            L.ref_gof_get_patch_matches(f, _p(m))
            o = np.zeros(max(1, L.ref_gof_get_patch_occupancy(f, None)), np.uint8)
            L.ref_gof_get_patch_occupancy(f, _p(o))
            out.append((pt, o, m))
        w, h = C.c_int(), C.c_int()
        L.ref_gof_frame_size(C.byref(w), C.byref(h))
        return out, (w.value, h.value)

    def smooth_and_transfer(self, xyz, btype, partition, colors16, grid_size=8, threshold=64.0):
        """smoothPointCloudPostprocess + transferColors16bitBP on an arbitrary cloud -> (xyz, boundary types, colours16)."""
        x = np.array(_i16(xyz), copy=True)
        bt = np.array(btype, dtype=np.uint16, copy=True)
        part = np.ascontiguousarray(partition, dtype=np.uint32)
        c = np.array(colors16, dtype=np.uint16, copy=True, order="C")
        self.L.ref_smooth_and_transfer(_p(x), _p(bt), _p(part), _p(c), C.c_size_t(len(x)), int(grid_size), C.c_double(threshold))
        return x, bt, c

    def adaptor_check_frame(self, frame, img, phase_b=None):
        """integration/tmc2hip_convert.cpp on what the C-ABI getters return for a frame (img: occupancy, occ_video,
        block_to_patch, geo0, geo1; phase_b: attribute, recon_xyz, recon_rgb, point_to_pixel) against the containers the reference
        filled for that frame (after phase_a / phase_b on the same GOF).  Returns a bit mask of containers that differ."""
        c = lambda a, t: np.ascontiguousarray(a, dtype=t)
        occ, ov, b2p = c(img["occupancy"], np.uint8), c(img["occ_video"], np.uint8), c(img["block_to_patch"], np.uint32)
        g0, g1 = c(img["geo0"], np.uint16), c(img["geo1"], np.uint16)
        if phase_b is None:
            return self.L.ref_adaptor_check_frame(int(frame), _p(occ), _p(ov), _p(b2p), _p(g0), _p(g1), None, None, None, None,
                                                  C.c_size_t(0))
        att = c(phase_b["attribute"], np.uint8)
        x, col, p2p = c(phase_b["recon_xyz"], np.int16), c(phase_b["recon_rgb"], np.uint8), c(phase_b["point_to_pixel"], np.uint32)
        return self.L.ref_adaptor_check_frame(int(frame), _p(occ), _p(ov), _p(b2p), _p(g0), _p(g1), _p(att), _p(x), _p(col), _p(p2p),
                                              C.c_size_t(len(x)))

    def ply_read(self, path, read_normals=False):
        """PCCPointSet3::read -> (xyz, rgb or None, normals or None), or None if the reference refuses the file."""
        L = self.L
        L.ref_ply_read.restype = C.c_int64
        flags = C.c_int()
        n = L.ref_ply_read(str(path).encode(), int(bool(read_normals)), C.byref(flags))
        if n < 0:
            return None
        xyz = np.zeros((n, 3), np.int16)
        rgb = np.zeros((n, 3), np.uint8) if flags.value & 1 else None
        nrm = np.zeros((n, 3), np.float64) if flags.value & 2 else None
        L.ref_ply_get(_p(xyz), None if rgb is None else _p(rgb), None if nrm is None else _p(nrm))
        return xyz, rgb, nrm

    def checksum(self, xyz, rgb=None, reorder=False):
        xyz = _i16(xyz)
        rgb = None if rgb is None else np.ascontiguousarray(rgb, dtype=np.uint8)
        d = np.zeros(16, np.uint8)
        self.L.ref_checksum(_p(xyz), None if rgb is None else _p(rgb), C.c_size_t(len(xyz)), int(bool(reorder)), _p(d))
        return d.tobytes()

    def phase_c(self, phase_b_out, decoded_attribute):
        """Post-reconstruction tail (must follow phase_b() on the same GOF): colorPointCloud from the given "decoded"
        attribute frames (u16 [2][3][H][W] per frame), grid geometry smoothing, transferColors16bitBP, convertYUV16ToRGB8."""
        L = self.L
        pre = []
        for i, b in enumerate(phase_b_out):
            M = len(b["recon_xyz"])
            bt, part = np.zeros(M, np.uint16), np.zeros(M, np.uint32)
            L.ref_gof_get_boundary_types(i, _p(bt))
            assert L.ref_gof_get_partition(i, _p(part)) == M
            att = np.ascontiguousarray(decoded_attribute[i], dtype=np.uint16)
            L.ref_gof_set_decoded_attribute(i, _p(att))
            pre.append((bt, part))
        rc = L.ref_gof_phase_c()
        assert rc == 0, rc
        out = []
        for i, b in enumerate(phase_b_out):
            M = len(b["recon_xyz"])
            xyz, c16, rgb, bt = np.zeros((M, 3), np.int16), np.zeros((M, 3), np.uint16), np.zeros((M, 3), np.uint8), np.zeros(M, np.uint16)
            L.ref_gof_get_post(i, _p(xyz), _p(c16), _p(rgb), _p(bt))
            out.append(dict(boundary_before=pre[i][0], partition=pre[i][1], xyz=xyz, colors16=c16, rgb=rgb, boundary=bt))
        return out

    def phase_a(self, frames, iterations=10, bits3d=11, occ_precision=4, min_w=1280, min_h=1280, constrained_pack=False, vox_dim=4):
        """S0..S16 through the reference's own PCCEncoder members (identity video codec)."""
        L = self.L
        L.ref_gof_begin2(len(frames), int(iterations), int(bits3d - 1), int(occ_precision), int(min_w), int(min_h),
                         int(constrained_pack))
        L.ref_gof_set_voxel_dimension_refine(int(vox_dim))
        L.ref_gof_set_nb_thread(self.nb_thread)
        keep = []
        for i, (xyz, rgb) in enumerate(frames):
            xyz = _i16(xyz)
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            keep.append((xyz, rgb))
            L.ref_gof_set_frame(i, _p(xyz), _p(rgb), C.c_size_t(len(xyz)))
        L.ref_gof_phase_a()
        w, h = C.c_int(), C.c_int()
        L.ref_gof_frame_size(C.byref(w), C.byref(h))
        W, H = w.value, h.value
        out = []
        for i in range(len(frames)):
            n = L.ref_gof_patch_count(i)
            pt = np.zeros(n, PATCH_DTYPE)
            L.ref_gof_get_patches(i, _p(pt))
            img = dict(occupancy=np.zeros((H, W), np.uint8), occ_video=np.zeros((H // occ_precision, W // occ_precision), np.uint8),
                       block_to_patch=np.zeros((H // 16, W // 16), np.uint32), geo0=np.zeros((H, W), np.uint16),
                       geo1=np.zeros((H, W), np.uint16))
            L.ref_gof_get_images(i, _p(img["occupancy"]), _p(img["occ_video"]), _p(img["block_to_patch"]),
                                 _p(img["geo0"]), _p(img["geo1"]))
            assert L.ref_gof_geometry_chroma_nonzero(i) == 0
            mt = np.zeros(n, np.int32)
            L.ref_gof_get_patch_matches(i, _p(mt))
            img.update(patches=pt, width=W, height=H, matches=mt)
            out.append(img)
        return out

    def segment(self, xyz, rgb, params):
        xyz = _i16(xyz)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        cnt = self.L.ref_segment(_p(xyz), _p(rgb), C.c_size_t(len(xyz)), C.byref(params))
        dc, oc = C.c_int64(), C.c_int64()
        self.L.ref_patch_pool_sizes(C.byref(dc), C.byref(oc))
        patches = np.zeros(cnt, PATCH_DTYPE)
        d0 = np.zeros(dc.value, np.int16)
        d1 = np.zeros(dc.value, np.int16)
        occ = np.zeros(oc.value, np.uint8)
        self.L.ref_get_patches(_p(patches), _p(d0), _p(d1), _p(occ))
        return dict(patches=patches, depth0=d0, depth1=d1, occupancy=occ)

    def knn(self, xyz, queries, k, with_dist=False):
        xyz, q = _i16(xyz), _i16(queries)
        idx = np.empty((len(q), k), np.uint32)
        d = np.empty((len(q), k), np.float64) if with_dist else None
        self.L.ref_knn(_p(xyz), C.c_size_t(len(xyz)), _p(q), C.c_size_t(len(q)), int(k), _p(idx),
                       None if d is None else _p(d))
        return (idx, d) if with_dist else idx

    def knn_self(self, xyz, k):
        return self.knn(xyz, xyz, k)

    def radius(self, xyz, queries, r2, cap):
        xyz, q = _i16(xyz), _i16(queries)
        cnt = np.zeros(len(q), np.int32)
        idx = np.zeros((len(q), cap), np.uint32)
        self.L.ref_radius(_p(xyz), C.c_size_t(len(xyz)), _p(q), C.c_size_t(len(q)), C.c_double(r2), int(cap), _p(cnt),
                          _p(idx))
        return cnt, idx

    def normals(self, xyz, k=16, oriented=True):
        xyz = _i16(xyz)
        out = np.empty((len(xyz), 3), np.float64)
        self.L.ref_normals(_p(xyz), C.c_size_t(len(xyz)), int(k), 1, 1 if oriented else 0, _p(out))
        return out

    def weight_normal(self, xyz, bits3d=11, min_weight=0.6):
        xyz = _i16(xyz)
        w = np.zeros(3)
        self.L.ref_weight_normal(_p(xyz), C.c_size_t(len(xyz)), int(bits3d), C.c_double(min_weight), _p(w))
        return w

    def initial_segmentation(self, normals, weight):
        nm = np.ascontiguousarray(normals, dtype=np.float64)
        w = np.ascontiguousarray(weight, dtype=np.float64)
        out = np.empty(len(nm), np.uint32)
        self.L.ref_initial_segmentation(_p(nm), C.c_size_t(len(nm)), _p(w), _p(out))
        return out

    def refine_grid(self, xyz, normals, partition, max_nn=1024, lam=3.0, iterations=10, vox_dim=4, radius=192):
        xyz = _i16(xyz)
        nm = np.ascontiguousarray(normals, dtype=np.float64)
        part = np.array(partition, dtype=np.uint32, order="C", copy=True)
        self.L.ref_refine_grid(_p(xyz), _p(nm), C.c_size_t(len(xyz)), _p(part), int(max_nn), C.c_double(lam),
                               int(iterations), int(vox_dim), int(radius))
        return part
