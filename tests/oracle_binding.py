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


def seg_params(iterations=10, bits3d=11, weight=(1.0, 1.0, 1.0)):
    p = SegParams(16, 1, 1, 1024, iterations, 4, 192, 16, 1, 1024, 16, 16, 16, 16, 4, 1, 64, 255, 8, bits3d, 9.0, 1.0,
                  3.0)
    p.weightNormal[0], p.weightNormal[1], p.weightNormal[2] = [float(x) for x in weight]
    return p


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libtmc2ref.so")


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


class Reference:
    def __init__(self):
        self.L = C.CDLL(REF_PATH)

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
