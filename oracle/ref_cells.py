"""ctypes binding of oracle/_ref/libll_ref_cells.so: the REFERENCE's own cell map and key-frame classes (Points_cloud_cell,
Points_cloud_map, Maps_keyframe of source/cell_map_keyframe.hpp) compiled verbatim from /root/reference against the stand-in third-party
headers of oracle/ref_stubs/ (recipe: `make -C oracle ref`, oracle/ref_cells_shim.cpp).

TEST INFRASTRUCTURE ONLY -- it pins oracle/orc_cellmap.py (and through the fixtures it writes, tests/golden/ref_cells*.npz, the cm_*
kernels) to the reference's text.  Git-ignored; (re)built only where /root/reference exists."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import ref as _ref

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libll_ref_cells.so")
_lib = None
F = np.float32


def available() -> bool:
    _ref.build()  # (one make target builds both libraries)
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libll_ref_cells.so is not built and /root/reference is absent")
        L = C.CDLL(_LIB_PATH)
        vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p
        L.refc_map_create.restype = vp
        L.refc_map_create.argtypes = [C.c_float, C.c_int]
        L.refc_map_destroy.argtypes = [vp]
        L.refc_map_append.argtypes = [vp, fp, C.c_int, C.c_int, fp, C.c_int]
        L.refc_map_n_cells.argtypes = [vp]
        L.refc_map_frame_idx.argtypes = [vp]
        L.refc_map_cells.argtypes = [vp, fp, ip, ip, C.c_int]
        L.refc_map_cell_points.argtypes = [vp, fp, fp, C.c_int]
        L.refc_map_cells_in_radius.argtypes = [vp, fp, C.c_float, fp, C.c_int]
        L.refc_cell_feature.argtypes = [vp, fp, ip, fp, fp, fp, fp, fp]
        L.refc_kf_create.restype = vp
        L.refc_kf_destroy.argtypes = [vp]
        L.refc_kf_add_cells.argtypes = [vp, vp, fp, C.c_int]
        L.refc_kf_n_cells.argtypes = [vp]
        L.refc_kf_analyze.argtypes = [vp, fp, fp, ip, fp]
        L.refc_kf_frames.argtypes = [vp, fp, fp]
        L.refc_max_similarity.argtypes = [fp, fp]
        L.refc_max_similarity.restype = C.c_float
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefCellMap:
    """Points_cloud_map<float> with set_resolution( resolution ) and m_minimum_revisit_threshold (laser_mapping.hpp:616-617)"""

    def __init__(self, resolution=1.0, minimum_revisit_threshold=2**31 - 1):
        self.L = lib()
        self.h = C.c_void_p(self.L.refc_map_create(float(resolution), int(minimum_revisit_threshold)))
        # find_cell_center's constants (cell_map_keyframe.hpp:559-560, 675-677), to turn the centres the reference reports into cell indices
        m_res = F(np.float64(F(resolution)) * 0.5)
        self.box, self.half = F(np.float64(m_res) * 1.0), F(np.float64(m_res) * 0.5)

    def close(self):
        if self.h:
            self.L.refc_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def index_of(self, centres):
        """cell index (the integer the centre was built from: centre = index * box + half, :566-568)"""
        c = np.asarray(centres, F).reshape(-1, 3)
        return np.rint((c.astype(np.float64) - float(self.half)) / float(self.box)).astype(np.int64)

    def centre_of(self, ijk):
        return (np.asarray(ijk, F) * self.box + self.half).astype(F)

    def append(self, cloud, want_cells=True):
        """append_cloud( pts, &cell_vec ); returns the cell indices of cell_vec [n,3] (sorted)"""
        xyz = np.ascontiguousarray(np.asarray(cloud, F).reshape(len(cloud), -1)[:, :3]) if len(cloud) else np.zeros((0, 3), F)
        cap = max(1, len(xyz))
        out = np.zeros((cap, 3), F)
        n = self.L.refc_map_append(self.h, _p(xyz), len(xyz), int(want_cells), _p(out), cap)
        idx = self.index_of(out[:n])
        return idx[np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))] if n else idx

    def cells(self):
        """(indices [c,3] sorted, point counts, last-update frames)"""
        n = self.L.refc_map_n_cells(self.h)
        ctr, cnt, last = np.zeros((max(n, 1), 3), F), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        self.L.refc_map_cells(self.h, _p(ctr), _p(cnt), _p(last), max(n, 1))
        idx = self.index_of(ctr[:n])
        o = np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))
        return idx[o], cnt[:n][o], last[:n][o]

    def frame_idx(self):
        return self.L.refc_map_frame_idx(self.h)

    def cell_points(self, ijk):
        c = np.ascontiguousarray(self.centre_of(ijk))
        n = self.L.refc_map_cell_points(self.h, _p(c), None, 0)
        if n < 0:
            raise KeyError(tuple(ijk))
        out = np.zeros((max(n, 1), 3), F)
        self.L.refc_map_cell_points(self.h, _p(c), _p(out), max(n, 1))
        return out[:n]

    def cells_in_radius(self, pt, radius):
        """find_cells_in_radius: indices (sorted) of the cells whose centre lies within `radius` of pt"""
        cap = max(1, self.L.refc_map_n_cells(self.h))
        out = np.zeros((cap, 3), F)
        p = np.ascontiguousarray(pt, F)
        n = self.L.refc_map_cells_in_radius(self.h, _p(p), float(radius), _p(out), cap)
        idx = self.index_of(out[:n])
        return idx[np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))] if n else idx

    def feature(self, ijk):
        """determine_feature( 1 ) of one cell: dict(type, vector, mean, cov [3,3], eigen_val, eigen_vec [3,3] columns, n)"""
        c = np.ascontiguousarray(self.centre_of(ijk))
        t = C.c_int(0)
        vec, mean, cov, ev, evec = np.zeros(3, F), np.zeros(3, F), np.zeros(9, F), np.zeros(3, F), np.zeros(9, F)
        n = self.L.refc_cell_feature(self.h, _p(c), C.byref(t), _p(vec), _p(mean), _p(cov), _p(ev), _p(evec))
        if n < 0:
            raise KeyError(tuple(ijk))
        return dict(type=t.value, vector=vec, mean=mean, cov=cov.reshape(3, 3), eigen_val=ev, eigen_vec=evec.reshape(3, 3), n=n)


class RefKeyframe:
    """Maps_keyframe<float>"""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.refc_kf_create())

    def close(self):
        if self.h:
            self.L.refc_kf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_cells(self, cmap: RefCellMap, ijk):
        c = np.ascontiguousarray(cmap.centre_of(np.asarray(ijk).reshape(-1, 3)))
        return self.L.refc_kf_add_cells(self.h, cmap.h, _p(c), len(c))

    def n_cells(self):
        return self.L.refc_kf_n_cells(self.h)

    def analyze(self):
        """update_features_of_each_cells + analyze: dict(images [4,60,60] line, plane, line_roi, plane_roi; ratio_nonzero [line, plane];
        n_vectors [4]; roi_range)"""
        img, ratio, nv, rr = np.zeros((4, 60, 60), F), np.zeros(2, F), np.zeros(4, np.int32), C.c_float(0)
        self.L.refc_kf_analyze(self.h, _p(img), _p(ratio), _p(nv), C.byref(rr))
        ctr, R = np.zeros(3, F), np.zeros((2, 3, 3), F)
        self.L.refc_kf_frames(self.h, _p(ctr), _p(R))
        return dict(images=img, ratio_nonzero=ratio, n_vectors=nv, roi_range=float(rr.value), centre=ctr, eigen_R=R)


def max_similarity(img_a, img_b) -> float:
    a, b = np.ascontiguousarray(img_a, F), np.ascontiguousarray(img_b, F)
    return float(lib().refc_max_similarity(_p(a), _p(b)))
