"""ctypes binding of tests/hostcheck/libhostcheck.so (TEST-ONLY host build of the device math headers)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhostcheck.so")
_SRC = os.path.join(_HERE, "hostcheck.cpp")
_CSRC = os.path.join(_HERE, "..", "..", "loam_livox_amd", "csrc")


def build():
    deps = [_SRC] + [os.path.join(_CSRC, f) for f in ("ll_fe_core.h", "ll_knn_core.h", "ll_knn_tile.h", "ll_reg_core.h", "ll_cellmap_core.h", "ll_voxel_core.h")]
    if not os.path.exists(_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-o", _LIB, _SRC])
    return _LIB


class FeParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("thr_corner_curvature", "thr_surface_curvature", "minimum_view_angle",
                                         "livox_min_allow_dis", "livox_min_sigma", "max_fov", "time_internal_pts")]


class RegParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("if_motion_deblur", "icp_max_iterations", "ceres_max_iterations",
                                       "ceres_prerun_times", "icp_line", "icp_plane", "force_all_iterations")] + \
               [(n, C.c_double) for n in ("max_d2_line", "max_d2_plane", "huber_a", "inliner_dis", "inlier_ratio",
                                          "minimum_icp_R_diff", "minimum_icp_T_diff", "bound")] + \
               [(n, C.c_float) for n in ("para_max_angular_rate", "max_final_cost", "min_ts", "max_ts")] + \
               [(n, C.c_int) for n in ("check_line_pca", "check_plane_pca", "max_blocks", "subsample_seed")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.hc_grid_build.restype = C.c_void_p
        L.hc_grid_build.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_float]
        L.hc_grid_free.argtypes = [C.c_void_p]
        L.hc_knn5.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.hc_fe_points.argtypes = [C.POINTER(FeParams), C.c_void_p, C.c_int, C.c_double] + [C.c_void_p] * 8
        L.hc_select.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 6
        L.hc_eval_blocks.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p]
        L.hc_make_block.argtypes = [C.c_int] + [C.c_void_p] * 6
        L.hc_reg_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(RegParams),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fe_points(xyzi, t0, prm: FeParams):
    xyzi = np.ascontiguousarray(xyzi, np.float32)
    n = xyzi.shape[0]
    out = dict(type=np.zeros(n, np.int32), label=np.zeros(n, np.int32), depth2=np.zeros(n, np.float32),
               curv=np.zeros(n, np.float32), view=np.zeros(n, np.float32), tstamp=np.zeros(n, np.float32),
               polar2_own=np.zeros(n, np.float32), flags=np.zeros(n, np.int32))
    lib().hc_fe_points(C.byref(prm), _p(xyzi), n, t0, _p(out["type"]), _p(out["label"]), _p(out["depth2"]), _p(out["curv"]),
                       _p(out["view"]), _p(out["tstamp"]), _p(out["polar2_own"]), _p(out["flags"]))
    return out


def select(type_, label, depth2, min_blur, max_blur):
    n = len(type_)
    ci, si, fi = (np.zeros(n, np.int32) for _ in range(3))
    nc, ns, nf = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    lib().hc_select(n, _p(type_), _p(label), _p(depth2), min_blur, max_blur, _p(ci), C.byref(nc), _p(si), C.byref(ns),
                    _p(fi), C.byref(nf))
    return ci[:nc.value].copy(), si[:ns.value].copy(), fi[:nf.value].copy()


class Grid:
    def __init__(self, xyz, cell):
        self.xyz = np.ascontiguousarray(xyz, np.float32)
        self.h = lib().hc_grid_build(_p(self.xyz), self.xyz.shape[1], self.xyz.shape[0], cell)

    def __del__(self):
        try:
            lib().hc_grid_free(self.h)
        except Exception:
            pass

    def set_guard(self, guard: float):
        """reuse guard band of the search in metres (Grid::guard; 0 = off)"""
        L = lib()
        L.hc_grid_set_guard.argtypes = [C.c_void_p, C.c_float]
        L.hc_grid_set_guard(self.h, float(guard))

    def knn5(self, q, max_d2):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        idx = np.zeros((q.shape[0], 5), np.int32)
        d2 = np.zeros((q.shape[0], 5), np.float32)
        lib().hc_knn5(self.h, _p(q), q.shape[0], max_d2, _p(idx), _p(d2))
        return idx, d2

    def knn5_tile(self, q, max_d2):
        """host model of the wavefront tile search (ll_knn_tile.h), 64 queries per wavefront in the order given:
        (idx5, d2, lb2, stats = [rounds, candidates staged, lanes that fell back to the per-lane search, wavefronts])"""
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        idx = np.zeros((q.shape[0], 5), np.int32)
        d2 = np.zeros((q.shape[0], 5), np.float32)
        lb2 = np.zeros(q.shape[0], np.float32)
        stats = np.zeros(4, np.int64)
        L = lib()
        L.hc_knn5_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 4
        rc = L.hc_knn5_tile(self.h, _p(q), q.shape[0], max_d2, _p(idx), _p(d2), _p(lb2), _p(stats))
        assert rc == 0, "tile of more than 25 rows"
        return idx, d2, lb2, stats

    def knn5_run_cands(self, q, max_d2):
        """candidates examined per run of the 3x3x3 block, [nq][9] (-1 = run not scanned)"""
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        out = np.zeros((q.shape[0], 9), np.int32)
        L = lib()
        L.hc_knn5_run_cands.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.hc_knn5_run_cands(self.h, _p(q), q.shape[0], max_d2, _p(out))
        return out

    def knn5_work(self, q, max_d2):
        """(rows looked up, candidates examined, deepest phase) per query -- instrumentation of the device search"""
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        out = [np.zeros(q.shape[0], np.int32) for _ in range(3)]
        L = lib()
        L.hc_knn5_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 3
        L.hc_knn5_work(self.h, _p(q), q.shape[0], max_d2, *[_p(o) for o in out])
        return out

    def knn5_bounds(self, q, max_d2):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        cand = np.zeros((q.shape[0], lib().hc_knn_k()), np.int32)
        out = [np.zeros(q.shape[0], np.float32) for _ in range(4)]
        L = lib()
        L.hc_knn5_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 5
        L.hc_knn5_bounds(self.h, _p(q), q.shape[0], max_d2, _p(cand), *[_p(o) for o in out])
        return [cand] + out  # candidates, lb2, out2, m_set, m_strong

    def knn5_reuse_chain(self, path, max_d2):
        """path [n_hops][nq][3] -> (idx5 [n_hops][nq][5], state [n_hops][nq]); state 0 kept, 1 re-sorted, 2 searched"""
        path = np.ascontiguousarray(path, np.float32)
        n_hops, nq = path.shape[0], path.shape[1]
        idx = np.zeros((n_hops, nq, 5), np.int32)
        st = np.zeros((n_hops, nq), np.int32)
        L = lib()
        L.hc_knn5_reuse_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.hc_knn5_reuse_chain(self.h, _p(path), n_hops, nq, max_d2, _p(idx), _p(st))
        return idx, st


def eval_blocks(kind, f, a, v, x, huber_a=0.1, sblur=None):
    kind = np.ascontiguousarray(kind, np.int32)
    sb = None if sblur is None else np.ascontiguousarray(sblur, np.float64)
    f, a, v = (np.ascontiguousarray(t, np.float64) for t in (f, a, v))
    x = np.ascontiguousarray(x, np.float64)
    acc = np.zeros(28)
    lib().hc_eval_blocks(len(kind), _p(kind), _p(f), _p(a), _p(v), _p(x), huber_a, _p(acc), 0 if sb is None else 1, None if sb is None else _p(sb))
    H = np.zeros((6, 6))
    k = 0
    for i in range(6):
        for j in range(i, 6):
            H[i, j] = H[j, i] = acc[k]
            k += 1
    return acc[27], acc[21:27].copy(), H


def make_block(kind, pose_last, pa, pb, pc=None):
    a, v = np.zeros(3), np.zeros(3)
    pc = np.zeros(3) if pc is None else np.asarray(pc, np.float64)
    ok = lib().hc_make_block(kind, _p(np.asarray(pose_last, np.float64)), _p(np.asarray(pa, np.float64)),
                             _p(np.asarray(pb, np.float64)), _p(pc), _p(a), _p(v))
    return ok, a, v


def reg_solve(gc: Grid, gs: Grid, corner, surf, prm: RegParams, pose_last, pose_curr, inc=None):
    corner = np.ascontiguousarray(corner, np.float32).reshape(-1, 4)
    surf = np.ascontiguousarray(surf, np.float32).reshape(-1, 4)
    pl = np.ascontiguousarray(pose_last, np.float64).copy()
    pc = np.ascontiguousarray(pose_curr, np.float64).copy()
    pi = np.array([0, 0, 0, 1, 0, 0, 0], np.float64) if inc is None else np.ascontiguousarray(inc, np.float64).copy()
    rep = np.zeros(10)
    ret = lib().hc_reg_solve(gc.h, gs.h, _p(corner), corner.shape[0], _p(surf), surf.shape[0], C.byref(prm), _p(pl), _p(pc),
                             _p(pi), _p(rep))
    return ret, pc, pi, rep


def pca_check(is_plane, pts5):
    p = np.ascontiguousarray(pts5, np.float32).reshape(15)
    ev = np.zeros(3)
    L = lib()
    L.hc_pca_check.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.hc_pca_check.restype = C.c_int
    return bool(L.hc_pca_check(int(is_plane), p.ctypes.data, ev.ctypes.data)), ev


class CellMap:
    """Serial stand-in for the device cell map (hostcheck.cpp, hc_cellmap_*)."""

    def __init__(self, resolution=1.0, minimum_revisit_threshold=2**31 - 1):
        L = lib()
        L.hc_cellmap_create.restype = C.c_void_p
        L.hc_cellmap_create.argtypes = [C.c_float, C.c_int]
        L.hc_cellmap_free.argtypes = [C.c_void_p]
        L.hc_cellmap_append.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hc_cellmap_query_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.hc_cellmap_sizes.argtypes = [C.c_void_p] * 4
        L.hc_cellmap_dump.argtypes = [C.c_void_p] * 5
        self.L, self.h = L, L.hc_cellmap_create(resolution, int(minimum_revisit_threshold))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hc_cellmap_free(self.h)
            self.h = None

    def append(self, cloud):
        c = np.ascontiguousarray(cloud, np.float32).reshape(-1, 4)
        self.L.hc_cellmap_append(self.h, c.ctypes.data, c.shape[0])

    def sizes(self):
        a, b, f = C.c_int(0), C.c_int(0), C.c_int(0)
        self.L.hc_cellmap_sizes(self.h, C.byref(a), C.byref(b), C.byref(f))
        return a.value, b.value, f.value

    def query_filter(self, pose, radius, max_fov, leaf, replace=1):
        pose = np.ascontiguousarray(pose, np.float64)
        cap = self.sizes()[1]
        out = np.zeros((max(cap, 1), 4), np.float32)
        nsel = C.c_int(0)
        n = self.L.hc_cellmap_query_filter(self.h, pose.ctypes.data, radius, max_fov, leaf, int(replace), out.ctypes.data, cap, C.byref(nsel))
        if n < 0:
            raise ValueError("leaf too small")
        return out[:n].copy(), nsel.value

    def features(self):
        nc = self.sizes()[0]
        o = np.zeros((max(nc, 1), 16), np.float32)
        self.L.hc_cellmap_features.argtypes = [C.c_void_p, C.c_void_p]
        self.L.hc_cellmap_features(self.h, o.ctypes.data)
        o = o[:nc]
        return dict(type=o[:, 0].astype(np.int32), vector=o[:, 1:4].copy(), mean=o[:, 4:7].copy(), cov=o[:, 7:13].copy(),
                    eigen_val=o[:, 13:16].copy())

    def keyframe_images(self, roi_ratio=0.9):
        img = np.zeros((4, 60, 60), np.float32)
        ratio, R, nv, cr = np.zeros(4, np.float32), np.zeros((2, 3, 3), np.float32), np.zeros(4, np.int32), np.zeros(4, np.float32)
        self.L.hc_cellmap_keyframe.argtypes = [C.c_void_p, C.c_float] + [C.c_void_p] * 5
        self.L.hc_cellmap_keyframe(self.h, roi_ratio, img.ctypes.data, ratio.ctypes.data, R.ctypes.data, nv.ctypes.data, cr.ctypes.data)
        return dict(images=img, ratio_nonzero=ratio, eigen_R=R, n_vectors=nv, centre=cr[:3].copy(), roi_range=float(cr[3]))

    def dump(self):
        nc, npts, _ = self.sizes()
        xyzi = np.zeros((max(npts, 1), 4), np.float32)
        ijk = np.zeros((max(nc, 1), 3), np.int32)
        start = np.zeros(nc + 1, np.int32)
        last = np.zeros(max(nc, 1), np.int32)
        self.L.hc_cellmap_dump(self.h, xyzi.ctypes.data, ijk.ctypes.data, start.ctypes.data, last.ctypes.data)
        return xyzi[:npts, :3].copy(), ijk[:nc].copy(), start, last[:nc].copy()


def quintic_min_step(f0, g0, x1, f1, g1, x2, f2, g2, lo, hi):
    """ll_reg_core.h lm_quintic_min_step compiled for the host"""
    L = lib()
    L.hc_quintic_min_step.restype = C.c_double
    L.hc_quintic_min_step.argtypes = [C.c_double] * 10
    return L.hc_quintic_min_step(f0, g0, x1, f1, g1, x2, f2, g2, lo, hi)


def plane_scaled(R, t, f, v, a0, huber_a, q_last):
    """(scaled accumulators [28], block_accumulate's [28], scaled L1, block_l1) of one plane block: ll_reg_core.h plane_* against the un-scaled forms"""
    L = lib()
    out = np.zeros(58)
    arrs = [np.ascontiguousarray(x, np.float64) for x in (R, t, f, v, q_last)]
    L.hc_plane_scaled.argtypes = [C.c_void_p] * 4 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.hc_plane_scaled(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data, float(a0), float(huber_a), arrs[4].ctypes.data, out.ctypes.data)
    return out[:28], out[28:56], out[56], out[57]
