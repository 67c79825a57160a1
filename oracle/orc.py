"""ctypes binding of the CPU ORACLE (oracle/libll_oracle.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package loam_livox_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libll_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("ll_oracle_fe.c", "ll_oracle_kdtree.c", "ll_oracle_reg.c", "ll_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class FeParams(C.Structure):
    _fields_ = [("thr_corner_curvature", C.c_float), ("thr_surface_curvature", C.c_float),
                ("minimum_view_angle", C.c_float), ("livox_min_allow_dis", C.c_float),
                ("livox_min_sigma", C.c_float), ("max_fov", C.c_float), ("time_internal_pts", C.c_float)]

    @staticmethod
    def node_defaults():
        """Values the extractor node sets (LFX:146-154,854,859)."""
        return FeParams(0.05, 0.01, 10.0, 0.1, 7e-4, 17.0, 1.0e-5)


class Timebase(C.Structure):
    _fields_ = [("first_receive_time", C.c_double), ("current_time", C.c_double),
                ("last_maximum_time_stamp", C.c_double)]


class FeTimebase:
    """The time base of one Livox_laser instance across messages (LFE:722-735): next(stamp) -> m_current_time."""

    def __init__(self):
        self.tb = Timebase()
        lib().orc_fe_timebase_init(C.byref(self.tb))

    def next(self, stamp: float) -> float:
        return float(lib().orc_fe_timebase_next(C.byref(self.tb), float(stamp)))

    def done(self, result) -> None:
        """after the extraction: m_last_maximum_time_stamp = time stamp of the scan's last point (LFE:482)"""
        self.tb.last_maximum_time_stamp = result.last_time_stamp


class RegParams(C.Structure):
    _fields_ = [("if_motion_deblur", C.c_int), ("icp_max_iterations", C.c_int), ("ceres_max_iterations", C.c_int),
                ("ceres_prerun_times", C.c_int), ("line_search_num", C.c_int), ("plane_search_num", C.c_int),
                ("icp_line", C.c_int), ("icp_plane", C.c_int), ("current_frame_index", C.c_int),
                ("mapping_init_accumulate_frames", C.c_int), ("force_all_iterations", C.c_int),
                ("maximum_dis_line_for_match", C.c_double), ("maximum_dis_plane_for_match", C.c_double),
                ("huber_a", C.c_double), ("inliner_dis", C.c_double), ("inlier_ratio", C.c_double),
                ("minimum_icp_R_diff", C.c_double), ("minimum_icp_T_diff", C.c_double),
                ("para_max_angular_rate", C.c_float), ("para_max_speed", C.c_float), ("max_final_cost", C.c_float),
                ("minimum_pt_time_stamp", C.c_float), ("maximum_pt_time_stamp", C.c_float),
                ("if_line_feature_check", C.c_int), ("if_plane_feature_check", C.c_int),
                ("maximum_allow_residual_block", C.c_int), ("subsample_seed", C.c_int)]

    @staticmethod
    def defaults(icp_iters=10, ceres_iters=20, force_all=0, deblur=0):
        """Code defaults (PCR:45-98; max_final_cost = class default 100, PCR:88) with launch/rosbag.launch
        bounds (max_allow_incre_R 20, max_allow_incre_T 0.3); sub-sampling disabled."""
        return RegParams(deblur, icp_iters, ceres_iters, 2, 5, 5, 1, 1, 100, 50, force_all,
                         2.0, 50.0, 0.1, 0.02, 0.8, 0.01, 0.01, 20.0, 0.3, 100.0, 0.0, 1.0, 0, 0, 99999, 0)


    @staticmethod
    def code_defaults():
        """The member initialisers of Point_cloud_registration alone (PCR:45-103): what a default-constructed registrar
        (Scene_alignment::m_pc_reg, scene_alignment.hpp:32) runs with."""
        return RegParams(0, 20, 100, 2, 5, 5, 1, 1, 101, 100, 0, 2.0, 50.0, 0.1, 0.02, 0.8, 0.01, 0.01, 200.0 / 50.0, 100.0 / 50.0, 100.0,
                         0.0, 1.0, 0, 0, 100000, 0)


class RegReport(C.Structure):
    _fields_ = [("final_cost", C.c_double), ("initial_cost", C.c_double), ("inlier_threshold", C.c_double),
                ("angular_diff_deg", C.c_double), ("t_diff", C.c_double), ("icp_iterations", C.c_int),
                ("n_blocks_last", C.c_int), ("corner_avail", C.c_int), ("surf_avail", C.c_int),
                ("lm_iterations_total", C.c_int), ("accepted", C.c_int), ("gated", C.c_int)]


class Block(C.Structure):
    _fields_ = [("kind", C.c_int), ("f", C.c_double * 3), ("a", C.c_double * 3), ("v", C.c_double * 3),
                ("s", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.orc_fe_timebase_init.argtypes = [C.POINTER(Timebase)]
        L.orc_fe_timebase_next.argtypes = [C.POINTER(Timebase), C.c_double]
        L.orc_fe_timebase_next.restype = C.c_double
        L.orc_fe_max_edge_polar_pos.argtypes = [C.c_float]
        L.orc_fe_max_edge_polar_pos.restype = C.c_float
        L.orc_fe_extract.argtypes = [C.POINTER(FeParams), fp, C.c_int, C.c_double, ip, ip, fp, fp, ip, fp, fp, fp, fp,
                                     fp, fp, ip, ip, fp]
        L.orc_fe_extract.restype = C.c_int
        L.orc_fe_get_features.argtypes = [C.c_int, ip, ip, fp, C.c_float, C.c_float, ip, ip, ip, ip, ip, ip]
        L.orc_fe_split_scan.argtypes = [C.c_int, C.c_int, fp, ip, fp, ip, ip]
        L.orc_fe_split_scan.restype = C.c_int
        L.orc_fe_piecewise.argtypes = [C.c_int, fp, C.c_int, ip, ip, C.c_int, fp, fp]
        L.orc_kdtree_build.argtypes = [fp, C.c_int, C.c_int64]
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_kdtree_knn.argtypes = [C.c_void_p, fp, C.c_int, ip, fp]
        L.orc_kdtree_knn.restype = C.c_int
        L.orc_bruteforce_knn.argtypes = [fp, C.c_int, C.c_int64, fp, C.c_int, ip, fp]
        L.orc_bruteforce_knn.restype = C.c_int
        L.orc_reg_solve.argtypes = [C.c_void_p, fp, C.c_int64, C.c_void_p, fp, C.c_int64, C.c_int, fp, C.c_int, fp,
                                    C.c_int, C.POINTER(RegParams), dp, dp, dp, C.POINTER(RegReport)]
        L.orc_reg_solve.restype = C.c_int
        L.orc_block_line.argtypes = [C.POINTER(Block), dp, dp, dp, C.c_double]
        L.orc_block_plane.argtypes = [C.POINTER(Block), dp, dp, dp, dp, C.c_double]
        L.orc_block_residual.argtypes = [C.POINTER(Block), dp, dp, C.c_int, dp]
        L.orc_blocks_eval.argtypes = [C.POINTER(Block), C.c_int, dp, dp, C.c_int, C.c_double, dp, dp, dp]
        L.orc_point_to_map.argtypes = [dp, fp, fp]
        L.orc_cloud_transform.argtypes = [dp, fp, fp, C.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class FeResult:
    pass


def fe_extract(xyzi: np.ndarray, current_time: float = 0.0, params: FeParams | None = None) -> FeResult:
    L = lib()
    p = params or FeParams.node_defaults()
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32)
    n = xyzi.shape[0]
    r = FeResult()
    r.n = n
    r.pt_type = np.zeros(n, np.int32)
    r.pt_label = np.zeros(n, np.int32)
    r.time_stamp = np.zeros(n, np.float32)
    r.polar_angle = np.zeros(n, np.float32)
    r.polar_direction = np.zeros(n, np.int32)
    r.polar_dis_sq2 = np.zeros(n, np.float32)
    r.depth_sq2 = np.zeros(n, np.float32)
    r.curvature = np.zeros(n, np.float32)
    r.view_angle = np.zeros(n, np.float32)
    r.sigma = np.zeros(n, np.float32)
    r.img2d = np.zeros((n, 2), np.float32)
    split = np.zeros(n + 2, np.int32)
    nsplit = C.c_int32(0)
    last_ts = C.c_float(0)
    r.n_petals = L.orc_fe_extract(C.byref(p), _fp(xyzi), n, current_time, _ip(r.pt_type), _ip(r.pt_label),
                                  _fp(r.time_stamp), _fp(r.polar_angle), _ip(r.polar_direction),
                                  _fp(r.polar_dis_sq2), _fp(r.depth_sq2), _fp(r.curvature), _fp(r.view_angle),
                                  _fp(r.sigma), _fp(r.img2d), _ip(split), C.byref(nsplit), C.byref(last_ts))
    r.split_idx = split[:nsplit.value].copy()
    r.last_time_stamp = last_ts.value
    r.xyzi = xyzi
    return r


def fe_get_features(r: FeResult, min_blur: float = 0.0, max_blur: float = 0.3):
    L = lib()
    n = r.n
    ci, si, fi = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    nc, ns, nf = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    L.orc_fe_get_features(n, _ip(r.pt_type), _ip(r.pt_label), _fp(r.depth_sq2), min_blur, max_blur, _ip(ci),
                          C.byref(nc), _ip(si), C.byref(ns), _ip(fi), C.byref(nf))
    return ci[:nc.value].copy(), si[:ns.value].copy(), fi[:nf.value].copy()


def fe_split_scan(r: FeResult):
    L = lib()
    cap = max(1, r.n_petals)
    first, last = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    s = L.orc_fe_split_scan(r.n, r.n_petals, _fp(r.xyzi), _ip(r.pt_type), _fp(r.polar_angle), _ip(first), _ip(last))
    return s, first[:s].copy(), last[:s].copy()


def fe_piecewise(r, first, last, pieces):
    # r: FeResult (its xyzi feeds the find_pt_info first-occurrence lookup)
    L = lib()
    n = r.n
    ps, pe = np.zeros(pieces, np.float32), np.zeros(pieces, np.float32)
    first = np.ascontiguousarray(first, np.int32)
    last = np.ascontiguousarray(last, np.int32)
    L.orc_fe_piecewise(n, _fp(r.xyzi), len(first), _ip(first), _ip(last), pieces, _fp(ps), _fp(pe))
    return ps, pe


def feature_cloud(r: FeResult, idx: np.ndarray) -> np.ndarray:
    """xyz + time stamp as intensity (LFE:244-246,254-255)."""
    out = r.xyzi[idx].copy()
    out[:, 3] = r.time_stamp[idx]
    return np.ascontiguousarray(out)


class KdTree:
    def __init__(self, xyz: np.ndarray):
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        self.stride = self.xyz.shape[1]
        self.h = lib().orc_kdtree_build(_fp(self.xyz), self.stride, self.xyz.shape[0])

    def __del__(self):
        try:
            if self.h:
                lib().orc_kdtree_free(self.h)
                self.h = None
        except Exception:
            pass

    def knn(self, q: np.ndarray, k: int = 5):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 3)
        idx = np.full((q.shape[0], k), -1, np.int32)
        d2 = np.full((q.shape[0], k), np.inf, np.float32)
        L = lib()
        for i in range(q.shape[0]):
            L.orc_kdtree_knn(self.h, _fp(q[i]), k, _ip(idx[i]), _fp(d2[i]))
        return idx, d2


def bruteforce_knn(xyz: np.ndarray, q: np.ndarray, k: int = 5):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 3)
    idx = np.full((q.shape[0], k), -1, np.int32)
    d2 = np.full((q.shape[0], k), np.inf, np.float32)
    L = lib()
    for i in range(q.shape[0]):
        L.orc_bruteforce_knn(_fp(xyz), xyz.shape[1], xyz.shape[0], _fp(q[i]), k, _ip(idx[i]), _fp(d2[i]))
    return idx, d2


def reg_solve(tree_c: KdTree, tree_s: KdTree, scan_corner: np.ndarray, scan_surf: np.ndarray, prm: RegParams,
              pose_last: np.ndarray, pose_curr: np.ndarray, pose_incre: np.ndarray | None = None):
    L = lib()
    sc = np.ascontiguousarray(scan_corner, np.float32).reshape(-1, 4)
    ss = np.ascontiguousarray(scan_surf, np.float32).reshape(-1, 4)
    pl = np.ascontiguousarray(pose_last, np.float64).copy()
    pc = np.ascontiguousarray(pose_curr, np.float64).copy()
    pi = np.array([0, 0, 0, 1, 0, 0, 0], np.float64) if pose_incre is None else np.ascontiguousarray(
        pose_incre, np.float64).copy()
    rep = RegReport()
    if tree_c is None or tree_s is None:  # an empty map: only the PCR:199 gate can be reached
        ret = L.orc_reg_solve(None, None, 0, None, None, 0, 4, _fp(sc), sc.shape[0], _fp(ss), ss.shape[0],
                              C.byref(prm), _dp(pl), _dp(pc), _dp(pi), C.byref(rep))
        return ret, pc, pi, rep
    assert tree_c.stride == tree_s.stride
    ret = L.orc_reg_solve(tree_c.h, _fp(tree_c.xyz), tree_c.xyz.shape[0], tree_s.h, _fp(tree_s.xyz),
                          tree_s.xyz.shape[0], tree_c.stride, _fp(sc), sc.shape[0], _fp(ss), ss.shape[0],
                          C.byref(prm), _dp(pl), _dp(pc), _dp(pi), C.byref(rep))
    return ret, pc, pi, rep


def make_block_line(f, a, b, s=1.0) -> Block:
    blk = Block()
    lib().orc_block_line(C.byref(blk), _dp(np.asarray(f, np.float64)), _dp(np.asarray(a, np.float64)),
                         _dp(np.asarray(b, np.float64)), s)
    return blk


def make_block_plane(f, a, b, c, s=1.0) -> Block:
    blk = Block()
    lib().orc_block_plane(C.byref(blk), _dp(np.asarray(f, np.float64)), _dp(np.asarray(a, np.float64)),
                          _dp(np.asarray(b, np.float64)), _dp(np.asarray(c, np.float64)), s)
    return blk


def block_residual(blk: Block, pose_last, x, deblur=0):
    r = np.zeros(3)
    lib().orc_block_residual(C.byref(blk), _dp(np.asarray(pose_last, np.float64)), _dp(np.asarray(x, np.float64)),
                             deblur, _dp(r))
    return r


def blocks_eval(blocks, pose_last, x, deblur=0, huber_a=0.1):
    arr = (Block * len(blocks))(*blocks)
    cost = C.c_double(0)
    g = np.zeros(6)
    H = np.zeros((6, 6))
    lib().orc_blocks_eval(arr, len(blocks), _dp(np.asarray(pose_last, np.float64)),
                          _dp(np.asarray(x, np.float64)), deblur, huber_a, C.byref(cost), _dp(g), _dp(H))
    return cost.value, g, H


def cloud_transform(pose, xyzi):
    xyzi = np.ascontiguousarray(xyzi, np.float32)
    out = np.empty_like(xyzi)
    lib().orc_cloud_transform(_dp(np.asarray(pose, np.float64)), _fp(xyzi), _fp(out), xyzi.shape[0])
    return out


def pca_check(is_plane, pts5):
    """PCR:259-292 / 357-389 on five points; returns (ok, eigenvalues ascending)."""
    p = np.ascontiguousarray(pts5, np.float32).reshape(15)
    ev = np.zeros(3)
    L = lib()
    L.orc_pca_check.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.orc_pca_check.restype = C.c_int
    ok = L.orc_pca_check(int(is_plane), p.ctypes.data, ev.ctypes.data)
    return bool(ok), ev


def voxel_grid(xyzi, leaf):
    """pcl::VoxelGrid<PointXYZI> (PCL 1.9 semantics, stable in-voxel order).  Returns (status, filtered cloud)."""
    pts = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    lf = np.ascontiguousarray(np.broadcast_to(np.asarray(leaf, np.float32), (3,)))
    out = np.zeros_like(pts)
    n_out = C.c_int32(0)
    L = lib()
    L.orc_voxel_grid.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.orc_voxel_grid.restype = C.c_int
    st = L.orc_voxel_grid(pts.ctypes.data, pts.shape[0], lf.ctypes.data, out.ctypes.data, C.byref(n_out))
    return st, out[:n_out.value].copy()
