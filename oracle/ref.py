"""ctypes binding of oracle/_ref/libll_ref.so: the REFERENCE's own hot-path classes (Livox_laser, the ceres_icp.hpp
functors, Point_cloud_registration) compiled verbatim from /root/reference against the stand-in third-party headers
of oracle/ref_stubs/ (recipe: `make -C oracle ref`).

TEST INFRASTRUCTURE ONLY -- it pins oracle/ (the C restatement) to the reference's text.  The library is git-ignored
and can only be (re)built where /root/reference exists; elsewhere `available()` is False unless a prebuilt copy
travelled with the tree.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libll_ref.so")
REFERENCE_ROOT = "/root/reference"
_lib = None


def can_build() -> bool:
    return os.path.exists(os.path.join(REFERENCE_ROOT, "source", "livox_feature_extractor.hpp"))


def build(force: bool = False) -> str | None:
    """Build the library when the reference sources are present; returns its path or None."""
    if can_build():
        deps = [os.path.join(_HERE, "ref_shim.cpp"), os.path.join(_HERE, "Makefile")]
        stubs = os.path.join(_HERE, "ref_stubs")
        for d, _, fs in os.walk(stubs):
            deps += [os.path.join(d, f) for f in fs]
        if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(_LIB_PATH) for d in deps):
            subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return _LIB_PATH if os.path.exists(_LIB_PATH) else None


def available() -> bool:
    return build() is not None


def lib():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/libll_ref.so is not built and /root/reference is absent")
        L = C.CDLL(path)
        vp, fp, ip, dp = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.ref_fe_create.restype = vp
        L.ref_fe_destroy.argtypes = [vp]
        L.ref_fe_set_params.argtypes = [vp] + [C.c_float] * 6
        L.ref_fe_max_edge_polar_pos.argtypes = [vp]
        L.ref_fe_max_edge_polar_pos.restype = C.c_float
        for f in ("ref_fe_current_time", "ref_fe_first_receive_time", "ref_fe_last_maximum_time_stamp"):
            getattr(L, f).argtypes = [vp]
            getattr(L, f).restype = C.c_double
        L.ref_fe_extract.argtypes = [vp, fp, C.c_int, C.c_double]
        L.ref_fe_num_points.argtypes = [vp]
        L.ref_fe_pts_info.argtypes = [vp, ip, ip, ip, fp, fp, fp, ip, fp, fp, fp, fp, fp, fp]
        L.ref_fe_get_features.argtypes = [vp, C.c_float, C.c_float, fp, ip, ip, fp, ip, ip, fp, ip]
        L.ref_fe_petal_size.argtypes = [vp, C.c_int]
        L.ref_fe_petal.argtypes = [vp, C.c_int, fp, ip]
        L.ref_icp_evaluate.argtypes = [C.c_int, dp, dp, dp, dp, C.c_double, dp, dp, dp, dp, dp, dp]
        L.ref_reg_create.restype = vp
        L.ref_reg_destroy.argtypes = [vp]
        L.ref_reg_set_params.argtypes = [vp] + [C.c_int] * 6 + [C.c_float] * 5 + [C.c_double] * 4 + [C.c_int] * 5
        L.ref_reg_set_maps.argtypes = [vp, fp, C.c_int64, fp, C.c_int64, C.c_int]
        L.ref_reg_solve.argtypes = [vp, fp, C.c_int, fp, C.c_int, dp, dp, dp, dp]
        L.ref_reg_cloud_transform.argtypes = [vp, dp, fp, fp, C.c_int]
        L.ref_reg_refine_blur.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float]
        L.ref_reg_refine_blur.restype = C.c_float
        L.ref_reg_inlier_threshold.argtypes = [vp, dp, C.c_int, C.c_double]
        L.ref_reg_inlier_threshold.restype = C.c_double
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class RefLivoxLaser:
    """One reference `Livox_laser` instance (keeps its time base across calls, LFE:722-736)."""

    def __init__(self, corner_curvature=0.05, surface_curvature=0.01, minimum_view_angle=10.0, min_dis=0.1, min_sigma=7e-4,
                 time_internal_pts=1.0e-5):
        self.L = lib()
        self.h = self.L.ref_fe_create()
        self.L.ref_fe_set_params(self.h, corner_curvature, surface_curvature, minimum_view_angle, min_dis, min_sigma,
                                 time_internal_pts)

    def __del__(self):
        try:
            self.L.ref_fe_destroy(self.h)
        except Exception:
            pass

    def extract(self, xyzi, stamp):
        """extract_laser_features: returns the number of petal clouds."""
        xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        self.n = xyzi.shape[0]
        return self.L.ref_fe_extract(self.h, _fp(xyzi), self.n, float(stamp))

    def current_time(self):
        return self.L.ref_fe_current_time(self.h)

    def max_edge_polar_pos(self):
        return self.L.ref_fe_max_edge_polar_pos(self.h)

    def pts_info(self):
        n = self.L.ref_fe_num_points(self.h)
        d = dict(pt_type=np.zeros(n, np.int32), pt_label=np.zeros(n, np.int32), idx=np.zeros(n, np.int32),
                 raw_intensity=np.zeros(n, np.float32), time_stamp=np.zeros(n, np.float32), polar_angle=np.zeros(n, np.float32),
                 polar_direction=np.zeros(n, np.int32), polar_dis_sq2=np.zeros(n, np.float32), depth_sq2=np.zeros(n, np.float32),
                 curvature=np.zeros(n, np.float32), view_angle=np.zeros(n, np.float32), sigma=np.zeros(n, np.float32),
                 img2d=np.zeros((n, 2), np.float32))
        self.L.ref_fe_pts_info(self.h, _ip(d["pt_type"]), _ip(d["pt_label"]), _ip(d["idx"]), _fp(d["raw_intensity"]),
                               _fp(d["time_stamp"]), _fp(d["polar_angle"]), _ip(d["polar_direction"]), _fp(d["polar_dis_sq2"]),
                               _fp(d["depth_sq2"]), _fp(d["curvature"]), _fp(d["view_angle"]), _fp(d["sigma"]), _fp(d["img2d"]))
        return d

    def get_features(self, min_blur=0.0, max_blur=0.3):
        n = max(1, self.L.ref_fe_num_points(self.h))
        pc, ps, pf = (np.zeros((n, 4), np.float32) for _ in range(3))
        ci, si = np.zeros(n, np.int32), np.zeros(n, np.int32)
        nc, ns, nf = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.L.ref_fe_get_features(self.h, min_blur, max_blur, _fp(pc), _ip(ci), C.byref(nc), _fp(ps), _ip(si), C.byref(ns),
                                   _fp(pf), C.byref(nf))
        return dict(pc_corners=pc[:nc.value].copy(), pc_surface=ps[:ns.value].copy(), pc_full=pf[:nf.value].copy(),
                    corner_idx=ci[:nc.value].copy(), surf_idx=si[:ns.value].copy())

    def petals(self, count):
        out = []
        for k in range(count):
            m = self.L.ref_fe_petal_size(self.h, k)
            pts, idx = np.zeros((max(m, 1), 4), np.float32), np.zeros(max(m, 1), np.int32)
            self.L.ref_fe_petal(self.h, k, _fp(pts), _ip(idx))
            out.append((pts[:m].copy(), idx[:m].copy()))
        return out


def icp_evaluate(kind, f, pa, pb, pc, s, pose_last, x, jac=True):
    """Residual (3) and AutoDiff Jacobians (3x4 wrt the stored quaternion x,y,z,w; 3x3 wrt t) of one ceres_icp.hpp functor.
    kind: 0 point2line, 1 point2plane, 2 point2line_mb, 3 point2plane_mb.  pose_last = {qx,qy,qz,qw,tx,ty,tz}."""
    L = lib()
    a = lambda v: np.ascontiguousarray(v, np.float64)
    pl = a(pose_last)
    qw = a([pl[3], pl[0], pl[1], pl[2]])
    tl = a(pl[4:7])
    r, jq, jt = np.zeros(3), np.zeros((3, 4)), np.zeros((3, 3))
    pcv = a(pc if pc is not None else [0, 0, 0])
    ok = L.ref_icp_evaluate(kind, _dp(a(f)), _dp(a(pa)), _dp(a(pb)), _dp(pcv), float(s), _dp(qw), _dp(tl), _dp(a(x)), _dp(r),
                            _dp(jq) if jac else None, _dp(jt) if jac else None)
    assert ok == 1
    return r, jq, jt


class RefRegistration:
    """One reference `Point_cloud_registration` plus its map clouds / k-d trees."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.ref_reg_create()

    def __del__(self):
        try:
            self.L.ref_reg_destroy(self.h)
        except Exception:
            pass

    def set_params(self, prm):
        """prm: oracle.orc.RegParams (same fields as the members LM:1266-1297 sets)."""
        self.L.ref_reg_set_params(self.h, prm.if_motion_deblur, prm.icp_max_iterations, prm.ceres_max_iterations, prm.ceres_prerun_times,
                                  prm.current_frame_index, prm.mapping_init_accumulate_frames, prm.para_max_angular_rate, prm.para_max_speed,
                                  prm.max_final_cost, prm.minimum_pt_time_stamp, prm.maximum_pt_time_stamp, prm.minimum_icp_R_diff,
                                  prm.minimum_icp_T_diff, prm.inliner_dis, prm.inlier_ratio, prm.maximum_allow_residual_block, prm.icp_line,
                                  prm.icp_plane, prm.if_line_feature_check, prm.if_plane_feature_check)

    def set_maps(self, corner, surf):
        c = np.ascontiguousarray(corner, np.float32)
        s = np.ascontiguousarray(surf, np.float32)
        assert c.shape[1] == s.shape[1]
        self.L.ref_reg_set_maps(self.h, _fp(c), c.shape[0], _fp(s), s.shape[0], c.shape[1])

    def solve(self, scan_corner, scan_surf, pose_last, pose_curr, pose_incre=None):
        sc = np.ascontiguousarray(scan_corner, np.float32).reshape(-1, 4)
        ss = np.ascontiguousarray(scan_surf, np.float32).reshape(-1, 4)
        pl = np.ascontiguousarray(pose_last, np.float64).copy()
        pc = np.ascontiguousarray(pose_curr, np.float64).copy()
        pi = np.array([0, 0, 0, 1, 0, 0, 0], np.float64) if pose_incre is None else np.ascontiguousarray(pose_incre, np.float64).copy()
        rep = np.zeros(8)
        ret = self.L.ref_reg_solve(self.h, _fp(sc), sc.shape[0], _fp(ss), ss.shape[0], _dp(pl), _dp(pc), _dp(pi), _dp(rep))
        report = dict(final_cost=rep[0], initial_cost=rep[1], inlier_threshold=rep[2], angular_diff_deg=rep[3], t_diff=rep[4],
                      n_blocks_last=int(rep[5]), lm_iterations_last=int(rep[6]))
        return ret, pc, pi, report

    def cloud_transform(self, pose, xyzi):
        xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        out = np.empty_like(xyzi)
        self.L.ref_reg_cloud_transform(self.h, _dp(np.ascontiguousarray(pose, np.float64)), _fp(xyzi), _fp(out), xyzi.shape[0])
        return out

    def refine_blur(self, deblur, in_blur, min_blur, max_blur):
        return self.L.ref_reg_refine_blur(self.h, int(deblur), float(in_blur), float(min_blur), float(max_blur))

    def inlier_threshold(self, residuals, ratio):
        r = np.ascontiguousarray(residuals, np.float64).reshape(-1)
        return self.L.ref_reg_inlier_threshold(self.h, _dp(r), r.shape[0], float(ratio))
