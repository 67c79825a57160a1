"""ctypes binding of libloamlivox_hip.so (include/loam_livox_hip.h).

The product path is the HIP library: if it is missing or cannot be loaded this module raises -- there is no
CPU fallback (the CPU restatement under oracle/ is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOAM_LIVOX_LIB", os.path.join(_HERE, "libloamlivox_hip.so"))  # override: experiments only


class LoamLivoxError(RuntimeError):
    pass


class FeParams(C.Structure):
    _fields_ = [("thr_corner_curvature", C.c_float), ("thr_surface_curvature", C.c_float),
                ("minimum_view_angle", C.c_float), ("livox_min_allow_dis", C.c_float), ("livox_min_sigma", C.c_float),
                ("max_fov", C.c_float), ("time_internal_pts", C.c_float), ("device", C.c_int32),
                ("max_points", C.c_int32), ("max_scans", C.c_int32), ("piecewise_number", C.c_int32)]


class RegParams(C.Structure):
    _fields_ = [("if_motion_deblur", C.c_int32), ("icp_max_iterations", C.c_int32), ("ceres_max_iterations", C.c_int32),
                ("ceres_prerun_times", C.c_int32), ("icp_line", C.c_int32), ("icp_plane", C.c_int32),
                ("current_frame_index", C.c_int32), ("mapping_init_accumulate_frames", C.c_int32),
                ("maximum_allow_residual_block", C.c_int32), ("force_all_iterations", C.c_int32),
                ("maximum_dis_line_for_match", C.c_double), ("maximum_dis_plane_for_match", C.c_double),
                ("huber_a", C.c_double), ("inliner_dis", C.c_double), ("inlier_ratio", C.c_double),
                ("minimum_icp_R_diff", C.c_double), ("minimum_icp_T_diff", C.c_double),
                ("para_max_angular_rate", C.c_float), ("para_max_speed", C.c_float), ("max_final_cost", C.c_float),
                ("minimum_pt_time_stamp", C.c_float), ("maximum_pt_time_stamp", C.c_float),
                ("if_line_feature_check", C.c_int32), ("if_plane_feature_check", C.c_int32), ("subsample_seed", C.c_int32)]


class RegReport(C.Structure):
    _fields_ = [("final_cost", C.c_double), ("initial_cost", C.c_double), ("inlier_threshold", C.c_double),
                ("angular_diff_deg", C.c_double), ("t_diff", C.c_double), ("icp_iterations", C.c_int32),
                ("n_blocks_last", C.c_int32), ("corner_avail", C.c_int32), ("surf_avail", C.c_int32),
                ("lm_iterations_total", C.c_int32), ("accepted", C.c_int32), ("gated", C.c_int32), ("aborted", C.c_int32)]


# name -> (restype, argtypes); every symbol include/loam_livox_hip.h declares
_vp, _i32, _i64, _f, _d = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
SYMBOLS = {
    "ll_fe_default_params": (None, [C.POINTER(FeParams)]),
    "ll_fe_create": (_i32, [C.POINTER(FeParams), C.POINTER(_vp)]),
    "ll_fe_destroy": (None, [_vp]),
    "ll_fe_extract": (_i32, [_vp, _vp, _i32, _d, C.POINTER(_i32)]),
    "ll_fe_labels": (_i32, [_vp, _i32] + [_vp] * 8),
    "ll_fe_splits": (_i32, [_vp, _i32, _vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp, _vp, _vp, _vp]),
    "ll_fe_select": (_i32, [_vp, _f, _f, _vp, C.POINTER(_i32), _vp, C.POINTER(_i32), _vp, C.POINTER(_i32), _vp, _vp]),
    "ll_fe_selection": (_i32, [_vp, _i32, _vp, C.POINTER(_i32), _vp, C.POINTER(_i32), _vp, C.POINTER(_i32), _vp, _vp]),
    "ll_fe_upload": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp]),
    "ll_fe_upload_async": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp]),
    "ll_fe_extract_batch": (_i32, [_vp, _i32]),
    "ll_fe_select_batch": (_i32, [_vp, _i32, _i32, _f, _f]),
    "ll_fe_counts": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "ll_fe_sync": (_i32, [_vp]),
    "ll_fe_resolve": (_i32, [_vp]),
    "ll_map_create": (_i32, [_i32, C.POINTER(_vp)]),
    "ll_map_destroy": (None, [_vp]),
    "ll_map_upload": (_i32, [_vp, _i32, _vp, _i32, _i64, _f]),
    "ll_map_upload_gen": (_i32, [_vp, _i32, _vp, _i32, _i64, _f, _vp]),
    "ll_map_size": (_i64, [_vp, _i32]),
    "ll_map_generation": (_i64, [_vp, _i32]),
    "ll_map_to_f16": (_i32, [_vp, _i32]),
    "ll_map_dequantized": (_i32, [_vp, _i32, _vp, _i64]),
    "ll_map_cells": (_i64, [_vp, _i32]),
    "ll_map_knn5": (_i32, [_vp, _i32, _vp, _i32, _f, _vp, _vp]),
    "ll_map_knn5_device": (_i32, [_vp, _i32, _vp, _i64, _f, _vp, _vp, _vp]),
    "ll_reg_default_params": (None, [C.POINTER(RegParams)]),
    "ll_reg_create": (_i32, [_i32, _i32, _i32, C.POINTER(_vp)]),
    "ll_reg_destroy": (None, [_vp]),
    "ll_reg_solve": (_i32, [_vp, _vp, _vp, _i32, _vp, _i32, C.POINTER(RegParams), _vp, _vp, _vp, C.POINTER(RegReport)]),
    "ll_reg_solve_batch_fe": (_i32, [_vp, _vp, _vp, _i32, C.POINTER(RegParams), _vp, _vp, _vp, _vp, _vp]),
    "ll_reg_solve_batch": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, C.POINTER(RegParams), _vp, _vp, _vp,
                                  _vp, _vp]),
    "ll_reg_upload_features": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32]),
    "ll_reg_enqueue_uploaded": (_i32, [_vp, _vp, _i32, C.POINTER(RegParams), _vp, _vp, _vp]),
    "ll_reg_enqueue_fe": (_i32, [_vp, _vp, _vp, _i32, C.POINTER(RegParams), _vp, _vp, _vp]),
    "ll_reg_enqueue_fe_merged": (_i32, [_vp, _vp, _vp, _i32, _i32, C.POINTER(RegParams), _vp, _vp, _vp]),
    "ll_reg_collect": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "ll_reg_debug_knn": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "ll_reg_set_debug": (_i32, [_vp, _i32]),
    "ll_reg_set_debug_knn_iteration": (_i32, [_vp, _i32]),
    "ll_cloud_transform": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "ll_cloud_transform_fe_device": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "ll_reg_set_profiling": (_i32, [_vp, _i32]),
    "ll_reg_kernel_times": (_i32, [_vp, _vp, _vp]),
    "ll_reg_debug_cycles": (_i32, [_vp, _i32, _vp]),
    "ll_reg_debug_worklists": (_i32, [_vp, _i32, _vp]),
    "ll_voxel_create": (_i32, [_i32, _i32, _i32, C.POINTER(_vp)]),
    "ll_voxel_destroy": (None, [_vp]),
    "ll_voxel_filter": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ll_voxel_counts": (_i32, [_vp, _i32, _vp, _vp]),
    "ll_reg_enqueue_fe_downsampled": (_i32, [_vp, _vp, _vp, _vp, _vp, _f, _f, _i32, C.POINTER(RegParams), _vp, _vp, _vp]),
    "ll_history_create": (_i32, [_i32, _i32, _i32, _f, _f, C.POINTER(_vp)]),
    "ll_history_destroy": (None, [_vp]),
    "ll_history_add": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, C.c_double, C.c_double, _vp]),
    "ll_history_add_fe": (_i32, [_vp, _vp, _i32, _vp, C.c_double, C.c_double, _vp]),
    "ll_history_add_voxel": (_i32, [_vp, _vp, _vp, _i32, _vp, C.c_double, C.c_double, _vp]),
    "ll_history_refresh": (_i32, [_vp, _vp, _vp, _vp]),
    "ll_history_size": (_i32, [_vp]),
    "ll_history_map_cloud": (_i64, [_vp, _i32, _vp, _i64]),
    "ll_history_set_gate_pose": (_i32, [_vp, _vp]),
    "ll_history_map_cloud_device": (_i32, [_vp, _i32, _vp, _vp]),
    "ll_cellmap_create": (_i32, [_i32, _i64, C.c_float, _i32, _vp]),
    "ll_cellmap_destroy": (None, [_vp]),
    "ll_cellmap_reserve": (_i32, [_vp, _i64]),
    "ll_cellmap_device_view": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "ll_cellmap_append": (_i32, [_vp, _vp, _i32]),
    "ll_cellmap_append_touched": (_i32, [_vp, _vp, _i32, _i32, _vp, _i64, C.POINTER(_i64)]),
    "ll_cellmap_query_filter": (_i32, [_vp, _vp, C.c_float, C.c_float, C.c_float, _i32, _vp, _vp]),
    "ll_cellmap_result": (_i64, [_vp, _vp, _i64]),
    "ll_cellmap_stats": (_i32, [_vp, _vp, _vp, _vp]),
    "ll_cellmap_dump": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _i64]),
    "ll_cellmap_features": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "ll_cellmap_keyframe_images": (_i32, [_vp, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "ll_keyframe_similarity": (_i32, [_i32, _vp, _vp, _vp]),
    "ll_history_enable_cell_map": (_i32, [_vp, _i64, C.c_float, _i32]),
    "ll_history_set_cell_map_async": (_i32, [_vp, _i32]),
    "ll_history_sync_cell_maps": (_i32, [_vp]),
    "ll_history_cell_map": (_vp, [_vp, _i32]),
    "ll_history_refresh_cells": (_i32, [_vp, _vp, _vp, C.c_float, C.c_float, C.c_float, _i32, _vp, _vp]),
    "ll_debug_quintic": (_i32, [_i32, _vp, _i32, _vp, _vp]),
    "ll_reg_stream": (_vp, [_vp]),
    "ll_fe_stream": (_vp, [_vp]),
    "ll_runtime_hint_hw_queues": (_i32, [_i32]),
    "ll_last_error": (C.c_char_p, []),
    "ll_version": (C.c_char_p, []),
}

_lib = None


# Importing this module does NOT touch the process environment.  Applications that keep batches in flight on several handles should set
# GPU_MAX_HW_QUEUES (bench*.py do, before their first HIP call) or call hint_hw_queues() below first thing: the ROCm runtime multiplexes a
# process's HIP streams onto 4 hardware queues by default (measured, profiles/README.md round 5: 44.0 k scans/s with 4 queues, 46.5 k with 12 - 32).


def hint_hw_queues(n: int = 16) -> bool:
    """ll_runtime_hint_hw_queues: GPU_MAX_HW_QUEUES=n for this process unless the environment sets it; effective only before the first HIP call"""
    return check(load().ll_runtime_hint_hw_queues(int(n)), "ll_runtime_hint_hw_queues") == 1


def load():
    """Load libloamlivox_hip.so and bind every symbol.  Raises LoamLivoxError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LoamLivoxError(
            f"{LIB_PATH} not found: build it with `python -m loam_livox_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise LoamLivoxError(f"cannot load {LIB_PATH}: {e} (ROCm runtime required; there is no CPU fallback)") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class use_library:
    """Context manager for tests and A/B runs: objects created inside bind to the library at `path` (a variant build of the same
    sources, e.g. a -DLL_SOLVE_TIMING build) instead of the product library.  Each object keeps the library it was created with."""

    def __init__(self, path: str):
        self.path = path

    def __enter__(self):
        global _lib, LIB_PATH
        self.saved = (_lib, LIB_PATH)
        _lib, LIB_PATH = None, self.path
        try:
            return load()
        except Exception:
            _lib, LIB_PATH = self.saved  # (a variant that does not load leaves the product library in place)
            raise

    def __exit__(self, *exc):
        global _lib, LIB_PATH
        _lib, LIB_PATH = self.saved
        return False


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = load().ll_last_error()
        raise LoamLivoxError(f"{what}: {msg.decode() if msg else 'error'}")
    return rc


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fe_default_params() -> FeParams:
    p = FeParams()
    load().ll_fe_default_params(C.byref(p))
    return p


def reg_default_params() -> RegParams:
    p = RegParams()
    load().ll_reg_default_params(C.byref(p))
    return p


def as_f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a
