"""Host-side mirror of the reference's call surface for the hot path, on top of the C ABI.

Class and method names follow hku-mars/loam_livox so that parity tests read like calls into the reference:

  Livox_laser               source/livox_feature_extractor.hpp:77     (extract_laser_features :722, get_features :219)
  Point_cloud_registration  source/point_cloud_registration.hpp:38    (find_out_incremental_transfrom :163/:585,
                                                                       pointcloudAssociateToMap :673)
  Map_buffer                the (cloud, KdTreeFLANN) pairs of source/laser_mapping.hpp:539-546

Poses are numpy float64[7] = (qx,qy,qz,qw,tx,ty,tz).  PyTorch is not needed here: device memory and streams are
owned by the library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import FeParams, RegParams, RegReport, check, ptr


class Livox_laser:
    """Device-backed Livox_laser.  Tunables are the public fields of the reference class
    (livox_feature_extractor.hpp:143-167); they are fixed at construction."""

    def __init__(self, max_points: int = 24000, max_scans: int = 1, device: int = 0, piecewise_number: int = 3, **tunables):
        self.L = capi.load()
        p = capi.fe_default_params()
        p.max_points, p.max_scans, p.device, p.piecewise_number = max_points, max_scans, device, piecewise_number
        for k, v in tunables.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        self.params = p
        self.h = C.c_void_p()
        check(self.L.ll_fe_create(C.byref(p), C.byref(self.h)), "ll_fe_create")
        self.m_input_points_size = 0
        self._n = [0] * max_scans

    def close(self):
        if getattr(self, "h", None):
            self.L.ll_fe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- per-message path (laser_feature_extractor.hpp:285) ---------------------------------------------------
    def extract_laser_features(self, xyzi: np.ndarray, time_stamp: float) -> int:
        """Returns laserCloudScans.size() (number of surviving petal clouds)."""
        xyzi = capi.as_f32(xyzi, 4)
        n = xyzi.shape[0]
        npc = C.c_int32(0)
        check(self.L.ll_fe_extract(self.h, ptr(xyzi), n, float(time_stamp), C.byref(npc)), "ll_fe_extract")
        self.m_input_points_size = n
        self._n[0] = n
        return npc.value

    def get_features(self, minimum_blur: float = 0.0, maximum_blur: float = 0.3, scan: int | None = None):
        """Returns dict(corner_idx, surf_idx, full_idx, pc_corners, pc_surface) for scan slot 0; with `scan` given: the selection
        select_batch() left in that slot (the blur window is then the one select_batch was called with)."""
        n = self.params.max_points
        ci, si, fi = (np.zeros(n, np.int32) for _ in range(3))
        cc, sc = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
        nc, ns, nf = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        if scan is None:
            check(self.L.ll_fe_select(self.h, minimum_blur, maximum_blur, ptr(ci), C.byref(nc), ptr(si), C.byref(ns), ptr(fi),
                                      C.byref(nf), ptr(cc), ptr(sc)), "ll_fe_select")
        else:
            check(self.L.ll_fe_selection(self.h, int(scan), ptr(ci), C.byref(nc), ptr(si), C.byref(ns), ptr(fi), C.byref(nf), ptr(cc), ptr(sc)),
                  "ll_fe_selection")
        return dict(corner_idx=ci[:nc.value].copy(), surf_idx=si[:ns.value].copy(), full_idx=fi[:nf.value].copy(),
                    pc_corners=cc[:nc.value].copy(), pc_surface=sc[:ns.value].copy())

    def pts_info(self, scan: int = 0):
        """m_pts_info_vec as a dict of arrays."""
        n = self._n[scan]
        out = dict(pt_type=np.zeros(n, np.int32), pt_label=np.zeros(n, np.int32), depth_sq2=np.zeros(n, np.float32),
                   polar_dis_sq2=np.zeros(n, np.float32), curvature=np.zeros(n, np.float32),
                   view_angle=np.zeros(n, np.float32), time_stamp=np.zeros(n, np.float32),
                   polar_angle=np.zeros(n, np.float32))
        check(self.L.ll_fe_labels(self.h, scan, ptr(out["pt_type"]), ptr(out["pt_label"]), ptr(out["depth_sq2"]),
                                  ptr(out["polar_dis_sq2"]), ptr(out["curvature"]), ptr(out["view_angle"]),
                                  ptr(out["time_stamp"]), ptr(out["polar_angle"])), "ll_fe_labels")
        return out

    def splits(self, scan: int = 0):
        cap = self.params.max_points // 50 + 8
        split = np.zeros(cap, np.int32)
        first, last = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        ps, pe = np.zeros(8, np.float32), np.zeros(8, np.float32)
        ns, cl, npc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(self.L.ll_fe_splits(self.h, scan, ptr(split), C.byref(ns), C.byref(cl), C.byref(npc), ptr(first), ptr(last),
                                  ptr(ps), ptr(pe)), "ll_fe_splits")
        P = self.params.piecewise_number
        return dict(split_idx=split[:ns.value].copy(), clutter_size=cl.value, n_petal_clouds=npc.value,
                    first_idx=first[:npc.value].copy(), last_idx=last[:npc.value].copy(), piece_start=ps[:P].copy(),
                    piece_end=pe[:P].copy())

    # -- batched, device-resident path ------------------------------------------------------------------------
    def upload(self, scans: np.ndarray, current_time: np.ndarray, first_scan: int = 0, wait: bool = True):
        """wait=False: ll_fe_upload_async -- the arrays are kept referenced on the object until the next upload / sync and
        should be page-locked (e.g. views of torch pinned tensors) for the copy to overlap other work."""
        assert scans.dtype == np.float32 and scans.flags["C_CONTIGUOUS"] if not wait else True
        scans = np.ascontiguousarray(scans, np.float32)
        assert scans.ndim == 3 and scans.shape[2] == 4
        ct = np.ascontiguousarray(current_time, np.float64)
        fn = self.L.ll_fe_upload if wait else self.L.ll_fe_upload_async
        check(fn(self.h, first_scan, scans.shape[0], ptr(scans), scans.shape[1], ptr(ct)), "ll_fe_upload")
        self._pending_upload = None if wait else (scans, ct)
        for i in range(scans.shape[0]):
            self._n[first_scan + i] = scans.shape[1]

    def extract_batch(self, n_scans: int):
        check(self.L.ll_fe_extract_batch(self.h, n_scans), "ll_fe_extract_batch")

    def resolve(self) -> int:
        return check(self.L.ll_fe_resolve(self.h), "ll_fe_resolve")

    def select_batch(self, n_scans: int, piece: int = -1, minimum_blur: float = 0.0, maximum_blur: float = 1.0):
        check(self.L.ll_fe_select_batch(self.h, n_scans, piece, minimum_blur, maximum_blur), "ll_fe_select_batch")

    def counts(self, n_scans: int):
        nc, ns, nf = (np.zeros(n_scans, np.int32) for _ in range(3))
        na = np.zeros(1, np.int32)
        check(self.L.ll_fe_counts(self.h, n_scans, ptr(nc), ptr(ns), ptr(nf), ptr(na)), "ll_fe_counts")
        return nc, ns, nf, int(na[0])

    def sync(self):
        check(self.L.ll_fe_sync(self.h), "ll_fe_sync")


class VoxelGrid:
    """pcl::VoxelGrid<pcl::PointXYZI> as the reference uses it (setLeafSize / setInputCloud / filter:
    laser_feature_extractor.hpp:192-193,372-381; laser_mapping.hpp:742-743,1367-1373,1434-1437,533-537),
    PCL 1.9 semantics with a deterministic in-leaf order (include/loam_livox_hip.h)."""

    def __init__(self, max_points: int = 100000, max_clouds: int = 1, device: int = 0):
        self.L = capi.load()
        self.h = C.c_void_p()
        self.max_points, self.max_clouds = max_points, max_clouds
        check(self.L.ll_voxel_create(device, max_clouds, max_points, C.byref(self.h)), "ll_voxel_create")
        self.leaf = np.array([0.4, 0.4, 0.4], np.float32)
        self._cloud = None
        self.status = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.ll_voxel_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setLeafSize(self, lx: float, ly: float, lz: float):
        self.leaf = np.array([lx, ly, lz], np.float32)

    def setInputCloud(self, cloud: np.ndarray):
        self._cloud = capi.as_f32(cloud, 4)

    def filter(self) -> np.ndarray:
        """Returns the filtered cloud (n_out x 4); self.status: 0 filtered, 1 leaf too small (copy of the input), 2 empty."""
        out, n_out, st = self.filter_batch(self._cloud[None], np.array([self._cloud.shape[0]], np.int32))
        self.status = int(st[0])
        return out[0, :n_out[0]].copy()

    def filter_batch(self, clouds: np.ndarray, n_points: np.ndarray):
        """clouds [B][stride][4], n_points [B] -> (out [B][stride][4], n_out [B], status [B])."""
        clouds = np.ascontiguousarray(clouds, np.float32)
        B, stride = clouds.shape[0], clouds.shape[1]
        n_points = np.ascontiguousarray(n_points, np.int32)
        if stride == 0:
            return np.zeros_like(clouds), np.zeros(B, np.int32), np.full(B, 2, np.int32)
        out = np.zeros_like(clouds)
        n_out = np.zeros(B, np.int32)
        st = np.zeros(B, np.int32)
        check(self.L.ll_voxel_filter(self.h, B, ptr(clouds), ptr(n_points), stride, ptr(self.leaf), ptr(out), ptr(n_out), ptr(st)),
              "ll_voxel_filter")
        return out, n_out, st

    def counts(self, n_clouds: int):
        n_out, st = np.zeros(n_clouds, np.int32), np.zeros(n_clouds, np.int32)
        check(self.L.ll_voxel_counts(self.h, n_clouds, ptr(n_out), ptr(st)), "ll_voxel_counts")
        return n_out, st


class History_buffer:
    """m_laser_cloud_{corner,surface}_history + update_buff_for_matching (history mode), laser_mapping.hpp:1417-1478,
    517-546, resident on the device (include/loam_livox_hip.h, ll_history_*)."""

    def __init__(self, maximum_history_size: int = 100, max_points_per_frame: int = 24000, line_res: float = 0.1,
                 plane_res: float = 0.4, device: int = 0):
        self.L = capi.load()
        self.h = C.c_void_p()
        self.device = device
        check(self.L.ll_history_create(device, maximum_history_size, max_points_per_frame, line_res, plane_res, C.byref(self.h)),
              "ll_history_create")

    def close(self):
        if getattr(self, "h", None):
            self.L.ll_history_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self.L.ll_history_size(self.h))

    def add(self, corner, surf, pose, history_add_t_step: float = 0.0, history_add_angle_step: float = 0.0) -> bool:
        corner, surf = capi.as_f32(corner, 4), capi.as_f32(surf, 4)
        pose = np.ascontiguousarray(pose, np.float64)
        added = C.c_int32(0)
        check(self.L.ll_history_add(self.h, ptr(corner), corner.shape[0], ptr(surf), surf.shape[0], ptr(pose), history_add_t_step,
                                    history_add_angle_step, C.byref(added)), "ll_history_add")
        return bool(added.value)

    def add_fe(self, fe: "Livox_laser", scan: int, pose, history_add_t_step: float = 0.0, history_add_angle_step: float = 0.0) -> bool:
        pose = np.ascontiguousarray(pose, np.float64)
        added = C.c_int32(0)
        check(self.L.ll_history_add_fe(self.h, fe.h, scan, ptr(pose), history_add_t_step, history_add_angle_step, C.byref(added)),
              "ll_history_add_fe")
        return bool(added.value)

    def add_voxel(self, vox_corner: "VoxelGrid", vox_surf: "VoxelGrid", cloud: int, pose, history_add_t_step: float = 0.0,
                  history_add_angle_step: float = 0.0) -> bool:
        pose = np.ascontiguousarray(pose, np.float64)
        added = C.c_int32(0)
        check(self.L.ll_history_add_voxel(self.h, vox_corner.h, vox_surf.h, cloud, ptr(pose), history_add_t_step,
                                          history_add_angle_step, C.byref(added)), "ll_history_add_voxel")
        return bool(added.value)

    def refresh(self, map_buffer: "Map_buffer"):
        nc, ns = C.c_int64(0), C.c_int64(0)
        check(self.L.ll_history_refresh(self.h, map_buffer.h, C.byref(nc), C.byref(ns)), "ll_history_refresh")
        return nc.value, ns.value

    def enable_cell_map(self, max_points: int = 1 << 20, cell_resolution: float = 1.0, threshold_cell_revisit: int = 5000) -> None:
        """m_pt_cell_map_corners / m_pt_cell_map_planes (laser_mapping.hpp:274-275, 617-624): fed by every add*()."""
        check(self.L.ll_history_enable_cell_map(self.h, max_points, cell_resolution, threshold_cell_revisit), "ll_history_enable_cell_map")
        self._cell_resolution = cell_resolution

    def set_cell_map_async(self, enable: bool = True) -> None:
        """feed the cell maps through the handle's service thread, beside the caller (matching mode 0: nothing reads them per frame)"""
        check(self.L.ll_history_set_cell_map_async(self.h, int(bool(enable))), "ll_history_set_cell_map_async")

    def sync_cell_maps(self) -> None:
        check(self.L.ll_history_sync_cell_maps(self.h), "ll_history_sync_cell_maps")

    def cell_map(self, kind: int) -> "Cell_map":
        h = self.L.ll_history_cell_map(self.h, kind)
        if not h:  # not enabled, or the service thread failed: the library says which
            msg = self.L.ll_last_error()
            raise capi.LoamLivoxError(f"ll_history_cell_map: {msg.decode() if msg else 'error'}")
        return Cell_map(resolution=self._cell_resolution, _borrowed=h)

    def refresh_cells(self, map_buffer: "Map_buffer", pose, maximum_search_range_corner: float = 100.0,
                      maximum_search_range_surface: float = 100.0, maximum_in_fov_angle: float = 30.0, down_sample_replace: int = 1):
        """update_buff_for_matching with m_matching_mode == 1 (laser_mapping.hpp:471-546)."""
        pose = np.ascontiguousarray(pose, np.float64)
        nc, ns = C.c_int64(0), C.c_int64(0)
        check(self.L.ll_history_refresh_cells(self.h, map_buffer.h, ptr(pose), maximum_search_range_corner, maximum_search_range_surface,
                                              maximum_in_fov_angle, int(down_sample_replace), C.byref(nc), C.byref(ns)),
              "ll_history_refresh_cells")
        return nc.value, ns.value

    def set_gate_pose(self, pose) -> None:
        """the node's pose before the registration whose result the next add() receives (laser_mapping.hpp:1439-1451)"""
        p = np.ascontiguousarray(pose, np.float64)
        check(self.L.ll_history_set_gate_pose(self.h, ptr(p)), "ll_history_set_gate_pose")

    def map_cloud(self, kind: int) -> np.ndarray:
        n = self.L.ll_history_map_cloud(self.h, kind, None, 0)
        out = np.zeros((max(n, 0), 4), np.float32)
        if n > 0:
            check(min(0, self.L.ll_history_map_cloud(self.h, kind, ptr(out), n)), "ll_history_map_cloud")
        return out


    def map_cloud_device(self, kind: int):
        """The same cloud as a torch tensor (n, 4) float32 ON THE DEVICE, copied device-to-device out of the handle's
        buffer (which the next refresh overwrites): the input of the multi-GPU sub-map gather, no host hop."""
        import torch
        p, n = C.c_void_p(), C.c_int64(0)
        check(self.L.ll_history_map_cloud_device(self.h, kind, C.byref(p), C.byref(n)), "ll_history_map_cloud_device")
        if n.value == 0:
            return torch.zeros((0, 4), dtype=torch.float32, device=f"cuda:{self.device}")
        return torch.as_tensor(_DeviceView(p.value, n.value), device=f"cuda:{self.device}").clone()


class _DeviceView:
    """(n, 4) float32 at a raw device address, for torch.as_tensor (zero copy)"""

    def __init__(self, address: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n), 4), "typestr": "<f4", "data": (int(address), False), "version": 2}


class _DeviceView64:
    """(n,) int64 at a raw device address"""

    def __init__(self, address: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(address), False), "version": 2}


class Cell_map:
    """Points_cloud_map<float> (cell_map_keyframe.hpp:477-790) as used by the "cube" matching mode, resident on the
    device (include/loam_livox_hip.h, ll_cellmap_*): append_cloud, find_cells_in_radius + if_pt_in_fov + per-cell
    VoxelGrid (laser_mapping.hpp:475-513)."""

    def __init__(self, max_points: int = 1 << 20, resolution: float = 1.0, minimum_revisit_threshold: int = 2**31 - 1, device: int = 0,
                 _borrowed=None):
        self.L = capi.load()
        self.owned = _borrowed is None
        self.resolution = resolution
        self.max_points = int(max_points)
        if self.owned:
            self.h = C.c_void_p()
            check(self.L.ll_cellmap_create(device, max_points, resolution, minimum_revisit_threshold, C.byref(self.h)), "ll_cellmap_create")
        else:
            self.h = C.c_void_p(_borrowed)

    def close(self):
        if getattr(self, "h", None) and self.owned:
            self.L.ll_cellmap_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_view(self, device: int = 0):
        """(points (n, 4) float32, cell keys (n,) int64) as torch DEVICE tensors, copied device-to-device out of the handle's arrays
        (which the next append rewrites): the input of multigpu.gather_cell_maps, no host hop"""
        import torch
        p, k, n, nc = C.c_void_p(), C.c_void_p(), C.c_int64(0), C.c_int64(0)
        check(self.L.ll_cellmap_device_view(self.h, C.byref(p), C.byref(k), C.byref(n), C.byref(nc)), "ll_cellmap_device_view")
        if n.value == 0:
            return torch.zeros((0, 4), dtype=torch.float32, device=f"cuda:{device}"), torch.zeros(0, dtype=torch.int64, device=f"cuda:{device}")
        pts = torch.as_tensor(_DeviceView(p.value, n.value), device=f"cuda:{device}").clone()
        keys = torch.as_tensor(_DeviceView64(k.value, n.value), device=f"cuda:{device}").clone()
        return pts, keys

    def reserve(self, max_points: int) -> None:
        """room for max_points points (no-op when not larger); content, revisit stamps and frame counter are kept"""
        check(self.L.ll_cellmap_reserve(self.h, int(max_points)), "ll_cellmap_reserve")
        self.max_points = max(self.max_points, int(max_points))

    def append_cloud(self, cloud) -> None:
        cloud = capi.as_f32(cloud, 4)
        check(self.L.ll_cellmap_append(self.h, ptr(cloud), cloud.shape[0]), "ll_cellmap_append")

    def append_cloud_touched(self, cloud, min_points: int = 3) -> np.ndarray:
        """append_cloud( pts, &cell_vec ) (cell_map_keyframe.hpp:619-672): the append, and the indices [n,3] of the cells that received
        at least min_points of this cloud's points (every touched cell on an empty map)."""
        cloud = capi.as_f32(cloud, 4)
        cap = max(1, cloud.shape[0])
        ijk = np.zeros((cap, 3), np.int32)
        n = C.c_int64(0)
        check(self.L.ll_cellmap_append_touched(self.h, ptr(cloud), cloud.shape[0], int(min_points), ptr(ijk), cap, C.byref(n)),
              "ll_cellmap_append_touched")
        return ijk[:n.value].copy()

    def stats(self):
        """(cells, points, m_current_frame_idx)"""
        nc, npts, fr = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        check(self.L.ll_cellmap_stats(self.h, C.byref(nc), C.byref(npts), C.byref(fr)), "ll_cellmap_stats")
        return nc.value, npts.value, fr.value

    def query_filter(self, pose, radius: float, maximum_in_fov_angle: float, leaf: float, down_sample_replace: int = 1):
        """Returns (concatenated per-cell filtered cloud [n,4], number of selected cells)."""
        pose = np.ascontiguousarray(pose, np.float64)
        nsel, nout = C.c_int64(0), C.c_int64(0)
        check(self.L.ll_cellmap_query_filter(self.h, ptr(pose), radius, maximum_in_fov_angle, leaf, int(down_sample_replace), C.byref(nsel),
                                             C.byref(nout)), "ll_cellmap_query_filter")
        out = np.zeros((nout.value, 4), np.float32)
        if nout.value > 0:
            check(min(0, self.L.ll_cellmap_result(self.h, ptr(out), nout.value)), "ll_cellmap_result")
        return out, nsel.value

    def features(self):
        """determine_feature for every cell (cell_map_keyframe.hpp:436-473), in the cell order of dump():
        dict(type [c] (0 sphere, 1 line, 2 plane), vector [c,3], mean [c,3], cov [c,6], eigen_val [c,3])."""
        nc = self.stats()[0]
        out = dict(type=np.zeros(nc, np.int32), vector=np.zeros((nc, 3), np.float32), mean=np.zeros((nc, 3), np.float32),
                   cov=np.zeros((nc, 6), np.float32), eigen_val=np.zeros((nc, 3), np.float32))
        if nc > 0:
            check(self.L.ll_cellmap_features(self.h, ptr(out["type"]), ptr(out["vector"]), ptr(out["mean"]), ptr(out["cov"]),
                                             ptr(out["eigen_val"]), nc), "ll_cellmap_features")
        return out

    def keyframe_images(self, roi_ratio: float = 0.9):
        """Maps_keyframe::analyze over the cells of this map (cell_map_keyframe.hpp:1385-1493): dict(images [4,60,60] =
        line, plane, line_roi, plane_roi; ratio_nonzero [4]; eigen_R [2,3,3]; n_vectors [4]; centre [3]; roi_range)."""
        img = np.zeros((4, 60, 60), np.float32)
        ratio, R, nv, cr = np.zeros(4, np.float32), np.zeros((2, 3, 3), np.float32), np.zeros(4, np.int32), np.zeros(4, np.float32)
        check(self.L.ll_cellmap_keyframe_images(self.h, roi_ratio, ptr(img), ptr(ratio), ptr(R), ptr(nv), ptr(cr)), "ll_cellmap_keyframe_images")
        return dict(images=img, ratio_nonzero=ratio, eigen_R=R, n_vectors=nv, centre=cr[:3].copy(), roi_range=float(cr[3]))

    def dump(self):
        """(xyz [n,3] in (cell, insertion) order, cell indices [c,3], first point of each cell [c+1], last-update frame [c])"""
        nc, npts, _ = self.stats()
        xyzi = np.zeros((max(npts, 1), 4), np.float32)
        ijk = np.zeros((max(nc, 1), 3), np.int32)
        start = np.zeros(nc + 1, np.int32)
        last = np.zeros(max(nc, 1), np.int32)
        check(self.L.ll_cellmap_dump(self.h, ptr(xyzi), xyzi.shape[0], ptr(ijk), ptr(start), ptr(last), ijk.shape[0]), "ll_cellmap_dump")
        return xyzi[:npts, :3].copy(), ijk[:nc].copy(), start, last[:nc].copy()


def keyframe_similarity(img_a, img_b, device: int = 0) -> float:
    """Maps_keyframe::max_similiarity_of_two_image (cell_map_keyframe.hpp:1155-1224) of two 60 x 60 direction images."""
    a, b = np.ascontiguousarray(img_a, np.float32), np.ascontiguousarray(img_b, np.float32)
    if a.shape != (60, 60) or b.shape != (60, 60):
        raise ValueError("direction images are 60 x 60")
    out = C.c_float(0)
    check(capi.load().ll_keyframe_similarity(device, ptr(a), ptr(b), C.byref(out)), "ll_keyframe_similarity")
    return float(out.value)


class Map_buffer:
    """m_laser_cloud_{corner,surf}_from_map + their kd-trees (laser_mapping.hpp:539-546) as device grids."""

    CORNER, SURF = 0, 1

    def __init__(self, device: int = 0):
        self.L = capi.load()
        self.h = C.c_void_p()
        check(self.L.ll_map_create(device, C.byref(self.h)), "ll_map_create")

    def close(self):
        if getattr(self, "h", None):
            self.L.ll_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setInputCloud(self, kind: int, cloud: np.ndarray, cell_size: float = 0.0):
        cloud = np.ascontiguousarray(cloud, np.float32)
        assert cloud.ndim == 2 and cloud.shape[1] >= 3
        check(self.L.ll_map_upload(self.h, kind, ptr(cloud), cloud.shape[1], cloud.shape[0], cell_size), "ll_map_upload")

    def size(self, kind: int) -> int:
        return int(self.L.ll_map_size(self.h, kind))

    def cells(self, kind: int) -> int:
        return int(self.L.ll_map_cells(self.h, kind))

    def to_f16(self, kind: int):
        """fp16-point records for this kind (BASELINE config C5); the registrar cannot use the map afterwards"""
        check(self.L.ll_map_to_f16(self.h, kind), "ll_map_to_f16")

    def dequantized(self, kind: int) -> np.ndarray:
        n = self.size(kind)
        out = np.zeros((n, 3), np.float32)
        check(self.L.ll_map_dequantized(self.h, kind, ptr(out), n), "ll_map_dequantized")
        return out

    def nearestKSearch(self, kind: int, queries: np.ndarray, max_sq_dis: float):
        q = capi.as_f32(queries, 3)
        idx = np.zeros((q.shape[0], 5), np.int32)
        d2 = np.zeros((q.shape[0], 5), np.float32)
        check(self.L.ll_map_knn5(self.h, kind, ptr(q), q.shape[0], max_sq_dis, ptr(idx), ptr(d2)), "ll_map_knn5")
        return idx, d2


    def nearestKSearch_device(self, kind: int, queries, max_sq_dis: float, idx_out, d2_out) -> float:
        """The same search on torch DEVICE tensors (queries (n, 3) float32, idx_out (n, 5) int32, d2_out (n, 5) float32, contiguous):
        nothing crosses PCIe.  Returns the search kernel's duration in ms (HIP events on the map's stream)."""
        n = int(queries.shape[0])
        assert queries.is_contiguous() and idx_out.is_contiguous() and d2_out.is_contiguous() and queries.shape[1] == 3
        ms = C.c_float(0)
        check(self.L.ll_map_knn5_device(self.h, kind, C.c_void_p(queries.data_ptr()), n, max_sq_dis, C.c_void_p(idx_out.data_ptr()),
                                        C.c_void_p(d2_out.data_ptr()), C.byref(ms)), "ll_map_knn5_device")
        return float(ms.value)


class Point_cloud_registration:
    """Device-backed Point_cloud_registration.  Configuration fields keep the reference names
    (point_cloud_registration.hpp:45-103); poses are the m_q_w_*/m_t_w_* pairs packed as float64[7]."""

    def __init__(self, max_scans: int = 1, max_features: int = 24000, device: int = 0):
        self.L = capi.load()
        self.h = C.c_void_p()
        check(self.L.ll_reg_create(device, max_scans, max_features, C.byref(self.h)), "ll_reg_create")
        self.params = capi.reg_default_params()
        self.max_scans = max_scans
        ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
        self.m_pose_w_last = ident.copy()   # m_q_w_last, m_t_w_last
        self.m_pose_w_curr = ident.copy()   # m_q_w_curr, m_t_w_curr
        self.m_para_buffer_incremental = ident.copy()
        self.report = RegReport()
        self.m_inlier_threshold = 0.0

    def close(self):
        if getattr(self, "h", None):
            self.L.ll_reg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def find_out_incremental_transfrom(self, map_buffer: Map_buffer, laserCloudCornerStack: np.ndarray,
                                       laserCloudSurfStack: np.ndarray) -> int:
        c = capi.as_f32(laserCloudCornerStack, 4)
        s = capi.as_f32(laserCloudSurfStack, 4)
        pl = np.ascontiguousarray(self.m_pose_w_last, np.float64)
        pc = np.ascontiguousarray(self.m_pose_w_curr, np.float64).copy()
        pi = np.ascontiguousarray(self.m_para_buffer_incremental, np.float64).copy()
        rep = RegReport()
        ret = check(self.L.ll_reg_solve(self.h, map_buffer.h, ptr(c), c.shape[0], ptr(s), s.shape[0], C.byref(self.params),
                                        ptr(pl), ptr(pc), ptr(pi), C.byref(rep)), "ll_reg_solve")
        self.m_pose_w_curr, self.m_para_buffer_incremental, self.report = pc, pi, rep
        self.m_inlier_threshold = rep.inlier_threshold
        return ret

    def solve_batch(self, map_buffer: Map_buffer, corners: list, surfs: list, poses_last: np.ndarray, poses_curr: np.ndarray):
        n = len(corners)
        nc = np.array([len(c) for c in corners], np.int32)
        ns = np.array([len(s) for s in surfs], np.int32)
        sc, ss = max(1, int(nc.max())), max(1, int(ns.max()))
        cbuf = np.zeros((n, sc, 4), np.float32)
        sbuf = np.zeros((n, ss, 4), np.float32)
        for i in range(n):
            cbuf[i, :nc[i]] = corners[i]
            sbuf[i, :ns[i]] = surfs[i]
        pl = np.ascontiguousarray(poses_last, np.float64).reshape(n, 7)
        pc = np.ascontiguousarray(poses_curr, np.float64).reshape(n, 7).copy()
        pi = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float64), (n, 1))
        reps = (RegReport * n)()
        res = np.zeros(n, np.int32)
        check(self.L.ll_reg_solve_batch(self.h, map_buffer.h, n, ptr(cbuf), ptr(nc), sc, ptr(sbuf), ptr(ns), ss,
                                        C.byref(self.params), ptr(pl), ptr(pc), ptr(pi), reps, ptr(res)), "ll_reg_solve_batch")
        return res, pc, pi, list(reps)

    def upload_features(self, corners: list, surfs: list):
        """Host feature clouds -> the registrar's HBM buffers (the upload half of solve_batch)."""
        n = len(corners)
        nc = np.array([len(c) for c in corners], np.int32)
        ns = np.array([len(s) for s in surfs], np.int32)
        sc, ss = max(1, int(nc.max())), max(1, int(ns.max()))
        cbuf = np.zeros((n, sc, 4), np.float32)
        sbuf = np.zeros((n, ss, 4), np.float32)
        for i in range(n):
            cbuf[i, :nc[i]] = corners[i]
            sbuf[i, :ns[i]] = surfs[i]
        check(self.L.ll_reg_upload_features(self.h, n, ptr(cbuf), ptr(nc), sc, ptr(sbuf), ptr(ns), ss), "ll_reg_upload_features")
        return n

    def enqueue_uploaded(self, map_buffer: Map_buffer, n_scans: int, poses_last, poses_curr):
        pl = np.ascontiguousarray(poses_last, np.float64).reshape(n_scans, 7)
        pc = np.ascontiguousarray(poses_curr, np.float64).reshape(n_scans, 7)
        check(self.L.ll_reg_enqueue_uploaded(self.h, map_buffer.h, n_scans, C.byref(self.params), ptr(pl), ptr(pc), None),
              "ll_reg_enqueue_uploaded")

    def enqueue_fe(self, map_buffer: Map_buffer, fe: Livox_laser, n_scans: int, poses_last, poses_curr):
        pl = np.ascontiguousarray(poses_last, np.float64).reshape(n_scans, 7)
        pc = np.ascontiguousarray(poses_curr, np.float64).reshape(n_scans, 7)
        check(self.L.ll_reg_enqueue_fe(self.h, map_buffer.h, fe.h, n_scans, C.byref(self.params), ptr(pl), ptr(pc), None),
              "ll_reg_enqueue_fe")

    def enqueue_fe_merged(self, map_buffer: Map_buffer, fe: Livox_laser, n_scans: int, heads: int, poses_last, poses_curr):
        """Mid-100 (laser_feature_extractor.hpp:348-358): registrar scan b = the selected features of extractor slots
        b * heads ... b * heads + heads - 1, concatenated on the device (corner clouds together, surface clouds together)."""
        pl = np.ascontiguousarray(poses_last, np.float64).reshape(n_scans, 7)
        pc = np.ascontiguousarray(poses_curr, np.float64).reshape(n_scans, 7)
        check(self.L.ll_reg_enqueue_fe_merged(self.h, map_buffer.h, fe.h, n_scans, int(heads), C.byref(self.params), ptr(pl), ptr(pc), None),
              "ll_reg_enqueue_fe_merged")

    def enqueue_fe_downsampled(self, map_buffer: Map_buffer, fe: Livox_laser, vox_corner: "VoxelGrid", vox_surf: "VoxelGrid",
                               line_res: float, plane_res: float, n_scans: int, poses_last, poses_curr):
        """m_if_input_downsample_mode (laser_mapping.hpp:1367-1373): voxel-filter the selected features on the device, then
        register them."""
        pl = np.ascontiguousarray(poses_last, np.float64).reshape(n_scans, 7)
        pc = np.ascontiguousarray(poses_curr, np.float64).reshape(n_scans, 7)
        check(self.L.ll_reg_enqueue_fe_downsampled(self.h, map_buffer.h, fe.h, vox_corner.h, vox_surf.h, line_res, plane_res, n_scans,
                                                   C.byref(self.params), ptr(pl), ptr(pc), None), "ll_reg_enqueue_fe_downsampled")

    def collect(self, n_scans: int):
        pc = np.zeros((n_scans, 7), np.float64)
        pi = np.zeros((n_scans, 7), np.float64)
        reps = (RegReport * n_scans)()
        res = np.zeros(n_scans, np.int32)
        check(self.L.ll_reg_collect(self.h, n_scans, ptr(pc), ptr(pi), reps, ptr(res)), "ll_reg_collect")
        return res, pc, pi, list(reps)

    def solve_batch_fe(self, map_buffer: Map_buffer, fe: Livox_laser, n_scans: int, poses_last, poses_curr):
        self.enqueue_fe(map_buffer, fe, n_scans, poses_last, poses_curr)
        return self.collect(n_scans)

    def set_debug(self, enable: bool = True, force_general_solver: bool = False, no_knn_reuse: bool = False,
                  reuse_from_iter1: bool = False, no_solver_groups: bool = False,
                  test_group_abort: bool = False, no_knn_coop: bool = False, no_knn_tile: bool = False,
                  knn_tile_with_reuse: bool = False, knn_tile_small_batches: bool = False, no_line_cache: bool = False,
                  no_small_solver: bool = False, small_solver_waves: int = 0, no_solve_order: bool = False):
        flags = (int(bool(enable)) | (2 if force_general_solver else 0) | (4 if no_knn_reuse else 0) | (8 if reuse_from_iter1 else 0)
                 | (32 if no_solver_groups else 0)
                 | (128 if test_group_abort else 0) | (256 if no_knn_coop else 0) | (512 if no_knn_tile else 0)
                 | (1024 if knn_tile_with_reuse else 0) | (2048 if knn_tile_small_batches else 0) | (4096 if no_line_cache else 0)
                 | (32768 if no_small_solver else 0) | {0: 0, 1: 65536, 4: 131072, 2: 65536 | 131072}[int(small_solver_waves)] | (262144 if no_solve_order else 0))
        self.set_debug_flags(flags)

    def set_debug_flags(self, flags: int):
        """ll_reg_set_debug with the raw flag word (kept in self.debug_flags, so a caller can add a bit and put the word back)"""
        check(self.L.ll_reg_set_debug(self.h, int(flags)), "ll_reg_set_debug")
        self.debug_flags = int(flags)

    def set_debug_knn_iteration(self, icp_iteration: int):
        """the ICP iteration whose neighbour lists debug_knn() returns (default 0)"""
        check(self.L.ll_reg_set_debug_knn_iteration(self.h, int(icp_iteration)), "ll_reg_set_debug_knn_iteration")

    def debug_knn(self, scan: int, n_corner: int, n_surf: int):
        ci, cd = np.zeros((n_corner, 5), np.int32), np.zeros((n_corner, 5), np.float32)
        si, sd = np.zeros((n_surf, 5), np.int32), np.zeros((n_surf, 5), np.float32)
        check(self.L.ll_reg_debug_knn(self.h, scan, ptr(ci), ptr(cd), ptr(si), ptr(sd)), "ll_reg_debug_knn")
        return ci, cd, si, sd

    def set_profiling(self, enable: bool = True):
        check(self.L.ll_reg_set_profiling(self.h, int(enable)), "ll_reg_set_profiling")

    def debug_worklists(self, n_scans=1):
        out = np.zeros(4, np.int64)
        check(self.L.ll_reg_debug_worklists(self.h, int(n_scans), ptr(out)), "ll_reg_debug_worklists")
        return int(out[0] + out[2]), int(out[1] + out[3])

    def debug_worklists_by_kind(self, n_scans=1):
        """((corner searched, corner re-sorted), (surface searched, surface re-sorted)) of the last ICP iteration"""
        out = np.zeros(4, np.int64)
        check(self.L.ll_reg_debug_worklists(self.h, int(n_scans), ptr(out)), "ll_reg_debug_worklists")
        return (int(out[0]), int(out[1])), (int(out[2]), int(out[3]))

    def debug_cycles(self, scan: int = 0):
        out = np.zeros(16, np.int64)
        check(self.L.ll_reg_debug_cycles(self.h, scan, ptr(out)), "ll_reg_debug_cycles")
        return out

    def kernel_times(self):
        ms = np.zeros(3, np.float32)
        n = np.zeros(3, np.int32)
        check(self.L.ll_reg_kernel_times(self.h, ptr(ms), ptr(n)), "ll_reg_kernel_times")
        return ms, n

    def append_to_submap_device(self, fe: "Livox_laser", n_scans: int, kind: int, accept: np.ndarray, poses: np.ndarray, out, n_used: int) -> int:
        """pointcloudAssociateToMap over an extractor's resident batch, device to device: the selected `kind` features
        of every accepted scan, moved to the map frame with its pose, are appended to the torch DEVICE tensor `out`
        ((capacity, 4) float32, contiguous) from row n_used on.  Returns the new row count."""
        acc = np.ascontiguousarray(accept, np.int32)
        ps = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        assert out.is_contiguous() and out.shape[1] == 4 and acc.shape[0] >= n_scans and ps.shape[0] >= n_scans
        n = C.c_int64(int(n_used))
        check(self.L.ll_cloud_transform_fe_device(self.h, fe.h, int(n_scans), int(kind), ptr(acc), ptr(ps), C.c_void_p(out.data_ptr()),
                                                  int(out.shape[0]), C.byref(n)), "ll_cloud_transform_fe_device")
        return int(n.value)

    def pointcloudAssociateToMap(self, pc_in: np.ndarray, pose: np.ndarray | None = None) -> np.ndarray:
        pc_in = capi.as_f32(pc_in, 4)
        out = np.empty_like(pc_in)
        p = np.ascontiguousarray(self.m_pose_w_curr if pose is None else pose, np.float64)
        check(self.L.ll_cloud_transform(self.h, ptr(pc_in), ptr(out), pc_in.shape[0], ptr(p)), "ll_cloud_transform")
        return out
