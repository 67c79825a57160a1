"""Host-side mirror of Scene_alignment::find_tranfrom_of_two_mappings (hku-mars/loam_livox source/scene_alignment.hpp:
269-391) on top of the C ABI: the line / plane cells of two key frames (device cell maps) are registered against each
other by the same registrar the mapping node uses, coarse to fine.  Loop-closure front half only: the pose graph
(ceres_pose_graph_3d.hpp) and the map refinement that consume the result are out of scope.

Differences from the reference, by design:
  * extract_specify_points walks a std::set of cell pointers (address order); cells come in cell-index order here;
  * the registrar object is created per call, so the previous pair's m_q_w_incre does not leak into the next one;
  * nothing is written to disk (if_save)."""
from __future__ import annotations

import numpy as np

from .api import Cell_map, Map_buffer, Point_cloud_registration, VoxelGrid

E_FEATURE_SPHERE, E_FEATURE_LINE, E_FEATURE_PLANE = 0, 1, 2   # Feature_type, cell_map_keyframe.hpp:46-51


def keyframe_clouds(km: Cell_map):
    """extract_specify_points( e_feature_line ), ( e_feature_plane ) and get_center() of a key frame held as a cell map
    (cell_map_keyframe.hpp:1263-1301).  Returns (line xyzi, plane xyzi, centre float32[3])."""
    xyz, ijk, start, _ = km.dump()
    f = km.features()
    per_point = np.repeat(f["type"], np.diff(start))
    cloud = np.c_[xyz, np.zeros(len(xyz), np.float32)].astype(np.float32)
    box = np.float32(np.float64(np.float32(km.resolution)) * 0.5)
    half = np.float32(np.float64(box) * 0.5)
    ctrs = (ijk.astype(np.float32) * box + half).astype(np.float32)
    centre = np.add.accumulate(ctrs, axis=0, dtype=np.float32)[-1] * np.float32(1.0 / float(np.float32(len(ctrs)))) if len(ctrs) else np.zeros(3, np.float32)
    return cloud[per_point == E_FEATURE_LINE], cloud[per_point == E_FEATURE_PLANE], centre


class Scene_alignment:
    def __init__(self, line_res: float = 0.4, plane_res: float = 0.4, maximum_icp_iteration: int = 10, accepted_threshold: float = 0.2,
                 maximum_residual_block: int = 5000, max_points: int = 1 << 18, device: int = 0, subsample_seed: int = 1,
                 registrar_init: bool = True):
        # registrar_init: apply the registrar settings of Scene_alignment::init (SA:233-243: ICP_LINE = 0, m_max_final_cost 20000,
        # m_para_max_speed 1000, m_para_max_angular_rate 360 * 57.3, m_inliner_dis 0.2) as the loop detector does before its first
        # alignment (laser_mapping.hpp:896); False = a default-constructed Scene_alignment (class defaults of the registrar)
        self.registrar_init = registrar_init
        self.m_line_res, self.m_plane_res = np.float32(line_res), np.float32(plane_res)         # SA:27-28
        self.m_maximum_icp_iteration, self.m_accepted_threshold = maximum_icp_iteration, accepted_threshold   # SA:35-36
        self.m_para_scene_alignments_maximum_residual_block = maximum_residual_block                # SA:34
        self.device, self.max_points, self.subsample_seed = device, max_points, subsample_seed
        self.pose = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)   # m_pc_reg.m_q_w_curr / m_t_w_curr after the call
        self.reports = []

    def find_tranfrom_of_two_mappings(self, keyframe_a: Cell_map, keyframe_b: Cell_map) -> float:
        """Registers key frame b (as the scan) against key frame a (as the map); returns m_inlier_threshold (SA:389)."""
        src_line, src_plane, centre_a = keyframe_clouds(keyframe_a)
        tgt_line, tgt_plane, centre_b = keyframe_clouds(keyframe_b)
        reg = Point_cloud_registration(max_scans=1, max_features=max(1, len(tgt_line), len(tgt_plane)), device=self.device)
        mp = Map_buffer(device=self.device)
        vox = VoxelGrid(max(1, len(src_line), len(src_plane), len(tgt_line), len(tgt_plane)), 1, device=self.device)
        p = reg.params
        if self.registrar_init:                                           # Scene_alignment::init, SA:233-243 (the detector calls it, laser_mapping.hpp:896)
            p.icp_line = 0
            p.max_final_cost = 20000.0
            p.para_max_speed = 1000.0
            p.para_max_angular_rate = 360 * 57.3
            p.inliner_dis = 0.2
        p.current_frame_index = 10000000                                  # SA:296
        p.icp_max_iterations = self.m_maximum_icp_iteration               # SA:300
        p.ceres_max_iterations, p.ceres_prerun_times = 50, 2              # SA:301-302
        p.maximum_allow_residual_block = self.m_para_scene_alignments_maximum_residual_block   # SA:303
        p.subsample_seed = self.subsample_seed
        ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
        t0 = (centre_a - centre_b).astype(np.float64)                     # SA:307
        reg.m_pose_w_last = ident.copy()                                  # SA:297-299
        reg.m_pose_w_curr = np.r_[ident[:4], t0]                          # SA:309-310
        reg.m_para_buffer_incremental = np.r_[ident[:4], t0]
        self.reports = []

        def ds(cloud, leaf):
            if len(cloud) == 0:
                return cloud
            vox.setLeafSize(leaf, leaf, leaf)
            vox.setInputCloud(cloud)
            return vox.filter()

        for scale in (8, 4, 0):                                           # SA:313
            line_res, plane_res = np.float32(self.m_line_res * np.float32(scale)), np.float32(self.m_plane_res * np.float32(scale))
            if line_res < self.m_line_res:
                line_res = self.m_line_res
            if plane_res < self.m_plane_res:
                plane_res = self.m_plane_res
                p.icp_max_iterations = self.m_maximum_icp_iteration * 2   # SA:327
            sl, sp = ds(src_line, float(line_res)), ds(src_plane, float(plane_res))
            tl, tp = ds(tgt_line, float(line_res)), ds(tgt_plane, float(plane_res))
            if len(sl) and len(sp):                                       # PCR:595-602: otherwise "return 1" without solving
                mp.setInputCloud(Map_buffer.CORNER, sl)
                mp.setInputCloud(Map_buffer.SURF, sp)
                reg.find_out_incremental_transfrom(mp, tl, tp)
                self.reports.append(reg.report)
            if reg.m_inlier_threshold > self.m_accepted_threshold * 2:     # SA:350-351
                break
        self.pose = reg.m_pose_w_curr.copy()
        thr = float(reg.m_inlier_threshold)
        for h in (reg, mp, vox):
            h.close()
        return thr
