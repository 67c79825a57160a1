"""CPU ORACLE (test infrastructure only, see ll_oracle.h) for the host logic around the hot path: a restatement of
Laser_mapping::process_new_scan / update_buff_for_matching of hku-mars/loam_livox (source/laser_mapping.hpp:1311-1520,
460-566; m_matching_mode 0 = history, 1 = cell maps via orc_cellmap), composed from the oracle's extraction,
VoxelGrid, k-d tree and registration.  PARITY UNPINNED.  Synchronous refresh after every accepted frame (the node's
service-thread timing is not reproducible)."""
import numpy as np

from . import orc
from .orc_cellmap import CellMap


def _angular_distance(a, b):
    # Eigen angularDistance: 2 atan2(|vec(a b*)|, |w(a b*)|)
    ax, ay, az, aw = a
    bx, by, bz, bw = -b[0], -b[1], -b[2], b[3]
    w = aw * bw - ax * bx - ay * by - az * bz
    x = aw * bx + ax * bw + ay * bz - az * by
    y = aw * by + ay * bw + az * bx - ax * bz
    z = aw * bz + az * bw + ax * by - ay * bx
    return 2.0 * np.arctan2(np.sqrt(x * x + y * y + z * z), abs(w))


class History:
    """m_laser_cloud_{corner,surface}_history + the match buffer (LM:1417-1478, 517-546)."""

    def __init__(self, maximum_history_size=100, line_res=0.1, plane_res=0.4):
        self.max_hist, self.res = maximum_history_size, (line_res, plane_res)
        self.frames = ([], [])
        self.last_q, self.last_t = np.array([0, 0, 0, 1.0]), np.zeros(3)
        self.cells = None

    def enable_cell_map(self, cell_resolution=1.0, threshold_cell_revisit=5000):      # LM:620-624
        self.cells = (CellMap(cell_resolution, threshold_cell_revisit), CellMap(cell_resolution, threshold_cell_revisit))

    def add(self, corner, surf, pose, t_step=0.0, angle_step=0.0, gate_pose=None):
        # LM:1439-1451 read Laser_mapping::m_q_w_curr / m_t_w_curr, which there is still the pose BEFORE this registration
        # (copied back at :1496-1500); the clouds move with the registered pose.  gate_pose = that earlier pose.
        gp = np.asarray(pose if gate_pose is None else gate_pose, np.float64)
        r_diff = _angular_distance(gp[:4], self.last_q) * 57.3            # LM:1439
        t_diff = np.linalg.norm(gp[4:] - self.last_t)                    # LM:1440
        push = len(self.frames[0]) < self.max_hist or t_diff > t_step or r_diff > angle_step * 57.3   # LM:1446-1448
        if not push and self.cells is None:
            return False
        if push:
            self.last_q, self.last_t = np.array(gp[:4], np.float64), np.array(gp[4:], np.float64)             # LM:1450-1451
        for kind, cloud in enumerate((corner, surf)):
            w = orc.cloud_transform(pose, cloud) if len(cloud) else np.zeros((0, 4), np.float32)  # LM:1421-1431
            w = orc.voxel_grid(w, self.res[kind])[1] if len(w) else w                               # LM:1434-1437
            if push:
                self.frames[kind].append(w)
                if len(self.frames[kind]) > self.max_hist:                                           # LM:1468-1478
                    self.frames[kind].pop(0)
            if self.cells is not None:                                                               # LM:1492-1493
                self.cells[kind].append(w)
        return push

    def refresh_cells(self, pose, ranges=(100.0, 100.0), maximum_in_fov_angle=30.0, down_sample_replace=1):
        out = []
        for kind in range(2):                                                                        # LM:471-513
            cat, _ = self.cells[kind].query_filter(pose, ranges[kind], maximum_in_fov_angle, self.res[kind], down_sample_replace)
            out.append(orc.voxel_grid(cat, self.res[kind])[1] if len(cat) else cat)                  # LM:533-537
        return out

    def refresh(self):
        out = []
        for kind in range(2):
            cat = np.concatenate(self.frames[kind], 0) if self.frames[kind] else np.zeros((0, 4), np.float32)  # LM:519-530
            out.append(orc.voxel_grid(cat, self.res[kind])[1] if len(cat) else cat)                          # LM:533-537
        return out


class LaserMapping:
    def __init__(self, maximum_history_size=100, line_res=0.1, plane_res=0.4, init_accumulate_frames=50, input_downsample_mode=1,
                 icp_max_iterations=20, ceres_max_iterations=100, max_allow_incre_R=4.0, max_allow_incre_T=2.0, max_allow_final_cost=100.0,
                 minimum_icp_R_diff=0.01, minimum_icp_T_diff=0.01, matching_mode=0, cell_resolution=1.0, threshold_cell_revisit=5000,
                 maximum_search_range_corner=100.0, maximum_search_range_surface=100.0, maximum_in_fov_angle=30.0, down_sample_replace=1,
                 maximum_residual_blocks=0, subsample_seed=1, history_add_t_step=0.0, history_add_angle_step=0.0):
        self.steps = (history_add_t_step, history_add_angle_step)
        self.hist = History(maximum_history_size, line_res, plane_res)
        self.mode = matching_mode
        self.cell_args = ((maximum_search_range_corner, maximum_search_range_surface), maximum_in_fov_angle, down_sample_replace)
        if matching_mode:
            self.hist.enable_cell_map(cell_resolution, threshold_cell_revisit)
        self.res = (line_res, plane_res)
        self.ds = input_downsample_mode
        self.prm = orc.RegParams.defaults(icp_iters=icp_max_iterations, ceres_iters=ceres_max_iterations, force_all=0)
        self.prm.para_max_angular_rate, self.prm.para_max_speed, self.prm.max_final_cost = max_allow_incre_R, max_allow_incre_T, max_allow_final_cost
        self.prm.mapping_init_accumulate_frames = init_accumulate_frames
        self.prm.minimum_icp_R_diff, self.prm.minimum_icp_T_diff = minimum_icp_R_diff, minimum_icp_T_diff
        if maximum_residual_blocks > 0:    # optimization/maximum_residual_blocks: sub-sampling on the reproducible stream (PCR:232-238)
            self.prm.maximum_allow_residual_block, self.prm.subsample_seed = maximum_residual_blocks, subsample_seed
        self.frame = 0
        self.pose = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
        self.maps = [np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32)]
        self.trees = [None, None]
        self.report = None

    def process_new_scan(self, xyzi, time_stamp=1.0):
        o = orc.fe_extract(xyzi, time_stamp)
        ci, si, _ = orc.fe_get_features(o, 0.0, 1.0)
        fc, fs = orc.feature_cloud(o, ci), orc.feature_cloud(o, si)
        if self.ds:                                                       # LM:1367-1373
            fc = orc.voxel_grid(fc, self.res[0])[1] if len(fc) else fc
            fs = orc.voxel_grid(fs, self.res[1])[1] if len(fs) else fs
        self.prm.current_frame_index = self.frame                          # LM:1276 precedes :1348
        self.frame += 1
        ret, pc, _, rep = orc.reg_solve(self.trees[0], self.trees[1], fc, fs, self.prm, self.pose, self.pose)
        self.report = rep
        if not ret:
            return 0
        self.hist.add(fc, fs, pc, self.steps[0], self.steps[1], gate_pose=self.pose)   # m_q_w_curr is updated after the add, LM:1496-1500
        self.pose = pc.copy()
        self.maps = self.hist.refresh_cells(self.pose, *self.cell_args) if self.mode else self.hist.refresh()
        self.trees = [orc.KdTree(m) if len(m) else None for m in self.maps]
        return 1
