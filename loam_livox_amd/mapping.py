"""Host-side mirror of Laser_mapping::process_new_scan for both match modes (m_matching_mode == 0: history,
1: cell map), hku-mars/loam_livox source/laser_mapping.hpp:1311-1520, on top of the C ABI: extractor -> (VoxelGrid)
-> registrar -> history / cell maps -> match-buffer refresh, everything resident on one device.

This is the unit of BASELINE config C4 (one sequence per GPU, local map growth).  Differences from the node, by design:
  * the match buffer is refreshed synchronously after every accepted frame; the node refreshes it on a service thread
    and registers against whichever buffer is newest (laser_mapping.hpp:568-594, 1395-1403), which makes its output
    depend on thread timing;
  * no ROS or logging; the full-cloud cell map, the key frames and the front half of loop detection are optional
    (loop_closure_if_enable, keyframes.py); the pose graph and the map refinement behind them are out of scope.
"""
from __future__ import annotations

import numpy as np

from .api import History_buffer, Livox_laser, Map_buffer, Point_cloud_registration, VoxelGrid


class Laser_mapping:
    def __init__(self, scan_points: int = 24000, device: int = 0, maximum_history_size: int = 100, line_res: float = 0.1,
                 plane_res: float = 0.4, init_accumulate_frames: int = 50, input_downsample_mode: int = 1, icp_max_iterations: int = 20,
                 ceres_max_iterations: int = 100, max_allow_incre_R: float = 200.0 / 50.0, max_allow_incre_T: float = 100.0 / 50.0,
                 max_allow_final_cost: float = 100.0, history_add_t_step: float = 0.0, history_add_angle_step: float = 0.0,
                 minimum_icp_R_diff: float = 0.01, minimum_icp_T_diff: float = 0.01, maximum_residual_blocks: int = 0,
                 subsample_seed: int = 1, matching_mode: int = 0, cell_resolution: float = 1.0, threshold_cell_revisit: int = 5000,
                 maximum_search_range_corner: float = 100.0, maximum_search_range_surface: float = 100.0,
                 maximum_in_fov_angle: float = 30.0, down_sample_replace: int = 1, cell_map_max_points: int = 1 << 21,
                 loop_closure_if_enable: int = 0, loop_closure: dict | None = None, keep_cell_maps: bool = False):
        self.fe = Livox_laser(max_points=scan_points, max_scans=1, device=device, piecewise_number=1)
        # The feature node and the mapping node are separate processes in the reference: scan k + 1 is extracted while scan k is
        # registered.  process_new_scan( scan, next_xyzi = ... ) does the same with a second extractor handle (own stream): the next
        # scan's upload, extraction and selection are issued between this scan's enqueue and its collect.
        self._fe_pair = [self.fe, None]
        self._prefetched = None  # (the array object that was prefetched, its time stamp, handle)
        self._scan_points, self._device = scan_points, device
        self.reg = Point_cloud_registration(max_scans=1, max_features=scan_points, device=device)
        self.map = Map_buffer(device=device)
        self.vox = (VoxelGrid(scan_points, 1, device=device), VoxelGrid(scan_points, 1, device=device))
        self.history = History_buffer(maximum_history_size, scan_points, line_res, plane_res, device=device)
        self.line_res, self.plane_res = line_res, plane_res
        # mapping/matching_mode (laser_mapping.hpp:689; 0 in the shipped configs): 1 = match against the cell maps
        self.m_matching_mode = matching_mode
        self.m_maximum_search_range = (maximum_search_range_corner, maximum_search_range_surface)  # :694-695
        self.m_maximum_in_fov_angle = maximum_in_fov_angle                                           # :691
        self.m_down_sample_replace = down_sample_replace                                             # :277
        # m_pt_cell_map_corners / m_pt_cell_map_planes receive every registered frame in BOTH match modes (:1492-1493); only mode 1 reads
        # them per frame.  keep_cell_maps: maintain them in mode 0 too -- the sub-map a batched map-building job hands over at its end
        # (BASELINE config C4, bench_c4.py); they grow with the sequence (ll_cellmap_reserve)
        self.keep_cell_maps = bool(matching_mode or keep_cell_maps)
        if self.keep_cell_maps:
            self.history.enable_cell_map(cell_map_max_points, cell_resolution, threshold_cell_revisit)  # :620-624
            if not matching_mode:  # nothing reads them between frames: fed by the handle's service thread, beside the loop
                self.history.set_cell_map_async(True)
        self.m_if_input_downsample_mode = input_downsample_mode
        self.history_add_t_step, self.history_add_angle_step = history_add_t_step, history_add_angle_step
        p = self.reg.params
        p.icp_max_iterations, p.ceres_max_iterations = icp_max_iterations, ceres_max_iterations
        p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = max_allow_incre_R, max_allow_incre_T, max_allow_final_cost
        p.mapping_init_accumulate_frames = init_accumulate_frames
        p.minimum_icp_R_diff, p.minimum_icp_T_diff = minimum_icp_R_diff, minimum_icp_T_diff  # PCR:94-95
        # optimization/maximum_residual_blocks (200 in the shipped configs): sub-sampling on a reproducible stream; 0 = keep all
        p.maximum_allow_residual_block = maximum_residual_blocks if maximum_residual_blocks > 0 else scan_points
        p.subsample_seed = subsample_seed if maximum_residual_blocks > 0 else 0
        self.m_current_frame_index = 0
        self.pose = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)  # m_q_w_curr / m_t_w_curr
        self.map_sizes = (0, 0)
        self.last_report = None
        self.aborted_solves = 0     # registrations abandoned by the grouped solver and repeated on one workgroup (ll_reg_report.aborted)
        self.stage_s = np.zeros(4)  # cumulative wall time: extract+register, history add, match-buffer refresh, frames
        self.m_last_time_stamp = 0.0
        self._host_vox = None
        # loop_closure/if_enable_loop_closure (laser_mapping.hpp:698; 0 in the shipped Mid-40 configs): the full-cloud cell map, the key frames
        # and the front half of loop detection (keyframes.py; laser_mapping.hpp:626, 1524-1562, 919-1060)
        self.keyframes = None
        self.loops = []
        if loop_closure_if_enable:
            from .keyframes import Keyframe_assembly
            self.keyframes = Keyframe_assembly(device=device, cell_resolution=cell_resolution, threshold_cell_revisit=threshold_cell_revisit,
                                               **(loop_closure or {}))

    def close(self):
        for h in tuple(f for f in self._fe_pair if f is not None) + (self.reg, self.map, self.vox[0], self.vox[1], self.history):
            h.close()
        if self.keyframes is not None:
            self.keyframes.close()

    def _keyframe_step(self, full_xyzi: np.ndarray) -> None:
        """laser_mapping.hpp:1442 + 1524-1562 (+ the detector's loop body for whatever key frame that closed): the scan's full cloud,
        moved into the map frame with the accepted pose, goes into the full cell map and the open key frames"""
        full = np.ascontiguousarray(full_xyzi, np.float32)
        full = full[np.isfinite(full[:, :3]).all(axis=1)]
        cloud = self.reg.pointcloudAssociateToMap(full, self.pose) if len(full) else full
        self.keyframes.add_scan(cloud, self.pose, self.m_current_frame_index)
        self.loops += self.keyframes.process_waiting()

    def sync(self) -> None:
        """every frame handed to the cell maps' service thread has been appended (keep_cell_maps in matching mode 0)"""
        if self.keep_cell_maps:
            self.history.sync_cell_maps()

    def _extract(self, fe, xyzi, time_stamp):
        fe.upload(xyzi[None], np.full(1, time_stamp))
        fe.extract_batch(1)
        fe.resolve()
        fe.select_batch(1, -1, 0.0, 1.0)

    def process_new_scan(self, xyzi: np.ndarray, time_stamp: float = 1.0, next_xyzi: np.ndarray | None = None, next_time_stamp: float = 1.0,
                         scan_id=None, next_scan_id=None) -> int:
        """One frame (laser_mapping.hpp:1311-1520).  Returns the registration result (1 accepted, 0 rejected).
        next_xyzi: the scan that will be passed next, extracted on the second handle while this one registers.  The prefetched extraction
        is used by the next call only if it is recognisably the same scan: the caller's token (scan_id of that call == next_scan_id of
        this one) when tokens are given -- the contract for callers that refill one buffer in place, e.g. a ring filled by a driver thread
        -- otherwise the same array OBJECT with the same time stamp, which the caller must then not have rewritten in between."""
        try:
            return self._process_new_scan(xyzi, time_stamp, next_xyzi, next_time_stamp, scan_id, next_scan_id)
        except Exception:
            self._prefetched = None  # (a failure between the prefetch and its use must not leave a stale extraction behind)
            raise

    def _process_new_scan(self, xyzi, time_stamp, next_xyzi, next_time_stamp, scan_id, next_scan_id) -> int:
        import time
        t0 = time.perf_counter()
        reg = self.reg
        pf = self._prefetched
        self._prefetched = None
        hit = pf is not None and pf[1] == time_stamp and ((scan_id is not None and pf[3] is not None and pf[3] == scan_id) or
                                                           (scan_id is None and pf[3] is None and pf[0] is xyzi))
        if hit:
            fe = pf[2]
        else:
            fe = self._fe_pair[0]
            self._extract(fe, xyzi, time_stamp)
        self.fe = fe  # the handle that holds this frame's features (history add, key frames)
        reg.params.current_frame_index = self.m_current_frame_index  # init_pointcloud_registration runs before the increment
        self.m_current_frame_index += 1
        pose = self.pose[None]
        if self.m_if_input_downsample_mode:  # :1367-1373
            reg.enqueue_fe_downsampled(self.map, fe, self.vox[0], self.vox[1], self.line_res, self.plane_res, 1, pose, pose)
        else:
            reg.enqueue_fe(self.map, fe, 1, pose, pose)
        if next_xyzi is not None:  # the next frame's extraction runs beside this frame's ICP kernels
            other = 1 if fe is self._fe_pair[0] else 0
            if self._fe_pair[other] is None:
                self._fe_pair[other] = Livox_laser(max_points=self._scan_points, max_scans=1, device=self._device, piecewise_number=1)
            self._extract(self._fe_pair[other], next_xyzi, next_time_stamp)
            self._prefetched = (next_xyzi, next_time_stamp, self._fe_pair[other], next_scan_id)
        res, pc, _, reps = reg.collect(1)
        flags = getattr(reg, "debug_flags", 0)
        if reps[0].aborted:
            self.aborted_solves += 1
        if reps[0].aborted and not (flags & 32):
            # not a rejection the reference would have made: a bounded wait of the grouped solver ran out (the device was oversubscribed,
            # e.g. by the prefetched extraction beside it).  Counted apart, and the scan is registered once more on one workgroup -- with
            # the caller's other debug / A-B flags left as they are.  (An abort with the groups already off has another cause -- the small
            # solver could not hold the scan -- which a repeat would not cure: the scan stays rejected.)
            reg.set_debug_flags(flags | 32)
            if self.m_if_input_downsample_mode:
                reg.enqueue_fe_downsampled(self.map, fe, self.vox[0], self.vox[1], self.line_res, self.plane_res, 1, pose, pose)
            else:
                reg.enqueue_fe(self.map, fe, 1, pose, pose)
            res, pc, _, reps = reg.collect(1)
            reg.set_debug_flags(flags)
        self.last_report = reps[0]
        t1 = time.perf_counter()
        self.stage_s[0] += t1 - t0
        self.stage_s[3] += 1
        if not res[0]:  # :1413-1416
            return 0
        self.history.set_gate_pose(self.pose)  # m_q_w_curr is still the pre-registration pose at LM:1439-1451
        if self.m_if_input_downsample_mode:
            self.history.add_voxel(self.vox[0], self.vox[1], 0, pc[0], self.history_add_t_step, self.history_add_angle_step)
        else:
            self.history.add_fe(fe, 0, pc[0], self.history_add_t_step, self.history_add_angle_step)
        self.pose = pc[0].copy()  # :1496-1500
        if self.keyframes is not None:
            self._keyframe_step(np.asarray(xyzi, np.float32)[fe.get_features(0.0, 1.0)["full_idx"]])  # /pc2_full of this scan
        t2 = time.perf_counter()
        if self.m_matching_mode:  # update_buff_for_matching (service thread in the node), synchronous here
            self.map_sizes = self.history.refresh_cells(self.map, self.pose, self.m_maximum_search_range[0], self.m_maximum_search_range[1],
                                                        self.m_maximum_in_fov_angle, self.m_down_sample_replace)
        else:
            self.map_sizes = self.history.refresh(self.map)
        t3 = time.perf_counter()
        self.stage_s[1] += t2 - t1
        self.stage_s[2] += t3 - t2
        return 1

    def process_clouds(self, full: np.ndarray, surface: np.ndarray, corners: np.ndarray) -> int:
        """process_new_scan as the mapping NODE runs it (laser_mapping.hpp:1316-1520): from the three clouds the
        feature node published (/pc2_full, /pc2_surface, /pc2_corners; feature_node.Laser_feature.laserCloudHandler),
        host clouds in, through the same C-ABI entry points tools/ll_node.cpp reaches through the adapter."""
        reg = self.reg
        max_t = float(np.max(full[:, 3])) if len(full) else 0.0  # find_min_max_intensity( full ), :1336
        reg.params.minimum_pt_time_stamp, reg.params.maximum_pt_time_stamp = self.m_last_time_stamp, max_t  # :1345-1346
        self.m_last_time_stamp = max_t
        reg.params.current_frame_index = self.m_current_frame_index
        self.m_current_frame_index += 1
        if self.m_if_input_downsample_mode:  # :1367-1373
            if self._host_vox is None:
                cap = int(self.fe.params.max_points)
                self._host_vox = (VoxelGrid(cap, 1), VoxelGrid(cap, 1))
                self._host_vox[0].setLeafSize(*([self.line_res] * 3))
                self._host_vox[1].setLeafSize(*([self.plane_res] * 3))
            stacks = []
            for vg, cloud in zip(self._host_vox, (corners, surface)):
                if len(cloud):
                    vg.setInputCloud(cloud)
                    cloud = vg.filter()
                stacks.append(cloud)
            corner_stack, surf_stack = stacks
        else:
            corner_stack, surf_stack = corners, surface
        self.stack_sizes = (len(corner_stack), len(surf_stack))
        reg.m_pose_w_last = self.pose.copy()
        reg.m_pose_w_curr = self.pose.copy()
        reg.m_para_buffer_incremental = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)  # a fresh Point_cloud_registration per scan (:1348)
        res = reg.find_out_incremental_transfrom(self.map, corner_stack, surf_stack)
        self.last_report = reg.report
        if not res:  # :1413-1416
            return 0
        self.history.set_gate_pose(self.pose)  # m_q_w_curr is still the pre-registration pose at LM:1439-1451
        self.pose = np.array(reg.m_pose_w_curr, np.float64)
        self.history.add(corner_stack, surf_stack, self.pose, self.history_add_t_step, self.history_add_angle_step)
        if self.keyframes is not None:
            self._keyframe_step(full)
        self.map_sizes = self.history.refresh(self.map)
        return 1
