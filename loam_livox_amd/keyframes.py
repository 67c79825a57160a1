"""Key-frame assembly of the mapping loop and the front half of loop detection, on top of the C ABI.

Mirrors hku-mars/loam_livox source/laser_mapping.hpp:
  :626          one open key frame from the start (m_keyframe_of_updating_list);
  :1524-1562    per accepted scan: append the scan's full cloud (map frame) to the full cell map and collect the cells it touched
                (Points_cloud_map::append_cloud( pts, &cell_vec ), cell_map_keyframe.hpp:619-672: >= 3 points of the scan, every cell on an
                empty map), add them to every open key frame (Maps_keyframe::add_cells, cell_map_keyframe.hpp:1243-1261), close the oldest
                key frame after scans_of_each_keyframe scans, open another every scans_between_two_keyframe scans, keep at most
                maximum_keyframe_in_waiting_list closed ones waiting;
  :919-1060     service_loop_detection's body for one waiting key frame: update_features_of_each_cells + analyze (the four direction images),
                then against every earlier key frame: the distance-in-time gate, the non-zero ratio gate, the ROI range gate, the image
                similarities, the cell-count gate, and Scene_alignment::find_tranfrom_of_two_mappings; a pair whose inlier threshold
                ends below map_alignment_inlier_threshold is reported as a loop with the alignment's transform (:1054-1068).
The pose graph, the map refinement and every file dump behind that point (:1069-1110; ceres_pose_graph_3d.hpp) are out of scope.

A key frame is a SET OF CELLS of the full cell map, not a copy: the reference's key frames hold shared pointers to the map's cells, so
what a key frame is analysed with is whatever those cells contain when it is processed (earlier and later scans included).  Here a key
frame keeps the cell indices; `materialize` reads those cells out of the device-resident full map into a cell map of its own, in the
map's (cell, insertion) order, which is the order determine_feature's float sums run in.

Memory (the shipped loop_closure settings keep a few hundred key frames before any pair is old enough to be compared,
minimum_keyframe_differen = 200): a processed key frame keeps its direction images, its cell set and the compacted points it was analysed
with (12 bytes per point, host memory) -- NOT a device cell map; `cell_map_of` builds one again from those points (same order, so the same
cells, features and clouds) for the pairs that pass the similarity gates, and it is closed after the alignment.  The full map starts at
max_points and doubles (ll_cellmap_reserve) when a scan would not fit, as the reference's heap-allocated cells grow.

Differences from the node, by design: everything runs synchronously (the node's service thread polls every millisecond);
m_accumulated_point_cloud (only consumed by the map refinement) is not kept."""
from __future__ import annotations

from collections import deque

import numpy as np

from .api import Cell_map, keyframe_similarity
from .scene_alignment import Scene_alignment


class Maps_keyframe:
    """cell_map_keyframe.hpp:1003-1261, the members the assembly and the detector touch."""

    def __init__(self):
        self.m_set_cell = set()          # cell indices (i, j, k) of the full map
        self.m_accumulate_frames = 0
        self.m_ending_frame_idx = 0
        self.m_pose_q = np.array([0, 0, 0, 1], np.float64)
        self.m_pose_t = np.zeros(3, np.float64)
        self.points = None               # xyz of its cells as they were when it was processed, in the full map's (cell, insertion) order
        self.analysis = None             # Cell_map.keyframe_images() of those cells

    def add_cells(self, cell_ijk: np.ndarray) -> None:   # :1243-1261
        self.m_set_cell.update(_pack_cells(cell_ijk).tolist())
        self.m_accumulate_frames += 1

    # what service_loop_detection reads
    @property
    def m_ratio_nonzero_line(self):
        return float(self.analysis["ratio_nonzero"][0])

    @property
    def m_ratio_nonzero_plane(self):
        return float(self.analysis["ratio_nonzero"][1])

    @property
    def m_roi_range(self):
        return float(self.analysis["roi_range"])

    @property
    def m_feature_img_line(self):
        return self.analysis["images"][0]

    @property
    def m_feature_img_plane(self):
        return self.analysis["images"][1]


class Keyframe_assembly:
    def __init__(self, device: int = 0, cell_resolution: float = 1.0, threshold_cell_revisit: int = 5000, max_points: int = 1 << 22,
                 scans_of_each_keyframe: int = 300, scans_between_two_keyframe: int = 100, maximum_keyframe_in_waiting_list: int = 3,
                 minimum_keyframe_differen: int = 200, minimum_similarity_linear: float = 0.65, minimum_similarity_planar: float = 0.95,
                 map_alignment_resolution: float = 0.2, map_alignment_inlier_threshold: float = 0.35,
                 map_alignment_maximum_icp_iteration: int = 2, scene_alignments_maximum_residual_block: int = 5000,
                 keyframe_max_points: int = 1 << 20, full_cell_map=None, avail_ratio_plane: float = 0.05, avail_ratio_line: float = 0.03):
        # parameter names and defaults: laser_mapping.hpp:698-710 (loop_closure/*), :686-687 (mapping/pt_cell_resolution, threshold_cell_revisit)
        self.device = device
        self.m_pt_cell_resolution = cell_resolution
        # (full_cell_map: anything with append_cloud_touched / dump / close -- tests drive the bookkeeping without a device)
        self.m_pt_cell_map_full = full_cell_map if full_cell_map is not None else \
            Cell_map(max_points, cell_resolution, threshold_cell_revisit, device=device)   # :616-617
        self.m_para_scans_of_each_keyframe = scans_of_each_keyframe
        self.m_para_scans_between_two_keyframe = scans_between_two_keyframe
        self.m_loop_closure_maximum_keyframe_in_wating_list = maximum_keyframe_in_waiting_list
        self.m_loop_closure_minimum_keyframe_differen = minimum_keyframe_differen
        self.m_loop_closure_minimum_similarity_linear = minimum_similarity_linear
        self.m_loop_closure_minimum_similarity_planar = minimum_similarity_planar
        self.m_loop_closure_map_alignment_resolution = map_alignment_resolution
        self.m_loop_closure_map_alignment_inlier_threshold = map_alignment_inlier_threshold
        self.m_loop_closure_map_alignment_maximum_icp_iteration = map_alignment_maximum_icp_iteration
        self.m_para_scene_alignments_maximum_residual_block = scene_alignments_maximum_residual_block
        self.keyframe_max_points = keyframe_max_points
        # locals of service_loop_detection (laser_mapping.hpp:887-888: "0.05 for 300 scans, 0.15 for 1000 scans"): a key frame whose direction
        # images are emptier than this is not compared against
        self.avail_ratio_plane, self.avail_ratio_line = avail_ratio_plane, avail_ratio_line
        self.m_keyframe_of_updating_list = deque([Maps_keyframe()])   # :626
        self.m_keyframe_need_precession_list = deque()
        self.keyframe_vec = []          # service_loop_detection's keyframe_vec
        self.pose3d_vec = []            # ... and the poses it pairs with them
        self.loops = []                 # detected loops: dict(his, last, inlier_threshold, icp_q, icp_t)
        self.if_end = False
        self.log = []                   # one record per compared pair (what the node writes to loop_closure.log)

    def state(self) -> str:
        """the lists as tests/verbatim_build.py's harness prints them: open key frames frames:cells, waiting ones frames:cells:ending-index"""
        u = " ".join(f"{kf.m_accumulate_frames}:{len(kf.m_set_cell)}" for kf in self.m_keyframe_of_updating_list)
        w = " ".join(f"{kf.m_accumulate_frames}:{len(kf.m_set_cell)}:{kf.m_ending_frame_idx}" for kf in self.m_keyframe_need_precession_list)
        return ("U " + u).rstrip() + " W" + (" " + w if w else "")

    def close(self):
        if self.m_pt_cell_map_full is not None:
            self.m_pt_cell_map_full.close()
            self.m_pt_cell_map_full = None

    # ---- laser_mapping.hpp:1524-1562 ------------------------------------------------------------------------------------------------
    def add_scan(self, full_cloud_map_frame: np.ndarray, pose: np.ndarray, current_frame_index: int) -> np.ndarray:
        """One accepted scan.  full_cloud_map_frame: current_laser_cloud_full after pointcloudAssociateToMap; pose: m_q_w_curr /
        m_t_w_curr as {qx, qy, qz, qw, tx, ty, tz}.  Returns the touched cells (cell_vec)."""
        full = self.m_pt_cell_map_full
        if hasattr(full, "reserve"):  # the reference's map grows on the heap: double the device map's capacity when this scan would not fit
            need = full.stats()[1] + len(full_cloud_map_frame)
            if need > full.max_points:
                full.reserve(max(2 * full.max_points, need))
        cell_vec = full.append_cloud_touched(full_cloud_map_frame, 3)
        for kf in self.m_keyframe_of_updating_list:
            kf.add_cells(cell_vec)
        front = self.m_keyframe_of_updating_list[0]
        if front.m_accumulate_frames >= self.m_para_scans_of_each_keyframe:
            front.m_ending_frame_idx = current_frame_index
            front.m_pose_q = np.array(pose[:4], np.float64)
            front.m_pose_t = np.array(pose[4:7], np.float64)
            self.m_keyframe_need_precession_list.append(front)
            self.m_keyframe_of_updating_list.popleft()
        # (with scans_of_each_keyframe <= scans_between_two_keyframe the list can run empty here; the node would dereference back() of
        #  an empty list -- the shipped 300 / 100 never does; a new key frame is opened instead)
        if not self.m_keyframe_of_updating_list:
            self.m_keyframe_of_updating_list.append(Maps_keyframe())
        elif self.m_keyframe_of_updating_list[-1].m_accumulate_frames >= self.m_para_scans_between_two_keyframe:
            if len(self.m_keyframe_need_precession_list) > self.m_loop_closure_maximum_keyframe_in_wating_list:
                self.m_keyframe_need_precession_list.popleft()
            self.m_keyframe_of_updating_list.append(Maps_keyframe())
        return cell_vec

    # ---- the key frame's view of the shared cells ------------------------------------------------------------------------------------
    def materialize(self, kf: Maps_keyframe) -> Cell_map:
        xyz, ijk, start, _ = self.m_pt_cell_map_full.dump()
        want = np.fromiter(kf.m_set_cell, np.int64, len(kf.m_set_cell))
        sel = np.flatnonzero(np.isin(_pack_cells(ijk), want))
        start = np.asarray(start, np.int64)
        lens = start[sel + 1] - start[sel]
        # the points of the selected cells, cell after cell: position p of the output lies in selected cell c(p) at offset p - first(c)
        first = np.cumsum(lens) - lens
        idx = np.repeat(start[sel] - first, lens) + np.arange(int(lens.sum()), dtype=np.int64)
        return self._cell_map_from_points(xyz[idx])

    def _cell_map_from_points(self, xyz: np.ndarray) -> Cell_map:
        km = Cell_map(max(1024, len(xyz) + 1), self.m_pt_cell_resolution, device=self.device)
        if len(xyz):
            km.append_cloud(np.c_[xyz, np.zeros(len(xyz), np.float32)].astype(np.float32))
        return km

    def cell_map_of(self, kf: Maps_keyframe):
        """a device cell map of a processed key frame, rebuilt from the points it was analysed with (the caller closes it); a key frame
        that kept no points (test stubs) is read out of the full map again"""
        return self._cell_map_from_points(kf.points) if kf.points is not None else self.materialize(kf)

    # ---- laser_mapping.hpp:919-1060 --------------------------------------------------------------------------------------------------
    def process_waiting(self):
        """Every waiting key frame through one pass of service_loop_detection's loop body.  Returns the loops found in this call."""
        found = []
        avail_ratio_plane, avail_ratio_line = self.avail_ratio_plane, self.avail_ratio_line   # :887-888
        while self.m_keyframe_need_precession_list and not self.if_end:
            last = self.m_keyframe_need_precession_list.popleft()
            cm = self.materialize(last)                     # update_features_of_each_cells + analyze read the cells as they are now
            last.analysis = cm.keyframe_images()
            last.points = cm.dump()[0] if hasattr(cm, "dump") else None
            _close(cm)
            self.keyframe_vec.append(last)
            self.pose3d_vec.append((last.m_pose_q.copy(), last.m_pose_t.copy()))
            n_kf = len(self.keyframe_vec)
            his = 0
            while his < n_kf - 1:
                if self.if_end:
                    break
                old = self.keyframe_vec[his]
                rec = dict(last=n_kf - 1, his=his)
                if n_kf - his < self.m_loop_closure_minimum_keyframe_differen:   # :994
                    his += 1
                    continue
                if old.m_ratio_nonzero_plane < avail_ratio_plane and old.m_ratio_nonzero_line < avail_ratio_line:   # :1001
                    his += 1
                    continue
                if abs(old.m_roi_range - last.m_roi_range) > 5.0:   # :1004
                    his += 1
                    continue
                sim_plane = keyframe_similarity(last.m_feature_img_plane, old.m_feature_img_plane, self.device)   # :1009-1010
                sim_line = keyframe_similarity(last.m_feature_img_line, old.m_feature_img_line, self.device)
                rec.update(sim_plane=sim_plane, sim_line=sim_line)
                self.log.append(rec)
                if (sim_line > self.m_loop_closure_minimum_similarity_linear and sim_plane > 0.92) or \
                        sim_plane > self.m_loop_closure_minimum_similarity_planar:   # :1012-1013
                    # :1030  ( a - b ) / ( a + b ) * 0.1 in size_t arithmetic: zero unless a < b, where a - b wraps around
                    if len(last.m_set_cell) < len(old.m_set_cell):
                        his += 1
                        continue
                    sa = Scene_alignment(self.m_loop_closure_map_alignment_resolution, self.m_loop_closure_map_alignment_resolution,   # :1034
                                         self.m_loop_closure_map_alignment_maximum_icp_iteration, self.m_loop_closure_map_alignment_inlier_threshold,
                                         self.m_para_scene_alignments_maximum_residual_block, device=self.device)   # :897-898, 1035
                    cm_last, cm_old = self.cell_map_of(last), self.cell_map_of(old)
                    thr = sa.find_tranfrom_of_two_mappings(cm_last, cm_old)   # :1036
                    _close(cm_last)
                    _close(cm_old)
                    rec.update(inlier_threshold=thr, pose=sa.pose.copy())
                    if thr > self.m_loop_closure_map_alignment_inlier_threshold * 2:   # :1048-1052
                        his += 10 + 1
                        continue
                    if thr < self.m_loop_closure_map_alignment_inlier_threshold:   # :1054: "I believe this is true loop."
                        q, t = sa.pose[:4], sa.pose[4:7]
                        qi = np.array([-q[0], -q[1], -q[2], q[3]])                # :1064-1065  ICP_t = ICP_q^-1 * (-ICP_t); ICP_q = ICP_q^-1
                        ti = _quat_rot(qi, -t)
                        loop = dict(his=his, last=n_kf - 1, inlier_threshold=thr, icp_q=qi, icp_t=ti, sim_plane=sim_plane, sim_line=sim_line)
                        self.loops.append(loop)
                        found.append(loop)
                        self.if_end = True   # :1108 (the pose graph optimisation and the map refinement in between are out of scope)
                        break
                    his += 5 + 1   # :1111-1114
                    continue
                his += 1
        return found


def _pack_cells(cell_ijk) -> np.ndarray:
    """(i, j, k) cell indices -> one int64 per cell (21 bits per axis, like the device's cell key)"""
    c = np.asarray(cell_ijk, np.int64).reshape(-1, 3) + (1 << 20)
    return c[:, 0] | (c[:, 1] << 21) | (c[:, 2] << 42)


def _close(cm) -> None:
    if hasattr(cm, "close"):
        cm.close()


def _quat_rot(q, v):
    """rotate v by the unit quaternion q = {x, y, z, w}"""
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R @ np.asarray(v, np.float64)
