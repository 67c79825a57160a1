"""Non-ROS core of Laser_feature::laserCloudHandler, Livox branch (hku-mars/loam_livox
source/laser_feature_extractor.hpp:241-389): the piece-wise / multi-lidar front-end of the feature-extraction node
(SURVEY 8(f) row 3), on top of the device extractor and the device VoxelGrid.

One call per incoming lidar message.  What the node publishes on /pc2_full, /pc2_surface, /pc2_corners comes back as a
list of (full, surface, corners) triples, one per published piece:
  * the first `m_para_system_delay` messages are dropped (:258-267);
  * a scan with 5 or fewer petal clouds is dropped (:287-290);
  * the scan is cut into `piecewise_number` pieces by petal count (1 piece when motion deblur is on, :305-309) and the
    features of piece i are selected with the window [piece_wise_start[i], piece_wise_end[i]] (:312-334);
  * the clouds are kept per (lidar, piece); only a message of lidar 0 publishes, and it publishes, per piece, the
    concatenation over all lidars of their latest clouds for that piece (:348-358) -- the Mid-100's three heads;
  * surface / corner clouds pass the voxel filters (leaf plane_res / 2 and line_res, :192-193, 372-381);
  * in odometry mode only the first piece is published (:385-388).
"""
from __future__ import annotations

import numpy as np

from .api import Livox_laser, VoxelGrid


class Laser_feature:
    def __init__(self, max_points: int = 24000, piecewise_number: int = 3, if_motion_deblur: int = 0,
                 maximum_input_lidar_pointcloud: int = 3, mapping_plane_resolution: float = 0.8, mapping_line_resolution: float = 0.8,
                 odom_mode: int = 0, para_system_delay: int = 20, device: int = 0, **livox_tunables):
        self.m_piecewise_number = piecewise_number
        self.m_if_motion_deblur = if_motion_deblur
        self.m_maximum_input_lidar_pointcloud = maximum_input_lidar_pointcloud
        self.m_odom_mode = odom_mode
        self.m_para_system_delay = para_system_delay
        self.m_para_system_init_count = 0
        self.m_para_systemInited = False
        self.piece_wise = 1 if if_motion_deblur else piecewise_number  # :305-309
        # one extractor for all lidars, as in the node (a single m_livox, :92): its time base runs across the messages
        self.m_livox = Livox_laser(max_points=max_points, device=device, piecewise_number=self.piece_wise, **livox_tunables)
        cap = max_points * maximum_input_lidar_pointcloud
        self.m_voxel_filter_for_surface = VoxelGrid(cap, 1, device=device)
        self.m_voxel_filter_for_corner = VoxelGrid(cap, 1, device=device)
        self.m_voxel_filter_for_surface.setLeafSize(*([mapping_plane_resolution / 2] * 3))  # :192
        self.m_voxel_filter_for_corner.setLeafSize(*([mapping_line_resolution] * 3))        # :193
        empty = np.zeros((0, 4), np.float32)
        L, P = maximum_input_lidar_pointcloud, piecewise_number
        self.m_map_pointcloud_full_vec_vec = [[empty for _ in range(P)] for _ in range(L)]
        self.m_map_pointcloud_surface_vec_vec = [[empty for _ in range(P)] for _ in range(L)]
        self.m_map_pointcloud_corner_vec_vec = [[empty for _ in range(P)] for _ in range(L)]
        self.m_laser_scan_number = 0

    def close(self):
        for h in (self.m_livox, self.m_voxel_filter_for_surface, self.m_voxel_filter_for_corner):
            h.close()

    def _filter(self, vg: VoxelGrid, cloud: np.ndarray) -> np.ndarray:
        if len(cloud) == 0:
            return cloud
        vg.setInputCloud(cloud)
        return vg.filter()

    def laserCloudHandler(self, laserCloudIn: np.ndarray, stamp: float, current_lidar_index: int = 0):
        """Returns the list of published (livox_full, livox_surface, livox_corners) triples of this message."""
        assert 0 <= current_lidar_index < self.m_maximum_input_lidar_pointcloud  # :254
        if not self.m_para_systemInited:  # :258-267
            self.m_para_system_init_count += 1
            if self.m_para_system_init_count >= self.m_para_system_delay:
                self.m_para_systemInited = True
            else:
                return []
        xyzi = np.ascontiguousarray(laserCloudIn, np.float32).reshape(-1, 4)
        n_clouds = self.m_livox.extract_laser_features(xyzi, stamp)  # :285
        if n_clouds <= 5:  # :287-290
            return []
        self.m_laser_scan_number = n_clouds
        sp = self.m_livox.splits()
        info = None
        for i in range(self.piece_wise):  # :326-334; the windows of :312-324 are computed on the device
            g = self.m_livox.get_features(float(sp["piece_start"][i]), float(sp["piece_end"][i]))
            if info is None:
                info = self.m_livox.pts_info()
            full = np.concatenate([xyzi[g["full_idx"], :3], info["time_stamp"][g["full_idx"], None]], axis=1).astype(np.float32)
            self.m_map_pointcloud_corner_vec_vec[current_lidar_index][i] = g["pc_corners"]
            self.m_map_pointcloud_surface_vec_vec[current_lidar_index][i] = g["pc_surface"]
            self.m_map_pointcloud_full_vec_vec[current_lidar_index][i] = full
        published = []
        for i in range(self.piece_wise):
            if current_lidar_index != 0:  # :348-351
                return published
            L = range(self.m_maximum_input_lidar_pointcloud)
            livox_full = np.concatenate([self.m_map_pointcloud_full_vec_vec[ii][i] for ii in L])        # :353-358
            livox_surface = np.concatenate([self.m_map_pointcloud_surface_vec_vec[ii][i] for ii in L])
            livox_corners = np.concatenate([self.m_map_pointcloud_corner_vec_vec[ii][i] for ii in L])
            livox_surface = self._filter(self.m_voxel_filter_for_surface, livox_surface)                  # :372-373
            livox_corners = self._filter(self.m_voxel_filter_for_corner, livox_corners)                   # :379-380
            published.append((livox_full, livox_surface, livox_corners))
            if self.m_odom_mode == 0:  # :385-388
                break
        return published
