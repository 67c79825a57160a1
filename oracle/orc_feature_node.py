"""CPU ORACLE (test infrastructure only, see ll_oracle.h) for the non-ROS core of Laser_feature::laserCloudHandler,
Livox branch (hku-mars/loam_livox source/laser_feature_extractor.hpp:241-389), composed from the oracle's extraction
and VoxelGrid.  PARITY UNPINNED."""
import numpy as np

from . import orc


class LaserFeature:
    def __init__(self, piecewise_number=3, if_motion_deblur=0, maximum_input_lidar_pointcloud=3, mapping_plane_resolution=0.8,
                 mapping_line_resolution=0.8, odom_mode=0, para_system_delay=20, params=None):
        self.P = piecewise_number
        self.piece_wise = 1 if if_motion_deblur else piecewise_number        # LFX:305-309
        self.L = maximum_input_lidar_pointcloud
        self.leaf_surface, self.leaf_corner = mapping_plane_resolution / 2, mapping_line_resolution  # LFX:192-193
        self.odom_mode, self.delay = odom_mode, para_system_delay
        self.init_count, self.inited = 0, False
        self.params = params
        self.tb = orc.FeTimebase()                                            # one Livox_laser for all lidars (LFX:92)
        empty = np.zeros((0, 4), np.float32)
        self.full = [[empty for _ in range(self.P)] for _ in range(self.L)]
        self.surf = [[empty for _ in range(self.P)] for _ in range(self.L)]
        self.corn = [[empty for _ in range(self.P)] for _ in range(self.L)]

    def handler(self, xyzi, stamp, lidar=0):
        if not self.inited:                                                   # LFX:258-267
            self.init_count += 1
            if self.init_count >= self.delay:
                self.inited = True
            else:
                return []
        xyzi = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        o = orc.fe_extract(xyzi, self.tb.next(stamp), self.params)            # LFX:285 (LFE:722-735 time base)
        self.tb.done(o)
        S, first, last = orc.fe_split_scan(o)
        if S <= 5:                                                            # LFX:287-290
            return []
        ps, pe = orc.fe_piecewise(o, first, last, self.piece_wise)            # LFX:312-324
        for i in range(self.piece_wise):                                      # LFX:326-334
            ci, si, fi = orc.fe_get_features(o, float(ps[i]), float(pe[i]))
            self.corn[lidar][i] = orc.feature_cloud(o, ci)
            self.surf[lidar][i] = orc.feature_cloud(o, si)
            self.full[lidar][i] = orc.feature_cloud(o, fi)
        out = []
        for i in range(self.piece_wise):
            if lidar != 0:                                                    # LFX:348-351
                return out
            full = np.concatenate([self.full[ii][i] for ii in range(self.L)])  # LFX:353-358
            surf = np.concatenate([self.surf[ii][i] for ii in range(self.L)])
            corn = np.concatenate([self.corn[ii][i] for ii in range(self.L)])
            surf = orc.voxel_grid(surf, self.leaf_surface)[1] if len(surf) else surf   # LFX:372-373
            corn = orc.voxel_grid(corn, self.leaf_corner)[1] if len(corn) else corn    # LFX:379-380
            out.append((full, surf, corn))
            if self.odom_mode == 0:                                           # LFX:385-388
                break
        return out
