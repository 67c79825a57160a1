"""CPU ORACLE (test infrastructure only, see ll_oracle.h) for the cell ("cube") match mode: a restatement of
Points_cloud_map<float> of hku-mars/loam_livox (source/cell_map_keyframe.hpp:477-790: set_resolution, find_cell_center,
find_cell with the revisit rule, append_cloud, find_cells_in_radius), of Laser_mapping::if_pt_in_fov
(source/laser_mapping.hpp:310-324) and of the cell branch of update_buff_for_matching (:471-513).  PARITY UNPINNED:
the reference cannot be built here (PCL / Eigen absent).  One thing is defined rather than restated: the order in
which find_cells_in_radius returns cells (a PCL octree traversal in the reference) is ascending (ix, iy, iz) here.
Plain dictionaries and per-cell calls of the oracle VoxelGrid: sized for small test cases."""
import numpy as np

from . import orc

F = np.float32
K_LIMIT = 1 << 20


def _round_half_away(v):
    """std::round on float32 values, exactly."""
    a = np.abs(v)
    r = np.floor(a)
    r = r + ((a - r) >= F(0.5)).astype(F)
    return np.copysign(r, v).astype(F)


class CellMap:
    def __init__(self, resolution=1.0, minimum_revisit_threshold=2**31 - 1):
        m_resolution = F(np.float64(F(resolution)) * 0.5)     # set_resolution, CMK:675-677
        self.box = F(np.float64(m_resolution) * 1.0)          # CMK:559
        self.half = F(np.float64(m_resolution) * 0.5)         # CMK:560
        self.thr = int(minimum_revisit_threshold)
        self.frame = 0                                         # m_current_frame_idx
        self.cells = {}                                        # (ix, iy, iz) -> {"pts": [n,3] float32, "last": int}

    # find_cell_center, CMK:566-568
    def cell_index(self, xyz):
        xyz = np.asarray(xyz, F).reshape(-1, 3)
        with np.errstate(invalid="ignore"):
            r = _round_half_away((xyz - self.half) / self.box)
        ok = np.all(np.abs(r) < F(K_LIMIT), axis=1) & np.all(np.isfinite(xyz), axis=1)
        return np.where(ok[:, None], r, 0).astype(np.int64), ok

    def centre(self, k):
        return (np.asarray(k, F) * self.box + self.half).astype(F)

    # append_cloud, CMK:619-672 (set_point_cloud :590-617 for the first cloud does the same on an empty map)
    def append(self, cloud):
        cloud = np.asarray(cloud, F)
        xyz = cloud.reshape(len(cloud), -1)[:, :3] if len(cloud) else np.zeros((0, 3), F)
        k, ok = self.cell_index(xyz)
        for i in np.nonzero(ok)[0]:
            key = (int(k[i, 0]), int(k[i, 1]), int(k[i, 2]))
            c = self.cells.get(key)
            if c is None:                                      # add_cell, CMK:686-713
                c = self.cells[key] = {"pts": [], "last": self.frame}
            elif self.frame - c["last"] < self.thr:            # CMK:737-741
                c["last"] = self.frame
            else:                                              # CMK:742-754: a fresh cell takes the place of the old one
                c = self.cells[key] = {"pts": [], "last": self.frame}
            c["pts"].append(xyz[i].copy())
        self.frame += 1

    def cell_points(self, key):
        p = self.cells[key]["pts"]
        return np.asarray(p, F).reshape(-1, 3)

    def n_points(self):
        return sum(len(c["pts"]) for c in self.cells.values())

    @staticmethod
    def in_fov(centre, pose, maximum_in_fov_angle):
        """if_pt_in_fov, LM:310-324 (double; Eigen's q.inverse() * v)."""
        q, t = np.asarray(pose[:4], np.float64), np.asarray(pose[4:], np.float64)
        v = centre.astype(np.float64) - t
        n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
        u, w = -q[:3] / n2, q[3] / n2
        uv = np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]])
        uv = uv + uv
        r = np.array([v[0] + w * uv[0] + (u[1] * uv[2] - u[2] * uv[1]), v[1] + w * uv[1] + (u[2] * uv[0] - u[0] * uv[2]),
                      v[2] + w * uv[2] + (u[0] * uv[1] - u[1] * uv[0])])
        if r[0] < 0:
            return False
        nrm = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
        angle = F(0.0) if nrm == 0 else F(np.arccos(abs(r[0]) / (nrm * 1.0)))   # eigen_math.hpp:25-46, stored in a float
        return float(angle) * 57.3 < maximum_in_fov_angle

    # find_cells_in_radius (CMK:761-788) + if_pt_in_fov
    def select(self, pose, radius, maximum_in_fov_angle):
        sp = np.asarray(pose[4:], np.float64).astype(F)        # eigen_to_pcl_pt<pcl::PointXYZ>
        out = []
        for key in sorted(self.cells):
            c = self.centre(key)
            d = c - sp
            d2 = F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2])
            if float(d2) > float(F(radius)) * float(F(radius)):
                continue
            if self.in_fov(c, pose, maximum_in_fov_angle):
                out.append(key)
        return out

    # LM:481-497 (corners) / :499-513 (planes)
    def query_filter(self, pose, radius, maximum_in_fov_angle, leaf, down_sample_replace=1):
        cat = []
        keys = self.select(pose, radius, maximum_in_fov_angle)
        for key in keys:
            p = self.cell_points(key)
            if len(p) == 0:
                continue
            cloud = np.concatenate([p, np.zeros((len(p), 1), F)], 1)
            f = orc.voxel_grid(cloud, leaf)[1]
            if down_sample_replace:                            # set_pointcloud, CMK:352-357
                self.cells[key]["pts"] = [r.copy() for r in f[:, :3]]
            cat.append(f)
        return (np.concatenate(cat, 0) if cat else np.zeros((0, 4), F)), keys

    def dump(self):
        """points in (cell, insertion) order, cell indices, starts, last-update frames"""
        keys = sorted(self.cells)
        pts = [self.cell_points(k) for k in keys]
        start = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.int32)
        xyz = np.concatenate(pts, 0) if pts else np.zeros((0, 3), F)
        return xyz, np.asarray(keys, np.int32).reshape(-1, 3), start, np.asarray([self.cells[k]["last"] for k in keys], np.int32)
