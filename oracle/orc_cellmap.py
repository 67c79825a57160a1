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
        was_empty = len(self.cells) == 0
        hits = {}
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
            hits[key] = hits.get(key, 0) + 1
        # m_current_frame_idx++ (CMK:667) -- and once more when the map was empty at the call: append_cloud then goes through
        # set_point_cloud, which increments it too (CMK:615).  Pinned by oracle/ref_cells.py (tests/test_ref_cells.py).
        self.frame += 2 if was_empty else 1
        # cell_vec of append_cloud( pts, &cell_vec ): every cell of the first cloud (CMK:606-609), afterwards the cells that received
        # at least 3 points of this cloud (CMK:640-662)
        return sorted(k for k, n in hits.items() if was_empty or n >= 3)

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
    def cells_in_radius(self, pt, radius):
        sp = np.asarray(pt, np.float64).astype(F)              # eigen_to_pcl_pt<pcl::PointXYZ>
        out = []
        for key in sorted(self.cells):
            d = self.centre(key) - sp
            d2 = F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2])
            if not float(d2) > float(F(radius)) * float(F(radius)):
                out.append(key)
        return out

    def select(self, pose, radius, maximum_in_fov_angle):
        if maximum_in_fov_angle >= 360.0:  # find_cells_in_radius on its own, as service_pub_surround_pts calls it (LM:1172): no field-of-view test
            return list(self.cells_in_radius(pose[4:], radius))
        return [key for key in self.cells_in_radius(pose[4:], radius) if self.in_fov(self.centre(key), pose, maximum_in_fov_angle)]

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

    # Points_cloud_cell::determine_feature( if_recompute = 1 ), CMK:436-473 (get_mean :225-237, get_covmat :280-315 with
    # COMP_TYPE float :41 and the non-incremental update :30, covmat_eig_decompose :239-249).  The eigen decomposition is
    # LAPACK's (numpy.linalg.eigh) on the float covariance in double -- independent of the device's Jacobi iteration.
    def features(self):
        keys = sorted(self.cells)
        n = len(keys)
        out = dict(type=np.zeros(n, np.int32), vector=np.zeros((n, 3), F), mean=np.zeros((n, 3), F), cov=np.zeros((n, 6), F),
                   eigen_val=np.zeros((n, 3), F), margin=np.full(n, np.inf))
        ia, ib = [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]
        for i, key in enumerate(keys):
            p = self.cell_points(key)
            cnt = len(p)
            if cnt == 0:
                continue
            mean = np.add.accumulate(p, axis=0, dtype=F)[-1] / F(cnt)              # m_xyz_sum / size, sequential float sums
            out["mean"][i] = mean
            if cnt < 5:                                                             # CMK:446-451
                continue
            c = np.add.accumulate(p[:, ia] * p[:, ib], axis=0, dtype=F)[-1]          # CMK:304-307
            cov = ((c - F(cnt) * (mean[ia] * mean[ib])) / F(cnt - 1)).astype(F)      # CMK:309-310
            out["cov"][i] = cov
            m = np.array([[cov[0], cov[1], cov[2]], [cov[1], cov[3], cov[4]], [cov[2], cov[4], cov[5]]], np.float64)
            w, v = np.linalg.eigh(m)
            ev = w.astype(F)
            out["eigen_val"][i] = ev
            d = self.centre(key) - mean
            dist = np.sqrt(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2]), dtype=F)
            lim = float(self.box) * 0.75
            third = 1.0 / 3.0
            # how far the cell is from each decision boundary (tests skip cells that sit on one)
            out["margin"][i] = min(abs(float(dist) - lim) / lim,
                                   abs(float(ev[1]) * third - float(ev[0])) / max(abs(float(ev[1])), 1e-30),
                                   abs(float(ev[2]) * third - float(ev[1])) / max(abs(float(ev[2])), 1e-30))
            if float(dist) > lim:                                                   # CMK:455-460
                continue
            if float(ev[1]) * third > float(ev[0]):                                  # CMK:462-467
                out["type"][i], out["vector"][i] = 2, v[:, 0].astype(F)
            elif float(ev[2]) * third > float(ev[1]):                                # CMK:468-472
                out["type"][i], out["vector"][i] = 1, v[:, 2].astype(F)
        return out

    # Maps_keyframe::analyze over the cells of this map: get_center / get_ratio_range_of_cell (CMK:1291-1319),
    # extract_feature_mapping_new (:1429-1484), generate_feature_img (:1385-1427), eigen_decompose_of_featurevector
    # (:1554-1567), feature_direction (:1071-1089), apply_guassian_blur (:1360-1372, as a circular convolution in double).
    # `features` may carry the labels / vectors of another implementation, to compare the image stage alone.
    def keyframe_images(self, roi_ratio=0.9, features=None):
        f = self.features() if features is None else features
        keys = sorted(self.cells)
        n = len(keys)
        out = dict(images=np.zeros((4, 60, 60), F), ratio_nonzero=np.zeros(4, F), eigen_R=np.zeros((2, 3, 3), F), n_vectors=np.zeros(4, np.int32),
                   centre=np.zeros(3, F), roi_range=0.0, near_bin_edge=0)
        if n == 0:
            return out
        ctrs = np.array([self.centre(k) for k in keys], F)
        member = [np.ones(n, bool)]
        if roi_ratio > 0:
            centre = np.add.accumulate(ctrs, axis=0, dtype=F)[-1] * F(1.0 / float(F(n)))
            d = ctrs - centre
            dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=F)
            dv = np.unique(dist)                                                   # std::set<float>
            rng_ = dv[int(np.ceil(F(len(dv) - 1) * F(roi_ratio)))]
            out["centre"], out["roi_range"] = centre, float(rng_)
            member.append(dist < rng_)
        x = np.arange(9) - 4.0
        gk = np.exp(-0.5 / 16.0 * x * x).astype(F)                                 # cv::getGaussianKernel( 9, 4, CV_32F )
        gk = (gk.astype(np.float64) * (1.0 / gk.astype(np.float64).sum())).astype(F).astype(np.float64)
        for roi, mem in enumerate(member):
            pl = mem & (f["type"] == 2)
            v = f["vector"][pl]
            prod = (v[:, :, None] * v[:, None, :]).astype(F).astype(np.float64)    # ( v v^T ) in float, cast to double
            w, V = np.linalg.eigh(np.eye(3) + prod.sum(0))
            V = V[:, ::-1].astype(F)                                               # rowwise().reverse(): largest first
            for j in range(2):                                                     # the device's sign convention
                lead = V[:, j][np.nonzero(V[:, j])[0][0]]
                if lead < 0:
                    V[:, j] = -V[:, j]
            V[:, 2] = np.array([V[1, 0] * V[2, 1] - V[2, 0] * V[1, 1], V[2, 0] * V[0, 1] - V[0, 0] * V[2, 1], V[0, 0] * V[1, 1] - V[1, 0] * V[0, 1]], F)
            out["eigen_R"][roi] = V
            for which, typ in ((0, 1), (1, 2)):
                vv = f["vector"][mem & (f["type"] == typ)]
                a = ((V[0][None, :] * vv[:, 0:1] + V[1][None, :] * vv[:, 1:2]) + V[2][None, :] * vv[:, 2:3]).astype(F)
                a = np.where(a[:, 0:1] < 0, (a * F(-1.0)).astype(F), a).astype(np.float64)
                phi = np.arctan2(a[:, 1], a[:, 0]) + np.pi / 2
                with np.errstate(invalid="ignore"):
                    theta = np.arcsin(a[:, 2]) + np.pi / 2
                pf, tf = phi / (np.pi / 60), theta / (np.pi / 60)
                out["near_bin_edge"] += int(np.sum(np.abs(pf - np.round(pf)) < 1e-9) + np.sum(np.abs(tf - np.round(tf)) < 1e-9))
                pi_ = np.clip(np.floor(pf), 0, 59).astype(int)
                ti_ = np.where(np.isnan(tf), np.where(a[:, 2] > 0, 59, 0), np.clip(np.floor(np.nan_to_num(tf)), 0, 59)).astype(int)
                h = np.zeros((60, 60))
                np.add.at(h, (pi_, ti_), 1.0)
                out["n_vectors"][2 * roi + which] = len(vv)
                out["ratio_nonzero"][2 * roi + which] = F(np.sum(h >= 1.0)) / F(3600)
                blur = sum(gk[k] * np.roll(h, 4 - k, axis=1) for k in range(9))
                blur = sum(gk[k] * np.roll(blur, 4 - k, axis=0) for k in range(9))
                out["images"][2 * roi + which] = blur.astype(F)
        return out

    @staticmethod
    def max_similarity(img_a, img_b):
        """max_similiarity_of_two_image (CMK:1155-1224): max over circular shifts of the normalised correlation."""
        a, b = np.asarray(img_a, np.float64), np.asarray(img_b, np.float64)
        t = np.sqrt((a * a).sum() * (b * b).sum())
        if not t > 0:
            return 0.0
        corr = np.real(np.fft.ifft2(np.conj(np.fft.fft2(a)) * np.fft.fft2(b)))   # corr[s] = sum a(i) b(i + s), all 60 x 60 shifts
        return float(corr.max() / t)

    def dump(self):
        """points in (cell, insertion) order, cell indices, starts, last-update frames"""
        keys = sorted(self.cells)
        pts = [self.cell_points(k) for k in keys]
        start = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.int32)
        xyz = np.concatenate(pts, 0) if pts else np.zeros((0, 3), F)
        return xyz, np.asarray(keys, np.int32).reshape(-1, 3), start, np.asarray([self.cells[k]["last"] for k in keys], np.int32)
