"""CPU ORACLE (test infrastructure only, see ll_oracle.h): restatement of Scene_alignment::find_tranfrom_of_two_mappings
(hku-mars/loam_livox source/scene_alignment.hpp:269-391) over two oracle cell maps (orc_cellmap.CellMap standing for the
key frames' cell sets), composed from the oracle's VoxelGrid, k-d tree and registration.  PARITY UNPINNED."""
import numpy as np

from . import orc

F = np.float32


def keyframe_clouds(km):
    """extract_specify_points (CMK:1263-1281) for lines and planes, get_center (CMK:1291-1301); cells in index order"""
    keys = sorted(km.cells)
    f = km.features()
    line, plane = [], []
    for i, k in enumerate(keys):
        p = km.cell_points(k)
        c = np.c_[p, np.zeros(len(p), F)].astype(F)
        if f["type"][i] == 1:
            line.append(c)
        elif f["type"][i] == 2:
            plane.append(c)
    cat = lambda v: np.concatenate(v, 0) if v else np.zeros((0, 4), F)
    ctrs = np.array([km.centre(k) for k in keys], F).reshape(-1, 3)
    centre = np.add.accumulate(ctrs, axis=0, dtype=F)[-1] * F(1.0 / float(F(len(ctrs)))) if len(ctrs) else np.zeros(3, F)
    return cat(line), cat(plane), centre


class SceneAlignment:
    def __init__(self, line_res=0.4, plane_res=0.4, maximum_icp_iteration=10, accepted_threshold=0.2, maximum_residual_block=5000,
                 subsample_seed=1, registrar_init=True):
        # registrar_init: the registrar settings of Scene_alignment::init (SA:233-243: plane blocks only, the bounds and the cost
        # gate wide open, m_inliner_dis 0.2) -- the loop detector calls init() before it aligns anything (laser_mapping.hpp:896);
        # False = a default-constructed Scene_alignment, whose m_pc_reg keeps the class defaults of point_cloud_registration.hpp
        self.registrar_init = registrar_init
        self.line_res, self.plane_res = F(line_res), F(plane_res)
        self.max_icp, self.accepted, self.max_blocks, self.seed = maximum_icp_iteration, accepted_threshold, maximum_residual_block, subsample_seed
        self.pose = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
        self.reports = []

    def find_tranfrom_of_two_mappings(self, keyframe_a, keyframe_b):
        src_line, src_plane, ca = keyframe_clouds(keyframe_a)
        tgt_line, tgt_plane, cb = keyframe_clouds(keyframe_b)
        prm = orc.RegParams.code_defaults()                                          # m_pc_reg, SA:32
        if self.registrar_init:                                                      # init(), SA:233-243 (laser_mapping.hpp:896 calls it)
            prm.icp_line = 0
            prm.max_final_cost = 20000.0
            prm.para_max_speed = 1000.0
            prm.para_max_angular_rate = 360 * 57.3
            prm.inliner_dis = 0.2
        prm.current_frame_index = 10000000                                           # SA:296
        prm.icp_max_iterations, prm.ceres_max_iterations, prm.ceres_prerun_times = self.max_icp, 50, 2   # SA:300-302
        prm.maximum_allow_residual_block, prm.subsample_seed = self.max_blocks, self.seed               # SA:303
        ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
        t0 = (ca - cb).astype(np.float64)                                            # SA:307
        last, curr, incre = ident.copy(), np.r_[ident[:4], t0], np.r_[ident[:4], t0]
        thr = 0.0
        self.reports = []
        ds = lambda c, leaf: orc.voxel_grid(c, float(leaf))[1] if len(c) else c
        for scale in (8, 4, 0):                                                      # SA:313
            lr, pr = F(self.line_res * F(scale)), F(self.plane_res * F(scale))
            if lr < self.line_res:
                lr = self.line_res
            if pr < self.plane_res:
                pr = self.plane_res
                prm.icp_max_iterations = self.max_icp * 2                            # SA:327
            sl, sp, tl, tp = ds(src_line, lr), ds(src_plane, pr), ds(tgt_line, lr), ds(tgt_plane, pr)
            if len(sl) and len(sp):                                                  # PCR:595-602
                _, curr, incre, rep = orc.reg_solve(orc.KdTree(sl), orc.KdTree(sp), tl, tp, prm, last, curr, incre)
                thr = rep.inlier_threshold
                self.reports.append(rep)
            if thr > self.accepted * 2:                                              # SA:350-351
                break
        self.pose = curr.copy()
        return float(thr)
