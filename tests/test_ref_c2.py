"""BASELINE config C2 size against the REFERENCE ITSELF: tests/golden/ref_c2_scene*.npz are outputs of the reference's own
extractor and registration driver (oracle/_ref/libll_ref.so = /root/reference compiled in the build container) for 24 000-point
Mid-40 scans against the 5 M-point map -- written by tests/golden/gen_ref_c2.py, which travels with them.  The inputs are
re-created here from their seeds and verified by checksum.

  CPU tier : the oracle reproduces them (index sets exactly, pose to 1e-9, counts and costs);
  GPU tier : the HIP path reproduces them through the C ABI -- not transitively through the oracle (missing #2)."""
import glob
import os
import zlib

import numpy as np
import pytest

from loam_livox_amd import synth
from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_c2_scene*.npz")))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@pytest.fixture(scope="module")
def c2_world():
    g = np.load(SCENES[0])
    world, corner, surf = synth.make_maps(int(g["map_points"]))
    assert crc(corner) == int(g["corner_crc"]) and crc(surf) == int(g["surf_crc"]), "the synthetic 5 M-point map changed"
    return world, corner, surf


def scan_of(world, g):
    sc = synth.make_scan(world, int(g["scan_seed"]))
    assert crc(sc.xyzi) == int(g["scan_crc"]), "the synthetic scan changed"
    assert np.array_equal(sc.pose_init, g["pose_init"])
    return sc


def test_fixtures_exist():
    assert len(SCENES) >= 2


def test_oracle_reproduces_the_reference_at_c2_size(c2_world):
    world, corner, surf = c2_world
    tc, ts = orc.KdTree(corner), orc.KdTree(surf)
    for path in SCENES:
        g = np.load(path)
        sc = scan_of(world, g)
        o = orc.fe_extract(sc.xyzi, float(g["current_time"]))
        ci, si, fi = orc.fe_get_features(o, 0.0, 1.0)
        assert np.array_equal(ci, g["corner_idx"]) and np.array_equal(si, g["surf_idx"])
        prm = orc.RegParams.defaults(icp_iters=int(g["icp_iters"]), ceres_iters=int(g["ceres_iters"]))
        prm.max_final_cost = float(g["max_final_cost"])
        ret, pc, pi, rep = orc.reg_solve(tc, ts, orc.feature_cloud(o, ci), orc.feature_cloud(o, si), prm, sc.pose_init, sc.pose_init)
        dt, dr = synth.pose_error(pc, g["pose_out"])
        assert ret == int(g["reg_ret"]) and dt < 1e-9 and dr < 1e-9, path
        assert rep.n_blocks_last == int(g["n_blocks_last"])
        assert abs(rep.final_cost - float(g["final_cost"])) < 1e-9 * max(1.0, float(g["final_cost"]))
        assert abs(rep.inlier_threshold - float(g["inlier_threshold"])) < 1e-9


@pytest.mark.gpu
def test_hip_path_reproduces_the_reference_at_c2_size(gpu_lib, c2_world):
    from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration
    world, corner, surf = c2_world
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, corner)
    m.setInputCloud(Map_buffer.SURF, surf)
    for path in SCENES:
        g = np.load(path)
        sc = scan_of(world, g)
        n = len(sc.xyzi)
        for groups in (True, False):  # a batch of one: the scan spread over 8 workgroups; and the one-workgroup form large batches take
            fe = Livox_laser(max_points=n, max_scans=1, piecewise_number=1)
            n_clouds = fe.extract_laser_features(sc.xyzi, float(g["stamp"]))  # first call of a fresh extractor, like the fixture
            assert n_clouds == int(g["n_petal_clouds"])
            f = fe.get_features(0.0, 1.0)
            assert np.array_equal(f["corner_idx"], g["corner_idx"]) and np.array_equal(f["surf_idx"], g["surf_idx"])  # integer artefact: exact
            reg = Point_cloud_registration(max_features=n)
            reg.set_debug(False, no_solver_groups=not groups)
            p = reg.params
            p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = int(g["icp_iters"]), int(g["ceres_iters"]), 0
            p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, float(g["max_final_cost"])
            p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
            p.maximum_allow_residual_block = n
            reg.m_pose_w_last = sc.pose_init.copy()
            reg.m_pose_w_curr = sc.pose_init.copy()
            ret = reg.find_out_incremental_transfrom(m, f["pc_corners"], f["pc_surface"])
            dt, dr = synth.pose_error(reg.m_pose_w_curr, g["pose_out"])
            assert ret == int(g["reg_ret"]), path
            assert dt < 1e-4 and dr < 1e-4  # north-star tolerance
            assert dt < 1e-7 and dr < 1e-7  # guard: the paths agree far inside it
            assert reg.report.n_blocks_last == int(g["n_blocks_last"])
            assert abs(reg.report.final_cost - float(g["final_cost"])) < 1e-7 * max(1.0, float(g["final_cost"]))
            assert abs(reg.report.inlier_threshold - float(g["inlier_threshold"])) < 1e-7
            fe.close(); reg.close()
    m.close()
