"""Committed outputs of the REFERENCE'S OWN CODE (tests/golden/ref_scene*.npz, written by tests/golden/gen_ref_golden.py
from oracle/_ref/libll_ref.so = /root/reference compiled in the build container).  They travel where /root/reference
does not:

  CPU tier  : the oracle reproduces them (bit-exact feature extraction; registration to 1e-9);
  GPU tier  : the HIP path reproduces them through the C ABI (index sets / labels bit-exact; pose within the
              north-star tolerance 1e-4 m / 1e-4 rad, with a 1e-7 guard).
"""
import glob
import os
import zlib

import numpy as np
import pytest

from loam_livox_amd import synth
from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_scene*.npz")))
BIT_FIELDS = ["pt_type", "pt_label", "time_stamp", "polar_direction", "polar_dis_sq2", "depth_sq2", "curvature", "sigma", "img2d"]
WINDOWS = ["all", "w03", "mid"]
_maps = {}


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def maps_for(g):
    key = (int(g["map_points"]), int(g["map_seed"]))
    if key not in _maps:
        _, corner, surf = synth.make_maps(key[0], seed=key[1])
        assert zlib.crc32(np.ascontiguousarray(corner).tobytes()) == int(g["corner_crc"]), "synthetic corner map changed"
        assert zlib.crc32(np.ascontiguousarray(surf).tobytes()) == int(g["surf_crc"]), "synthetic surface map changed"
        _maps[key] = (corner, surf)
    return _maps[key]


def reg_params(g):
    prm = orc.RegParams.defaults(icp_iters=int(g["icp_iters"]), ceres_iters=int(g["ceres_iters"]), deblur=int(g["deblur"]))
    prm.minimum_pt_time_stamp, prm.maximum_pt_time_stamp = float(g["t_min"]), float(g["t_max"])
    return prm


def test_fixtures_exist():
    assert len(SCENES) >= 3


@pytest.mark.parametrize("path", SCENES)
def test_oracle_reproduces_reference_outputs(path):
    g = np.load(path)
    tb = orc.FeTimebase()
    cur = tb.next(float(g["stamp"]))
    assert cur == float(g["current_time"])
    o = orc.fe_extract(g["xyzi"], cur)
    for f in BIT_FIELDS + ["view_angle", "polar_angle"]:
        assert np.array_equal(bits(getattr(o, f)), bits(g[f])), f
    s, first_idx, last_idx = orc.fe_split_scan(o)
    first = {}
    for i, p in enumerate(map(tuple, g["xyzi"][:, :3])):
        first.setdefault(p, i)
    assert s == int(g["n_petal_clouds"])
    assert [first[tuple(g["xyzi"][i, :3])] for i in first_idx] == g["petal_first"].tolist()
    assert [first[tuple(g["xyzi"][i, :3])] for i in last_idx] == g["petal_last"].tolist()
    for tag in WINDOWS:
        lo, hi = g[f"{tag}_window"]
        ci, si, fi = orc.fe_get_features(o, float(lo), float(hi))
        assert np.array_equal(bits(orc.feature_cloud(o, ci)), bits(g[f"{tag}_corners"]))
        assert np.array_equal(bits(orc.feature_cloud(o, si)), bits(g[f"{tag}_surface"]))
        assert np.array_equal(bits(orc.feature_cloud(o, fi)), bits(g[f"{tag}_full"]))
        assert [first[tuple(g["xyzi"][i, :3])] for i in ci] == g[f"{tag}_corner_idx"].tolist()
        assert [first[tuple(g["xyzi"][i, :3])] for i in si] == g[f"{tag}_surf_idx"].tolist()
    corner, surf = maps_for(g)
    ret, pc, pi, rep = orc.reg_solve(orc.KdTree(corner), orc.KdTree(surf), g["all_corners"], g["all_surface"], reg_params(g), g["pose_init"],
                                     g["pose_init"])
    dt, dr = synth.pose_error(pc, g["pose_out"])
    assert ret == int(g["reg_ret"]) and dt < 1e-9 and dr < 1e-9
    assert np.allclose(pi, g["pose_incre"], rtol=0, atol=1e-9)
    assert rep.n_blocks_last == int(g["n_blocks_last"])
    assert abs(rep.final_cost - float(g["final_cost"])) < 1e-9 and abs(rep.initial_cost - float(g["initial_cost"])) < 1e-9
    assert abs(rep.inlier_threshold - float(g["inlier_threshold"])) < 1e-9 and abs(rep.t_diff - float(g["t_diff"])) < 1e-9


@pytest.mark.parametrize("path", SCENES)
def test_device_math_on_host_reproduces_reference_labels(path):
    # the per-thread device arithmetic (ll_fe_core.h compiled by g++, tests/hostcheck) against the reference's labels
    from tests.hostcheck import hc
    g = np.load(path)
    p = orc.FeParams.node_defaults()
    h = hc.fe_points(g["xyzi"], float(g["current_time"]), hc.FeParams(*[getattr(p, f[0]) for f in p._fields_]))
    assert np.array_equal(h["type"], g["pt_type"]) and np.array_equal(h["label"], g["pt_label"])
    first = {}
    for i, q in enumerate(map(tuple, g["xyzi"][:, :3])):
        first.setdefault(q, i)
    for tag in WINDOWS:
        lo, hi = g[f"{tag}_window"]
        ci, si, fi = hc.select(h["type"], h["label"], h["depth2"], float(lo), float(hi))
        assert [first[tuple(g["xyzi"][i, :3])] for i in ci] == g[f"{tag}_corner_idx"].tolist()
        assert [first[tuple(g["xyzi"][i, :3])] for i in si] == g[f"{tag}_surf_idx"].tolist()
        assert len(fi) == len(g[f"{tag}_full"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", SCENES)
def test_hip_path_reproduces_reference_outputs(gpu_lib, path):
    from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration
    g = np.load(path)
    n = len(g["xyzi"])
    fe = Livox_laser(max_points=n, max_scans=1, piecewise_number=1)
    n_clouds = fe.extract_laser_features(g["xyzi"], float(g["stamp"]))  # first call of a fresh extractor, like the fixture
    info = fe.pts_info()
    assert np.array_equal(info["pt_type"], g["pt_type"]) and np.array_equal(info["pt_label"], g["pt_label"])
    for dev_name, f in (("depth_sq2", "depth_sq2"), ("polar_dis_sq2", "polar_dis_sq2"), ("curvature", "curvature"), ("time_stamp", "time_stamp")):
        assert np.array_equal(info[dev_name], g[f], equal_nan=True), f
    assert np.allclose(info["view_angle"], g["view_angle"], rtol=3e-6, atol=1e-5, equal_nan=True)  # acosf: device vs host libm
    assert n_clouds == int(g["n_petal_clouds"])
    sp = fe.splits()
    first = {}
    for i, q in enumerate(map(tuple, g["xyzi"][:, :3])):
        first.setdefault(q, i)
    assert [first[tuple(g["xyzi"][i, :3])] for i in sp["first_idx"]] == g["petal_first"].tolist()
    assert [first[tuple(g["xyzi"][i, :3])] for i in sp["last_idx"]] == g["petal_last"].tolist()
    feats = None
    for tag in WINDOWS:
        lo, hi = g[f"{tag}_window"]
        f = fe.get_features(float(lo), float(hi))
        assert np.array_equal(f["pc_corners"], g[f"{tag}_corners"], equal_nan=True)
        assert np.array_equal(f["pc_surface"], g[f"{tag}_surface"], equal_nan=True)
        assert [first[tuple(g["xyzi"][i, :3])] for i in f["corner_idx"]] == g[f"{tag}_corner_idx"].tolist()
        assert [first[tuple(g["xyzi"][i, :3])] for i in f["surf_idx"]] == g[f"{tag}_surf_idx"].tolist()
        assert len(f["full_idx"]) == len(g[f"{tag}_full"])
        if tag == "all":
            feats = f
    corner, surf = maps_for(g)
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, corner)
    m.setInputCloud(Map_buffer.SURF, surf)
    reg = Point_cloud_registration(max_features=n)
    p = reg.params
    p.if_motion_deblur = int(g["deblur"])
    p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = int(g["icp_iters"]), int(g["ceres_iters"]), 0
    p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 100.0
    p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
    p.minimum_pt_time_stamp, p.maximum_pt_time_stamp = float(g["t_min"]), float(g["t_max"])
    reg.m_pose_w_last = g["pose_init"].copy()
    reg.m_pose_w_curr = g["pose_init"].copy()
    ret = reg.find_out_incremental_transfrom(m, feats["pc_corners"], feats["pc_surface"])
    dt, dr = synth.pose_error(reg.m_pose_w_curr, g["pose_out"])
    assert ret == int(g["reg_ret"])
    assert dt < 1e-4 and dr < 1e-4  # north-star tolerance
    assert dt < 1e-7 and dr < 1e-7  # guard: the paths agree far inside it
    assert reg.report.n_blocks_last == int(g["n_blocks_last"])
    assert abs(reg.report.final_cost - float(g["final_cost"])) < 1e-7
    assert abs(reg.report.inlier_threshold - float(g["inlier_threshold"])) < 1e-7
    fe.close(); m.close(); reg.close()
