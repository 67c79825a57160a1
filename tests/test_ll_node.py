"""tools/ll_node.cpp: the compiled, ROS-free host of the feature node's Livox handler (laser_feature_extractor.hpp:241-392)
and the mapping node's process_new_scan (laser_mapping.hpp:1316-1520) on top of include/loam_livox_adapter.hpp.
CPU tier: it builds and links against the C-ABI library and rejects bad input.  GPU tier: replaying a recorded synthetic
sequence, every published cloud and every registered pose equals what the Python mirrors (feature_node.py, mapping.py)
give for the same messages -- both sit on the same C ABI, so equality is bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ll_sequence import cloud_hash, parse_log, read_sequence, write_sequence  # noqa: E402

SRC = os.path.join(ROOT, "tools", "ll_node.cpp")
EXE = os.path.join(ROOT, "tools", "ll_node")


def build_node():
    from loam_livox_amd import build
    lib = build.build()
    deps = [SRC, os.path.join(ROOT, "include", "loam_livox_adapter.hpp"), os.path.join(ROOT, "include", "loam_livox_hip.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++14", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", EXE, SRC, lib,
                               "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_ll_node_builds_and_checks_its_input(tmp_path):
    exe = build_node()
    assert subprocess.run([exe], capture_output=True).returncode == 2
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a sequence")
    r = subprocess.run([exe, "--in", str(bad), "--out", str(tmp_path / "o.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "LLSEQ001" in r.stderr
    seq = tmp_path / "s.bin"
    msgs = [(0, 1.5, np.arange(12, dtype=np.float32).reshape(3, 4)), (2, 2.5, np.zeros((0, 4), np.float32))]
    write_sequence(seq, msgs)
    back = read_sequence(seq)
    assert [(m[0], m[1]) for m in back] == [(0, 1.5), (2, 2.5)] and np.array_equal(back[0][2], msgs[0][2]) and len(back[1][2]) == 0
    assert cloud_hash(msgs[0][2]) == sum(int(w) * (2 * i + 1) for i, w in enumerate(msgs[0][2].view(np.uint32).ravel())) % (1 << 64)


def sequence(world, n_frames, lidars, seed):
    """a sensor backing away from a room corner; with three lidars every frame is one message per head (Mid-100)"""
    from loam_livox_amd import synth
    rng = np.random.default_rng(seed)
    start = synth.sensor_pose_in_world(world, rng)
    step = np.r_[synth.quat_from_axis_angle(np.array([0.1, 0.2, 1.0]), np.deg2rad(0.2)), np.array([-0.02, 0.01, 0.0])]
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
    yaws = np.deg2rad([0.0, -38.4, 38.4])[:lidars]
    msgs, cur = [], start
    for k in range(n_frames):
        if k >= 3:
            cur = synth.pose_compose(cur, step)
        for li in reversed(range(lidars)):  # lidar 0 last: its message publishes the merged clouds of the frame
            sc = synth.make_moving_scan(world, seed + 10 * k + li, 24000, inc_true=ident, pose_start=cur, yaw_offset=float(yaws[li]), t_phase=0.13 * k)
            msgs.append((li, 0.1 * k, sc.xyzi))
    return msgs


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["piecewise3_single_lidar", "mid100_three_lidars"])
def test_ll_node_equals_python_mirrors(tmp_path, gpu_lib, case):
    from loam_livox_amd import synth
    from loam_livox_amd.feature_node import Laser_feature
    from loam_livox_amd.mapping import Laser_mapping
    exe = build_node()
    world = synth.world_for_map_size(200_000)
    if case == "piecewise3_single_lidar":
        lidars, frames, fe_kw = 1, 6, dict(piecewise_number=3, odom_mode=1)
    else:
        lidars, frames, fe_kw = 3, 5, dict(piecewise_number=1, odom_mode=0)
    msgs = sequence(world, frames, lidars, 4100 if lidars == 1 else 5200)
    seq, log = tmp_path / "seq.bin", tmp_path / "log.txt"
    write_sequence(seq, msgs)
    prm = {"common/piecewise_number": fe_kw["piecewise_number"], "common/odom_mode": fe_kw["odom_mode"], "common/maximum_input_lidar_pointcloud": lidars,
           "feature_extraction/system_delay": 2, "feature_extraction/mapping_plane_resolution": 0.3, "feature_extraction/mapping_line_resolution": 0.2,
           "mapping/init_accumulate_frames": 2, "mapping/maximum_histroy_buffer": 20, "mapping/mapping_line_resolution": 0.1,
           "mapping/mapping_plane_resolution": 0.15, "mapping/max_allow_incre_R": 20.0, "mapping/max_allow_incre_T": 0.3,
           "optimization/icp_maximum_iteration": 10, "optimization/ceres_maximum_iteration": 20, "mapping/minimum_icp_R_diff": 1e-3,
           "mapping/minimum_icp_T_diff": 1e-4}
    cmd = [exe, "--in", str(seq), "--out", str(log)]
    for k, v in prm.items():
        cmd += ["--param", f"{k}={v}"]
    subprocess.check_call(cmd, timeout=300)
    pubs, regs = parse_log(log)

    fn = Laser_feature(max_points=24000, if_motion_deblur=0, maximum_input_lidar_pointcloud=lidars, mapping_plane_resolution=0.3,
                       mapping_line_resolution=0.2, para_system_delay=2, **fe_kw)
    lm = Laser_mapping(scan_points=3 * 24000, maximum_history_size=20, line_res=0.1, plane_res=0.15, init_accumulate_frames=2, icp_max_iterations=10,
                       ceres_max_iterations=20, max_allow_incre_R=20.0, max_allow_incre_T=0.3, minimum_icp_R_diff=1e-3, minimum_icp_T_diff=1e-4)
    want_pub, want_reg = [], []
    for lidar, stamp, xyzi in msgs:
        for full, surf, corn in fn.laserCloudHandler(xyzi, stamp, lidar):
            want_pub.append((len(full), len(surf), len(corn), cloud_hash(full), cloud_hash(surf), cloud_hash(corn)))
            res = lm.process_clouds(full, surf, corn)
            want_reg.append((res, lm.pose.copy(), lm.stack_sizes, tuple(int(v) for v in lm.map_sizes), int(lm.last_report.icp_iterations)))
    fn.close(); lm.close()
    assert len(pubs) == len(want_pub) > 0 and pubs == want_pub
    assert len(regs) == len(want_reg)
    for got, (res, pose, stacks, maps, its) in zip(regs, want_reg):
        assert got["res"] == res and np.array_equal(got["pose"], pose)
        assert (got["n_corner"], got["n_surf"]) == stacks and (got["map_corner"], got["map_surf"]) == maps and got["icp_iterations"] == its
    assert sum(r["res"] for r in regs) >= len(regs) - 1 and regs[-1]["map_surf"] > 1000  # the sequence really registers and the map grows
    moved = np.linalg.norm(regs[-1]["pose"][4:])
    assert 0.01 < moved < 1.0
