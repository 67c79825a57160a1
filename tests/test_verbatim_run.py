"""Link-and-run form of the drop-in claim (tests/test_adapter_verbatim.py checks that the excerpts COMPILE): the verbatim
node code of laser_mapping.hpp:1266-1297, 1405-1445, 1494-1512 -- init_pointcloud_registration and the registration call /
"Add new frame" / pose read-back of process_new_scan -- is executed twice from one translation unit (tests/verbatim_build.py):
once with the adapter class on the GPU, once with the reference's own Point_cloud_registration (compiled from /root/reference
against oracle/ref_stubs) on the CPU, on the same inputs.  The binaries are built where /root/reference exists
(__graft_entry__.build()) and travel with the tree.

Also the adapter's map identity: the node allocates NEW clouds for every scan and deep-copies its match
buffer into them (laser_mapping.hpp:1391-1392, 1396-1401); equal contents in fresh objects must not be uploaded again, a
single moved point must be."""
import os
import subprocess

import numpy as np
import pytest

from tests import verbatim_build


def test_verbatim_binaries_build_where_the_reference_is():
    if not verbatim_build.have_reference():
        pytest.skip("/root/reference absent")
    a, b = verbatim_build.build()
    assert os.path.exists(a) and os.path.exists(b)
    a, b = verbatim_build.build_sequence()
    assert os.path.exists(a) and os.path.exists(b)
    a, b = verbatim_build.build_feature()
    assert os.path.exists(a) and os.path.exists(b)


def _inputs(tmp_path, small_world, sc):
    from oracle import orc
    fe = orc.fe_extract(sc.xyzi, 1.0)
    ci, si, fi = orc.fe_get_features(fe, 0.0, 1.0)
    fc, fs = orc.feature_cloud(fe, ci), orc.feature_cloud(fe, si)
    fc, fs = orc.voxel_grid(fc, 0.1)[1], orc.voxel_grid(fs, 0.4)[1]  # the node's input down-sampling (laser_mapping.hpp:1367-1373)
    paths = [str(tmp_path / n) for n in ("mc.bin", "ms.bin", "fc.bin", "fs.bin", "pose.bin")]
    np.c_[small_world["corner"], np.zeros(len(small_world["corner"]), np.float32)].astype(np.float32).tofile(paths[0])
    np.c_[small_world["surf"], np.zeros(len(small_world["surf"]), np.float32)].astype(np.float32).tofile(paths[1])
    fc.astype(np.float32).tofile(paths[2])
    fs.astype(np.float32).tofile(paths[3])
    sc.pose_init.astype(np.float64).tofile(paths[4])
    return paths, fc, fs


def _run(exe, paths, out):
    subprocess.check_call([exe] + paths + [out], timeout=600)
    rows = [l.split() for l in open(out).read().strip().split("\n")]
    return [(int(r[0]), int(r[1]), int(r[2]), np.array([float(v) for v in r[3:10]]), np.array([float(v) for v in r[10:13]])) for r in rows]


def test_reference_class_through_the_excerpt_equals_the_oracle(tmp_path, small_world, scans):
    """CPU tier: the reference's own class, driven by the reference's own call site, lands where the oracle does"""
    from loam_livox_amd import synth
    from oracle import orc
    _, exe_b = verbatim_build.build()
    if not exe_b:
        pytest.skip("verbatim binaries not built (no /root/reference here and none travelled)")
    sc = scans[0]
    paths, fc, fs = _inputs(tmp_path, small_world, sc)
    ref = _run(exe_b, paths, str(tmp_path / "ref.txt"))
    prm = orc.RegParams.defaults(icp_iters=20, ceres_iters=20, force_all=0)
    prm.max_final_cost = 1000.0
    ret, opc, _, _ = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    dt, dr = synth.pose_error(ref[0][3], opc)
    assert ref[0][0] == ret == 1 and dt < 1e-9 and dr < 1e-9


@pytest.mark.gpu
def test_excerpt_on_the_adapter_matches_the_reference_class_and_uploads_once(tmp_path, small_world, scans):
    from loam_livox_amd import synth
    exe_a, exe_b = verbatim_build.build()
    if not exe_a or not exe_b:
        pytest.skip("verbatim binaries not built (no /root/reference here and none travelled)")
    sc = scans[0]
    paths, _, _ = _inputs(tmp_path, small_world, sc)
    ref = _run(exe_b, paths, str(tmp_path / "ref.txt"))
    ada = _run(exe_a, paths, str(tmp_path / "ada.txt"))
    for k in range(3):  # same return value, pose <= 1e-7 (north_star: 1e-4), same transformed full cloud
        dt, dr = synth.pose_error(ada[k][3], ref[k][3])
        assert ada[k][0] == ref[k][0] == 1 and dt < 1e-7 and dr < 1e-7
        assert np.allclose(ada[k][4], ref[k][4], atol=2e-6)
    # pass 1 uploads both clouds; pass 2 (fresh objects, equal contents) uploads nothing; pass 3 (one surface point moved) uploads
    # the surface cloud again and only that
    assert (ada[0][1], ada[0][2]) == (1, 1)
    assert (ada[1][1], ada[1][2]) == (1, 1)
    assert (ada[2][1], ada[2][2]) == (1, 2)


# ---- the mapping loop (process_new_scan from its first line, history add rule, update_buff_for_matching's history branch) --------
def _sequence_inputs(tmp_path, small_world):
    from oracle import orc
    from oracle.orc_mapping import LaserMapping
    from tests.test_mapping_sequence import MAP_ARGS, make_sequence
    scans, _ = make_sequence(small_world["world"])
    om = LaserMapping(**MAP_ARGS)
    expect = []
    path = str(tmp_path / "frames.bin")
    with open(path, "wb") as f:
        np.array([len(scans)], np.int32).tofile(f)
        for xyzi in scans:
            o = orc.fe_extract(xyzi, 1.0)
            ci, si, fi = orc.fe_get_features(o, 0.0, 1.0)
            clouds = [orc.feature_cloud(o, ix).astype(np.float32) for ix in (ci, si, fi)]  # what /pc2_corners, /pc2_surface, /pc2_full carry
            np.array([len(c) for c in clouds], np.int32).tofile(f)
            for c in clouds:
                c.tofile(f)
            r = om.process_new_scan(xyzi)
            expect.append((r, om.pose.copy(), len(om.hist.frames[0]), len(om.hist.frames[1]), len(om.maps[0]), len(om.maps[1]),
                           [float((m[:, 0].astype(np.float64) + 2.0 * m[:, 1] + 3.0 * m[:, 2]).sum()) if len(m) else 0.0 for m in om.maps]))
    args = [path, None, str(MAP_ARGS["line_res"]), str(MAP_ARGS["plane_res"]), str(MAP_ARGS["init_accumulate_frames"]),
            str(MAP_ARGS["maximum_history_size"]), str(MAP_ARGS["icp_max_iterations"]), "100.0"]
    return args, expect


def _run_sequence(exe, args, out):
    a = list(args)
    a[1] = out
    subprocess.check_call([exe] + a, timeout=900, stdout=subprocess.DEVNULL)
    rows = [l.split() for l in open(out).read().strip().split("\n")]
    return [([int(v) for v in r[:5]], np.array([float(v) for v in r[5:12]]), [float(r[12]), float(r[13])]) for r in rows]


def test_reference_mapping_loop_text_equals_the_oracle_restatement(tmp_path, small_world):
    """CPU tier: laser_mapping.hpp's own process_new_scan / history rule / match-buffer refresh lines (with the reference's own
    registrar and the stand-in pcl::VoxelGrid) against oracle/orc_mapping.py over a nine-frame sequence: every return value,
    history and match-buffer size equal, poses to 1e-9, match-buffer contents by checksum -- the pin orc_mapping.py did not have"""
    from loam_livox_amd import synth
    _, exe_b = verbatim_build.build_sequence()
    if not exe_b:
        pytest.skip("verbatim binaries not built (no /root/reference here and none travelled)")
    args, expect = _sequence_inputs(tmp_path, small_world)
    ref = _run_sequence(exe_b, args, str(tmp_path / "ref.txt"))
    assert len(ref) == len(expect) == 9
    for (ints, pose, sums), e in zip(ref, expect):
        assert ints == [e[0], e[2], e[3], e[4], e[5]]
        dt, dr = synth.pose_error(pose, e[1])
        assert dt < 1e-9 and dr < 1e-9
        assert np.allclose(sums, e[6], rtol=1e-9, atol=1e-5)
    assert ref[-1][0][1] == 5 and ref[3][0][3] > 300  # the FIFO filled up, and the buffer holds a map by the first registered frame


@pytest.mark.gpu
def test_mapping_loop_text_on_the_adapter_matches_the_reference_classes(tmp_path, small_world):
    """the same translation unit built with loam_livox_hip::Point_cloud_registration and loam_livox_hip::VoxelGrid (device), frame
    after frame against the build with the reference's registrar: identical history / match-buffer sizes, poses < 1e-7"""
    from loam_livox_amd import synth
    exe_a, exe_b = verbatim_build.build_sequence()
    if not exe_a or not exe_b:
        pytest.skip("verbatim binaries not built (no /root/reference here and none travelled)")
    args, _ = _sequence_inputs(tmp_path, small_world)
    ref = _run_sequence(exe_b, args, str(tmp_path / "ref.txt"))
    ada = _run_sequence(exe_a, args, str(tmp_path / "ada.txt"))
    for (ai, ap, asum), (ri, rp, rsum) in zip(ada, ref):
        assert ai == ri
        dt, dr = synth.pose_error(ap, rp)
        assert dt < 1e-7 and dr < 1e-7
        assert np.allclose(asum, rsum, rtol=1e-6, atol=1e-3)


# ---- the feature node (Laser_feature::laserCloudHandler, laser_feature_extractor.hpp:256-392 as one verbatim range) ----------------
FEATURE_CFGS = [dict(piecewise_number=3, odom_mode=1, maximum_input_lidar_pointcloud=3, if_motion_deblur=0),
                dict(piecewise_number=3, odom_mode=0, maximum_input_lidar_pointcloud=1, if_motion_deblur=0),
                dict(piecewise_number=2, odom_mode=1, maximum_input_lidar_pointcloud=3, if_motion_deblur=1)]


def _feature_inputs(tmp_path, small_world, cfg):
    from tests.test_feature_node import messages
    msgs = messages(small_world["world"], 8, cfg["maximum_input_lidar_pointcloud"])
    path = str(tmp_path / "msgs.bin")
    with open(path, "wb") as f:
        np.array([len(msgs)], np.int32).tofile(f)
        for xyzi, stamp, lidar in msgs:
            np.array([len(xyzi), lidar], np.int32).tofile(f)
            np.array([stamp], np.float64).tofile(f)
            np.ascontiguousarray(xyzi, np.float32).tofile(f)
    args = [path, None, str(cfg["piecewise_number"]), str(cfg["odom_mode"]), str(cfg["maximum_input_lidar_pointcloud"]), "2", "0.4", "0.2",
            str(cfg["if_motion_deblur"])]
    return msgs, args


def _run_feature(exe, args, out):
    a = list(args)
    a[1] = out
    subprocess.check_call([exe] + a, timeout=900, stdout=subprocess.DEVNULL)
    raw = np.fromfile(out, np.int32)
    pos, res = 0, []
    while pos < len(raw):
        n_pub = int(raw[pos]); pos += 1
        pubs = []
        for _ in range(n_pub):
            trio = []
            for _ in range(3):
                n = int(raw[pos]); pos += 1
                trio.append(raw[pos:pos + 4 * n].view(np.float32).reshape(n, 4).copy()); pos += 4 * n
            pubs.append(tuple(trio))
        res.append(pubs)
    return res


def _same_clouds(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("cfg", FEATURE_CFGS)
def test_reference_feature_node_text_equals_the_oracle_restatement(tmp_path, small_world, cfg):
    """CPU tier: laserCloudHandler's own lines (start-up delay, extract, piece windows, per-lidar stores, merge, voxel filters, publish
    order, odometry-mode break) with the reference's own Livox_laser against oracle/orc_feature_node.py: every published cloud
    bit-identical, message by message -- the pin orc_feature_node.py did not have"""
    from oracle import orc
    from oracle.orc_feature_node import LaserFeature
    _, exe_b = verbatim_build.build_feature()
    if not exe_b:
        pytest.skip("verbatim binaries not built (no /root/reference here and none travelled)")
    msgs, args = _feature_inputs(tmp_path, small_world, cfg)
    ref = _run_feature(exe_b, args, str(tmp_path / "ref.bin"))
    ora = LaserFeature(para_system_delay=2, mapping_plane_resolution=0.4, mapping_line_resolution=0.2, params=orc.FeParams.node_defaults(), **cfg)
    n_pub = 0
    for m, pubs in zip(msgs, ref):
        exp = ora.handler(*m)
        assert len(pubs) == len(exp)
        for (fa, sa, ca), (fb, sb, cb) in zip(pubs, exp):
            assert _same_clouds(fa, fb) and _same_clouds(sa, sb) and _same_clouds(ca, cb)
            n_pub += 1
    assert len(ref) == len(msgs) and n_pub >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", FEATURE_CFGS)
def test_feature_node_text_on_the_adapter_matches_the_reference_classes(tmp_path, small_world, cfg):
    exe_a, exe_b = verbatim_build.build_feature()
    if not exe_a or not exe_b:
        pytest.skip("verbatim binaries not built (no /root/reference here and none travelled)")
    msgs, args = _feature_inputs(tmp_path, small_world, cfg)
    ref = _run_feature(exe_b, args, str(tmp_path / "ref.bin"))
    ada = _run_feature(exe_a, args, str(tmp_path / "ada.bin"))
    assert len(ada) == len(ref) == len(msgs)
    for pa, pr in zip(ada, ref):
        assert len(pa) == len(pr)
        for ta, tr in zip(pa, pr):
            for x, y in zip(ta, tr):
                assert _same_clouds(x, y)
