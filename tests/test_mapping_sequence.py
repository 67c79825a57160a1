"""History match buffer + the per-frame mapping loop (BASELINE config C4's unit).
CPU tier: the oracle restatement (oracle/orc_mapping.py) tracks a synthetic trajectory.  GPU tier: the device history
and the device mapping loop against it -- match-buffer clouds bit-identical, poses within 1e-7."""
import numpy as np
import pytest

from loam_livox_amd import synth
from oracle import orc
from oracle.orc_mapping import History, LaserMapping

IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
N_PTS = 12000


def pose_inv(p):
    R = synth.quat_to_mat(p[:4])
    q = np.array([-p[0], -p[1], -p[2], p[3]])
    return np.r_[q, -(R.T @ p[4:])]


def make_sequence(world, n_frames=9, n_static=3):
    """frames 0..n_static-1 from the start pose (they seed the map, PCR:199), then a slow drift"""
    rng = np.random.default_rng(77)
    start = synth.sensor_pose_in_world(world, rng)
    step = np.r_[synth.quat_from_axis_angle(np.array([0.1, 0.2, 1.0]), np.deg2rad(0.4)), np.array([0.04, 0.015, 0.0])]
    poses, scans, cur = [], [], start
    for k in range(n_frames):
        if k >= n_static:
            cur = synth.pose_compose(cur, step)
        sc = synth.make_moving_scan(world, 500 + k, N_PTS, inc_true=IDENT, pose_start=cur, t_phase=0.13 * k)
        scans.append(sc.xyzi)
        poses.append(synth.pose_compose(pose_inv(start), cur))  # relative to frame 0 = the map frame
    return scans, poses


MAP_ARGS = dict(maximum_history_size=5, init_accumulate_frames=2, line_res=0.1, plane_res=0.15, icp_max_iterations=6, ceres_max_iterations=20,
                max_allow_incre_R=20.0, max_allow_incre_T=0.3)


@pytest.fixture(scope="module")
def sequence(small_world):
    return make_sequence(small_world["world"])


@pytest.fixture(scope="module")
def oracle_run(sequence):
    scans, _ = sequence
    om = LaserMapping(**MAP_ARGS)
    out = []
    for xyzi in scans:
        r = om.process_new_scan(xyzi)
        out.append((r, om.pose.copy(), [m.copy() for m in om.maps], om.report.gated, om.report.n_blocks_last))
    return out


def test_oracle_mapping_tracks_the_trajectory(sequence, oracle_run):
    _, truth = sequence
    gated = [o[3] for o in oracle_run]
    assert gated[:3] == [1, 1, 1] and not any(gated[3:])      # frame index must exceed init_accumulate_frames (PCR:199)
    for k, (r, pose, maps, _, nb) in enumerate(oracle_run):
        assert r == 1
        dt, dr = synth.pose_error(pose, truth[k])
        assert dt < 0.03 and dr < 0.006                        # odometry drift over a few frames stays at the cm level
        if k >= 3:
            assert nb > 100
    # FIFO of 5 frames: the buffer grows while the history fills, then follows the sensor
    sizes = [len(o[2][1]) for o in oracle_run]
    assert sizes[0] > 100 and sizes[2] > sizes[0] and min(sizes[3:]) > 0.5 * sizes[2]


def test_history_fifo_and_add_rule():
    rng = np.random.default_rng(4)
    h = History(maximum_history_size=3, line_res=0.2, plane_res=0.5)
    pose = IDENT.copy()
    for k in range(5):
        c = rng.uniform(-3, 3, (200, 4)).astype(np.float32)
        s = rng.uniform(-3, 3, (800, 4)).astype(np.float32)
        assert h.add(c, s, pose, t_step=0.5, angle_step=0.1) == (k < 3)   # full history + no motion -> rejected (LM:1446-1448)
    pose[4] = 0.6
    assert h.add(c, s, pose, t_step=0.5, angle_step=0.1) and len(h.frames[0]) == 3
    pose2 = pose.copy(); pose2[:4] = synth.quat_from_axis_angle(np.array([0, 0, 1.0]), np.deg2rad(7.0))
    assert h.add(c, s, pose2, t_step=0.5, angle_step=0.1)                  # 7 deg > 0.1 * 57.3 deg
    mc, ms = h.refresh()
    assert 0 < len(mc) <= 600 and 0 < len(ms) <= 2400
    # LM:1439-1451 gate on (and record) the node's pose BEFORE the registration, not the pose the clouds are moved with
    far = pose2.copy(); far[4] += 5.0
    assert not h.add(c, s, far, t_step=0.5, angle_step=0.1, gate_pose=pose2)      # registered pose far away, node pose unchanged
    assert h.add(c, s, pose2, t_step=0.5, angle_step=0.1, gate_pose=far) and np.array_equal(h.last_t, far[4:])


# ------------------------------------------------------------------------------------------------------ GPU tier
def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.gpu
def test_device_history_bit_exact(gpu_lib):
    from loam_livox_amd.api import History_buffer, Map_buffer
    rng = np.random.default_rng(4)
    dev, ora = History_buffer(3, 4000, 0.2, 0.5), History(3, 0.2, 0.5)
    m = Map_buffer()
    pose = IDENT.copy()
    for k in range(7):
        c = rng.uniform(-6, 6, (300 + 50 * k, 4)).astype(np.float32)
        s = rng.uniform(-6, 6, (3000 + 100 * k, 4)).astype(np.float32)
        if k == 2:
            c = np.zeros((0, 4), np.float32)                   # a frame without corner features
        if k in (4, 5):                                        # motion below / above the add thresholds
            pose = synth.pose_compose(pose, np.r_[synth.quat_from_axis_angle(np.array([0, 0, 1.0]), np.deg2rad(1.0 if k == 4 else 8.0)), [0.1, 0, 0]])
        if k == 6:  # the add-frame rule reads the pose before the registration (ll_history_set_gate_pose, one-shot)
            gate = synth.pose_compose(pose, np.r_[0, 0, 0, 1, 0.7, 0, 0])
            dev.set_gate_pose(gate)
            assert dev.add(c, s, pose, 0.5, 0.1) == ora.add(c, s, pose, 0.5, 0.1, gate_pose=gate) == True
        else:
            assert dev.add(c, s, pose, 0.5, 0.1) == ora.add(c, s, pose, 0.5, 0.1)
        assert len(dev) == len(ora.frames[0])
        nc, ns = dev.refresh(m)
        mc, ms = ora.refresh()
        assert (nc, ns) == (len(mc), len(ms))
        assert np.array_equal(bits(dev.map_cloud(0)), bits(mc)) and np.array_equal(bits(dev.map_cloud(1)), bits(ms))
    # the refreshed device grids answer k-NN queries like a k-d tree over the oracle's buffer
    q = rng.uniform(-6, 6, (500, 3)).astype(np.float32)
    idx, d2 = m.nearestKSearch(1, q, 50.0)
    oi, od = orc.KdTree(ms).knn(q, 5)
    assert np.array_equal(np.where(od < 50.0, oi, -1), idx)
    dev.close(); m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("downsample", [1, 0])
def test_device_mapping_loop_matches_oracle(gpu_lib, sequence, oracle_run, downsample):
    from loam_livox_amd.mapping import Laser_mapping
    scans, truth = sequence
    if downsample:
        ref = oracle_run
    else:
        om = LaserMapping(input_downsample_mode=0, **MAP_ARGS)
        ref = []
        for xyzi in scans[:6]:
            r = om.process_new_scan(xyzi)
            ref.append((r, om.pose.copy(), [m.copy() for m in om.maps], om.report.gated, om.report.n_blocks_last))
    lm = Laser_mapping(scan_points=N_PTS, input_downsample_mode=downsample, **MAP_ARGS)
    for k, o in enumerate(ref):
        r = lm.process_new_scan(scans[k])
        dt, dr = synth.pose_error(lm.pose, o[1])
        assert r == o[0] and dt < 1e-7 and dr < 1e-7
        assert lm.map_sizes == (len(o[2][0]), len(o[2][1]))
        assert lm.last_report.n_blocks_last == o[4]
        if dt == 0.0 and dr == 0.0:   # identical poses -> identical transforms -> bit-identical match buffer
            assert np.array_equal(bits(lm.history.map_cloud(1)), bits(o[2][1]))
    dt, dr = synth.pose_error(lm.pose, truth[len(ref) - 1])
    assert dt < 0.03 and dr < 0.006
    lm.close()


@pytest.mark.gpu
def test_cell_maps_fed_by_the_service_thread_equal_the_inline_ones(gpu_lib, sequence):
    """matching mode 0 with keep_cell_maps: the frames reach m_pt_cell_map_corners / _planes (laser_mapping.hpp:1492-1493) through the
    history handle's service thread, beside the loop; the maps -- which also outgrow their first allocation on the way -- are bit for bit
    the ones the inline path builds, and the poses do not notice"""
    from loam_livox_amd.mapping import Laser_mapping
    scans, _ = sequence
    dumps, poses = [], []
    for async_ in (True, False):
        lm = Laser_mapping(scan_points=N_PTS, keep_cell_maps=True, cell_map_max_points=N_PTS, **MAP_ARGS)
        if not async_:
            lm.history.set_cell_map_async(False)
        for xyzi in scans:
            lm.process_new_scan(xyzi)
        lm.sync()
        dumps.append([lm.history.cell_map(k).dump() for k in (0, 1)])
        poses.append(lm.pose.copy())
        assert lm.history.cell_map(1).stats()[1] > N_PTS // 4
        lm.close()
    assert np.array_equal(poses[0], poses[1])
    for k in (0, 1):
        assert all(np.array_equal(a, b) for a, b in zip(dumps[0][k], dumps[1][k]))
    assert len(dumps[0][1][0]) > 1000
