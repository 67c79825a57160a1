"""Piece-wise / multi-lidar front-end of the feature-extraction node (LFX:241-389).
CPU tier: the oracle restatement's message logic.  GPU tier: the device mirror against it, clouds bit-identical."""
import numpy as np
import pytest

from loam_livox_amd import synth
from oracle.orc_feature_node import LaserFeature

N = 12000


def messages(world, n_msgs, lidars=3):
    """(scan, stamp, lidar index) in arrival order: the three heads of a Mid-100 publish in turn"""
    out = []
    for k in range(n_msgs):
        lidar = k % lidars
        sc = synth.make_moving_scan(world, 900 + k, N, inc_true=np.array([0, 0, 0, 1, 0, 0, 0.0]),
                                    yaw_offset=float(np.deg2rad([0.0, -38.4, 38.4][lidar])), pose_start=synth.make_scan(world, 1).pose_true)
        out.append((sc.xyzi, 10.0 + 0.05 * k, lidar))
    return out


def test_oracle_handler_message_logic(small_world):
    msgs = messages(small_world["world"], 7)
    node = LaserFeature(piecewise_number=3, para_system_delay=2, odom_mode=1, maximum_input_lidar_pointcloud=3,
                        mapping_plane_resolution=0.4, mapping_line_resolution=0.2)
    outs, sizes3 = [], None
    for k, m in enumerate(msgs):
        outs.append(node.handler(*m))
        if k == 3:   # per-lidar piece-0 cloud sizes as they were when message 3 published
            sizes3 = [(len(node.full[ii][0]), len(node.surf[ii][0])) for ii in range(3)]
    assert outs[0] == []                                   # dropped by the start-up delay (LFX:258-267)
    assert outs[1] == [] and outs[2] == []                 # lidars 1 and 2 never publish (LFX:348-351)
    assert len(outs[3]) == 3                               # lidar 0, mapping mode: one triple per piece
    full, surf, corn = outs[3][0]
    # piece 0 of message 3 merges lidar 0 (this message) with the latest clouds of lidars 1 and 2 (messages 1, 2)
    assert len(full) > sizes3[0][0] and len(full) == sum(s[0] for s in sizes3)
    assert 0 < len(surf) < sum(s[1] for s in sizes3)                          # voxel-filtered
    # time stamps of the merged clouds are on one time base and pieces are ordered in time
    t0 = [o[0][:, 3].mean() for o in outs[3]]
    assert t0[0] < t0[1] < t0[2]
    # odometry mode publishes the first piece only (LFX:385-388); deblur forces a single piece (LFX:305-309)
    node2 = LaserFeature(piecewise_number=3, para_system_delay=1, odom_mode=0, maximum_input_lidar_pointcloud=1)
    assert len(node2.handler(*msgs[0])) == 1
    node3 = LaserFeature(piecewise_number=3, if_motion_deblur=1, para_system_delay=1, odom_mode=1, maximum_input_lidar_pointcloud=1)
    out3 = node3.handler(*msgs[0])
    assert len(out3) == 1 and len(out3[0][0]) > len(node2.full[0][0])         # the single piece spans the whole scan


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(piecewise_number=3, odom_mode=1, maximum_input_lidar_pointcloud=3),
                                 dict(piecewise_number=3, odom_mode=0, maximum_input_lidar_pointcloud=1),
                                 dict(piecewise_number=2, if_motion_deblur=1, odom_mode=1, maximum_input_lidar_pointcloud=3)])
def test_device_handler_matches_oracle(gpu_lib, small_world, cfg):
    from loam_livox_amd.feature_node import Laser_feature
    kw = dict(para_system_delay=2, mapping_plane_resolution=0.4, mapping_line_resolution=0.2, **cfg)
    lidars = cfg["maximum_input_lidar_pointcloud"]
    msgs = messages(small_world["world"], 8, lidars)
    dev, ora = Laser_feature(max_points=N, **kw), LaserFeature(**kw)
    n_pub = 0
    for m in msgs:
        a, b = dev.laserCloudHandler(*m), ora.handler(*m)
        assert len(a) == len(b)
        for (fa, sa, ca), (fb, sb, cb) in zip(a, b):
            for x, y in ((fa, fb), (sa, sb), (ca, cb)):
                assert x.shape == y.shape and np.array_equal(x.view(np.uint32), np.ascontiguousarray(y, np.float32).view(np.uint32))
            n_pub += 1
    assert n_pub >= 2
    dev.close()
