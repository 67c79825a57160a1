"""The mapping node's message surface in tools/ll_node.cpp (output side): the three cloud handlers with the
queue of complete triples (laser_mapping.hpp:89-120, 633-647, 749-780), the loop body of Laser_mapping::process with the
maximum_mapping_buffer drop rule (:1701-1735), and what process_new_scan publishes after a registration -- the registered full
cloud on /velodyne_cloud_registered (:1570-1575), nav_msgs/Odometry on /aft_mapped_to_init, nav_msgs/Path on /aft_mapped_path every
tenth frame, the camera_init -> aft_mapped transform (:1613-1653) -- plus /laser_cloud_surround from the full-cloud cell map (:1151-1200).

The checker is the reference's own text: tests/verbatim_build.py pulls those line ranges out of /root/reference and compiles them
against stub ROS message types into tests/cpp/_verbatim/verbatim_mapping_io (the binary travels to the GPU box like oracle/_ref).
Both programs are driven by the same io record ("LLIO0001": handler calls, process passes, publish calls) and log every message they
hand to a publisher; the logs must be equal line for line.

CPU tier: random records through ll_node --replay-io (no device).  GPU tier: a real sequence through ll_node --dump-io; the
reference text replayed on the dump must reproduce the run's own output lines, and the surround cloud must equal the one the CPU
oracle's cell map and VoxelGrid give for the dumped registered clouds."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ll_sequence import cloud_hash, write_sequence  # noqa: E402

from tests import verbatim_build  # noqa: E402
from tests.test_ll_node import build_node, sequence  # noqa: E402


def harness():
    exe = verbatim_build.build_mapping_io()
    if exe is None:
        pytest.skip("tests/cpp/_verbatim/verbatim_mapping_io not built (needs /root/reference at build time)")
    return exe


def write_io(path, events):
    with open(path, "wb") as f:
        f.write(b"LLIO0001")
        for ev in events:
            t = ev[0]
            f.write(t.encode())
            if t in "CSF":
                _, stamp, cloud = ev
                a = np.ascontiguousarray(cloud, np.float32).reshape(-1, 4)
                f.write(struct.pack("<di", float(stamp), a.shape[0]) + a.tobytes())
            elif t == "U":
                _, frame, stamp, pose, cloud = ev
                a = np.ascontiguousarray(cloud, np.float32).reshape(-1, 4)
                f.write(struct.pack("<id7di", int(frame), float(stamp), *[float(v) for v in pose], a.shape[0]) + a.tobytes())


def read_io(path):
    ev = []
    with open(path, "rb") as f:
        assert f.read(8) == b"LLIO0001"
        while True:
            t = f.read(1)
            if not t:
                break
            t = t.decode()
            if t in "CSF":
                stamp, n = struct.unpack("<di", f.read(12))
                ev.append((t, stamp, np.frombuffer(f.read(16 * n), np.float32).reshape(n, 4).copy()))
            elif t == "P":
                ev.append(("P",))
            elif t == "U":
                w = struct.unpack("<id7di", f.read(4 + 8 + 56 + 4))
                ev.append(("U", w[0], w[1], np.array(w[2:9]), np.frombuffer(f.read(16 * w[9]), np.float32).reshape(w[9], 4).copy()))
            else:
                raise AssertionError("unknown event " + t)
    return ev


def lines(path, kinds):
    return [ln.rstrip("\n") for ln in open(path) if ln.split(" ", 1)[0] in kinds]


def random_record(seed, n_triples, buffer_size):
    """triples whose three clouds arrive interleaved and out of order, process passes falling behind (so the drop rule fires), and
    publish calls on frame indices around the every-tenth-frame rule"""
    rng = np.random.default_rng(seed)
    stamps = 100.0 + np.cumsum(rng.uniform(0.05, 0.15, n_triples))
    arrivals = [(k, s) for s in stamps for k in "CSF"]
    # shuffle locally: a message may overtake up to five others
    for i in range(len(arrivals) - 1):
        j = min(len(arrivals) - 1, i + int(rng.integers(0, 6)))
        arrivals[i], arrivals[j] = arrivals[j], arrivals[i]
    events, frame = [], int(rng.integers(0, 7))
    for i, (k, s) in enumerate(arrivals):
        events.append((k, s, rng.normal(size=(int(rng.integers(0, 9)), 4)).astype(np.float32)))
        if rng.random() < 0.25:   # the mapping loop gets a turn now and then
            for _ in range(int(rng.integers(1, 3))):
                events.append(("P",))
                if rng.random() < 0.8:  # most scans register: process_new_scan reaches its publish section
                    frame += 1
                    q = rng.normal(size=4); q /= np.linalg.norm(q)
                    events.append(("U", frame, s + 0.001, np.r_[q, rng.normal(size=3) * 5], rng.normal(size=(int(rng.integers(1, 40)), 4)).astype(np.float32)))
    events += [("P",)] * (buffer_size + 2)
    return events


@pytest.mark.parametrize("seed,buffer_size", [(1, 2), (2, 3), (3, 5), (4, 50)])
def test_handlers_drop_rule_and_published_fields_equal_the_reference_text(tmp_path, seed, buffer_size):
    ref, exe = harness(), build_node()
    events = random_record(seed, 40, buffer_size)
    io = tmp_path / "io.bin"
    write_io(io, events)
    assert [e[0] for e in read_io(io)] == [e[0] for e in events]
    log_ref, log_node = tmp_path / "ref.txt", tmp_path / "node.txt"
    subprocess.check_call([ref, str(io), str(log_ref), str(buffer_size)], timeout=60)
    subprocess.check_call([exe, "--replay-io", str(io), "--out", str(log_node), "--param", f"mapping/maximum_mapping_buffer={buffer_size}"], timeout=60)
    kinds = {"TAKE", "DROP", "CLOUD", "ODOM", "PATH", "TF"}
    a, b = lines(log_ref, kinds), lines(log_node, kinds)
    assert a == b and len(a) > 50
    got = {k: sum(1 for ln in a if ln.startswith(k)) for k in kinds}
    assert got["TAKE"] > 5 and got["ODOM"] == got["TF"] == got["CLOUD"] > 5 and got["PATH"] >= 1
    if buffer_size <= 3:
        assert got["DROP"] > 0     # the slow mapper really dropped frames
    if buffer_size == 50:
        assert got["DROP"] == 0 and got["TAKE"] == 40   # nothing dropped: every complete triple is taken once, in completion order
    # spot-check the line formats against the reference's constants (frame ids, topic names, PCL's PointXYZI layout)
    odom = next(ln for ln in b if ln.startswith("ODOM")).split()
    assert odom[1] == "/aft_mapped_to_init" and odom[3] == "camera_init" and odom[4] == "aft_mapped"
    tf = next(ln for ln in b if ln.startswith("TF")).split()
    assert tf[2] == "camera_init" and tf[3] == "aft_mapped"
    cl = next(ln for ln in b if ln.startswith("CLOUD")).split()
    assert cl[1] == "/velodyne_cloud_registered" and cl[3] == "camera_init" and cl[6] == "32" and cl[7] == "4"
    path = [ln.split() for ln in b if ln.startswith("PATH")]
    assert [int(p[4]) for p in path] == list(range(1, len(path) + 1))   # one more pose on every publication


def test_path_is_published_on_every_tenth_frame_only(tmp_path):
    ref, exe = harness(), build_node()
    events = [("U", f, 10.0 + f, np.array([0, 0, 0, 1, f, 0, 0.0]), np.zeros((1, 4), np.float32)) for f in range(1, 35)]
    io = tmp_path / "io.bin"
    write_io(io, events)
    outs = []
    for cmd in ([ref, str(io), str(tmp_path / "r.txt")], [exe, "--replay-io", str(io), "--out", str(tmp_path / "n.txt")]):
        subprocess.check_call(cmd, timeout=60)
        outs.append(lines(cmd[-1], {"PATH", "ODOM"}))
    assert outs[0] == outs[1]
    path = [ln.split() for ln in outs[1] if ln.startswith("PATH")]
    assert [float(p[7]) for p in path] == [10.0, 20.0, 30.0] and sum(ln.startswith("ODOM") for ln in outs[1]) == 34


@pytest.mark.gpu
def test_real_run_publishes_what_the_reference_text_publishes(tmp_path, gpu_lib):
    from loam_livox_amd import synth
    from oracle import orc
    from oracle.orc_cellmap import CellMap
    ref, exe = harness(), build_node()
    world = synth.world_for_map_size(200_000)
    msgs = sequence(world, 14, 1, 4100)
    seq, log, io = tmp_path / "seq.bin", tmp_path / "log.txt", tmp_path / "io.bin"
    write_sequence(seq, msgs)
    prm = {"common/piecewise_number": 3, "common/odom_mode": 1, "common/maximum_input_lidar_pointcloud": 1, "feature_extraction/system_delay": 2,
           "feature_extraction/mapping_plane_resolution": 0.3, "feature_extraction/mapping_line_resolution": 0.2, "mapping/init_accumulate_frames": 2,
           "mapping/maximum_histroy_buffer": 20, "mapping/mapping_line_resolution": 0.1, "mapping/mapping_plane_resolution": 0.15,
           "mapping/max_allow_incre_R": 20.0, "mapping/max_allow_incre_T": 0.3, "optimization/icp_maximum_iteration": 10,
           "optimization/ceres_maximum_iteration": 20, "mapping/minimum_icp_R_diff": 1e-3, "mapping/minimum_icp_T_diff": 1e-4,
           "mapping/surround_pointcloud_resolution": 0.4, "ll/surround_every_frames": 12, "mapping/maximum_mapping_buffer": 5}
    cmd = [exe, "--in", str(seq), "--out", str(log), "--dump-io", str(io)]
    for k, v in prm.items():
        cmd += ["--param", f"{k}={v}"]
    subprocess.check_call(cmd, timeout=600)
    log_ref = tmp_path / "ref.txt"
    subprocess.check_call([ref, str(io), str(log_ref), "5"], timeout=60)
    kinds = {"TAKE", "DROP", "CLOUD", "ODOM", "PATH", "TF"}
    want = lines(log_ref, kinds)
    got = [ln for ln in lines(log, kinds) if "/laser_cloud_surround" not in ln]
    assert got == want and len(want) > 100
    # every registered frame: odometry = the pose of its REG line, stamp = the stamp of its triple; the three clouds of a piece were paired
    regs = [ln.split() for ln in open(log) if ln.startswith("REG ")]
    odom = [ln.split() for ln in got if ln.startswith("ODOM")]
    takes = [ln.split() for ln in got if ln.startswith("TAKE")]
    ok = [r for r in regs if int(r[2]) != 0]
    assert len(takes) == len(regs) >= 30 and len(odom) == len(ok) >= 25 and not any(ln.startswith("DROP") for ln in got)
    assert all(t[1] == t[2] == t[3] for t in takes)
    for r, o in zip(ok, odom):
        assert [float(v) for v in r[3:10]] == [float(v) for v in o[8:12]] + [float(v) for v in o[5:8]]
    assert sum(ln.startswith("PATH") for ln in got) == len([r for r in ok if (int(r[1]) + 1) % 10 == 0]) >= 2
    # /laser_cloud_surround: the full-cloud cell map around the pose, every cell voxel-filtered, then the union (laser_mapping.hpp:1151-1200)
    sur = [ln.split() for ln in open(log) if ln.startswith("CLOUD /laser_cloud_surround")]
    assert len(sur) >= 2
    pubs = [e for e in read_io(io) if e[0] == "U"]
    cm = CellMap(resolution=1.0)   # the CPU oracle's restatement of Points_cloud_map (oracle/orc_cellmap.py)
    k, last = 0, 0
    for e in pubs:
        cm.append(e[4])
        if e[1] - last >= 12:
            last = e[1]
            cells, _ = cm.query_filter(e[3], 1000.0, 360.0, 0.4, 0)
            out = orc.voxel_grid(cells, 0.4)[1]
            assert int(sur[k][4]) == len(out) and int(sur[k][8], 16) == cloud_hash(out) and float(sur[k][2]) == e[2] and sur[k][3] == "camera_init"
            k += 1
    assert k == len(sur)
