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
