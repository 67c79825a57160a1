"""include/loam_livox_adapter.hpp: the C++ host-side mirror of Livox_laser / Point_cloud_registration.
CPU tier: it compiles and links against the C-ABI library.  GPU tier: the demo (written like the reference's
node code) reproduces what the Python mirror and the oracle give."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "adapter_demo.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_demo")


def build_demo():
    from loam_livox_amd import build
    lib = build.build()
    deps = [SRC, os.path.join(ROOT, "include", "loam_livox_adapter.hpp"), os.path.join(ROOT, "include", "loam_livox_hip.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++14", "-Wall", "-o", EXE, SRC, lib, "-Wl,-rpath," + os.path.dirname(lib),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_adapter_compiles_and_links():
    assert os.path.exists(build_demo())


@pytest.mark.gpu
@pytest.mark.parametrize("downsample", [False, True])
def test_adapter_demo_matches_oracle(tmp_path, small_world, scans, downsample):
    from loam_livox_amd import synth
    from oracle import orc
    exe = build_demo()
    sc = scans[0]
    paths = [str(tmp_path / n) for n in ("scan.bin", "corner.bin", "surf.bin", "pose.bin", "out.txt")]
    sc.xyzi.astype(np.float32).tofile(paths[0])
    np.c_[small_world["corner"], np.zeros(len(small_world["corner"]), np.float32)].astype(np.float32).tofile(paths[1])
    np.c_[small_world["surf"], np.zeros(len(small_world["surf"]), np.float32)].astype(np.float32).tofile(paths[2])
    sc.pose_init.astype(np.float64).tofile(paths[3])
    subprocess.check_call([exe] + paths + (["0.1", "0.4"] if downsample else []), timeout=120)
    lines = open(paths[4]).read().split("\n")
    n_clouds, n_c, n_s, n_f, reg_res, n_c_used, n_s_used = [int(v) for v in lines[0].split()]
    pose = np.array([float(v) for v in lines[1].split()])
    ps, pe = [np.float32(v) for v in lines[2].split()]
    # oracle: first call -> current_time = stamp + 1 (LFE:731)
    o = orc.fe_extract(sc.xyzi, 6.0)
    S, first, last = orc.fe_split_scan(o)
    ops, ope = orc.fe_piecewise(o, first, last, 1)
    assert n_clouds == S and ps == ops[0] and pe == ope[0]
    ci, si, fi = orc.fe_get_features(o, float(ops[0]), float(ope[0]))
    assert (n_c, n_s, n_f) == (len(ci), len(si), len(fi))
    prm = orc.RegParams.defaults(icp_iters=5, ceres_iters=20, force_all=0)
    fc, fs = orc.feature_cloud(o, ci), orc.feature_cloud(o, si)
    if downsample:
        fc, fs = orc.voxel_grid(fc, 0.1)[1], orc.voxel_grid(fs, 0.4)[1]
    assert (n_c_used, n_s_used) == (len(fc), len(fs))
    ret, opc, _, _ = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    dt, dr = synth.pose_error(pose, opc)
    assert reg_res == ret and dt < 1e-7 and dr < 1e-7
    # the history buffer through the adapter: one frame pushed with the registered pose, refreshed into a map
    from oracle.orc_mapping import History
    h = History(4, 0.1, 0.4)
    assert h.add(fc, fs, pose)
    mc, ms = h.refresh()
    assert [int(v) for v in lines[3].split()] == [len(mc), len(ms)]
    # ... and the cell maps (m_matching_mode == 1) fed by the same add()
    h = History(4, 0.1, 0.4)
    h.enable_cell_map(1.0, 5000)
    h.add(fc, fs, pose)
    cc, cs = h.refresh_cells(pose, (100.0, 100.0), 45.0, 1)
    assert [int(v) for v in lines[4].split()] == [len(cc), len(cs), len(h.cells[1].cells), h.cells[1].n_points()]
    assert len(cs) > 0.5 * len(ms)
    # ... and Points_cloud_map: cells, labels, key-frame image of the scan's full cloud
    from oracle.orc_cellmap import CellMap
    km = CellMap(1.0)
    touched_first = km.append(orc.feature_cloud(o, fi))
    f = km.features()
    n_cells, n_line, n_plane, self_sim, rz_line, rz_plane, n_vec_first, n_vec_second = [float(v) for v in lines[5].split()]
    # append_cloud( pts, &cell_vec ) through the adapter: every cell of the first cloud, then the cells with >= 3 points of the cloud
    km2 = CellMap(1.0)
    km2.append(orc.feature_cloud(o, fi))
    assert n_vec_first == len(touched_first) == len(km.cells) and n_vec_second == len(km2.append(orc.feature_cloud(o, fi))) > 20
    loose = int(np.sum(f["margin"] <= 1e-3))                  # cells on a decision boundary may fall either way
    assert n_cells == len(km.cells) > 100
    assert abs(n_line - np.sum(f["type"] == 1)) <= loose and abs(n_plane - np.sum(f["type"] == 2)) <= loose and n_plane > 20
    ko = km.keyframe_images(0.9)
    if loose == 0:
        assert abs(rz_line - ko["ratio_nonzero"][0]) < 1e-6 and abs(rz_plane - ko["ratio_nonzero"][1]) < 1e-6
    assert abs(self_sim - 1.0) < 1e-5
