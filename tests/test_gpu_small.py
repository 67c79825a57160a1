"""-m gpu: the solver for small scans (ll_reg_small_kernels.hip) -- the reference's operating point: voxel-filtered feature clouds
(laser_mapping.hpp:1367-1373, leaf 0.1 / 0.4 m), a few hundred residual blocks, optionally capped at maximum_residual_blocks = 200
(config/performance_precision.yaml:23).  One wavefront per scan (batches of >= 512) and four wavefronts per scan (below), both held to the
oracle's point_cloud_registration.hpp:163-583 restatement -- pose, block / LM / ICP counts, inlier threshold -- and to the 512-thread
solver the same scans took until round 4."""
import numpy as np
import pytest

from loam_livox_amd import synth
from loam_livox_amd.api import Map_buffer, Point_cloud_registration
from oracle import orc
from tests.conftest import oracle_features

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev_map(gpu_lib, small_world):
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, small_world["corner"])
    m.setInputCloud(Map_buffer.SURF, small_world["surf"])
    yield m
    m.close()


@pytest.fixture(scope="module")
def filtered(scans):
    """the registrar's input in input_downsample_mode: VoxelGrid 0.1 m on the corner cloud, 0.4 m on the surface cloud"""
    out = []
    for sc in scans:
        _, _, _, _, fc, fs = oracle_features(sc)
        out.append((orc.voxel_grid(fc, 0.1)[1], orc.voxel_grid(fs, 0.4)[1]))
    return out


def set_params(reg, icp=10, ceres=20, force=1):
    p = reg.params
    p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = icp, ceres, force
    p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 100.0
    p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
    return p


def check_against(rep, orep, pose, opose, res, ret, tol=1e-7):
    dt, dr = synth.pose_error(pose, opose)
    assert res == ret and dt <= 1e-4 and dr <= 1e-4  # north-star tolerance
    assert dt < tol and dr < tol, (dt, dr)
    assert rep.n_blocks_last == orep.n_blocks_last and rep.lm_iterations_total == orep.lm_iterations_total
    assert rep.icp_iterations == orep.icp_iterations
    assert rep.corner_avail == orep.corner_avail and rep.surf_avail == orep.surf_avail
    assert np.isclose(rep.inlier_threshold, orep.inlier_threshold, rtol=1e-9)
    assert np.isclose(rep.final_cost, orep.final_cost, rtol=1e-7)


@pytest.mark.parametrize("waves", [1, 2, 4])
@pytest.mark.parametrize("force", [0, 1])
def test_small_solver_matches_oracle_and_the_512_thread_solver(dev_map, small_world, scans, filtered, waves, force):
    B = len(scans)
    init = np.stack([s.pose_init for s in scans])
    outs = {}
    for kw in ({"small_solver_waves": waves}, {"no_small_solver": True}):
        reg = Point_cloud_registration(max_scans=B, max_features=2048)
        reg.set_debug(False, **kw)
        set_params(reg, 10, 20, force)
        outs[tuple(kw)] = reg.solve_batch(dev_map, [f[0] for f in filtered], [f[1] for f in filtered], init, init)
        reg.close()
    res, pc, pi, reps = outs[("small_solver_waves",)]
    res5, pc5, _, reps5 = outs[("no_small_solver",)]
    prm = orc.RegParams.defaults(icp_iters=10, ceres_iters=20, force_all=force)
    for b, sc in enumerate(scans):
        fc, fs = filtered[b]
        assert len(fc) + len(fs) <= 1024
        ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
        check_against(reps[b], orep, pc[b], opc, res[b], ret)
        dt, dr = synth.pose_error(pc[b], pc5[b])
        assert dt < 1e-9 and dr < 1e-9 and res[b] == res5[b]
        assert (reps[b].n_blocks_last, reps[b].lm_iterations_total, reps[b].icp_iterations) == (reps5[b].n_blocks_last, reps5[b].lm_iterations_total, reps5[b].icp_iterations)


@pytest.mark.parametrize("waves", [1, 2, 4])
@pytest.mark.parametrize("sizes", [(40, 150), (150, 330), (230, 760), (0, 300), (200, 0)])
def test_every_size_class(dev_map, small_world, scans, waves, sizes):
    """the three capacities of a kernel form (256 / 512 / 1024 candidate blocks), a scan without corner and one without surface features"""
    sc = scans[2]
    _, _, _, _, fc, fs = oracle_features(sc)
    fc, fs = fc[:sizes[0]], fs[::max(1, len(fs) // max(1, sizes[1]))][:sizes[1]]
    assert len(fc) == sizes[0] and len(fs) == sizes[1]
    prm = orc.RegParams.defaults(icp_iters=6, ceres_iters=20, force_all=1)
    ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    reg = Point_cloud_registration(max_scans=1, max_features=1024)
    reg.set_debug(False, small_solver_waves=waves)
    set_params(reg, 6, 20, 1)
    res, pc, _, reps = reg.solve_batch(dev_map, [fc], [fs], sc.pose_init[None], sc.pose_init[None])
    reg.close()
    check_against(reps[0], orep, pc[0], opc, res[0], ret)


@pytest.mark.parametrize("sizes", [(340, 1600), (100, 1948), (0, 1025)])
@pytest.mark.parametrize("force", [0, 1])
def test_eight_wavefronts_for_up_to_2048_blocks(dev_map, small_world, scans, sizes, force):
    """the sequential mapping loop's scans (VoxelGrid 0.1 / 0.15 m: 1 000 - 2 000 features): eight wavefronts per scan, the sort in LDS"""
    sc = scans[3]
    _, _, _, _, fc, fs = oracle_features(sc)
    fc, fs = fc[:sizes[0]], fs[::max(1, len(fs) // sizes[1])][:sizes[1]]
    assert len(fc) == sizes[0] and len(fs) == sizes[1]
    prm = orc.RegParams.defaults(icp_iters=10, ceres_iters=20, force_all=force)
    ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    outs = []
    for kw in ({}, {"no_small_solver": True}):
        reg = Point_cloud_registration(max_scans=1, max_features=2048)
        reg.set_debug(False, **kw)
        set_params(reg, 10, 20, force)
        outs.append(reg.solve_batch(dev_map, [fc], [fs], sc.pose_init[None], sc.pose_init[None]))
        reg.close()
    res, pc, _, reps = outs[0]
    check_against(reps[0], orep, pc[0], opc, res[0], ret)
    dt, dr = synth.pose_error(pc[0], outs[1][1][0])
    assert dt < 1e-9 and dr < 1e-9


def test_eight_wavefronts_duplicate_residuals(dev_map, small_world, scans):
    """the hash de-duplication + radix select of the four / eight-wavefront forms on a scan with many exact duplicates"""
    sc = scans[1]
    _, _, _, _, fc, fs = oracle_features(sc)
    fs = fs[::14][:1100]
    rng = np.random.default_rng(8)
    rep = rng.choice(len(fs), 500, replace=False)
    fs2 = np.concatenate([fs, fs[rep], fs[rep[:200]]])  # 1 800 surface features: 500 of them twice, 200 three times
    fc2 = np.concatenate([fc[:120], fc[:60]])
    prm = orc.RegParams.defaults(icp_iters=6, ceres_iters=20, force_all=1)
    ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc2, fs2, prm, sc.pose_init, sc.pose_init)
    _, _, _, orep_plain = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc[:120], fs, prm, sc.pose_init, sc.pose_init)
    assert orep.inlier_threshold != orep_plain.inlier_threshold
    reg = Point_cloud_registration(max_scans=1, max_features=2048)
    set_params(reg, 6, 20, 1)
    res, pc, _, reps = reg.solve_batch(dev_map, [fc2], [fs2], sc.pose_init[None], sc.pose_init[None])
    reg.close()
    check_against(reps[0], orep, pc[0], opc, res[0], ret)


@pytest.mark.parametrize("waves", [1, 2, 4])
def test_shipped_block_cap_200(dev_map, small_world, scans, filtered, waves):
    """a13 at the shipped setting: more than 200 candidate blocks -> the reproducible block drop of PCR:438-458"""
    prm = orc.RegParams.defaults(icp_iters=8, ceres_iters=20, force_all=1)
    prm.maximum_allow_residual_block, prm.subsample_seed = 200, 11
    for b in (0, 1):
        sc, (fc, fs) = scans[b], filtered[b]
        assert len(fc) + len(fs) > 200 and max(len(fc), len(fs)) <= 400  # only the block drop fires (the feature skip needs n > 2 M per kind)
        ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
        reg = Point_cloud_registration(max_scans=1, max_features=1024)
        reg.set_debug(False, small_solver_waves=waves)
        p = set_params(reg, 8, 20, 1)
        p.maximum_allow_residual_block, p.subsample_seed = 200, 11
        res, pc, _, reps = reg.solve_batch(dev_map, [fc], [fs], sc.pose_init[None], sc.pose_init[None])
        reg.close()
        check_against(reps[0], orep, pc[0], opc, res[0], ret)
        assert reps[0].n_blocks_last < 260


@pytest.mark.parametrize("waves", [1, 2, 4])
def test_duplicate_residuals_follow_std_set_semantics(dev_map, small_world, scans, filtered, waves):
    """compute_inlier_residual_threshold (PCR:155-160) ranks the DISTINCT values: features repeated verbatim give exact duplicates, which
    the sort has to count once"""
    sc, (fc, fs) = scans[0], filtered[0]
    rng = np.random.default_rng(3)
    rep = rng.choice(len(fs), len(fs) // 3, replace=False)
    fs2 = np.concatenate([fs, fs[rep], fs[rep[: len(rep) // 2]]])  # some features twice, some three times
    fc2 = np.concatenate([fc, fc[: len(fc) // 2]])
    assert len(fc2) + len(fs2) <= 1024
    prm = orc.RegParams.defaults(icp_iters=6, ceres_iters=20, force_all=1)
    ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc2, fs2, prm, sc.pose_init, sc.pose_init)
    _, _, _, orep_plain = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    assert orep.inlier_threshold != orep_plain.inlier_threshold  # the duplicates do matter
    reg = Point_cloud_registration(max_scans=1, max_features=1024)
    reg.set_debug(False, small_solver_waves=waves)
    set_params(reg, 6, 20, 1)
    res, pc, _, reps = reg.solve_batch(dev_map, [fc2], [fs2], sc.pose_init[None], sc.pose_init[None])
    reg.close()
    check_against(reps[0], orep, pc[0], opc, res[0], ret)


@pytest.mark.parametrize("waves", [1, 2, 4])
def test_bounded_line_search(dev_map, small_world, scans, filtered, waves):
    """start 0.25 m outside a 0.05 m bound on t_inc: repeated contractions of the projected line search, the three-sample fit on the
    controller's wavefront"""
    sc, (fc, fs) = scans[0], filtered[0]
    start = sc.pose_init.copy()
    start[4:7] += [0.25, -0.2, 0.1]
    prm = orc.RegParams.defaults(icp_iters=4)
    prm.para_max_speed = 0.05
    ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, start, start)
    reg = Point_cloud_registration(max_scans=1, max_features=1024)
    reg.set_debug(False, small_solver_waves=waves)
    p = set_params(reg, 4, 20, 0)
    p.para_max_speed = 0.05
    res, pc, pi, reps = reg.solve_batch(dev_map, [fc], [fs], start[None], start[None])
    reg.close()
    dt, dr = synth.pose_error(pc[0], opc)
    assert res[0] == ret and dt < 1e-7 and dr < 1e-7
    assert reps[0].lm_iterations_total == orep.lm_iterations_total and reps[0].icp_iterations == orep.icp_iterations


def test_large_batch_one_wavefront_per_scan_by_default(dev_map, small_world, scans, filtered):
    """B = 640: the library picks one wavefront per scan on its own; every slot's answer is the oracle's, equal scans in different slots
    give equal bits, and the four-wavefront form of a small batch agrees to rounding"""
    B, S = 640, len(scans)
    rng = np.random.default_rng(9)
    inits = [sc.pose_init for sc in scans]
    for _ in range(4):  # four more starting points per scan: 20 distinct registrations
        for sc in scans[:S]:
            inits.append(synth.pose_compose(sc.pose_true, np.r_[synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 1.0))), rng.uniform(-0.1, 0.1, 3)]))
    D = len(inits)
    pl = np.stack([inits[b % D] for b in range(B)])
    reg = Point_cloud_registration(max_scans=B, max_features=1024)
    set_params(reg, 10, 20, 1)
    res, pc, _, reps = reg.solve_batch(dev_map, [filtered[b % S][0] for b in range(B)], [filtered[b % S][1] for b in range(B)], pl, pl)
    reg.close()
    prm = orc.RegParams.defaults(icp_iters=10, ceres_iters=20, force_all=1)
    for d in range(D):
        fc, fs = filtered[d % S]
        ret, opc, _, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, inits[d], inits[d])
        check_against(reps[d], orep, pc[d], opc, res[d], ret)
    for b in range(D, B):
        assert np.array_equal(pc[b], pc[b % D]) and res[b] == res[b % D]
    # the order the workgroups start in (longest first from the second ICP iteration on) decides nothing
    reg = Point_cloud_registration(max_scans=B, max_features=1024)
    reg.set_debug(False, no_solve_order=True)
    set_params(reg, 10, 20, 1)
    res_o, pc_o, _, _ = reg.solve_batch(dev_map, [filtered[b % S][0] for b in range(B)], [filtered[b % S][1] for b in range(B)], pl, pl)
    reg.close()
    assert np.array_equal(pc_o, pc) and np.array_equal(res_o, res)
    reg4 = Point_cloud_registration(max_scans=D, max_features=1024)
    set_params(reg4, 10, 20, 1)
    res4, pc4, _, reps4 = reg4.solve_batch(dev_map, [filtered[b % S][0] for b in range(D)], [filtered[b % S][1] for b in range(D)], pl[:D], pl[:D])
    reg4.close()
    for d in range(D):
        dt, dr = synth.pose_error(pc[d], pc4[d])
        assert dt < 1e-9 and dr < 1e-9 and reps[d].lm_iterations_total == reps4[d].lm_iterations_total
