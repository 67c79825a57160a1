"""-m gpu: grid 5-NN and the registrar (through the C ABI) against the CPU oracle.
Neighbour index lists must be identical; the final SE(3) pose must agree within 1e-4 m / 1e-4 rad
(BASELINE.json north_star) -- in practice the agreement is ~1e-9."""
import numpy as np
import pytest

from loam_livox_amd import capi, synth
from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration
from oracle import orc
from tests.conftest import oracle_features

pytestmark = pytest.mark.gpu
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-4


@pytest.fixture(scope="module")
def dev_map(gpu_lib, small_world):
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, small_world["corner"])
    m.setInputCloud(Map_buffer.SURF, small_world["surf"])
    yield m
    m.close()


def set_params(reg, icp=10, ceres=20, force=1):
    p = reg.params
    p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = icp, ceres, force
    p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, 100.0
    p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
    return p


def test_knn5_identical_to_oracle(dev_map, small_world, scans):
    for sc in scans[:2]:
        _, _, _, _, fc, fs = oracle_features(sc)
        qs = synth.transform_points(sc.pose_init, fs[:, :3])
        oi, od = small_world["tree_s"].knn(qs, 5)
        gi, gd = dev_map.nearestKSearch(Map_buffer.SURF, qs, 50.0)
        assert np.array_equal(oi, gi) and np.array_equal(od, gd)
        qc = synth.transform_points(sc.pose_init, fc[:, :3])
        oi, od = small_world["tree_c"].knn(qc, 5)
        gi, gd = dev_map.nearestKSearch(Map_buffer.CORNER, qc, 2.0)   # a few hundred queries: one WAVEFRONT per query (ll_knn_coop.h)
        inside = od < 2.0
        assert np.array_equal(np.where(inside, oi, -1), gi)
        assert np.array_equal(np.where(inside, od, np.inf), gd)
        # both forms of the search on both maps: batches of up to 8192 queries go one per wavefront, larger ones one per lane
        reps = 8192 // len(qc) + 1
        gi2, gd2 = dev_map.nearestKSearch(Map_buffer.CORNER, np.tile(qc, (reps, 1)), 2.0)
        assert np.array_equal(gi2.reshape(reps, -1, 5), np.broadcast_to(gi, (reps,) + gi.shape))
        assert np.array_equal(gd2.reshape(reps, -1, 5), np.broadcast_to(gd, (reps,) + gd.shape))
        oi, od = small_world["tree_s"].knn(qs[:3000], 5)
        gi, gd = dev_map.nearestKSearch(Map_buffer.SURF, qs[:3000], 50.0)
        assert np.array_equal(oi, gi) and np.array_equal(od, gd)


def test_knn5_sparse_outside_ties_nonfinite(gpu_lib):
    rng = np.random.default_rng(2)
    pts = rng.uniform(0, 30, (4000, 3)).astype(np.float32)
    pts[100:104] = pts[100]
    pts[7, 0] = np.nan
    m = Map_buffer()
    m.setInputCloud(Map_buffer.SURF, pts, 0.7)
    tree = orc.KdTree(np.where(np.isfinite(pts), pts, 1e9).astype(np.float32))
    q = np.concatenate([rng.uniform(-8, 38, (500, 3)), pts[100:101], [[1e6, 0, 0]], [[np.nan, 0, 0]]]).astype(np.float32)
    gi, gd = m.nearestKSearch(Map_buffer.SURF, q, 50.0)      # 503 queries: one wavefront per query, rings and the cube sweep included
    oi, od = tree.knn(np.nan_to_num(q, nan=1e9), 5)
    inside = od < 50.0
    assert np.array_equal(np.where(inside, oi, -1)[:-2], gi[:-2])
    assert np.array_equal(np.where(inside, od, np.inf)[:-2], gd[:-2])
    gi_l, gd_l = m.nearestKSearch(Map_buffer.SURF, np.tile(q, (17, 1)), 50.0)   # 8551 queries: one lane per query
    assert np.array_equal(gi_l.reshape(17, -1, 5), np.broadcast_to(gi, (17,) + gi.shape))
    assert np.array_equal(gd_l.reshape(17, -1, 5), np.broadcast_to(gd, (17,) + gd.shape))
    assert np.all(gi[-2:] == -1)
    assert gi[500].tolist()[:4] == [100, 101, 102, 103]
    # xyzi stride-4 input gives the same answer
    m2 = Map_buffer()
    m2.setInputCloud(Map_buffer.SURF, np.c_[pts, np.ones(len(pts), np.float32)], 0.7)
    gi2, _ = m2.nearestKSearch(Map_buffer.SURF, q, 50.0)
    assert np.array_equal(gi, gi2)
    m.close(); m2.close()


@pytest.mark.parametrize("k", [0, 1, 2, 3])
@pytest.mark.parametrize("force", [0, 1])
@pytest.mark.parametrize("general", [False, "single", True])
def test_registration_matches_oracle(request, dev_map, small_world, scans, k, force, general):
    """general=False: round-3 compact path (plane table: {n', c} once per distinct neighbour triple, 18-byte block records) in
    the form a batch of one takes: the scan spread over a group of 8 workgroups; "single": the same path with one workgroup
    per scan, as batches of more than 16 scans run it; True: the HBM-resident path used by scans with more than 24576
    residual blocks (forced here on a normal scan)."""
    sc = scans[k]
    _, _, _, _, fc, fs = oracle_features(sc)
    prm = orc.RegParams.defaults(icp_iters=10, ceres_iters=20, force_all=force)
    ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    reg = Point_cloud_registration(max_scans=1, max_features=24000)
    reg.set_debug(True, force_general_solver=(general is True), no_solver_groups=(general == "single"))
    set_params(reg, 10, 20, force)
    reg.m_pose_w_last = sc.pose_init.copy()
    reg.m_pose_w_curr = sc.pose_init.copy()
    gret = reg.find_out_incremental_transfrom(dev_map, fc, fs)
    dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
    assert gret == ret
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
    assert dt < 1e-7 and dr < 1e-7  # what we actually expect from identical algorithms in fp64
    g = reg.report
    assert g.icp_iterations == rep.icp_iterations and g.n_blocks_last == rep.n_blocks_last
    assert g.corner_avail == rep.corner_avail and g.surf_avail == rep.surf_avail
    assert g.lm_iterations_total == rep.lm_iterations_total
    assert np.isclose(g.final_cost, rep.final_cost, rtol=1e-8) and np.isclose(g.initial_cost, rep.initial_cost, rtol=1e-8)
    assert np.isclose(g.inlier_threshold, rep.inlier_threshold, rtol=1e-8)
    assert np.isclose(g.angular_diff_deg, rep.angular_diff_deg, atol=1e-6) and np.isclose(g.t_diff, rep.t_diff, atol=1e-8)
    # neighbour lists of the first ICP iteration are identical (inside the match radius)
    ci, cd, si, sd = reg.debug_knn(0, len(fc), len(fs))
    qs = synth.transform_points(sc.pose_init, fs[:, :3])
    oi, od = small_world["tree_s"].knn(qs, 5)
    assert np.array_equal(oi, si) and np.array_equal(od, sd)
    qc = synth.transform_points(sc.pose_init, fc[:, :3])
    oi, od = small_world["tree_c"].knn(qc, 5)
    inside = od < 2.0
    assert np.array_equal(np.where(inside, oi, -1), ci)
    reg.close()


def test_group_barrier_abort_rejects_the_scan(request, dev_map, scans):
    """A group barrier of the small-batch solver that does not complete (bounded spin; forced here) must not hand back a NaN
    pose as an accepted result: the scan is rejected with its pose restored and ll_reg_collect reports the failure."""
    import ctypes as C
    from loam_livox_amd.api import RegReport
    from loam_livox_amd.capi import ptr
    sc = scans[0]
    _, _, _, _, fc, fs = oracle_features(sc)
    assert len(fc) + len(fs) >= 6000  # large enough for the grouped form
    reg = Point_cloud_registration(max_scans=1, max_features=24000)
    reg.set_debug(False, test_group_abort=True)
    set_params(reg, 4, 20, 1)
    reg.upload_features([fc], [fs])
    pl = sc.pose_init.reshape(1, 7).copy()
    reg.enqueue_uploaded(dev_map, 1, pl, pl)
    pc, pi = np.zeros((1, 7)), np.zeros((1, 7))
    reps, res = (RegReport * 1)(), np.ones(1, np.int32)
    rc = reg.L.ll_reg_collect(reg.h, 1, ptr(pc), ptr(pi), reps, ptr(res))
    assert rc == 1 and b"timed out" in reg.L.ll_last_error()   # one aborted scan: reported, not an error of the call
    assert res[0] == 0 and reps[0].accepted == 0 and reps[0].aborted == 1 and np.array_equal(pc[0], sc.pose_init) and np.all(np.isfinite(pc))
    # the handle is usable afterwards
    reg.set_debug(False)
    reg.enqueue_uploaded(dev_map, 1, pl, pl)
    res2, pc2, _, reps2 = reg.collect(1)
    assert res2[0] == 1 and np.all(np.isfinite(pc2)) and not np.array_equal(pc2[0], sc.pose_init) and reps2[0].aborted == 0
    reg.close()


@pytest.mark.parametrize("groups,deblur", [(False, 0), (True, 0), (False, 1)])
@pytest.mark.parametrize("world", ["rooms", "random_cloud"])
def test_plane_table_agrees_with_per_block_records(dev_map, scans, groups, deblur, world):
    """The plane-table solver path evaluates, block for block, the numbers of the general path, which stores {n', c} with every
    block (different evaluation loops and summation grouping, so they agree to rounding, not to the bit) -- on the synthetic rooms (a few thousand distinct neighbour triples: the whole table in LDS) and
    against a uniform random cloud, where nearly every block has a triple of its own: the LDS hash table fills up (private
    table entries), the table overflows its LDS part (planes gathered from HBM) and nothing is de-duplicated.
    deblur = 1: the same for solve_big (ll_reg_big_path.h: use-ranked ids, the clamped L2 gather for what LDS does not hold, private entries
    after 32 probes) with the motion-deblur residuals."""
    sc = scans[1]
    fe_o, _, _, _, fc, fs = oracle_features(sc)
    if world == "rooms":
        m = dev_map
    else:
        rng = np.random.default_rng(5)
        lo, hi = synth.transform_points(sc.pose_init, fs[:, :3]).min(0) - 1.0, synth.transform_points(sc.pose_init, fs[:, :3]).max(0) + 1.0
        m = Map_buffer()
        m.setInputCloud(Map_buffer.CORNER, rng.uniform(lo, hi, (60000, 3)).astype(np.float32))
        m.setInputCloud(Map_buffer.SURF, rng.uniform(lo, hi, (400000, 3)).astype(np.float32))
    out = []
    for per_block in (False, True):
        reg = Point_cloud_registration(max_scans=1, max_features=24000)
        reg.set_debug(False, force_general_solver=per_block, no_solver_groups=not groups)
        pp = set_params(reg, 2 if world != "rooms" else 4, 20, 1)
        if deblur:
            pp.if_motion_deblur, pp.minimum_pt_time_stamp, pp.maximum_pt_time_stamp = 1, float(fe_o.time_stamp.min()), float(fe_o.time_stamp.max())
        reg.m_pose_w_last = sc.pose_init.copy()
        reg.m_pose_w_curr = sc.pose_init.copy()
        ret = reg.find_out_incremental_transfrom(m, fc, fs)
        g = reg.report
        out.append((ret, reg.m_pose_w_curr.copy(), g.final_cost, g.initial_cost, g.inlier_threshold, g.n_blocks_last, g.surf_avail, g.corner_avail,
                    g.lm_iterations_total))
        reg.close()
    a, b = out
    dt, dr = synth.pose_error(a[1], b[1])
    tol = 1e-9 if world == "rooms" else 1e-6  # (registration against noise is ill-conditioned: rounding differences grow)
    assert a[0] == b[0] and dt < tol and dr < tol and np.all(np.isfinite(a[1]))
    assert a[6:8] == b[6:8] and a[6] > 1000  # blocks found: decided before the solver
    assert np.isclose(a[3], b[3], rtol=1e-9)  # cost at the first evaluation: every block's constants agree
    if world == "rooms":
        assert a[5] == b[5] and a[8] == b[8] and np.isclose(a[2], b[2], rtol=1e-8) and np.isclose(a[4], b[4], rtol=1e-8)
    if world != "rooms":
        m.close()


def test_reject_gate_and_bounds(dev_map, small_world, scans):
    sc = scans[0]
    _, _, _, _, fc, fs = oracle_features(sc)
    reg = Point_cloud_registration()
    p = set_params(reg, 3, 20, 0)
    p.max_final_cost = 1e-6
    reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
    assert reg.find_out_incremental_transfrom(dev_map, fc, fs) == 0          # PCR:561-573
    assert np.array_equal(reg.m_pose_w_curr, sc.pose_init) and reg.report.accepted == 0
    p = set_params(reg, 3, 20, 0)
    p.current_frame_index = 10                                                # PCR:199 gate
    reg.m_pose_w_curr = sc.pose_init.copy(); reg.m_para_buffer_incremental = np.array([0, 0, 0, 1, 0, 0, 0.0])
    assert reg.find_out_incremental_transfrom(dev_map, fc, fs) == 1 and reg.report.gated == 1
    assert np.array_equal(reg.m_pose_w_curr, sc.pose_init)
    p = set_params(reg, 2, 20, 0)
    p.para_max_speed = 0.01                                                   # PCR:143-151
    prm = orc.RegParams.defaults(icp_iters=2); prm.para_max_speed = 0.01
    ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    reg.m_pose_w_curr = sc.pose_init.copy(); reg.m_para_buffer_incremental = np.array([0, 0, 0, 1, 0, 0, 0.0])
    reg.find_out_incremental_transfrom(dev_map, fc, fs)
    assert np.all(np.abs(reg.m_para_buffer_incremental[4:]) <= 0.01 + 1e-15)
    dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
    assert dt < 1e-7 and dr < 1e-7
    reg.close()


@pytest.mark.parametrize("groups", [True, False])
def test_bounded_line_search_three_sample_interpolation(dev_map, small_world, scans, groups):
    """start 0.25 m outside a 0.05 m bound on t_inc: the projected line search contracts repeatedly -- from the second contraction
    on with Ceres' three-sample quintic (lm_quintic_min_step on the controller lane) -- and lands where the oracle does"""
    sc = scans[0]
    _, _, _, _, fc, fs = oracle_features(sc)
    start = sc.pose_init.copy()
    start[4:7] += [0.25, -0.2, 0.1]
    prm = orc.RegParams.defaults(icp_iters=4)
    prm.para_max_speed = 0.05
    ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, start, start)
    reg = Point_cloud_registration(max_scans=1, max_features=24000)
    reg.set_debug(False, no_solver_groups=not groups)
    p = set_params(reg, 4, 20, 0)
    p.para_max_speed = 0.05
    reg.m_pose_w_last = start.copy()
    reg.m_pose_w_curr = start.copy()
    gret = reg.find_out_incremental_transfrom(dev_map, fc, fs)
    dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
    assert gret == ret and dt < 1e-7 and dr < 1e-7
    assert reg.report.lm_iterations_total == rep.lm_iterations_total and reg.report.icp_iterations == rep.icp_iterations
    assert np.all(np.abs(reg.m_para_buffer_incremental[4:]) <= float(np.float32(0.05)) + 1e-15)
    reg.close()


def test_empty_and_nonfinite_features(dev_map, scans):
    sc = scans[0]
    reg = Point_cloud_registration(max_features=1000)
    set_params(reg, 2, 5, 0)
    reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
    empty = np.zeros((0, 4), np.float32)
    assert reg.find_out_incremental_transfrom(dev_map, empty, empty) == 1
    assert np.allclose(reg.m_pose_w_curr, sc.pose_init, atol=1e-15) and reg.report.n_blocks_last == 0
    bad = np.full((10, 4), np.nan, np.float32)
    assert reg.find_out_incremental_transfrom(dev_map, bad, bad) == 1 and reg.report.surf_avail == 0
    reg.close()


def test_batch_pipeline_equals_single_calls(dev_map, small_world, scans):
    """extractor -> device-resident selection -> batched registration == oracle per scan"""
    B = 4
    fe = Livox_laser(max_points=24000, max_scans=B, piecewise_number=1)
    batch = np.stack([s.xyzi for s in scans])
    ct = np.full(B, 1.0)
    fe.upload(batch, ct)
    fe.extract_batch(B); fe.resolve(); fe.select_batch(B, -1, 0.0, 1.0)
    reg = Point_cloud_registration(max_scans=B, max_features=24000)
    set_params(reg, 10, 20, 1)
    init = np.stack([s.pose_init for s in scans])
    res, pc, pi, reps = reg.solve_batch_fe(dev_map, fe, B, init, init)
    for b, sc in enumerate(scans):
        _, _, _, _, fc, fs = oracle_features(sc)
        prm = orc.RegParams.defaults(icp_iters=10, ceres_iters=20, force_all=1)
        ret, opc, opi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
        dt, dr = synth.pose_error(pc[b], opc)
        assert res[b] == ret and dt < 1e-7 and dr < 1e-7
        assert reps[b].n_blocks_last == rep.n_blocks_last and reps[b].lm_iterations_total == rep.lm_iterations_total
    # host-feature batch entry point gives the same poses
    feats = [oracle_features(sc)[4:] for sc in scans]
    res2, pc2, _, _ = reg.solve_batch(dev_map, [f[0] for f in feats], [f[1] for f in feats], init, init)
    assert np.array_equal(res, res2) and np.allclose(pc, pc2, atol=1e-12)
    fe.close(); reg.close()


def test_grouped_and_single_workgroup_solver_agree_across_batch_sizes(dev_map, small_world, scans):
    """Batches of up to 16 scans give every scan a group of 8 solver workgroups (partial sums exchanged through memory),
    larger ones a single workgroup.  The grouping of the floating-point sums differs, nothing else: same iteration and block
    counts, poses equal to rounding, and inside one form a scan's answer does not depend on its slot or on the batch size."""
    feats = [oracle_features(sc)[4:] for sc in scans]
    def run(n, groups=True):
        reg = Point_cloud_registration(max_scans=n, max_features=24000)
        reg.set_debug(False, no_solver_groups=not groups)
        set_params(reg, 10, 20, 1)
        pl = np.stack([scans[i % len(scans)].pose_init for i in range(n)])
        out = reg.solve_batch(dev_map, [feats[i % len(scans)][0] for i in range(n)], [feats[i % len(scans)][1] for i in range(n)], pl, pl)
        reg.close()
        return out
    r16, p16, _, rep16 = run(16)            # grouped: 128 workgroups
    r17, p17, _, rep17 = run(17)            # one workgroup per scan
    r1, p1, _, rep1 = run(1)                # grouped, a single group
    r1s, p1s, _, rep1s = run(1, False)      # one workgroup
    assert list(r16) == [1] * 16 and list(r17) == [1] * 17
    assert np.array_equal(p16[0], p1[0]) and np.array_equal(p16[0], p16[len(scans)])       # same form: bit-identical
    assert np.array_equal(p17[0], p1s[0]) and np.array_equal(p17[0], p17[len(scans)])
    for a, b_ in ((p16[:16], p17[:16]),):
        for i in range(16):
            dt, dr = synth.pose_error(a[i], b_[i])
            assert dt < 1e-9 and dr < 1e-9
    for i in range(16):
        assert (rep16[i].lm_iterations_total, rep16[i].n_blocks_last, rep16[i].icp_iterations) == \
               (rep17[i].lm_iterations_total, rep17[i].n_blocks_last, rep17[i].icp_iterations)
    assert np.all(np.isfinite(p16)) and np.all(np.isfinite(p1))


@pytest.mark.parametrize("mode", ["no_reuse", "reuse_from_iter1"])
def test_knn_reuse_is_exact(dev_map, small_world, scans, mode):
    """the neighbour reuse across ICP iterations must not change anything: same pose bits as with a full search in
    every iteration, whether it starts at iteration 1 or 2 (per-lane search: the tile search is switched off here)"""
    sc = scans[2]
    _, _, _, _, fc, fs = oracle_features(sc)
    poses = []
    for kw in ({}, {"no_knn_reuse": True} if mode == "no_reuse" else {"reuse_from_iter1": True}):
        reg = Point_cloud_registration()
        reg.set_debug(False, no_knn_tile=True, **kw)
        set_params(reg, 10, 20, 1)
        reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
        reg.find_out_incremental_transfrom(dev_map, fc, fs)
        poses.append((reg.m_pose_w_curr.copy(), reg.report.lm_iterations_total, reg.report.n_blocks_last))
        reg.close()
    assert np.array_equal(poses[0][0], poses[1][0]) and poses[0][1:] == poses[1][1:]


@pytest.mark.parametrize("n", [1, 3, 20])
def test_tile_search_changes_nothing(dev_map, small_world, scans, n):
    """The tile search of the surface queries (ll_knn_tile.h: queries sorted by map cell, one wavefront per 64 of them against the
    LDS-staged points of their cells' neighbourhood; default: in every ICP iteration, no reuse records) returns the neighbour lists
    of the per-lane search -- same pose bits, same counts, with the reuse machinery behind it (tile at iterations 0 / 1 only) or
    without, and the neighbour lists of iteration 0 equal to the k-d tree's."""
    feats = [(f[4], f[5]) for f in (oracle_features(sc) for sc in scans)]
    outs = []
    for kw in ({"knn_tile_small_batches": True}, {"knn_tile_small_batches": True, "knn_tile_with_reuse": True}, {"no_knn_tile": True},
               {"no_knn_tile": True, "no_knn_reuse": True}):
        reg = Point_cloud_registration(max_scans=n, max_features=24000)
        reg.set_debug(True, **kw)
        set_params(reg, 10, 20, 1)
        pl = np.stack([scans[i % len(scans)].pose_init for i in range(n)])
        res, pc, _, reps = reg.solve_batch(dev_map, [feats[i % len(scans)][0] for i in range(n)], [feats[i % len(scans)][1] for i in range(n)], pl, pl)
        knn = [reg.debug_knn(i, len(feats[i % len(scans)][0]), len(feats[i % len(scans)][1])) for i in range(min(n, 3))]
        outs.append((res.copy(), pc.copy(), [(r.lm_iterations_total, r.n_blocks_last, r.icp_iterations) for r in reps], knn))
        reg.close()
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) and outs[0][2] == o[2]
        for a, b in zip(outs[0][3], o[3]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for i in range(min(n, 3)):
        qs = synth.transform_points(scans[i % len(scans)].pose_init, feats[i % len(scans)][1][:, :3])
        oi, od = small_world["tree_s"].knn(qs, 5)
        assert np.array_equal(oi, outs[0][3][i][2]) and np.array_equal(od, outs[0][3][i][3])


def test_tile_search_sparse_map_ties_and_strays(gpu_lib):
    """a uniform random cloud (most lanes need the rings), exact duplicates among the map points (ties by index) and queries outside
    the grid / not finite: the tile kernel's fall-back lanes; checked against the k-d tree through the registrar's debug tap"""
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 20, (60000, 3)).astype(np.float32)
    pts[100:104] = pts[100]
    corner = rng.uniform(0, 20, (500, 3)).astype(np.float32)
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, corner)
    m.setInputCloud(Map_buffer.SURF, pts, 0.6)
    tree = orc.KdTree(pts)
    fs = np.zeros((5000, 4), np.float32)
    fs[:, :3] = rng.uniform(-1, 21, (5000, 3))
    fs[0, :3] = pts[100]
    fs[1, :3] = [1e5, 0, 0]
    fs[2, 0] = np.nan
    fc = np.zeros((10, 4), np.float32)
    fc[:, :3] = rng.uniform(0, 20, (10, 3))
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
    res = []
    for kw in ({"knn_tile_small_batches": True}, {"no_knn_tile": True}):
        reg = Point_cloud_registration(max_scans=1, max_features=8192)
        reg.set_debug(True, **kw)
        set_params(reg, 1, 4, 1)
        reg.m_pose_w_last = ident.copy(); reg.m_pose_w_curr = ident.copy()
        reg.find_out_incremental_transfrom(m, fc, fs)
        res.append(reg.debug_knn(0, len(fc), len(fs)))
        reg.close()
    si, sd = res[0][2], res[0][3]
    assert np.array_equal(si, res[1][2]) and np.array_equal(sd, res[1][3])
    ok = np.isfinite(fs[:, 0])
    oi, od = tree.knn(fs[ok, :3], 5)
    inside = od < 50.0
    assert np.array_equal(np.where(inside, oi, -1), si[ok]) and np.array_equal(np.where(inside, od, np.inf), sd[ok])
    assert np.all(si[2] == -1) and np.all(si[1] == -1)
    assert si[0].tolist()[:4] == [100, 101, 102, 103]
    m.close()


def test_corner_queries_take_the_tile_search_in_large_batches(gpu_lib):
    """Batches of more than 16 scans: the corner queries go through the tile search on the corner map inside reg_knn_lane_kernel
    (cell-ordered by reg_qsort_corner_kernel; scans with more than 2 048 corner queries keep the feature order).  A sparse corner map:
    most queries have fewer than five neighbours inside the line radius -- the tile settles those too, because a 3 x 3 x 3 block of
    1.45 m cells covers the radius --, exact duplicates (ties by index), queries outside the grid and non-finite ones.  The lists of
    ICP iteration 0 against the per-lane search and against the k-d tree, for a scan on either side of the 2 048 limit."""
    rng = np.random.default_rng(11)
    corner = rng.uniform(0, 20, (9000, 3)).astype(np.float32)
    corner[50:54] = corner[50]
    surf = rng.uniform(0, 20, (40000, 3)).astype(np.float32)
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, corner)
    m.setInputCloud(Map_buffer.SURF, surf, 0.6)
    tree = orc.KdTree(corner)
    n = 20
    fcs, fss = [], []
    for b in range(n):
        nc = 3000 if b % 2 == 0 else 700
        fc = np.zeros((nc, 4), np.float32)
        fc[:, :3] = rng.uniform(-2, 22, (nc, 3))
        fc[0, :3] = corner[50]
        fc[1, :3] = [1e5, 0, 0]
        fc[2, 1] = np.nan
        fs = np.zeros((2000, 4), np.float32)
        fs[:, :3] = rng.uniform(0, 20, (2000, 3))
        fcs.append(fc)
        fss.append(fs)
    ident = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float64), (n, 1))
    res = []
    for kw in ({}, {"no_knn_tile": True}):
        reg = Point_cloud_registration(max_scans=n, max_features=4096)
        reg.set_debug(True, **kw)
        set_params(reg, 1, 4, 1)
        reg.solve_batch(m, fcs, fss, ident, ident)
        res.append([reg.debug_knn(b, len(fcs[b]), len(fss[b])) for b in (0, 1, 19)])
        reg.close()
    for (ci, cd, si, sd), (ci2, cd2, si2, sd2), b in zip(res[0], res[1], (0, 1, 19)):
        assert np.array_equal(ci, ci2) and np.array_equal(cd, cd2) and np.array_equal(si, si2) and np.array_equal(sd, sd2)
        fc = fcs[b]
        ok = np.isfinite(fc[:, :3]).all(axis=1)
        oi, od = tree.knn(fc[ok, :3], 5)
        inside = od < 2.0
        assert np.array_equal(np.where(inside, oi, -1), ci[ok]) and np.array_equal(np.where(inside, od, np.inf), cd[ok])
        assert (inside.sum(axis=1) < 5).mean() > 0.3            # most lists are short ones
        assert np.all(ci[1] == -1) and np.all(ci[2] == -1)
        assert ci[0].tolist()[:4] == [50, 51, 52, 53]
    m.close()


@pytest.mark.parametrize("n,thin", [(1, 1), (20, 1), (1, 12), (16, 12)])
def test_wavefront_search_changes_nothing(dev_map, scans, n, thin):
    """Searches by whole wavefronts (ll_knn_coop.h: every corner query of ICP iterations 0 / 1 for batches of up to 16 scans --
    and every surface query too when the scans are small, thin = 12 --, the late iterations' search lists while they are short,
    their corner entries otherwise) return the neighbour lists of the per-lane search; only the reuse budgets differ (larger),
    i.e. which queries are searched again later.  Same pose bits."""
    feats = [(f[4][::thin], f[5][::thin]) for f in (oracle_features(sc) for sc in scans)]
    outs = []
    for kw in ({"no_knn_tile": True}, {"no_knn_tile": True, "no_knn_coop": True}, {"knn_tile_small_batches": True},
               {"knn_tile_small_batches": True, "no_knn_coop": True}):
        reg = Point_cloud_registration(max_scans=n, max_features=24000)
        reg.set_debug(False, **kw)
        set_params(reg, 10, 20, 1)
        pl = np.stack([scans[i % len(scans)].pose_init for i in range(n)])
        res, pc, _, reps = reg.solve_batch(dev_map, [feats[i % len(scans)][0] for i in range(n)], [feats[i % len(scans)][1] for i in range(n)], pl, pl)
        outs.append((res.copy(), pc.copy(), [(r.lm_iterations_total, r.n_blocks_last, r.icp_iterations) for r in reps]))
        reg.close()
    for o in outs[1:]:
        assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) and outs[0][2] == o[2]


def test_run_to_run_determinism(dev_map, scans):
    sc = scans[1]
    _, _, _, _, fc, fs = oracle_features(sc)
    outs = []
    for _ in range(3):
        reg = Point_cloud_registration()
        set_params(reg, 5, 20, 1)
        reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
        reg.find_out_incremental_transfrom(dev_map, fc, fs)
        outs.append(reg.m_pose_w_curr.copy())
        reg.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_cloud_transform_bit_exact(gpu_lib):
    rng = np.random.default_rng(4)
    pose = np.r_[synth.quat_from_axis_angle(rng.normal(size=3), 1.1), 100.0, -50.0, 2.0]
    pts = rng.uniform(-40, 40, (5000, 4)).astype(np.float32)
    reg = Point_cloud_registration(max_features=16)
    out = reg.pointcloudAssociateToMap(pts, pose)
    assert np.array_equal(out, orc.cloud_transform(pose, pts))  # PCR:629,656-659
    reg.close()


def test_errors_are_reported_not_fatal(gpu_lib, dev_map):
    reg = Point_cloud_registration(max_features=10)
    with pytest.raises(capi.LoamLivoxError):
        reg.find_out_incremental_transfrom(dev_map, np.zeros((11, 4), np.float32), np.zeros((1, 4), np.float32))
    reg.close()


@pytest.mark.parametrize("k", [0, 1, 2])
@pytest.mark.parametrize("general", [False, True])
def test_motion_deblur_matches_oracle(dev_map, small_world, scans, k, general):
    """if_motion_deblur = 1: the *_mb residuals (ceres_icp.hpp:81-233, slerp-interpolated increment) and the Rodrigues
    query transform (PCR:641-646).  The device uses SO(3) left-Jacobians where the reference differentiates slerp."""
    sc = scans[k]
    fe, ci, si, fi, fc, fs = oracle_features(sc)
    tmin, tmax = float(fe.time_stamp.min()), float(fe.time_stamp.max())
    prm = orc.RegParams.defaults(icp_iters=6, ceres_iters=20, force_all=1, deblur=1)
    prm.minimum_pt_time_stamp, prm.maximum_pt_time_stamp = tmin, tmax
    ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    reg = Point_cloud_registration(max_scans=1, max_features=24000)
    reg.set_debug(False, force_general_solver=general)
    p = set_params(reg, 6, 20, 1)
    p.if_motion_deblur, p.minimum_pt_time_stamp, p.maximum_pt_time_stamp = 1, tmin, tmax
    reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
    gret = reg.find_out_incremental_transfrom(dev_map, fc, fs)
    dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
    assert gret == ret and dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and dt < 1e-7 and dr < 1e-7
    g = reg.report
    assert g.n_blocks_last == rep.n_blocks_last and g.lm_iterations_total == rep.lm_iterations_total
    assert np.isclose(g.final_cost, rep.final_cost, rtol=1e-8)
    reg.close()


def test_moving_sensor_deblur_matches_oracle_and_lowers_cost(dev_map, small_world):
    """a scan taken from a moving sensor (constant twist, the motion model of the *_mb residuals): GPU == oracle with
    and without deblur, and compensating the distortion explains the data better (lower final cost)"""
    sc = synth.make_moving_scan(small_world["world"], 1)
    o = orc.fe_extract(sc.xyzi, 0.0)
    ci, si, fi = orc.fe_get_features(o, 0.0, 1.0)
    fc, fs = orc.feature_cloud(o, ci), orc.feature_cloud(o, si)
    tmin, tmax = float(o.time_stamp[0]), float(o.time_stamp[-1])
    costs = {}
    for deblur in (0, 1):
        prm = orc.RegParams.defaults(icp_iters=8, ceres_iters=20, force_all=1, deblur=deblur)
        prm.minimum_pt_time_stamp, prm.maximum_pt_time_stamp = tmin, tmax
        ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
        reg = Point_cloud_registration()
        p = set_params(reg, 8, 20, 1)
        p.if_motion_deblur, p.minimum_pt_time_stamp, p.maximum_pt_time_stamp = deblur, tmin, tmax
        reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
        assert reg.find_out_incremental_transfrom(dev_map, fc, fs) == ret
        dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
        assert dt < 1e-7 and dr < 1e-7 and reg.report.n_blocks_last == rep.n_blocks_last
        costs[deblur] = reg.report.final_cost
        reg.close()
    assert costs[1] < costs[0]


@pytest.mark.parametrize("checks", [(1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("general", [False, True])
def test_pca_feature_checks_match_oracle(gpu_lib, small_world, scans, checks, general):
    """K7: IF_LINE_FEATURE_CHECK / IF_PLANE_FEATURE_CHECK (PCR:46,48,259-292,357-389), with the plane check on the
    surface cloud.  A noisy corner map makes the line test reject neighbourhoods too."""
    from tests.test_hostcheck import noisy_corner_map
    sc = scans[1]
    _, _, _, _, fc, fs = oracle_features(sc)
    corner = noisy_corner_map(small_world)
    tree_c = orc.KdTree(corner)
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, corner)
    m.setInputCloud(Map_buffer.SURF, small_world["surf"])
    prm = orc.RegParams.defaults(icp_iters=8, ceres_iters=20, force_all=1)
    prm.if_line_feature_check, prm.if_plane_feature_check = checks
    ret, pc, pi, rep = orc.reg_solve(tree_c, small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    base = orc.RegParams.defaults(icp_iters=8, ceres_iters=20, force_all=1)
    _, _, _, rep0 = orc.reg_solve(tree_c, small_world["tree_s"], fc, fs, base, sc.pose_init, sc.pose_init)
    assert rep.corner_avail + rep.surf_avail < rep0.corner_avail + rep0.surf_avail
    poses = []
    for no_reuse in (False, True):
        reg = Point_cloud_registration(max_scans=1, max_features=24000)
        reg.set_debug(False, force_general_solver=general, no_knn_reuse=no_reuse)
        p = set_params(reg, 8, 20, 1)
        p.if_line_feature_check, p.if_plane_feature_check = checks
        reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
        gret = reg.find_out_incremental_transfrom(m, fc, fs)
        dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
        g = reg.report
        assert gret == ret and dt < 1e-7 and dr < 1e-7
        assert g.n_blocks_last == rep.n_blocks_last and g.corner_avail == rep.corner_avail and g.surf_avail == rep.surf_avail
        assert g.lm_iterations_total == rep.lm_iterations_total
        poses.append(reg.m_pose_w_curr.copy())
        reg.close()
    assert np.array_equal(poses[0], poses[1])
    m.close()


@pytest.mark.parametrize("general", [False, True])
def test_duplicate_residuals_follow_std_set_semantics(dev_map, small_world, scans, general):
    """compute_inlier_residual_threshold (PCR:155-160) ranks the DISTINCT loss-corrected residuals (it inserts them into a
    std::set).  Repeating features verbatim produces exact duplicates, which shift the rank of the 80 % threshold; the device
    de-duplication (LDS hash table / HBM table) has to land on the same value as the oracle's set."""
    sc = scans[0]
    _, _, _, _, fc, fs = oracle_features(sc)
    rng = np.random.default_rng(3)
    rep = rng.choice(len(fs), len(fs) // 3, replace=False)
    fs2 = np.concatenate([fs, fs[rep], fs[rep[: len(rep) // 2]]])          # some features twice, some three times
    fc2 = np.concatenate([fc, fc[: len(fc) // 2]])
    prm = orc.RegParams.defaults(icp_iters=6, ceres_iters=20, force_all=1)
    ret, pc, pi, orep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc2, fs2, prm, sc.pose_init, sc.pose_init)
    _, pc_plain, _, orep_plain = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    assert orep.inlier_threshold != orep_plain.inlier_threshold            # the duplicates do matter
    reg = Point_cloud_registration(max_scans=1, max_features=40000)
    reg.set_debug(False, force_general_solver=general)
    set_params(reg, 6, 20, 1)
    reg.params.maximum_allow_residual_block = 40000
    reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
    gret = reg.find_out_incremental_transfrom(dev_map, fc2, fs2)
    g = reg.report
    dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
    assert gret == ret and dt < 1e-7 and dr < 1e-7
    assert g.n_blocks_last == orep.n_blocks_last and g.lm_iterations_total == orep.lm_iterations_total
    assert np.isclose(g.inlier_threshold, orep.inlier_threshold, rtol=1e-9)
    reg.close()


@pytest.mark.parametrize("max_blocks,general", [(200, False), (3000, False), (200, True), (6000, False)])
def test_subsampling_matches_oracle(dev_map, small_world, scans, max_blocks, general):
    """a13: maximum_allow_residual_block below the feature count (200 in the shipped configs).  With a seed the library
    sub-samples like the reference does (feature skip when n > 2 M, block drop when blocks > M) on a reproducible stream;
    the oracle makes the same choices.  6000: only the block drop fires, so the neighbour reuse stays on."""
    sc = scans[1]
    _, _, _, _, fc, fs = oracle_features(sc)
    prm = orc.RegParams.defaults(icp_iters=8, ceres_iters=20, force_all=1)
    prm.maximum_allow_residual_block, prm.subsample_seed = max_blocks, 11
    ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    reg = Point_cloud_registration(max_scans=1, max_features=24000)
    reg.set_debug(False, force_general_solver=general)
    p = set_params(reg, 8, 20, 1)
    p.maximum_allow_residual_block, p.subsample_seed = max_blocks, 11
    reg.m_pose_w_last = sc.pose_init.copy(); reg.m_pose_w_curr = sc.pose_init.copy()
    gret = reg.find_out_incremental_transfrom(dev_map, fc, fs)
    g = reg.report
    dt, dr = synth.pose_error(reg.m_pose_w_curr, pc)
    assert gret == ret and dt < 1e-7 and dr < 1e-7
    assert g.n_blocks_last == rep.n_blocks_last and g.corner_avail == rep.corner_avail and g.surf_avail == rep.surf_avail
    assert g.lm_iterations_total == rep.lm_iterations_total
    assert g.n_blocks_last < 1.3 * max_blocks
    # strict mode (seed 0) refuses instead
    p.subsample_seed = 0
    with pytest.raises(Exception):
        reg.find_out_incremental_transfrom(dev_map, fc, fs)
    reg.close()


def test_map_refresh_while_registering_uses_immutable_snapshots(gpu_lib, small_world, scans):
    """update_buff_for_matching refreshes the match buffer on its own thread (laser_mapping.hpp:568) while
    process_new_scan threads register against it (:1737-1742).  ll_map_upload publishes a new immutable snapshot and a
    solve keeps the one it started with: with one thread re-uploading two different maps in turn and two threads
    registering the same scan over and over, every result must be bit-identical to the serial result against one map
    or the other -- never a mixture, never a crash."""
    import threading
    sc = scans[0]
    _, _, _, _, fc, fs = oracle_features(sc)
    maps = [(small_world["corner"], small_world["surf"])]
    shift = np.array([0.004, -0.003, 0.002], np.float32)  # a second, slightly displaced map: a different answer
    maps.append((small_world["corner"] + shift, small_world["surf"] + shift))

    def solve(reg, m):
        reg.m_pose_w_last = sc.pose_init.copy()
        reg.m_pose_w_curr = sc.pose_init.copy()
        ret = reg.find_out_incremental_transfrom(m, fc, fs)
        return ret, reg.m_pose_w_curr.copy(), reg.report.n_blocks_last

    serial = []
    for c, s in maps:
        m = Map_buffer()
        m.setInputCloud(Map_buffer.CORNER, c)
        m.setInputCloud(Map_buffer.SURF, s)
        reg = Point_cloud_registration(max_scans=1, max_features=24000)
        set_params(reg, 4, 20, 1)
        serial.append(solve(reg, m))
        reg.close(); m.close()
    assert not np.array_equal(serial[0][1], serial[1][1])

    shared = Map_buffer()
    shared.setInputCloud(Map_buffer.CORNER, maps[0][0])
    shared.setInputCloud(Map_buffer.SURF, maps[0][1])
    stop = threading.Event()
    errors, results = [], []

    def refresher():
        k = 0
        try:
            while not stop.is_set():
                k ^= 1
                # both kinds of one map are swapped one after the other: a solve may see corner of one and surface of
                # the other map only between the two calls, so the kinds are uploaded under the registrars' pause lock
                with gate:
                    shared.setInputCloud(Map_buffer.CORNER, maps[k][0])
                    shared.setInputCloud(Map_buffer.SURF, maps[k][1])
        except Exception as e:  # pragma: no cover
            errors.append(e)

    gate = threading.Lock()

    def registrar():
        try:
            reg = Point_cloud_registration(max_scans=1, max_features=24000)
            set_params(reg, 4, 20, 1)
            reg.upload_features([fc], [fs])
            for _ in range(12):
                with gate:  # pin a consistent pair of snapshots ...
                    reg.enqueue_uploaded(shared, 1, sc.pose_init[None], sc.pose_init[None])
                res, pc, _, reps = reg.collect(1)  # ... the kernels run while the refresher is already uploading again
                results.append((int(res[0]), pc[0].copy(), reps[0].n_blocks_last))
            reg.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=refresher)] + [threading.Thread(target=registrar) for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads[1:]:
        t.join()
    stop.set()
    threads[0].join()
    shared.close()
    assert not errors, errors
    assert len(results) == 24
    seen = set()
    for ret, pose, nb in results:
        match = [i for i in range(2) if ret == serial[i][0] and np.array_equal(pose, serial[i][1]) and nb == serial[i][2]]
        assert match, "a registration mixed two map snapshots"
        seen.add(match[0])
    assert seen == {0, 1}  # the refresher really did swap maps under the registrars


def test_library_refuses_the_retired_solver_forms(gpu_lib):
    """ll_reg_set_debug bits 4 / 6 selected the round-1 / round-2 solver forms, retired in round 6: refused, not ignored"""
    from loam_livox_amd.capi import LoamLivoxError, check
    reg = Point_cloud_registration(max_scans=1, max_features=1000)
    for bits in (16, 64):
        with pytest.raises(LoamLivoxError, match="retired"):
            check(reg.L.ll_reg_set_debug(reg.h, bits), "ll_reg_set_debug")
    reg.set_debug(False)
    reg.close()


def test_wavefront_line_search_fit_equals_the_sequential_one(gpu_lib):
    """The solver fits a line search's three-sample interpolant on the controller's whole wavefront (grid points side by side, the cells
    with a root bisected side by side): to the bit the step of the sequential form on the device, and of the same code compiled for
    the host (which tests/test_hostcheck.py / test_ref_pin.py hold to the oracle and to the Ceres stand-in)."""
    import ctypes as C
    from loam_livox_amd.capi import ptr
    from tests.hostcheck import hc
    rng = np.random.default_rng(31)
    args = []
    for i in range(4000):
        x1 = float(10.0 ** rng.uniform(-4, 0))
        x2 = x1 * float(rng.uniform(1.2, 8.0))
        if i % 5 == 0:   # samples of a wiggly function: several roots of the derivative inside [lo, hi]
            w, ph, a0, s = rng.uniform(5, 60) / x1, rng.uniform(0, 6.28), rng.uniform(0.1, 3.0), rng.uniform(-2, 0.5)
            f = lambda x: a0 * np.sin(w * x + ph) + s * x / x1
            g = lambda x: a0 * w * np.cos(w * x + ph) + s / x1
        else:            # samples of a random quintic (descent at 0)
            c = rng.normal(size=6) * np.array([1, 1, 1 / x1, 1 / x1 ** 2, 1 / x1 ** 3, 1 / x1 ** 4])
            c[1] = -abs(c[1]) - 1e-3
            f = lambda x: float(np.polyval(c[::-1], x))
            g = lambda x: float(np.polyval(np.polyder(c[::-1]), x))
        row = [f(0.0), g(0.0), x1, f(x1), g(x1), x2, f(x2), g(x2), 1e-3 * x1, 0.6 * x1]
        if i % 97 == 0:
            row[5] = row[2]      # coincident samples
        if i % 101 == 0:
            row[3] = np.inf      # a non-finite sample value
        if i % 103 == 0:
            row[1] = row[4] = row[7] = 0.0   # flat derivative samples
        args.append(row)
    a = np.ascontiguousarray(args, np.float64)
    seq, wav = np.zeros(len(a)), np.zeros(len(a))
    assert gpu_lib.ll_debug_quintic(0, ptr(a), len(a), ptr(seq), ptr(wav)) == 0
    host = np.array([hc.quintic_min_step(*r) for r in a])
    same = lambda x, y: np.array_equal(x.view(np.uint64), y.view(np.uint64))
    assert same(wav, seq) and same(seq, host)
    inner = (seq != a[:, 8]) & (seq != a[:, 9])
    assert inner.sum() > 500     # a root of the derivative won, not an end point, in a good share of the cases
