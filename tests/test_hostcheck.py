"""No-GPU tier: the device math headers (loam_livox_amd/csrc/ll_*_core.h), compiled for the host by
tests/hostcheck, against the oracle.  This checks the arithmetic the kernels execute per thread; the kernels
themselves (launch geometry, LDS, reductions, compaction) are checked by the -m gpu tests."""
import numpy as np
import pytest

from loam_livox_amd import synth
from oracle import orc
from tests.conftest import oracle_features
from tests.hostcheck import hc


def hc_fe_params():
    p = orc.FeParams.node_defaults()
    return hc.FeParams(*[getattr(p, f[0]) for f in p._fields_])


def hc_reg_params(icp=10, ceres=20, force=1, bound=0.3, max_cost=100.0):
    return hc.RegParams(0, icp, ceres, 2, 1, 1, force, 2.0, 50.0, 0.1, 0.02, 0.8, 0.01, 0.01, bound, 20.0, max_cost, 0.0, 1.0)


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_point_math_bit_exact(scans, k):
    sc = scans[k]
    fe = orc.fe_extract(sc.xyzi, 1.0)
    h = hc.fe_points(sc.xyzi, 1.0, hc_fe_params())
    assert np.array_equal(h["type"], fe.pt_type)
    assert np.array_equal(h["label"], fe.pt_label)
    assert np.array_equal(h["depth2"], fe.depth_sq2)
    assert np.array_equal(h["curv"], fe.curvature)
    assert np.array_equal(h["view"], fe.view_angle)
    assert np.array_equal(h["tstamp"], fe.time_stamp)
    for (lo, hi) in ((0.0, 1.0), (0.0, 0.3), (0.31, 0.67)):
        o = orc.fe_get_features(fe, lo, hi)
        d = hc.select(h["type"], h["label"], h["depth2"], lo, hi)
        assert all(np.array_equal(a, b) for a, b in zip(o, d))


def test_point_math_edge_cases():
    rng = np.random.default_rng(11)
    n = 3000
    p = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    p[:, 0] = rng.uniform(0.0, 6.0, n)
    p[:, 3] = rng.uniform(0, 100, n)
    p[rng.uniform(size=n) < 0.05, :3] = 0.0
    p[rng.uniform(size=n) < 0.03, 0] = 0.0          # x == 0 but y,z != 0
    p[rng.uniform(size=n) < 0.03, 1] = np.nan
    p[rng.uniform(size=n) < 0.01, 2] = np.inf
    p[0, :3] = 0.0                                    # first point zero: division by zero path (LFE:495-504)
    fe = orc.fe_extract(p, 3.0)
    h = hc.fe_points(p, 3.0, hc_fe_params())
    assert np.array_equal(h["type"], fe.pt_type) and np.array_equal(h["label"], fe.pt_label)
    assert np.array_equal(h["curv"], fe.curvature, equal_nan=True)


def test_grid_knn_identical_to_kdtree(small_world, scans):
    gs, gc = hc.Grid(small_world["surf"], 1.0), hc.Grid(small_world["corner"], 0.5)
    for sc in scans[:2]:
        _, _, _, _, fc, fs = oracle_features(sc)
        qs = synth.transform_points(sc.pose_init, fs[::7, :3])
        oi, od = small_world["tree_s"].knn(qs, 5)
        hi, hd = gs.knn5(qs, 50.0)
        assert np.array_equal(oi, hi) and np.array_equal(od, hd)
        qc = synth.transform_points(sc.pose_init, fc[:, :3])
        oi, od = small_world["tree_c"].knn(qc, 5)
        hi, hd = gc.knn5(qc, 2.0)
        # within the match radius the lists are identical; beyond it entries are -1/inf (never used: PCR:254)
        inside = od < 2.0
        assert np.array_equal(np.where(inside, oi, -1), hi)
        assert np.array_equal(np.where(inside, od, np.inf), hd)


def test_grid_knn_sparse_outside_and_ties():
    rng = np.random.default_rng(2)
    pts = rng.uniform(0, 30, (4000, 3)).astype(np.float32)
    pts[100:104] = pts[100]                      # exact duplicates -> ties by index
    pts[7, 0] = np.nan                           # non-finite map point is ignored
    tree = orc.KdTree(np.where(np.isfinite(pts), pts, 1e9).astype(np.float32))
    g = hc.Grid(pts, 0.7)
    q = np.concatenate([rng.uniform(-8, 38, (500, 3)), pts[100:101], [[1e6, 0, 0]], [[np.nan, 0, 0]]]).astype(np.float32)
    hi, hd = g.knn5(q, 50.0)
    oi, od = tree.knn(np.nan_to_num(q, nan=1e9), 5)
    inside = od < 50.0
    assert np.array_equal(np.where(inside, oi, -1)[:-2], hi[:-2])
    assert np.all(hi[-2:] == -1)
    assert hi[500].tolist()[:4] == [100, 101, 102, 103]


def _cell_order(q, cell, origin):
    """queries grouped by grid cell, x fastest (what reg_qsort_kernel produces; any order is valid, this one is the efficient one)"""
    c = np.floor((np.nan_to_num(q, nan=0.0) - origin) / cell).astype(np.int64)
    key = (c[:, 2] * 100003 + c[:, 1]) * 100003 + c[:, 0]
    return np.argsort(key, kind="stable")


def test_tile_search_equals_per_lane_search_on_the_scan_workload(small_world, scans):
    """ll_knn_tile.h (host model of the wavefront: rounds, tiles, min / med3 network, k = 1 termination test, fall-back): the same
    lists as knn5_search, index for index and bit for bit, valid lb2, and nearly every lane settled by its tile"""
    gs = hc.Grid(small_world["surf"], 0.6)
    for sc in scans[:2]:
        _, _, _, _, fc, fs = oracle_features(sc)
        q = synth.transform_points(sc.pose_init, fs[:, :3]).astype(np.float32)
        q = q[_cell_order(q, 0.6, small_world["surf"][:, :3].min(axis=0))]
        hi, hd = gs.knn5(q, 50.0)
        ti, td, lb2, stats = gs.knn5_tile(q, 50.0)
        assert np.array_equal(hi, ti) and np.array_equal(hd, td)
        assert stats[2] < 0.2 * len(q), stats  # (the small test map is sparser than C2's: more lanes need the rings)
        # also in scan order (many cells per wavefront -> several rounds per wavefront): still exact
        q2 = synth.transform_points(sc.pose_init, fs[:, :3]).astype(np.float32)
        hi2, hd2 = gs.knn5(q2, 50.0)
        ti2, td2, _, stats2 = gs.knn5_tile(q2, 50.0)
        assert np.array_equal(hi2, ti2) and np.array_equal(hd2, td2)
        assert stats2[0] >= stats[0]


@pytest.mark.parametrize("cell,max_d2,npts", [(0.7, 50.0, 4000), (0.5, 2.0, 30000), (1.3, 9.0, 1500), (0.6, 50.0, 200000),
                                               (1.5, 9.0, 200000),   # tiles of more than 256 points, several passes
                                               (1.45, 2.0, 3000), (1.45, 2.0, 40000)])  # the corner map: the radius lies inside one cell -- a tile
                                                                                         # settles "fewer than five" too
def test_tile_search_sparse_dense_ties_and_outside(cell, max_d2, npts):
    rng = np.random.default_rng(21)
    pts = rng.uniform(0, 30, (npts, 3)).astype(np.float32)
    pts[100:104] = pts[100]          # exact duplicates: ties by original index, decided by the fall-back
    pts[200:206, :] = pts[200] + np.float32(0.0)
    pts[7, 0] = np.nan
    g = hc.Grid(pts, cell)
    q = np.concatenate([rng.uniform(-3, 33, (3000, 3)), pts[100:101], pts[200:201] + 1e-3, [[1e6, 0, 0]], [[np.nan, 0, 0]],
                        rng.uniform(10, 12, (1000, 3))]).astype(np.float32)
    for order in (np.arange(len(q)), _cell_order(q, cell, np.nanmin(pts, axis=0))):
        qq = q[order]
        hi, hd = g.knn5(qq, max_d2)
        ti, td, lb2, stats = g.knn5_tile(qq, max_d2)
        assert np.array_equal(hi, ti) and np.array_equal(hd, td)
        # lb2 bounds every point outside a full list
        fin = np.isfinite(qq).all(axis=1) & (ti[:, 4] >= 0)
        sel = np.flatnonzero(fin)[:300]
        P = np.where(np.isfinite(pts), pts, 1e9).astype(np.float64)
        for i in sel:
            d2_all = ((qq[i].astype(np.float64) - P) ** 2).sum(-1)
            rest = np.delete(d2_all, ti[i])
            assert lb2[i] <= rest.min() * (1 + 1e-5) + 1e-6
    if npts >= 200000:
        assert stats[2] < 0.5 * len(q)  # a dense cloud: the tiles settle most lanes
    if cell == 1.45:
        # only ties, queries more than a cell outside and non-finite ones are left to the per-lane search -- although most lists are short
        assert (ti[:, 4] < 0).sum() > 0.1 * len(q) or npts > 3000
        assert stats[2] < 0.2 * len(q), stats


def test_tile_search_settles_queries_just_outside_the_grid():
    """The grid is the bounding box of the map's points, so scan points on an outer wall fall outside it half of the time (1 % of the
    C2 queries).  tile_query adopts the nearest cell for a query up to one cell outside: same lists as knn5_search, and the tile
    settles them -- only queries farther out (or with their fifth neighbour beyond one cell) take the per-lane search."""
    rng = np.random.default_rng(77)
    n = 60000
    wall = np.c_[rng.normal(0.0, 0.01, n), rng.uniform(0, 20, n), rng.uniform(0, 6, n)].astype(np.float32)      # the outer wall x = 0
    floor = np.c_[rng.uniform(0, 20, n), rng.uniform(0, 20, n), rng.normal(0.0, 0.01, n)].astype(np.float32)   # and the floor z = 0
    pts = np.concatenate([wall, floor])
    g = hc.Grid(pts, 0.6)
    lo = pts.min(axis=0)
    nq = 4096
    q = np.c_[lo[0] - rng.uniform(0.0, 0.05, nq), rng.uniform(1, 19, nq), rng.uniform(1, 5, nq)].astype(np.float32)     # outside in -x
    q2 = np.c_[rng.uniform(1, 19, nq), rng.uniform(1, 19, nq), lo[2] - rng.uniform(0.0, 0.5, nq)].astype(np.float32)    # outside in -z, up to a cell
    q3 = np.c_[lo[0] - rng.uniform(0.61, 3.0, 256), rng.uniform(1, 19, 256), rng.uniform(1, 5, 256)].astype(np.float32)  # more than a cell out
    for qq, settled in ((q, True), (q2, None), (q3, False)):
        qq = qq[_cell_order(qq, 0.6, lo)]
        hi, hd = g.knn5(qq, 50.0)
        ti, td, lb2, stats = g.knn5_tile(qq, 50.0)
        assert np.array_equal(hi, ti) and np.array_equal(hd, td)
        assert (ti[:, 4] >= 0).all()
        if settled is True:
            assert stats[2] < 0.02 * len(qq), stats
        if settled is False:
            assert stats[2] == len(qq), stats
        P = pts.astype(np.float64)
        for i in range(0, len(qq), 97):  # lb2 stays a valid bound on everything outside the list
            d2_all = ((qq[i].astype(np.float64) - P) ** 2).sum(-1)
            assert lb2[i] <= np.delete(d2_all, ti[i]).min() * (1 + 1e-5) + 1e-6


def test_tile_offer_network_equals_ordered_insertion():
    """tile5_offer against a sort: random streams with repeated values"""
    rng = np.random.default_rng(5)
    pts = np.zeros((600, 3), np.float32)
    pts[:, 0] = rng.integers(0, 40, 600).astype(np.float32) * 0.01   # many equal distances along one axis
    pts[:, 1:] = rng.uniform(0, 0.3, (600, 2)).astype(np.float32) * 0
    g = hc.Grid(pts + np.float32(5.0), 0.6)
    q = np.array([[5.2, 5.0, 5.0], [5.0, 5.0, 5.0], [5.39, 5.0, 5.0]], np.float32)
    hi, hd = g.knn5(q, 50.0)
    ti, td, _, stats = g.knn5_tile(q, 50.0)
    assert np.array_equal(hi, ti) and np.array_equal(hd, td)
    assert stats[2] == 3  # every list has ties: all three lanes fall back


@pytest.mark.parametrize("guard", [0.0, 0.05, 0.5])
@pytest.mark.parametrize("cell,max_d2,npts", [(0.7, 50.0, 4000), (0.5, 2.0, 30000), (1.3, 9.0, 1500)])
def test_grid_knn_reuse_bounds_are_valid(cell, max_d2, npts, guard):
    """the search keeps LL_KNN_K candidates; lb2 / out2 must bound every point outside that list whatever was pruned -- with and
    without the reuse guard band (runs pruned only beyond the 5th best + guard: same lists, larger displacement budgets)"""
    rng = np.random.default_rng(11)
    pts = rng.uniform(0, 30, (npts, 3)).astype(np.float32)
    g = hc.Grid(pts, cell)
    q = rng.uniform(-2, 32, (400, 3)).astype(np.float32)
    hi, hd = g.knn5(q, max_d2)
    m0 = g.knn5_bounds(q, max_d2)[3]
    g.set_guard(guard)
    gi, gd = g.knn5(q, max_d2)
    assert np.array_equal(gi, hi) and np.array_equal(gd, hd)
    cand, lb2, out2, m_set, m_strong = g.knn5_bounds(q, max_d2)
    assert np.all(m_set >= m0 - 1e-7)  # never a smaller budget (on a sparse cloud like this one mostly the same: rings decide)
    assert np.array_equal(cand[:, :5], hi)
    d2_all = ((q[:, None, :].astype(np.float64) - pts[None, :, :]) ** 2).sum(-1)
    for i in range(len(q)):
        listed = cand[i][cand[i] >= 0]
        if len(listed) >= 5:
            rest = np.delete(d2_all[i], listed)
            assert rest.size == 0 or lb2[i] <= rest.min() * (1 + 1e-5) + 1e-6
        else:  # out2 only has to hold while fewer than five neighbours are inside the radius (knn5_reuse_margin)
            far = d2_all[i][d2_all[i] >= max_d2]
            assert far.size == 0 or out2[i] <= far.min() * (1 + 1e-5) + 1e-6
        inside = np.sort(d2_all[i][d2_all[i] < max_d2])
        assert len(listed) == min(cand.shape[1], len(inside))
    assert (m_set > 0).mean() > 0.5 and np.all(m_strong <= m_set)


@pytest.mark.parametrize("cell,max_d2,npts,step", [(0.7, 50.0, 4000, 0.05), (0.5, 2.0, 30000, 0.02), (1.3, 9.0, 1500, 0.3),
                                                   (0.6, 50.0, 60000, 0.01)])
@pytest.mark.parametrize("guard", [0.0, 0.05])
def test_knn_reuse_chain_is_exact(cell, max_d2, npts, step, guard):
    """the registrar's reuse over a chain of moves (keep / re-sort the 8 candidates / search) returns at every hop exactly
    what a fresh search returns"""
    rng = np.random.default_rng(12)
    pts = rng.uniform(0, 30, (npts, 3)).astype(np.float32)
    g = hc.Grid(pts, cell)
    g.set_guard(guard)
    nq, n_hops = 600, 6
    path = np.zeros((n_hops, nq, 3), np.float32)
    path[0] = rng.uniform(-1, 31, (nq, 3))
    for h in range(1, n_hops):  # shrinking moves, like successive ICP iterations
        d = rng.normal(size=(nq, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        path[h] = path[h - 1] + (step / h) * rng.uniform(0.2, 1.0, (nq, 1)) * d
    idx, st = g.knn5_reuse_chain(path, max_d2)
    for h in range(n_hops):
        hi, hd = g.knn5(path[h], max_d2)
        fresh = np.where(((hi >= 0).sum(1) == 5)[:, None], hi, -1)
        assert np.array_equal(idx[h], fresh), f"hop {h}"
    assert np.all(st[0] == 2)
    assert (st[1:] == 2).mean() < 0.6 and (st[1:] < 2).any()
    if step <= 0.02 and hc.lib().hc_knn_k() >= 8:
        assert (st[2:] == 2).mean() < 0.05   # small moves: 8 candidates almost always cover the answer


def test_analytic_gauss_newton_matches_jets():
    rng = np.random.default_rng(9)
    pose_last = np.r_[synth.quat_from_axis_angle(rng.normal(size=3), 0.8), rng.uniform(-50, 50, 3)]
    x = np.r_[synth.quat_from_axis_angle(rng.normal(size=3), 0.03), rng.uniform(-0.1, 0.1, 3)]
    oblocks, kind, F, A, V = [], [], [], [], []
    for i in range(200):
        f = rng.uniform(-8, 8, 3)
        pw = synth.quat_to_mat(pose_last[:4]) @ f + pose_last[4:]
        a = pw + rng.normal(0, 0.04 if i % 4 else 0.4, 3)
        b, c = a + rng.normal(size=3), a + rng.normal(size=3)
        if i % 2:
            oblocks.append(orc.make_block_line(f, a, b))
            ok, aa, vv = hc.make_block(1, pose_last, a, b)
        else:
            oblocks.append(orc.make_block_plane(f, a, b, c))
            ok, aa, vv = hc.make_block(2, pose_last, a, b, c)
        assert ok
        kind.append(1 if i % 2 else 2)
        F.append(f); A.append(aa); V.append(vv)
    oc, og, oH = orc.blocks_eval(oblocks, pose_last, x)
    hc_c, hg, hH = hc.eval_blocks(kind, np.array(F), np.array(A), np.array(V), x)
    assert np.isclose(oc, hc_c, rtol=1e-12)
    assert np.allclose(og, hg, rtol=1e-9, atol=1e-12)
    assert np.allclose(oH, hH, rtol=1e-9, atol=1e-10)


def test_degenerate_blocks_are_skipped():
    ident = np.array([0, 0, 0, 1, 0, 0, 0], float)
    assert hc.make_block(1, ident, [1, 1, 1], [1, 1, 1 + 5e-5])[0] == 0      # |a-b| < 1e-4 (PCR:302)
    assert hc.make_block(2, ident, [1, 1, 1], [1, 1, 1], [2, 2, 2])[0] == 0  # a == b: NaN normal in the reference


@pytest.mark.parametrize("k", [0, 1, 2])
def test_full_registration_matches_oracle(small_world, scans, k):
    sc = scans[k]
    _, _, _, _, fc, fs = oracle_features(sc)
    gc, gs = hc.Grid(small_world["corner"], 0.5), hc.Grid(small_world["surf"], 1.0)
    for force in (0, 1):
        prm = orc.RegParams.defaults(icp_iters=6, ceres_iters=20, force_all=force)
        ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
        hret, hpc, hpi, hrep = hc.reg_solve(gc, gs, fc, fs, hc_reg_params(6, 20, force), sc.pose_init, sc.pose_init)
        dt, dr = synth.pose_error(pc, hpc)
        assert ret == hret and dt < 1e-9 and dr < 1e-9
        assert rep.icp_iterations == hrep[3] and rep.n_blocks_last == hrep[4]
        assert rep.corner_avail == hrep[5] and rep.surf_avail == hrep[6] and rep.lm_iterations_total == hrep[7]
        assert np.isclose(rep.final_cost, hrep[0], rtol=1e-9) and np.isclose(rep.inlier_threshold, hrep[2], rtol=1e-9)


def test_bounded_line_search_with_three_sample_interpolation_matches_oracle(small_world, scans):
    """A start 0.25 m outside a 0.05 m bound on t_inc (PCR:143-151): the projected Armijo line search contracts up to ten times,
    from the second contraction on through the three-sample quintic of Ceres' line_search.cc (lm_quintic_min_step in
    ll_reg_core.h, quintic_min_step in the oracle, quintic_min in the Ceres stand-in of oracle/_ref -- tests/test_ref_pin.py holds
    the oracle to that build on the same case)."""
    import ctypes as C
    sc = scans[0]
    _, _, _, _, fc, fs = oracle_features(sc)
    gc, gs = hc.Grid(small_world["corner"], 0.5), hc.Grid(small_world["surf"], 1.0)
    L = orc.lib()
    L.orc_dbg_ls_max_contractions.argtypes, L.orc_dbg_ls_max_contractions.restype = [C.c_int], C.c_int
    for bound, off in ((0.05, [0.25, -0.2, 0.1]), (0.02, [0.3, 0.3, 0.3])):
        start = sc.pose_init.copy()
        start[4:7] += off
        prm = orc.RegParams.defaults(icp_iters=4)
        prm.para_max_speed = bound
        L.orc_dbg_ls_max_contractions(1)
        ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, start, start)
        assert L.orc_dbg_ls_max_contractions(0) >= 2  # the quintic ran
        hret, hpc, hpi, hrep = hc.reg_solve(gc, gs, fc, fs, hc_reg_params(4, 20, 0, float(np.float32(bound))), start, start)
        dt, dr = synth.pose_error(pc, hpc)
        assert ret == hret and dt < 1e-9 and dr < 1e-9
        assert rep.lm_iterations_total == hrep[7] and rep.icp_iterations == hrep[3]
        assert np.all(np.abs(hpi[4:7]) <= float(np.float32(bound)) + 1e-15)


def test_rejection_matches_oracle(small_world, scans):
    sc = scans[0]
    _, _, _, _, fc, fs = oracle_features(sc)
    gc, gs = hc.Grid(small_world["corner"], 0.5), hc.Grid(small_world["surf"], 1.0)
    prm = orc.RegParams.defaults(icp_iters=2)
    prm.max_final_cost = 1e-6
    ret, pc, _, _ = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    hret, hpc, _, _ = hc.reg_solve(gc, gs, fc, fs, hc_reg_params(2, 20, 0, 0.3, 1e-6), sc.pose_init, sc.pose_init)
    assert ret == hret == 0 and np.array_equal(pc, hpc)


def test_motion_deblur_gauss_newton_matches_jets():
    """SO(3)-Jacobian form of the *_mb blocks vs the oracle's dual numbers through Eigen's slerp (ceres_icp.hpp:116)."""
    rng = np.random.default_rng(19)
    pose_last = np.r_[synth.quat_from_axis_angle(rng.normal(size=3), 0.8), rng.uniform(-50, 50, 3)]
    for ang in (0.0, 1e-9, 1e-5, 0.03, 0.5):
        x = np.r_[synth.quat_from_axis_angle(rng.normal(size=3), ang), rng.uniform(-0.1, 0.1, 3)]
        oblocks, kind, F, A, V, S = [], [], [], [], [], []
        for i in range(120):
            f, s = rng.uniform(-8, 8, 3), rng.uniform(0.0, 1.0)
            pw = synth.quat_to_mat(pose_last[:4]) @ f + pose_last[4:]
            a = pw + rng.normal(0, 0.04 if i % 4 else 0.4, 3)
            b, c = a + rng.normal(size=3), a + rng.normal(size=3)
            if i % 2:
                oblocks.append(orc.make_block_line(f, a, b, s)); ok, aa, vv = hc.make_block(1, pose_last, a, b)
            else:
                oblocks.append(orc.make_block_plane(f, a, b, c, s)); ok, aa, vv = hc.make_block(2, pose_last, a, b, c)
            kind.append(1 if i % 2 else 2); F.append(f); A.append(aa); V.append(vv); S.append(s)
        oc, og, oH = orc.blocks_eval(oblocks, pose_last, x, deblur=1)
        c2, g2, H2 = hc.eval_blocks(kind, np.array(F), np.array(A), np.array(V), x, sblur=S)
        assert np.isclose(oc, c2, rtol=1e-12)
        assert np.allclose(og, g2, rtol=1e-9, atol=1e-11) and np.allclose(oH, H2, rtol=1e-9, atol=1e-10)


def test_motion_deblur_registration_matches_oracle(small_world, scans):
    sc = scans[1]
    fe, ci, si, fi, fc, fs = oracle_features(sc)
    tmin, tmax = float(fe.time_stamp.min()), float(fe.time_stamp.max())
    gc, gs = hc.Grid(small_world["corner"], 0.5), hc.Grid(small_world["surf"], 0.6)
    prm = orc.RegParams.defaults(icp_iters=5, ceres_iters=20, force_all=1, deblur=1)
    prm.minimum_pt_time_stamp, prm.maximum_pt_time_stamp = tmin, tmax
    ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    hp = hc.RegParams(1, 5, 20, 2, 1, 1, 1, 2.0, 50.0, 0.1, 0.02, 0.8, 0.01, 0.01, 0.3, 20.0, 100.0, tmin, tmax)
    hret, hpc, hpi, hrep = hc.reg_solve(gc, gs, fc, fs, hp, sc.pose_init, sc.pose_init)
    dt, dr = synth.pose_error(pc, hpc)
    assert ret == hret and dt < 1e-9 and dr < 1e-9 and rep.lm_iterations_total == hrep[7] and rep.n_blocks_last == hrep[4]


def _pca_cases(rng):
    cases = []
    for _ in range(200):
        kind = rng.integers(0, 5)
        c = rng.uniform(-50, 50, 3)
        if kind == 0:      # blob
            p = c + rng.normal(0, rng.uniform(0.01, 1.0), (5, 3))
        elif kind == 1:    # noisy line
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            p = c + np.outer(rng.uniform(-1, 1, 5), d) + rng.normal(0, rng.uniform(0, 0.2), (5, 3))
        elif kind == 2:    # noisy plane
            u, v = rng.normal(size=3), rng.normal(size=3)
            p = c + np.outer(rng.uniform(-1, 1, 5), u) + np.outer(rng.uniform(-1, 1, 5), v) + rng.normal(0, rng.uniform(0, 0.1), (5, 3))
        elif kind == 3:    # exactly collinear on an axis (repeated zero eigenvalues)
            p = c + np.outer(np.arange(5.0), [0.25, 0, 0])
        else:              # all the same point
            p = np.tile(c, (5, 1))
        cases.append(p.astype(np.float32))
    return cases


def test_pca_feature_checks_match_oracle_and_numpy():
    """K7 (PCR:259-292, 357-389): Jacobi eigenvalues of the device header, the oracle's closed form and LAPACK agree,
    and so do the pass/fail decisions"""
    rng = np.random.default_rng(77)
    n_pass = [0, 0]
    for p in _pca_cases(rng):
        z = p.astype(np.float64) - p.astype(np.float64).sum(0) / 5.0
        ev_np = np.linalg.eigvalsh(z.T @ z)
        scale = max(ev_np[2], 1e-300)
        for is_plane in (0, 1):
            ok_o, ev_o = orc.pca_check(is_plane, p)
            ok_h, ev_h = hc.pca_check(is_plane, p)
            assert np.all(np.abs(ev_o - ev_np) <= 1e-9 * scale + 1e-300) and np.all(np.abs(ev_h - ev_np) <= 1e-12 * scale + 1e-300)
            expect = (ev_np[2] > 3 * ev_np[0] and ev_np[2] < 10 * ev_np[1]) if is_plane else ev_np[2] > 3 * ev_np[1]
            margin = min(abs(ev_np[2] - 3 * ev_np[0]), abs(ev_np[2] - 10 * ev_np[1])) if is_plane else abs(ev_np[2] - 3 * ev_np[1])
            if margin > 1e-6 * scale:  # away from the decision boundary all three must agree
                assert ok_o == expect and ok_h == expect
            n_pass[is_plane] += ok_h
    assert 20 < n_pass[0] < 190 and 20 < n_pass[1] < 190  # both outcomes are exercised


def noisy_corner_map(small_world, sigma=0.12):
    c = small_world["corner"].copy()
    c[:, :3] += np.random.default_rng(5).normal(0, sigma, (len(c), 3)).astype(np.float32)
    return c


@pytest.mark.parametrize("checks", [(1, 0), (0, 1), (1, 1)])
def test_registration_with_pca_checks_matches_oracle(small_world, scans, checks):
    sc = scans[1]
    _, _, _, _, fc, fs = oracle_features(sc)
    corner = noisy_corner_map(small_world)  # the clean synthetic edges always pass the line test
    tree_c = orc.KdTree(corner)
    gc, gs = hc.Grid(corner, 0.5), hc.Grid(small_world["surf"], 0.6)
    prm = orc.RegParams.defaults(icp_iters=5, ceres_iters=20, force_all=1)
    prm.if_line_feature_check, prm.if_plane_feature_check = checks
    ret, pc, pi, rep = orc.reg_solve(tree_c, small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    hp = hc.RegParams(0, 5, 20, 2, 1, 1, 1, 2.0, 50.0, 0.1, 0.02, 0.8, 0.01, 0.01, 0.3, 20.0, 100.0, 0.0, 1.0, *checks)
    hret, hpc, hpi, hrep = hc.reg_solve(gc, gs, fc, fs, hp, sc.pose_init, sc.pose_init)
    dt, dr = synth.pose_error(pc, hpc)
    assert ret == hret and dt < 1e-9 and dr < 1e-9
    assert rep.n_blocks_last == hrep[4] and rep.corner_avail == hrep[5] and rep.surf_avail == hrep[6]
    base = orc.RegParams.defaults(icp_iters=5, ceres_iters=20, force_all=1)
    _, _, _, rep0 = orc.reg_solve(tree_c, small_world["tree_s"], fc, fs, base, sc.pose_init, sc.pose_init)
    if checks[0]:  # both checks reject some neighbourhoods
        assert rep.corner_avail < rep0.corner_avail
    if checks[1]:
        assert rep.surf_avail < rep0.surf_avail


@pytest.mark.parametrize("max_blocks", [200, 3000])
def test_registration_with_subsampling_matches_oracle(small_world, scans, max_blocks):
    """a13 (PCR:232-238, 339-345, 438-458): with a seed the keep / drop rules run on the shared counter-based stream; the host
    build of the device math and the oracle make identical choices.  200 = the shipped configs' maximum_residual_blocks
    (both the feature skip and the block drop fire), 3000 exercises the block drop alone for the corner-poor scan."""
    sc = scans[1]
    _, _, _, _, fc, fs = oracle_features(sc)
    gc, gs = hc.Grid(small_world["corner"], 1.45), hc.Grid(small_world["surf"], 0.6)
    prm = orc.RegParams.defaults(icp_iters=5, ceres_iters=20, force_all=1)
    prm.maximum_allow_residual_block, prm.subsample_seed = max_blocks, 7
    ret, pc, pi, rep = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    hp = hc.RegParams(0, 5, 20, 2, 1, 1, 1, 2.0, 50.0, 0.1, 0.02, 0.8, 0.01, 0.01, 0.3, 20.0, 100.0, 0.0, 1.0, 0, 0, max_blocks, 7)
    hret, hpc, hpi, hrep = hc.reg_solve(gc, gs, fc, fs, hp, sc.pose_init, sc.pose_init)
    dt, dr = synth.pose_error(pc, hpc)
    assert ret == hret and dt < 1e-9 and dr < 1e-9
    assert rep.n_blocks_last == hrep[4] and rep.corner_avail == hrep[5] and rep.surf_avail == hrep[6] and rep.lm_iterations_total == hrep[7]
    assert rep.n_blocks_last <= max_blocks * 1.3                      # ~M blocks survive the drop (then 20 % are pruned)
    base = orc.RegParams.defaults(icp_iters=5, ceres_iters=20, force_all=1)
    _, pc0, _, rep0 = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, base, sc.pose_init, sc.pose_init)
    assert rep0.n_blocks_last > 3 * rep.n_blocks_last
    if max_blocks == 200:
        assert rep.surf_avail < 0.1 * rep0.surf_avail                 # the feature skip fired (n > 2 M): ~400 of 17 k kept
    dt, dr = synth.pose_error(pc, pc0)
    if max_blocks == 3000:
        assert dt < 0.05 and dr < 0.01                                # a fifth of the blocks, nearly the same pose
    else:
        assert dt < 0.5                                               # ~180 blocks constrain this room view only loosely
    # another seed makes other choices
    prm.subsample_seed = 8
    _, pc2, _, rep2 = orc.reg_solve(small_world["tree_c"], small_world["tree_s"], fc, fs, prm, sc.pose_init, sc.pose_init)
    assert not np.array_equal(pc2, pc)


def test_three_sample_interpolation_matches_a_dense_vandermonde_solve():
    """Ceres (polynomial.cc FindInterpolatingPolynomial + MinimizePolynomial) solves the 6 x 6 system and takes companion-matrix
    roots; the shared restatement uses divided differences and bracketed bisection.  Same minimiser on [lo, hi] to rounding."""
    rng = np.random.default_rng(5)
    worst = 0.0
    for trial in range(300):
        x1 = 10.0 ** rng.uniform(-4, 0)
        x2 = x1 * rng.uniform(1.5, 12.0)                      # the previous (larger) trial step
        f0, g0 = rng.uniform(0.1, 10.0), -rng.uniform(0.1, 10.0) / x2
        f1, f2 = f0 * rng.uniform(0.9, 1.5), f0 * rng.uniform(1.0, 4.0)
        g1, g2 = rng.normal(0, 3.0) / x2, rng.uniform(0.0, 20.0) / x2
        lo, hi = 1e-3 * x1, 0.6 * x1
        # dense solve in u = x / x2 (keeps the Vandermonde matrix well scaled): rows [u^5 .. 1] per value, [5 u^4 .. 0] per gradient
        rows, rhs = [], []
        for x, f, g in ((0.0, f0, g0), (x1, f1, g1), (x2, f2, g2)):
            u = x / x2
            rows.append([u ** k for k in range(5, -1, -1)]); rhs.append(f)
            rows.append([k * u ** (k - 1) if k > 0 else 0.0 for k in range(5, -1, -1)]); rhs.append(g * x2)
        coef_u = np.linalg.solve(np.array(rows), np.array(rhs))
        coef = coef_u / x2 ** np.arange(5, -1, -1)
        cand = [lo, hi] + [float(r.real) * x2 for r in np.roots(np.polyder(coef_u)) if abs(r.imag) < 1e-12 and lo < r.real * x2 < hi]
        vals = [np.polyval(coef, c) for c in cand]
        want_v = min(vals)
        got = hc.quintic_min_step(f0, g0, x1, f1, g1, x2, f2, g2, lo, hi)
        assert lo <= got <= hi
        # the minimum VALUE is what the line search acts on (two near-equal minima may swap places under rounding)
        got_v = float(np.polyval(coef, got))
        scale = max(abs(f0), abs(f1), abs(f2))
        worst = max(worst, (got_v - want_v) / scale)
        assert got_v - want_v <= 1e-9 * scale, (trial, got, cand, vals)
    assert worst < 1e-9


def _quintic_samples_from_poly(coef, x1, x2):
    """(f0, g0, x1, f1, g1, x2, f2, g2) of the quintic with monomial coefficients coef (highest first)"""
    d = np.polyder(coef)
    return (float(np.polyval(coef, 0.0)), float(np.polyval(d, 0.0)), x1, float(np.polyval(coef, x1)), float(np.polyval(d, x1)), x2,
            float(np.polyval(coef, x2)), float(np.polyval(d, x2)))


def _companion_minimiser(coef, lo, hi):
    """MinimizePolynomial as Ceres does it (polynomial.cc): the better end point, then the REAL PART of every eigenvalue of the derivative's
    companion matrix that lies inside [lo, hi] -- numpy.roots is that eigenvalue computation"""
    best_x, best_v = (lo, np.polyval(coef, lo)) if np.polyval(coef, lo) < np.polyval(coef, hi) else (hi, np.polyval(coef, hi))
    for r in np.roots(np.polyder(coef)):
        x = float(r.real)
        if x < lo or x > hi:
            continue
        v = np.polyval(coef, x)
        if v < best_v:
            best_x, best_v = x, v
    return best_x, float(best_v)


def test_adversarial_quintics_stationary_points_inside_one_grid_cell():
    """VERDICT r5 (weak 1, next 9 iii): rounds 2 - 5 bracketed the roots of the derivative by sign changes on a 32-cell grid; two
    stationary points inside one cell went unseen, three looked like one sign change and the bisection ended on whichever it met --
    possibly the maximum between two minima, or the shallower minimum.  Fits built to do exactly that: p' = k (x - a)(x - b)(x - c)(x - r)
    with a < b < c inside ONE of the old cells (a minimum, a maximum, a minimum; unequal spacing, so one minimum is deeper) and r outside
    the interval, and pairs (maximum + minimum) inside one cell.  On all three restatements -- the device arithmetic compiled for the host
    (ll_reg_core.h lm_quintic_min_step), the oracle (ll_oracle_reg.c quintic_min_step) and the Ceres stand-in of oracle/_ref
    (quintic_min): the same bits, and the minimiser a dense companion-matrix root finder returns (numpy.roots, what Ceres'
    FindPolynomialRoots computes), to 1e-7 of the interval in POSITION."""
    import ctypes as C
    from oracle import ref
    L = orc.lib()
    L.orc_dbg_quintic_min_step.restype = C.c_double
    L.orc_dbg_quintic_min_step.argtypes = [C.c_double] * 10
    R = ref.lib() if ref.available() else None
    if R is not None:
        R.ref_quintic_min.restype = C.c_double
        R.ref_quintic_min.argtypes = [C.c_double] * 10
    rng = np.random.default_rng(2026)
    n_triple = n_pair = 0
    for trial in range(600):
        x1 = 10.0 ** rng.uniform(-3, 0)
        x2 = x1 * rng.uniform(1.5, 8.0)
        lo, hi = 1e-3 * x1, 0.6 * x1
        w = hi - lo
        cell = w / 32.0
        k = int(rng.integers(1, 31))
        base = lo + k * cell
        if trial % 2 == 0:   # minimum, maximum, minimum inside cell k; the fourth root left of the interval
            a, b, c = np.sort(base + cell * rng.uniform(0.05, 0.95, 3))
            if min(b - a, c - b) < 0.02 * cell or abs((b - a) - (c - b)) < 0.05 * cell:
                continue
            roots = [a, b, c, -rng.uniform(0.5, 3.0) * x1]
        else:                # maximum + minimum inside cell k, the other two roots complex
            a, b = np.sort(base + cell * rng.uniform(0.05, 0.95, 2))
            if b - a < 0.02 * cell:
                continue
            roots = [a, b]
        dpoly = np.poly(roots)
        if trial % 2 == 1:
            cc, ss = rng.uniform(-2, 3) * x1, rng.uniform(0.3, 2.0) * x1
            dpoly = np.polymul(dpoly, [1.0, -2 * cc, cc * cc + ss * ss])
        coef = np.polyint(dpoly)
        coef = coef / np.abs(np.polyval(coef, [lo, hi])).max()
        args = _quintic_samples_from_poly(coef, x1, x2)
        want_x, want_v = _companion_minimiser(coef, lo, hi)
        got = [hc.quintic_min_step(*args, lo, hi), L.orc_dbg_quintic_min_step(*args, lo, hi)]
        if R is not None:
            got.append(R.ref_quintic_min(*args, lo, hi))
        assert all(g == got[0] for g in got), (trial, got)          # the three restatements: the same bits
        assert lo <= got[0] <= hi
        scale = np.abs(np.polyval(coef, [lo, hi])).max()
        assert np.polyval(coef, got[0]) - want_v <= 1e-12 * scale, (trial, got[0], want_x)
        if trial % 2 == 0:
            n_triple += 1
            va, vc = np.polyval(coef, a), np.polyval(coef, c)
            assert abs(want_x - (a if va < vc else c)) < 1e-6 * cell          # the companion matrix picks the deeper of the two minima
            assert abs(got[0] - want_x) <= 1e-7 * w, (trial, got[0], want_x, (a, b, c))   # ... and so do we
        else:
            n_pair += 1
            assert abs(got[0] - want_x) <= 1e-7 * w, (trial, got[0], want_x, (a, b))
    assert n_triple > 150 and n_pair > 150


def test_quintic_minimiser_degenerate_shapes():
    """the derivative chain's corner cases: a cubic / quadratic interpolant (leading coefficients exactly zero), a stationary point on
    an end of the interval, a triple root, no stationary point at all -- against the companion-matrix minimiser"""
    x1, x2 = 0.5, 1.25
    lo, hi = 1e-3 * x1, 0.6 * x1
    shapes = {
        "quadratic": np.array([0, 0, 0, 3.0, -1.2, 2.0]),                     # minimum at 0.2
        "cubic": np.array([0, 0, 1.0, -0.9, 0.1, 1.0]),
        "monotone": np.array([0, 0, 0, 0, -1.0, 3.0]),                        # no stationary point: the better end
        "root_on_hi": np.polyint(np.polymul([1.0, -hi], [1.0, 0, 1.0])),       # p' vanishes exactly at hi
        "triple_root": np.polyint(np.polymul(np.poly([0.1, 0.1, 0.1]), [1.0, 2.0])) + np.array([0, 0, 0, 0, 0, 1.0]),
        "two_minima": np.polyint(np.poly([0.05, 0.1, 0.2, 0.25])) * 1e3 + np.array([0, 0, 0, 0, 0, 1.0]),
    }
    for name, coef in shapes.items():
        coef = np.asarray(coef, np.float64)
        coef = np.r_[np.zeros(6 - len(coef)), coef]
        args = _quintic_samples_from_poly(coef, x1, x2)
        want_x, want_v = _companion_minimiser(coef, lo, hi)
        got = hc.quintic_min_step(*args, lo, hi)
        assert lo <= got <= hi, name
        assert np.polyval(coef, got) - want_v <= 1e-12 * max(1.0, abs(want_v)), (name, got, want_x)


def test_scaled_plane_blocks_equal_the_unscaled_forms():
    """Round 6: solve_fast3 stores the SCALED plane {m = |n'| n', beta = |n'| c} per distinct neighbour triple and evaluates a plane block as
    the scalar residual e = m.p - beta with Jacobian row (B^T m)^T, the factor 2 of the rotation rows put back once per evaluation
    (ll_reg_core.h plane_accumulate_scaled / plane_unfold2 / plane_l1_scaled).  Same cost / gradient / Gauss-Newton terms and the same
    loss-corrected L1 value as block_accumulate / block_l1 (which tests above hold to the oracle's forward-mode duals), to rounding --
    in the quadratic and in the linear region of the Huber loss, for un-normalised normals of any length."""
    rng = np.random.default_rng(66)
    worst = 0.0
    n_linear = 0
    for trial in range(400):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        ang = rng.uniform(0, 0.05)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        qi = np.r_[np.sin(ang / 2) * ax, np.cos(ang / 2)]
        x, y, z, w = qi
        R = np.array([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                      2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)])
        t = rng.normal(size=3) * 0.1
        f = rng.uniform(-30, 30, 3)
        n = rng.normal(size=3); n *= rng.uniform(0.05, 1.0) / np.linalg.norm(n)       # |n'| = sin of the angle between the two edges: anything in (0, 1]
        p = R.reshape(3, 3) @ f + t
        dist = rng.choice([rng.normal() * 0.02, rng.normal() * 0.5])                  # a few centimetres, or well into the linear Huber region
        a0 = float(n @ p - dist * np.linalg.norm(n))
        acc, ref, l1, l1_ref = hc.plane_scaled(R, t, f, n, a0, 0.1, q)
        n_linear += abs(dist) * np.linalg.norm(n) ** 2 > 0.1
        scale = np.abs(ref).max() + 1e-300
        # (the residual n'.p - c cancels two numbers of size |n'| |p| ~ 20: both forms carry that rounding, 1e-15 absolute; it is what a
        #  residual of a few microns is measured against, so the L1 values are compared on that scale)
        l1_tol = 1e-13 * np.linalg.norm(n) * (np.linalg.norm(p) + 1.0) + 1e-12 * abs(l1_ref)
        worst = max(worst, np.abs(acc - ref).max() / scale, abs(l1 - l1_ref) / l1_tol * 1e-12)
        assert np.allclose(acc, ref, rtol=0, atol=1e-11 * scale), trial
        assert abs(l1 - l1_ref) <= l1_tol, trial
    assert worst < 1e-11 and n_linear > 50
