"""-m gpu: the configuration bench.py's headline number is measured in, under pytest.

bench.py runs batches of 256 24 000-point scans against the 5 M-point map with the library's DEFAULT switches: for batches of
more than 16 scans that is the query sort + the tile search (ll_knn_tile.h) in every ICP iteration and the one-workgroup
plane-table solver -- a path the small parity batches (B = 1 ... 4, where ll_api.hip switches the tile search off) never reach --
and it keeps three batches in flight on three extractor handles / registrars / stream sets that share one map.

  (a) B = 32 slots, default switches, slots filled from the four scans the REFERENCE'S OWN build registered
      (tests/golden/ref_c2_scene*.npz, written by tests/golden/gen_ref_c2.py from oracle/_ref/libll_ref.so): index sets exact,
      pose < 1e-7 m / rad from the reference's, block counts and costs; the 5-NN lists of ICP iterations 0 and 9 equal to the k-d
      tree's (point_cloud_registration.hpp:249, 351), index and squared distance, bit for bit.
  (b) three slots in flight in bench.py's own schedule (bench.pipeline_schedule), DIFFERENT scans per slot, the map re-published
      (ll_map_upload) while batches are in flight: every batch bit-equal to the same batch run alone.
  (c) the Mid-100 / motion-deblur / 20 M-point configuration (C3) at B = 17 is in tests/test_gpu_c3_c5.py.
"""
import glob
import os
import zlib

import numpy as np
import pytest

from loam_livox_amd import synth
from loam_livox_amd.api import Livox_laser, Map_buffer, Point_cloud_registration
from oracle import orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_c2_scene*.npz")))
N = 24000


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@pytest.fixture(scope="module")
def c2(gpu_lib):
    g0 = np.load(SCENES[0])
    world, corner, surf = synth.make_maps(int(g0["map_points"]))
    assert crc(corner) == int(g0["corner_crc"]) and crc(surf) == int(g0["surf_crc"]), "the synthetic 5 M-point map changed"
    m = Map_buffer()
    m.setInputCloud(Map_buffer.CORNER, corner)
    m.setInputCloud(Map_buffer.SURF, surf)
    fixtures = [np.load(p) for p in SCENES]
    scans = []
    for g in fixtures:
        sc = synth.make_scan(world, int(g["scan_seed"]))
        assert crc(sc.xyzi) == int(g["scan_crc"]), "the synthetic scan changed"
        scans.append(sc)
    yield dict(world=world, corner=corner, surf=surf, map=m, fixtures=fixtures, scans=scans)
    m.close()


def set_params(reg, g, force, icp=None):
    p = reg.params
    p.icp_max_iterations, p.ceres_max_iterations, p.force_all_iterations = int(g["icp_iters"]) if icp is None else icp, int(g["ceres_iters"]), force
    p.para_max_angular_rate, p.para_max_speed, p.max_final_cost = 20.0, 0.3, float(g["max_final_cost"])
    p.current_frame_index, p.mapping_init_accumulate_frames = 100, 50
    p.maximum_allow_residual_block = N
    return p


def extracted_batch(scans, B, stamp):
    fe = Livox_laser(max_points=N, max_scans=B, piecewise_number=1)
    fe.upload(np.stack([scans[b % len(scans)].xyzi for b in range(B)]), np.full(B, stamp))
    fe.extract_batch(B); fe.resolve(); fe.select_batch(B, -1, 0.0, 1.0)
    return fe


def test_default_fast_path_b32_reproduces_the_reference_fixtures(c2):
    """(a): tile search + one-workgroup plane-table solver, as B = 256 runs them, against the reference's own outputs"""
    B, S = 32, len(c2["scans"])
    assert S >= 4
    fe = extracted_batch(c2["scans"], B, float(c2["fixtures"][0]["stamp"]))
    nc, ns, nf, _ = fe.counts(B)
    for b in range(B):  # integer artefacts: exact
        f, g = fe.get_features(0.0, 1.0, scan=b), c2["fixtures"][b % S]
        assert np.array_equal(f["corner_idx"], g["corner_idx"]) and np.array_equal(f["surf_idx"], g["surf_idx"]), b
    inits = np.stack([c2["scans"][b % S].pose_init for b in range(B)])

    # -- the reference's configuration (the convergence break of PCR:521-526 enabled), default switches
    reg = Point_cloud_registration(max_scans=B, max_features=N)
    set_params(reg, c2["fixtures"][0], force=0)
    res, pc, pi, reps = reg.solve_batch_fe(c2["map"], fe, B, inits, inits)
    for b in range(B):
        g = c2["fixtures"][b % S]
        dt, dr = synth.pose_error(pc[b], g["pose_out"])
        assert res[b] == int(g["reg_ret"]), b
        assert dt < 1e-7 and dr < 1e-7, (b, dt, dr)  # (north-star tolerance: 1e-4)
        assert reps[b].n_blocks_last == int(g["n_blocks_last"])
        assert abs(reps[b].final_cost - float(g["final_cost"])) < 1e-7 * max(1.0, float(g["final_cost"]))
        assert abs(reps[b].inlier_threshold - float(g["inlier_threshold"])) < 1e-7
        if b >= S:  # a scan's answer does not depend on its slot
            assert np.array_equal(pc[b], pc[b % S])
    reg.close()

    # -- neighbour lists against the k-d tree at ICP iterations 0 and 9 (forced iterations, so that iteration 9 exists for every scan)
    tc, ts = orc.KdTree(c2["corner"]), orc.KdTree(c2["surf"])
    feats = [fe.get_features(0.0, 1.0, scan=b) for b in range(S)]
    reg9 = Point_cloud_registration(max_scans=B, max_features=N)
    set_params(reg9, c2["fixtures"][0], force=1, icp=9)
    _, pose9, _, _ = reg9.solve_batch_fe(c2["map"], fe, B, inits, inits)  # the pose ICP iteration 9 transforms the queries with
    reg9.close()
    for it, poses in ((0, inits), (9, pose9)):
        reg = Point_cloud_registration(max_scans=B, max_features=N)
        reg.set_debug(True)
        reg.set_debug_knn_iteration(it)
        set_params(reg, c2["fixtures"][0], force=1, icp=10)
        reg.solve_batch_fe(c2["map"], fe, B, inits, inits)
        for b in (0, 1, 2, 3, B - 1):
            ci, cd, si, sd = reg.debug_knn(b, int(nc[b]), int(ns[b]))
            f = feats[b % S]
            qs = synth.transform_points(poses[b], f["pc_surface"][:, :3])
            oi, od = ts.knn(qs, 5)
            assert np.array_equal(oi, si) and np.array_equal(od, sd), (it, b)
            qc = synth.transform_points(poses[b], f["pc_corners"][:, :3])
            oi, od = tc.knn(qc, 5)
            inside = od < float(np.float32(reg.params.maximum_dis_line_for_match))
            # (a corner list is recorded as found only when all five lie inside the match radius: compare those)
            full = inside.all(axis=1)
            assert np.array_equal(oi[full], ci[full]) and np.array_equal(od[full], cd[full]), (it, b)
        reg.close()
    fe.close()


def test_three_batches_in_flight_equal_one_at_a_time(c2):
    """(b): bench.py's loop -- three slots (extractor handle + registrar + streams each) sharing one map, batch i + 2 started before
    batch i is collected -- with different scans per slot and the map re-published while batches are in flight"""
    import bench
    B, D, K = 32, 3, 6
    world = c2["world"]
    extra = [synth.make_scan(world, 2000 + k) for k in range(8)]
    pool = c2["scans"] + extra
    rng = np.random.default_rng(77)

    def batch_inputs(j):  # slot j's own scans and initial guesses
        idx = [(5 * j + 3 * b) % len(pool) for b in range(B)]
        xyzi = np.stack([pool[i].xyzi for i in idx])
        init = np.stack([synth.pose_compose(pool[i].pose_true, np.r_[synth.quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 1.0))),
                                                                       rng.uniform(-0.1, 0.1, 3)]) for i in idx])
        return xyzi, init

    inputs = [batch_inputs(j) for j in range(D)]

    def make_slot(j):
        fe = Livox_laser(max_points=N, max_scans=B, piecewise_number=1)
        fe.upload(inputs[j][0], np.full(B, 1.0))
        reg = Point_cloud_registration(max_scans=B, max_features=N)
        set_params(reg, c2["fixtures"][0], force=1)
        return fe, reg

    def start(fe, reg, init):
        fe.extract_batch(B); fe.resolve(); fe.select_batch(B, -1, 0.0, 1.0)
        reg.enqueue_fe(c2["map"], fe, B, init, init)

    # one at a time
    alone = []
    for j in range(D):
        fe, reg = make_slot(j)
        start(fe, reg, inputs[j][1])
        res, pc, pi, reps = reg.collect(B)
        alone.append((res.copy(), pc.copy(), pi.copy(), [(r.lm_iterations_total, r.n_blocks_last, r.icp_iterations) for r in reps]))
        fe.close(); reg.close()
    assert not np.array_equal(alone[0][1], alone[1][1])  # the slots really hold different work

    # in flight, the surface map re-published (same points: a new snapshot, the old one pinned by the batches that started on it)
    slots = [make_slot(j) for j in range(D)]
    got = {}
    n_started = [0]

    def start_slot(j):
        start(slots[j][0], slots[j][1], inputs[j][1])
        n_started[0] += 1
        if n_started[0] in (2, 4):
            c2["map"].setInputCloud(Map_buffer.SURF, c2["surf"])

    def collect_slot(j):
        res, pc, pi, reps = slots[j][1].collect(B)
        got.setdefault(j, []).append((res.copy(), pc.copy(), pi.copy(), [(r.lm_iterations_total, r.n_blocks_last, r.icp_iterations) for r in reps]))

    bench.pipeline_schedule(K, D, start_slot, collect_slot)
    for j in range(D):
        assert len(got[j]) == K // D
        for o in got[j]:
            assert np.array_equal(o[0], alone[j][0]) and np.array_equal(o[1], alone[j][1]) and np.array_equal(o[2], alone[j][2]) and o[3] == alone[j][3], j
    for fe, reg in slots:
        fe.close(); reg.close()
