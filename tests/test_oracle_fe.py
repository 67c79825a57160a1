"""Known-answer tests that pin the CPU oracle's feature extractor to the reference's source text
(hku-mars/loam_livox source/livox_feature_extractor.hpp).  The reference ships no tests or fixtures, so these
cases are authored from the code: each cites the lines it exercises."""
import numpy as np

from oracle import orc


def line_scan(n=40, x=5.0, dy=0.01, z=0.2, inten=50.0):
    """points on a straight line across the view at constant x"""
    p = np.zeros((n, 4), np.float32)
    p[:, 0] = x
    p[:, 1] = (np.arange(n) - n / 2) * dy
    p[:, 2] = z
    p[:, 3] = inten
    return p


def test_collinear_points_are_surface():
    # LFE:419-423: acc = sum(neigh) - 4p = 0 on a line with equal spacing -> curvature 0 < 0.01 -> surface (LFE:436)
    r = orc.fe_extract(line_scan(), 0.0)
    inner = slice(2, -2)
    assert np.all(r.curvature[inner] < 1e-9)
    assert np.all(r.pt_label[inner] == 2)
    assert np.all(r.pt_label[:2] == 0) and np.all(r.pt_label[-2:] == 0)  # loop bounds LFE:368
    assert np.all(r.pt_type == 0)


def test_depth_and_projection_fields():
    p = line_scan()
    r = orc.fe_extract(p, 0.0)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    assert np.array_equal(r.depth_sq2, (x * x + y * y + z * z).astype(np.float32))  # LFE:516 evaluation order
    assert np.array_equal(r.img2d[:, 0], y / x) and np.array_equal(r.img2d[:, 1], z / x)  # LFE:518
    assert np.array_equal(r.polar_dis_sq2, r.img2d[:, 0] ** 2 + r.img2d[:, 1] ** 2)  # LFE:519
    assert np.array_equal(r.sigma, p[:, 3] / r.polar_dis_sq2)  # LFE:351


def test_convex_corner_is_corner():
    # a 90 degree convex wedge pointing at the sensor: depth minimum at the apex, large curvature (LFE:443-452)
    n = 41
    p = np.zeros((n, 4), np.float32)
    s = (np.arange(n) - n // 2) * 0.1
    p[:, 0] = 5.0 + np.abs(s)          # apex closest to the sensor
    p[:, 1] = s
    p[:, 2] = 0.1
    p[:, 3] = 80.0
    r = orc.fe_extract(p, 0.0)
    apex = n // 2
    assert r.curvature[apex] > 0.05
    assert r.pt_label[apex] & 1
    # far from the apex the wedge arms are straight: surface, not corner
    assert r.pt_label[5] == 2 and r.pt_label[n - 6] == 2


def test_concave_corner_is_not_corner():
    # depth maximum at the apex fails the `depth <= both neighbours` test (LFE:445-446)
    n = 41
    p = np.zeros((n, 4), np.float32)
    s = (np.arange(n) - n // 2) * 0.1
    p[:, 0] = 9.0 - np.abs(s)
    p[:, 1] = s
    p[:, 2] = 0.1
    p[:, 3] = 80.0
    r = orc.fe_extract(p, 0.0)
    assert r.curvature[n // 2] > 0.05 and not (r.pt_label[n // 2] & 1)


def test_zero_point_neighbours():
    # (0,0,0) at +-1 -> near_zero (8) and the partial sum is still used (LFE:384-392); at +-2 -> invalid (-1)
    p = line_scan()
    k = 20
    p[k, :3] = 0.0
    r = orc.fe_extract(p, 0.0)
    assert r.pt_type[k] == 1 and r.pt_label[k] == 0  # 000 mask, skipped by compute_features (LFE:370)
    assert r.pt_label[k - 1] & 8 and r.pt_label[k + 1] & 8
    assert r.pt_label[k - 2] == -1 and r.pt_label[k + 2] == -1
    # i==1 break leaves acc = 0 -> curvature = |4p|^2 (quirk (i) of SURVEY 8a-a4)
    q = p[k - 1, :3]
    acc = -4.0 * q
    assert np.isclose(r.curvature[k - 1], np.float32(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]))
    # the zero point inherits the previous point's projection (LFE:507-508) and has depth 0
    assert r.polar_dis_sq2[k] == r.polar_dis_sq2[k - 1] and r.depth_sq2[k] == 0.0
    assert np.array_equal(r.img2d[k], r.img2d[k - 1])


def test_nan_point_neighbours():
    p = line_scan()
    k = 15
    p[k, 1] = np.nan
    r = orc.fe_extract(p, 0.0)
    assert r.pt_type[k] == 32
    assert r.pt_label[k - 1] & 4 and r.pt_label[k + 1] & 4  # near_nan LFE:396-399
    assert r.pt_label[k - 2] == -1 and r.pt_label[k + 2] == -1  # LFE:400-403
    assert r.polar_dis_sq2[k] == 0.0 and r.depth_sq2[k] == 0.0  # value-initialised, never written (LFE:485-491)


def test_invalid_label_passes_both_bit_tests():
    # e_label_invalid = -1 has all bits set: get_features takes it as corner AND surface (LFE:96,238,250)
    p = line_scan()
    p[20, :3] = 0.0
    r = orc.fe_extract(p, 0.0)
    ci, si, fi = orc.fe_get_features(r, 0.0, 1.0)
    assert 18 in ci and 18 in si and 22 in ci and 22 in si


def test_fov_edge_mask_smear():
    # polar^2 > tan(17/57.3)^2 marks the point and idx-2, idx-1, idx+1 (LFE:523-526, 330-339)
    p = line_scan(n=30, dy=0.0)
    edge = orc.lib().orc_fe_max_edge_polar_pos(17.0)
    k = 12
    p[k, 1] = 5.0 * np.sqrt(edge) * 1.05  # y/x slightly past the edge
    r = orc.fe_extract(p, 0.0)
    masked = np.nonzero(r.pt_type & 16)[0]
    assert masked.tolist() == [k - 2, k - 1, k, k + 1]
    # corner features need pt_type == 0 exactly, surfaces do not (LFE:240 vs :250)
    ci, si, _ = orc.fe_get_features(r, 0.0, 1.0)
    assert not set(masked.tolist()) & set(ci.tolist())


def test_too_near_and_low_sigma_masks():
    p = line_scan(n=12)
    p[3, :3] = [0.05, 0.0, 0.01]      # depth^2 < 0.1^2 (LFE:345)
    p[7, 3] = 1e-7                      # sigma = I / polar^2 < 7e-4 (LFE:351-353)
    r = orc.fe_extract(p, 0.0)
    assert r.pt_type[3] & 2
    assert r.pt_type[7] & 4
    ci, si, fi = orc.fe_get_features(r, 0.0, 1.0)
    assert 3 not in si and 3 in fi     # too_near removed from features but pc_full keeps it (LFE:263-265)


def test_time_stamps():
    # LFE:481: float(current_time + float(idx) * 1e-5f)
    p = line_scan(n=1000)
    r = orc.fe_extract(p, 12.5)
    idx = np.arange(1000, dtype=np.float32)
    expect = (np.float64(12.5) + (idx * np.float32(1e-5)).astype(np.float64)).astype(np.float32)
    assert np.array_equal(r.time_stamp, expect)
    assert r.last_time_stamp == expect[-1]


def test_timebase_state_machine():
    # LFE:724-736 with m_last_maximum_time_stamp defined as 0
    L = orc.lib()
    tb = orc.Timebase()
    L.orc_fe_timebase_init(tb)
    assert L.orc_fe_timebase_next(tb, 100.0) == 101.0  # first call: stamp - (-1)
    assert tb.first_receive_time == 100.0
    tb.last_maximum_time_stamp = 101.2
    assert L.orc_fe_timebase_next(tb, 100.5) == 101.2  # older than the last point seen -> reuse (LFE:725-727)
    assert L.orc_fe_timebase_next(tb, 150.0) == 50.0


def test_get_features_window_and_depth_limits():
    p = line_scan(n=100)
    p[60:, 0] = 40.0                    # depth^2 = 1600 > 30^2: no corners there (LFE:242)
    r = orc.fe_extract(p, 0.0)
    ci, si, fi = orc.fe_get_features(r, 0.2, 0.5)
    assert fi.min() == 20 and fi.max() == 50  # inclusive float window (LFE:232-233)
    assert all(20 <= i <= 50 for i in si)


def test_petal_split_and_windows(scans):
    sc = scans[0]
    r = orc.fe_extract(sc.xyzi, 1.0)
    s = r.split_idx
    assert r.n_petals == len(s) - 1 and s[-1] == r.n - 1  # LFE:565,606
    d = np.diff(s[:-1])
    assert np.all(d[2:] > 50)           # hysteresis LFE:545,555 (the first edge / zero split are unconditional)
    S, first, last = orc.fe_split_scan(r)
    assert 0 < S <= r.n_petals - 1      # last petal dropped (LFE:681), empty ones removed
    assert np.all(first <= last) and np.all(np.diff(first) > 0)
    ps, pe = orc.fe_piecewise(r, first, last, 3)
    assert ps[0] == np.float32(first[0]) / r.n and pe[2] == np.float32(last[-1]) / r.n
    assert np.all(ps <= pe) and np.all(ps[1:] > pe[:-1])


def test_too_few_petals_returns_zero():
    r = orc.fe_extract(line_scan(n=200), 0.0)
    assert r.n_petals == 0 and np.all(r.polar_angle == 0)  # LFE:572
    # compute_features still ran (LFE:755-757)
    assert np.any(r.pt_label == 2)


def test_tiny_and_empty_scans():
    for n in (0, 1, 4):
        p = line_scan(n=max(n, 1))[:n]
        r = orc.fe_extract(p.reshape(-1, 4), 0.0)
        assert np.all(r.pt_label == 0)
