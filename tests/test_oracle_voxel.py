"""No-GPU tier: the VoxelGrid oracle (oracle/ll_oracle_voxel.c, PCL 1.9 semantics restated) against hand-worked
cases and an independent numpy restatement.  PARITY UNPINNED: PCL is absent, the reference has no fixtures."""
import numpy as np

from oracle import orc


def np_voxel_grid(xyzi, leaf):
    """independent restatement: numpy float32 ops, stable argsort, sequential float32 sums"""
    p = np.asarray(xyzi, np.float32).reshape(-1, 4)
    leaf = np.broadcast_to(np.asarray(leaf, np.float32), (3,))
    ok = np.isfinite(p[:, :3]).all(1)
    q = p[ok]
    if len(q) == 0:
        return 2, np.zeros((0, 4), np.float32)
    inv = (np.float32(1.0) / leaf).astype(np.float32)
    mn, mx = q[:, :3].min(0), q[:, :3].max(0)
    d = ((mx - mn) * inv).astype(np.float32).astype(np.int64) + 1
    if int(d[0]) * int(d[1]) * int(d[2]) > 2**31 - 1:
        return 1, p.copy()
    min_b = np.floor(mn * inv).astype(np.int32)
    max_b = np.floor(mx * inv).astype(np.int32)
    div = max_b - min_b + 1
    ijk = (np.floor(q[:, :3] * inv).astype(np.float32) - min_b.astype(np.float32)).astype(np.int32)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    out = []
    k = 0
    while k < len(order):
        e = k
        s = np.zeros(4, np.float32)
        while e < len(order) and idx[order[e]] == idx[order[k]]:
            s = (s + q[order[e]]).astype(np.float32)
            e += 1
        out.append(s / np.float32(e - k))
        k = e
    return 0, np.array(out, np.float32)


def test_hand_worked_centroids_and_order():
    # leaf 1: voxels by floor(); output ascending in x-fastest leaf index; centroid includes intensity
    pts = np.array([[0.2, 0.2, 0.2, 10], [0.8, 0.4, 0.6, 30],      # voxel (0,0,0)
                    [1.5, 0.5, 0.5, 7],                            # voxel (1,0,0)
                    [0.5, 1.5, 0.5, 9],                            # voxel (0,1,0)
                    [-0.5, 0.5, 0.5, 1], [-0.25, 0.25, 0.75, 3]],  # voxel (-1,0,0)
                   np.float32)
    st, out = orc.voxel_grid(pts, 1.0)
    assert st == 0 and out.shape == (4, 4)
    # min_b = (-1,0,0), div = (3,2,1): indices (-1,0,0)->0, (0,0,0)->1, (1,0,0)->2, (0,1,0)->4
    expect = np.array([[-0.375, 0.375, 0.625, 2], [0.5, 0.3, 0.4, 20], [1.5, 0.5, 0.5, 7], [0.5, 1.5, 0.5, 9]], np.float32)
    assert np.allclose(out, expect, atol=1e-6)


def test_boundaries_negative_zero_and_nonfinite():
    pts = np.array([[1.0, 0, 0, 1], [0.99999994, 0, 0, 1], [-0.0, 0, 0, 5], [0.0, 0, 0, 7], [np.nan, 0, 0, 9], [0, np.inf, 0, 9],
                    [0.5, 0, np.nan, 9]], np.float32)
    st, out = orc.voxel_grid(pts, 1.0)
    assert st == 0 and len(out) == 2                   # [0,1) holds -0, +0 and 0.99999994; 1.0 opens the next leaf
    assert out[0, 3] == np.float32(13.0) / np.float32(3) and out[1, 0] == 1.0
    assert orc.voxel_grid(pts[4:], 1.0)[0] == 2        # no finite point: empty (defined deviation)
    st, out = orc.voxel_grid(np.zeros((0, 4), np.float32), 1.0)
    assert st == 2 and out.shape == (0, 4)


def test_leaf_too_small_copies_input():
    rng = np.random.default_rng(3)
    pts = rng.uniform(-100, 100, (50, 4)).astype(np.float32)
    pts[3, 1] = np.nan
    st, out = orc.voxel_grid(pts, 0.01)     # 20000^3 cells > INT32_MAX
    assert st == 1 and out.shape == pts.shape and np.array_equal(out, pts, equal_nan=True)
    st, out = orc.voxel_grid(pts, 0.2)      # 1000^3 = 1e9 fits
    assert st == 0


def test_float_summation_is_sequential_in_input_order():
    # 3 points in one voxel whose float sum depends on the order: (a + b) + c != a + (b + c)
    a, b, c = np.float32(1e8), np.float32(-1e8), np.float32(0.3)
    pts = np.array([[0.1, 0.1, 0.1, a], [0.2, 0.2, 0.2, b], [0.3, 0.3, 0.3, c]], np.float32)
    st, out = orc.voxel_grid(pts, 1.0)
    assert out[0, 3] == ((a + b) + c) / np.float32(3)
    st, out = orc.voxel_grid(pts[[2, 1, 0]], 1.0)
    assert out[0, 3] == ((c + b) + a) / np.float32(3)


def test_matches_numpy_restatement_on_random_clouds():
    rng = np.random.default_rng(8)
    for n, span, leaf in ((5000, 20.0, 0.4), (20000, 60.0, (0.2, 0.4, 0.8)), (300, 2.0, 0.1), (1, 1.0, 0.4), (4000, 500.0, 0.3)):
        p = rng.uniform(-span, span, (n, 4)).astype(np.float32)
        p[rng.random(n) < 0.01, rng.integers(0, 3)] = np.nan
        st, out = orc.voxel_grid(p, leaf)
        st2, out2 = np_voxel_grid(p, leaf)
        assert st == st2 and out.shape == out2.shape and np.array_equal(out, out2, equal_nan=True)


def test_idempotent_on_its_own_output_when_leafs_hold_one_point():
    rng = np.random.default_rng(9)
    p = rng.uniform(-10, 10, (3000, 4)).astype(np.float32)
    _, o1 = orc.voxel_grid(p, 0.5)
    _, o2 = orc.voxel_grid(o1, 0.5)
    # a centroid stays inside its leaf except for rounding at the walls: the count can only shrink marginally
    assert len(o2) <= len(o1) and len(o2) >= 0.99 * len(o1)
