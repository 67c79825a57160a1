"""Synthetic Livox Mid-40 data for parity tests and bench.py.

Nothing here comes from the reference: the reference ships no data (bags are external,
README.md:76-136).  The world is an axis-aligned building of planar patches + explicit edge
lines; the surface map samples patches at ~0.4 m, the corner map samples edges at ~0.1 m
(leaf sizes of config/performance_precision.yaml:12-13); the scan is a Mid-40 rosette
(two counter-rotating Risley prisms) ray-cast into the world.  Everything is seeded.

numpy only; this module is host-side test/bench plumbing, not part of the GPU hot path.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

WORLD_SEED = 20260923
ROOM_PITCH = 10.0
ROOM_HEIGHT = 4.0

# Mid-40 rosette
ROSETTE_ALPHA_DEG = 9.6
ROSETTE_F1 = 121.6
ROSETTE_F2 = 77.7
SAMPLE_RATE = 100e3


@dataclass
class World:
    """Axis-aligned rectangles: rect[i] = (axis, offset, lo_u, hi_u, lo_v, hi_v) where (u, v) are the two
    remaining axes in increasing order; edges[i] = (x0,y0,z0,x1,y1,z1)."""

    rects: np.ndarray
    edges: np.ndarray
    extent: tuple
    nx: int
    ny: int
    pillars: np.ndarray = None  # (K,4) x0,x1,y0,y1


def make_world(n_rooms_x: int, n_rooms_y: int, seed: int = WORLD_SEED) -> World:
    rng = np.random.default_rng(seed)
    P, H = ROOM_PITCH, ROOM_HEIGHT
    W, L = n_rooms_x * P, n_rooms_y * P
    xs = np.arange(n_rooms_x + 1) * P + np.r_[0.0, rng.uniform(-1.0, 1.0, n_rooms_x - 1), 0.0]
    ys = np.arange(n_rooms_y + 1) * P + np.r_[0.0, rng.uniform(-1.0, 1.0, n_rooms_y - 1), 0.0]
    rects = []
    edges = []
    pillars = []
    # floor / ceiling (axis 2): u = x, v = y
    rects.append((2, 0.0, 0.0, W, 0.0, L))
    rects.append((2, H, 0.0, W, 0.0, L))
    # walls normal to x (axis 0): u = y, v = z ; normal to y (axis 1): u = x, v = z
    for x in xs:
        rects.append((0, x, 0.0, L, 0.0, H))
        edges.append((x, 0.0, 0.0, x, L, 0.0))
        edges.append((x, 0.0, H, x, L, H))
    for y in ys:
        rects.append((1, y, 0.0, W, 0.0, H))
        edges.append((0.0, y, 0.0, W, y, 0.0))
        edges.append((0.0, y, H, W, y, H))
    for x in xs:
        for y in ys:
            edges.append((x, y, 0.0, x, y, H))
    # pillars: 1-2 boxes per room, floor to ceiling
    for i in range(n_rooms_x):
        for j in range(n_rooms_y):
            for _ in range(int(rng.integers(1, 3))):
                sx, sy = rng.uniform(0.4, 1.0, 2)
                cx = rng.uniform(xs[i] + 1.5, xs[i + 1] - 1.5)
                cy = rng.uniform(ys[j] + 1.5, ys[j + 1] - 1.5)
                x0, x1, y0, y1 = cx - sx / 2, cx + sx / 2, cy - sy / 2, cy + sy / 2
                pillars.append((x0, x1, y0, y1))
                rects.append((0, x0, y0, y1, 0.0, H))
                rects.append((0, x1, y0, y1, 0.0, H))
                rects.append((1, y0, x0, x1, 0.0, H))
                rects.append((1, y1, x0, x1, 0.0, H))
                for (ex, ey) in ((x0, y0), (x0, y1), (x1, y0), (x1, y1)):
                    edges.append((ex, ey, 0.0, ex, ey, H))
    return World(np.asarray(rects, dtype=np.float64), np.asarray(edges, dtype=np.float64), (W, L, H),
                 n_rooms_x, n_rooms_y, np.asarray(pillars, dtype=np.float64))


def world_for_map_size(m_total: int, seed: int = WORLD_SEED) -> World:
    """Building sized so that ~0.8*m_total surface points at 0.4 m spacing fit."""
    per_room = 1900.0  # surface points per 10x10x4 room at 0.16 m^2/pt (floor+ceiling+2 walls+pillars)
    n_rooms = max(4.0, 0.8 * m_total / per_room)
    n = max(2, int(round(np.sqrt(n_rooms))))
    return make_world(n, n, seed)


def _rect_area(r):
    return (r[:, 3] - r[:, 2]) * (r[:, 5] - r[:, 4])


def sample_surface_map(world: World, n_target: int, seed: int = WORLD_SEED + 1) -> np.ndarray:
    """Jittered-grid samples of every rectangle: spacing chosen to hit ~n_target points; in-plane jitter
    U(-0.25,0.25)*spacing, sigma = 0.01 m normal noise. Returns float32 (M,3)."""
    rng = np.random.default_rng(seed)
    area = float(_rect_area(world.rects).sum())
    h = np.sqrt(area / n_target)
    out = []
    for r in world.rects:
        axis = int(r[0])
        nu = max(1, int(round((r[3] - r[2]) / h)))
        nv = max(1, int(round((r[5] - r[4]) / h)))
        du, dv = (r[3] - r[2]) / nu, (r[5] - r[4]) / nv
        # chunk big rectangles to bound memory
        rows_per = max(1, 2_000_000 // nv)
        for u0 in range(0, nu, rows_per):
            u_idx = np.arange(u0, min(nu, u0 + rows_per))
            uu, vv = np.meshgrid((u_idx + 0.5) * du + r[2], (np.arange(nv) + 0.5) * dv + r[4], indexing="ij")
            uu = uu.ravel() + rng.uniform(-0.25, 0.25, uu.size) * du
            vv = vv.ravel() + rng.uniform(-0.25, 0.25, vv.size) * dv
            nn = r[1] + rng.normal(0.0, 0.01, uu.size)
            pts = np.empty((uu.size, 3), dtype=np.float32)
            others = [a for a in range(3) if a != axis]
            pts[:, axis] = nn
            pts[:, others[0]] = uu
            pts[:, others[1]] = vv
            out.append(pts)
    pts = np.concatenate(out, axis=0)
    # deterministic shuffle: the reference's map cloud has no spatial ordering guarantee either
    perm = np.random.default_rng(seed + 7).permutation(pts.shape[0])
    return np.ascontiguousarray(pts[perm])


def sample_corner_map(world: World, n_target: int, seed: int = WORLD_SEED + 2) -> np.ndarray:
    rng = np.random.default_rng(seed)
    e = world.edges
    length = np.linalg.norm(e[:, 3:6] - e[:, 0:3], axis=1)
    h = float(length.sum()) / n_target
    out = []
    for k in range(e.shape[0]):
        n = max(1, int(round(length[k] / h)))
        t = (np.arange(n) + 0.5 + rng.uniform(-0.2, 0.2, n)) / n
        p = e[k, 0:3][None, :] + t[:, None] * (e[k, 3:6] - e[k, 0:3])[None, :]
        p = p + rng.normal(0.0, 0.01, p.shape)
        out.append(p.astype(np.float32))
    pts = np.concatenate(out, axis=0)
    perm = np.random.default_rng(seed + 7).permutation(pts.shape[0])
    return np.ascontiguousarray(pts[perm])


def make_maps(m_total: int, seed: int = WORLD_SEED):
    """(world, corner_map float32 (Mc,3), surf_map float32 (Ms,3)) with Ms ~ 0.8 M, Mc ~ 0.2 M."""
    world = world_for_map_size(m_total, seed)
    surf = sample_surface_map(world, int(0.8 * m_total), seed + 1)
    corner = sample_corner_map(world, int(0.2 * m_total), seed + 2)
    return world, corner, surf


# ----------------------------------------------------------------------------- poses

def quat_from_rpy(roll, pitch, yaw):
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy])  # x,y,z,w


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.r_[axis * np.sin(angle / 2), np.cos(angle / 2)]


def pose_compose(a, b):
    """pose = [qx,qy,qz,qw,tx,ty,tz]; returns a o b (apply b first, then a)."""
    q = quat_mul(a[:4], b[:4])
    t = quat_to_mat(a[:4]) @ b[4:] + a[4:]
    return np.r_[q, t]


def pose_error(a, b):
    """(translation error [m], rotation error [rad]) between two poses."""
    dt = float(np.linalg.norm(np.asarray(a[4:]) - np.asarray(b[4:])))
    qa, qb = np.asarray(a[:4], dtype=np.float64), np.asarray(b[:4], dtype=np.float64)
    d = quat_mul(qa, np.r_[-qb[:3], qb[3]])
    ang = 2.0 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3]))
    return dt, float(ang)


# ----------------------------------------------------------------------------- scans

def rosette_dirs(n: int, t0: float = 0.0) -> np.ndarray:
    """Unit ray directions in the sensor frame (X forward), n consecutive 100 kHz samples."""
    t = t0 + np.arange(n) / SAMPLE_RATE
    a = np.deg2rad(ROSETTE_ALPHA_DEG)
    a1 = 2 * np.pi * ROSETTE_F1 * t
    a2 = -2 * np.pi * ROSETTE_F2 * t
    az = a * (np.cos(a1) + np.cos(a2))
    el = a * (np.sin(a1) + np.sin(a2))
    d = np.stack([np.ones(n), np.tan(az), np.tan(el)], axis=1)
    return d / np.linalg.norm(d, axis=1, keepdims=True)


def raycast(world: World, origin: np.ndarray, dirs_w: np.ndarray, max_range: float = 200.0,
            prefilter_range: float = 40.0) -> np.ndarray:
    """Range along each ray to the nearest rectangle (inf if none).  `origin` is one point (3,) or one per ray (n,3)
    (a sensor that moves during the scan)."""
    origin = np.asarray(origin, dtype=np.float64)
    per_ray = origin.ndim == 2
    o_ref = origin.mean(axis=0) if per_ray else origin
    r = world.rects
    # prefilter rectangles by distance from the origin to their bounding box
    lo = np.empty((r.shape[0], 3))
    hi = np.empty((r.shape[0], 3))
    for axis in range(3):
        others = [a for a in range(3) if a != axis]
        m = r[:, 0] == axis
        lo[m, axis] = r[m, 1]
        hi[m, axis] = r[m, 1]
        lo[m, others[0]] = r[m, 2]
        hi[m, others[0]] = r[m, 3]
        lo[m, others[1]] = r[m, 4]
        hi[m, others[1]] = r[m, 5]
    dbox = np.linalg.norm(np.maximum(np.maximum(lo - o_ref, o_ref - hi), 0.0), axis=1)
    keep = dbox < prefilter_range
    r = r[keep]
    best = np.full(dirs_w.shape[0], np.inf)
    for axis in range(3):
        ra = r[r[:, 0] == axis]
        if ra.shape[0] == 0:
            continue
        others = [a for a in range(3) if a != axis]
        dk = dirs_w[:, axis][:, None]
        oa = origin[:, axis][:, None] if per_ray else origin[axis]
        ou = origin[:, others[0]][:, None] if per_ray else origin[others[0]]
        ov = origin[:, others[1]][:, None] if per_ray else origin[others[1]]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (ra[:, 1][None, :] - oa) / dk  # (n, R)
        u = ou + t * dirs_w[:, others[0]][:, None]
        v = ov + t * dirs_w[:, others[1]][:, None]
        ok = (t > 1e-3) & (u >= ra[:, 2][None, :]) & (u <= ra[:, 3][None, :]) & (v >= ra[:, 4][None, :]) & (
            v <= ra[:, 5][None, :])
        t = np.where(ok, t, np.inf)
        best = np.minimum(best, t.min(axis=1))
    best[best > max_range] = np.inf
    return best


@dataclass
class Scan:
    xyzi: np.ndarray          # float32 (N,4) sensor frame, intensity = reflectivity
    pose_true: np.ndarray     # [qx,qy,qz,qw,tx,ty,tz] sensor -> world
    pose_init: np.ndarray     # perturbed initial guess
    seed: int = 0
    meta: dict = field(default_factory=dict)


def sensor_pose_in_world(world: World, rng) -> np.ndarray:
    """Sensor inside a room, looking (nose slightly down) towards one of the room's corners so that the narrow
    Mid-40 FOV sees the floor and two walls: all six degrees of freedom are constrained."""
    i = int(rng.integers(1, max(2, world.nx - 1))) if world.nx > 2 else int(rng.integers(0, world.nx))
    j = int(rng.integers(1, max(2, world.ny - 1))) if world.ny > 2 else int(rng.integers(0, world.ny))
    for _ in range(100):  # keep 1 m clear of every pillar
        px = (i + rng.uniform(0.35, 0.65)) * ROOM_PITCH
        py = (j + rng.uniform(0.35, 0.65)) * ROOM_PITCH
        pl = world.pillars
        if pl is None or pl.size == 0:
            break
        dx = np.maximum(np.maximum(pl[:, 0] - px, px - pl[:, 1]), 0.0)
        dy = np.maximum(np.maximum(pl[:, 2] - py, py - pl[:, 3]), 0.0)
        if np.min(np.hypot(dx, dy)) > 1.0:
            break
    pz = rng.uniform(1.2, 1.8)
    cx = (i + int(rng.integers(0, 2))) * ROOM_PITCH
    cy = (j + int(rng.integers(0, 2))) * ROOM_PITCH
    yaw = np.arctan2(cy - py, cx - px) + rng.uniform(-0.15, 0.15)
    q = quat_from_rpy(rng.uniform(-0.03, 0.03), rng.uniform(0.12, 0.30), yaw)
    return np.r_[q, px, py, pz]


def make_scan(world: World, k: int = 0, n: int = 24000, range_sigma: float = 0.02, p_zero: float = 0.005,
              p_nan: float = 0.001, pose_true: np.ndarray | None = None) -> Scan:
    """Scan k uses seed 1000+k."""
    rng = np.random.default_rng(1000 + k)
    if pose_true is None:
        pose_true = sensor_pose_in_world(world, rng)
    dirs = rosette_dirs(n, t0=rng.uniform(0.0, 1.0))
    R = quat_to_mat(pose_true[:4])
    dirs_w = dirs @ R.T
    rng_m = raycast(world, pose_true[4:], dirs_w)
    hit = np.isfinite(rng_m)
    r = np.where(hit, rng_m + rng.normal(0.0, range_sigma, n), 0.0)
    pts = (dirs * r[:, None]).astype(np.float32)
    pts[~hit] = 0.0
    inten = rng.uniform(5.0, 150.0, n).astype(np.float32)
    u = rng.uniform(0.0, 1.0, n)
    pts[u < p_zero] = 0.0
    pts[(u >= p_zero) & (u < p_zero + p_nan)] = np.nan
    xyzi = np.concatenate([pts, inten[:, None]], axis=1).astype(np.float32)
    # initial guess = T* o delta, delta ~ U(-0.1,0.1)^3 m x axis-angle U(0,1 deg)
    dq = quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0.0, 1.0)))
    dt = rng.uniform(-0.1, 0.1, 3)
    pose_init = pose_compose(pose_true, np.r_[dq, dt])
    return Scan(np.ascontiguousarray(xyzi), pose_true, pose_init, seed=1000 + k)


def quat_slerp_from_identity(q, s):
    """Eigen's Quaterniond::Identity().slerp(s, q) for an array of ratios s -> (n,4) (x,y,z,w)."""
    q = np.asarray(q, dtype=np.float64)
    if q[3] < 0:
        q = -q
    nv = np.linalg.norm(q[:3])
    if nv < 1e-15:
        return np.tile(np.array([0, 0, 0, 1.0]), (len(s), 1))
    th = np.arctan2(nv, q[3])
    axis = q[:3] / nv
    return np.concatenate([np.sin(s * th)[:, None] * axis[None, :], np.cos(s * th)[:, None]], axis=1)


def make_moving_scan(world: World, k: int = 0, n: int = 24000, inc_true=None, yaw_offset: float = 0.0, t_phase: float | None = None,
                     range_sigma: float = 0.02, p_zero: float = 0.005, p_nan: float = 0.001, pose_start=None) -> Scan:
    """Scan taken while the sensor moves with a constant twist: point i (blur ratio s_i = i/(n-1)) is measured from
    T(s_i) = T_start o (slerp(I, q_inc, s_i), s_i t_inc) -- exactly the motion model of the reference's *_mb residuals
    (ceres_icp.hpp:116-121) -- and is expressed in the instantaneous sensor frame.  `yaw_offset` rotates the rosette
    about the sensor Z axis (the three heads of a Mid-100 are ~38.4 degrees apart).  pose_true = pose at scan END,
    pose_init = pose at scan START (what the registrar receives as pose_last / initial guess)."""
    rng = np.random.default_rng(3000 + k)
    if pose_start is None:
        pose_start = sensor_pose_in_world(world, rng)
    if inc_true is None:
        inc_true = np.r_[quat_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0.3, 1.0))), rng.uniform(-0.08, 0.08, 3)]
    s = np.arange(n) / max(1, n - 1)
    dirs = rosette_dirs(n, t0=rng.uniform(0.0, 1.0) if t_phase is None else t_phase)
    if yaw_offset != 0.0:
        cy, sy = np.cos(yaw_offset), np.sin(yaw_offset)
        dirs = dirs @ np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]).T
    qs = quat_slerp_from_identity(inc_true[:4], s)                       # (n,4) incremental rotation at ratio s
    ts = s[:, None] * inc_true[4:][None, :]
    R0 = quat_to_mat(pose_start[:4])
    # rotate each direction by its own incremental rotation: v + 2w(qv x v) + 2 qv x (qv x v)
    qv, qw = qs[:, :3], qs[:, 3:4]
    uv = 2.0 * np.cross(qv, dirs)
    d_inc = dirs + qw * uv + np.cross(qv, uv)
    dirs_w = d_inc @ R0.T
    origins = ts @ R0.T + pose_start[4:]
    rng_m = raycast(world, origins, dirs_w)
    hit = np.isfinite(rng_m)
    r = np.where(hit, rng_m + rng.normal(0.0, range_sigma, n), 0.0)
    pts = (dirs * r[:, None]).astype(np.float32)
    pts[~hit] = 0.0
    inten = rng.uniform(5.0, 150.0, n).astype(np.float32)
    u = rng.uniform(0.0, 1.0, n)
    pts[u < p_zero] = 0.0
    pts[(u >= p_zero) & (u < p_zero + p_nan)] = np.nan
    xyzi = np.concatenate([pts, inten[:, None]], axis=1).astype(np.float32)
    pose_end = pose_compose(pose_start, inc_true)
    return Scan(np.ascontiguousarray(xyzi), pose_end, np.asarray(pose_start, dtype=np.float64), seed=3000 + k,
                meta=dict(inc_true=np.asarray(inc_true, dtype=np.float64)))


def transform_points(pose, pts):
    return (pts.astype(np.float64) @ quat_to_mat(pose[:4]).T + pose[4:]).astype(np.float32)
