"""SURVEY 8(d) roofline inputs, counted on the host (test / bench infrastructure, not product code).

  U = number of DISTINCT map points lying in the 27-cell neighbourhoods of all occupied query cells (each counted once)
  C = number of distinct cells touched (their start/count table entries are read)
  Q = number of queries (every extracted corner / surface feature of the scan, "Q-full")

for one scan registered against the corner + surface maps, with the uniform grid geometry of the device maps
(origin = minimum corner of the finite map points, cubic cells of the library's default size: 1.45 m corner, 0.6 m
surface -- loam_livox_amd/csrc/ll_api.hip ll_map_upload).  Queries are the oracle's features of the scan transformed
with the given pose (the first ICP iteration's query positions; later iterations move them by centimetres).
"""
from __future__ import annotations

import numpy as np


class MapCells:
    """Occupancy of a uniform grid over a map cloud: per-cell point counts, addressable by integer cell coordinates."""

    def __init__(self, xyz: np.ndarray, h: float):
        p = np.asarray(xyz, np.float32)[:, :3]
        p = p[np.isfinite(p).all(axis=1)]
        self.h = float(h)
        self.origin = p.min(axis=0).astype(np.float64) if len(p) else np.zeros(3)
        c = np.floor((p.astype(np.float64) - self.origin) / self.h).astype(np.int64)
        self.dims = (c.max(axis=0) + 1) if len(p) else np.ones(3, np.int64)
        key = (c[:, 2] * self.dims[1] + c[:, 1]) * self.dims[0] + c[:, 0]
        self.keys, self.counts = np.unique(key, return_counts=True)

    def neighbourhood(self, q_xyz: np.ndarray):
        """(U, C) of the union of the 3x3x3 neighbourhoods of the cells the queries fall into."""
        q = np.asarray(q_xyz, np.float64)
        q = q[np.isfinite(q).all(axis=1)]
        if len(q) == 0:
            return 0, 0
        c = np.floor((q - self.origin) / self.h).astype(np.int64)
        c = np.unique(c, axis=0)
        offs = np.array([(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)], np.int64)
        n = (c[:, None, :] + offs[None, :, :]).reshape(-1, 3)
        inside = ((n >= 0) & (n < self.dims)).all(axis=1)
        n = n[inside]
        key = np.unique((n[:, 2] * self.dims[1] + n[:, 1]) * self.dims[0] + n[:, 0])
        pos = np.searchsorted(self.keys, key)
        pos[pos >= len(self.keys)] = len(self.keys) - 1
        hit = self.keys[pos] == key
        return int(self.counts[pos[hit]].sum()), int(len(key))


def scan_u_c(map_corner: MapCells, map_surf: MapCells, scan_xyzi: np.ndarray, pose: np.ndarray):
    """(U, C, Q) of one scan: oracle feature extraction (LFE), every corner / surface feature a query (Q-full)."""
    from loam_livox_amd import synth
    from oracle import orc
    o = orc.fe_extract(scan_xyzi, 1.0)
    ci, si, _ = orc.fe_get_features(o, 0.0, 1.0)
    qc = synth.transform_points(pose, scan_xyzi[ci, :3])
    qs = synth.transform_points(pose, scan_xyzi[si, :3])
    uc, cc = map_corner.neighbourhood(qc)
    us, cs = map_surf.neighbourhood(qs)
    return uc + us, cc + cs, len(ci) + len(si)
