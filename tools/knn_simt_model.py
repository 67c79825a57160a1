#!/usr/bin/env python3
"""CPU model of how a wavefront executes the per-lane 5-NN search (ll_knn_core.h) on the C2 workload: 64 consecutive queries of
a scan run the nine x-runs of the 3x3x3 block in lockstep.  Prints, per query kind, the candidate slots a lane EXECUTES against
the candidates it USES -- the figure behind the k-NN kernels' VALU-issue bound (profiles/r03*_pmc_*)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loam_livox_amd import synth
from tests.hostcheck import hc
from tests.conftest import oracle_features

world, corner, surf = synth.make_maps(5_000_000)
gs, gc = hc.Grid(surf[:, :3], 0.6), hc.Grid(corner[:, :3], 1.45)
gs.set_guard(0.05); gc.set_guard(0.05)
for k in range(2):
    sc = synth.make_scan(world, 1000 + k, n=24000)
    _, _, _, _, fc, fs = oracle_features(sc)
    for name, g, f in (("surface", gs, fs), ("corner", gc, fc)):
        pw = synth.transform_points(sc.pose_init, f[:, :3]).astype(np.float32)
        c9 = g.knn5_run_cands(pw, 25.0)
        n = len(pw) // 64 * 64
        c = c9[:n].reshape(-1, 64, 9)
        visited = c >= 0
        trips = np.where(visited, (np.maximum(c, 0) + 3) // 4, 0)            # 4 candidates per trip (0 trips for an empty run)
        used = np.maximum(c, 0).sum(axis=2).mean()
        lock_trips = trips.max(axis=1).sum(axis=1).mean()                    # sum over runs of max over lanes
        lock_runs = visited.any(axis=1).sum(axis=1).mean()
        flat_trips = trips.sum(axis=2).max(axis=1).mean()                    # max over lanes of a lane's own total
        own_trips = trips.sum(axis=2).mean()
        own_runs = visited.sum(axis=2).mean()
        print(f"scan {k} {name:8s} queries {len(pw):6d}: candidates used/lane {used:6.1f}; runs visited/lane {own_runs:4.1f}, by the wavefront {lock_runs:4.1f}; "
              f"trips own {own_trips:5.1f}, lockstep {lock_trips:5.1f} (= {4 * lock_trips:5.0f} slots), flat schedule {flat_trips:5.1f}")
