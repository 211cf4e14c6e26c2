#!/usr/bin/env python3
"""Solver check on the GPU: dxi from the blocked Cholesky vs NumPy on the same reduced system, and the
LU-fallback counter (must stay 0 for these SPD systems).  Prints as it goes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
import numpy as np
from lib.bundle_adjustment import BundleAdjuster
from lib.synthetic import make_scene

cases = [(300, 2, 1.0), (400, 4, 1.0), (600, 10, 0.8), (2000, 15, 0.6), (3000, 20, 0.5), (5000, 29, 0.5), (5000, 40, 0.4), (20000, 100, 0.2)]
for n, m, p in ([cases[int(a)] for a in sys.argv[1:]] or cases):
    sc = make_scene(n, m, vis_p=p)
    ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
    eng = ba._engine
    eng.cost(); eng.linearize()
    E1 = eng.try_step(1e-4)
    m9 = 9 * m
    A = eng.debug_read("A_full").reshape(m9, m9); b = eng.debug_read("b_full"); dxi = eng.debug_read("dxi")
    A = np.triu(A) + np.triu(A, 1).T
    removed = [3, 4, 5, 6, 7, 8, 12 + (0 if sc.axis.startswith("x-right") else 1)]
    keep = np.setdiff1d(np.arange(m9), removed)
    ref = np.zeros(m9); ref[keep] = np.linalg.solve(A[np.ix_(keep, keep)], b[keep])
    # sign convention: compare up to the library's own sign by testing both
    err = min(np.abs(dxi - ref).max(), np.abs(dxi + ref).max()) / np.abs(ref).max()
    print(f"m={m:4d} D={9*m-7:5d} lu_fallback={eng.stats()['counts']['lu_fallback']} rel err {err:.2e}", flush=True)
