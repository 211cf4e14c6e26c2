#!/usr/bin/env python3
"""Dense-solve A/B at a given camera count: solve ms per try_step and dxi vs NumPy.
usage: [MVBA_CHOL=launches] [MVBA_TRAIL32_MAX=n] python tools/ab_solve.py m"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
import numpy as np
from lib.bundle_adjustment import BundleAdjuster
from lib.synthetic import make_scene
m = int(sys.argv[1]); n = 40000 if m >= 300 else 100000
sc = make_scene(n, m, vis_p=0.05 if m >= 300 else 0.1)
ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
eng = ba._engine
eng.cost(); eng.linearize(); eng.try_step(1e-4)
m9 = 9 * m
A = eng.debug_read("A_full").reshape(m9, m9); b = eng.debug_read("b_full"); dxi = eng.debug_read("dxi")
A = np.triu(A) + np.triu(A, 1).T
keep = np.setdiff1d(np.arange(m9), [3, 4, 5, 6, 7, 8, 13])
ref = np.zeros(m9); ref[keep] = np.linalg.solve(A[np.ix_(keep, keep)], b[keep])
err = np.abs(dxi - ref).max() / np.abs(ref).max()
eng.set_profiling(True); eng.reset_stats()
for _ in range(10): eng.try_step(1e-4)
st = eng.stats()
print(f"{' '.join(k + '=' + v for k, v in os.environ.items() if k.startswith('MVBA_'))} m={m} D={9*m-7} solve {st['solve']['ms']/10:.3f} ms  schur {st['schur']['ms']/10:.3f}  dxi rel err {err:.2e}  lu {st['counts']['lu_fallback']}")
