"""Wall and device time of svd_factorize for wide matrices (many columns = 2m / 3m image rows of W): where the one-workgroup
eigen-solver stops being usable.   usage: python tools/time_svd_wide.py [rows] [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib._mvba import svd_factorize  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ns = [int(a) for a in sys.argv[2:]] or [64, 128, 256, 512]
rng = np.random.default_rng(0)
for n in ns:
    # rank-4 signal + noise, like a measurement matrix
    Wt = rng.standard_normal((rows, 4)) @ rng.standard_normal((4, n)) + 1e-3 * rng.standard_normal((rows, n))
    t0 = time.perf_counter()
    M, sigma, S, _mu, tm = svd_factorize(Wt, 4)
    t1 = time.perf_counter()
    s_ref = np.linalg.svd(Wt, compute_uv=False)
    err = np.max(np.abs(sigma[:8] - s_ref[:8]) / s_ref[0])
    rec = np.linalg.norm(S.T @ M.T - Wt) / np.linalg.norm(Wt)
    print(f"n {n:5d} rows {rows}: wall {t1 - t0:8.3f} s  timings_ms {tm}  sigma err {err:.2e}  rank-4 residual {rec:.2e}", flush=True)
