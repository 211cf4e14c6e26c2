"""Wall and device time of svd_factorize for wide matrices (many columns = 2m / 3m image rows of W): the Gram + Jacobi route up to
64 columns, the block iteration beyond.   usage: python tools/time_svd_wide.py [rows] [n ...] [--dtype f32|f64] [--reps k]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib._mvba import svd_factorize  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("rows", nargs="?", type=int, default=20000)
ap.add_argument("ns", nargs="*", type=int, default=[64, 128, 256, 512])
ap.add_argument("--dtype", default="f64")
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
dt = np.float32 if a.dtype == "f32" else np.float64
rng = np.random.default_rng(0)
for n in a.ns:
    # rank-4 signal + noise, like a measurement matrix
    Wt = (rng.standard_normal((a.rows, 4), dtype=np.float32) @ rng.standard_normal((4, n), dtype=np.float32)
          + np.float32(1e-3) * rng.standard_normal((a.rows, n), dtype=np.float32)).astype(dt)
    for rep in range(a.reps):
        t0 = time.perf_counter()
        M, sigma, S, _mu, tm = svd_factorize(Wt, 4)
        t1 = time.perf_counter()
        gb = Wt.nbytes / 1e9
        print(f"n {n:5d} rows {a.rows} {a.dtype}: wall {t1 - t0:8.3f} s  H2D {tm['h2d_ms']:.1f} ms  eigen / iteration {tm['jacobi_ms']:.2f} ms ({tm['sweeps']} sweeps / iterations)"
              f"  final {tm['refine_ms']:.2f} ms   W = {gb:.2f} GB", flush=True)
    if a.rows * n <= 5e7:
        s_ref = np.linalg.svd(Wt.astype(np.float64), compute_uv=False)
        print(f"        sigma[:4] rel. error vs LAPACK {np.max(np.abs(sigma[:4] - s_ref[:4]) / s_ref[:4]):.1e}", flush=True)
