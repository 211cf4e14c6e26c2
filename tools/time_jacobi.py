#!/usr/bin/env python3
"""Eigen-solver time and sweep count of the factorisation at a few column counts (rows fixed, small).
usage: python tools/time_jacobi.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
import numpy as np
from lib._mvba import SvdWorkspace

rng = np.random.default_rng(0)
for n, dt in ((24, np.float32), (24, np.float64), (20, np.float64), (30, np.float32), (12, np.float64)):
    rows = 200000
    W = (rng.standard_normal((rows, 4)) @ rng.standard_normal((4, n)) + 1e-3 * rng.standard_normal((rows, n))).astype(dt)
    ws = SvdWorkspace(rows, n, dt)
    ws.load(W)
    best = None
    for _ in range(5):
        M, sigma, S, mu, tm = ws.run(3)
        if best is None or tm["jacobi_ms"] < best["jacobi_ms"]:
            best = tm
    sw = best["sweeps"]
    steps = sw * (((n + 1) & ~1) - 1)
    print(f"n = {n:2d} {np.dtype(dt).name}: jacobi {best['jacobi_ms'] * 1e3:7.1f} us, {sw} sweeps = {steps} steps, {best['jacobi_ms'] * 1e3 / steps:.3f} us per step")
