#!/usr/bin/env python3
"""Soak test of the dense solve: the same reduced system solved `iters` times, every result compared bit for bit with the
first (the persistent back-substitution synchronises its workgroups point to point: an ordering fault would show as a
differing or non-finite dxi, or as a barrier fallback).
usage: python tools/soak_solve.py m [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
import numpy as np
from lib.bundle_adjustment import BundleAdjuster
from lib.synthetic import make_scene
m = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = 40000 if m >= 300 else 100000
sc = make_scene(n, m, vis_p=0.05 if m >= 300 else 0.1)
ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
eng = ba._engine
eng.cost(); eng.linearize()
E0 = eng.try_step(1e-4); ref = eng.debug_read("dxi").copy()
assert np.isfinite(ref).all()
bad = 0; t0 = time.time()
for i in range(iters):
    E = eng.try_step(1e-4)
    d = eng.debug_read("dxi")
    if E != E0 or not np.array_equal(d, ref):
        bad += 1
        print(f"iteration {i}: E {E!r} vs {E0!r}, max |ddxi| {np.abs(d - ref).max():.3e}")
st = eng.stats()["counts"]
print(f"m={m} D={9*m-7}: {iters} solves in {time.time()-t0:.1f} s, {bad} differing, barrier_fallback {st['barrier_fallback']}, lu_fallback {st['lu_fallback']}")
sys.exit(1 if bad or st["barrier_fallback"] else 0)
