#!/usr/bin/env python3
"""mvba_create phase times (MVBA_CREATE_TIMING=1) at a given scene size, device-built against host-built Schur index.
usage: python tools/time_create.py <points> <cameras> <visibility>      (config-4 per-GPU shard: 1250000 500 0.05)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib.bundle_adjustment import BundleAdjuster
from lib.synthetic import make_scene

n, m, vis = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
sc = make_scene(n, m, vis_p=vis)
os.environ["MVBA_CREATE_TIMING"] = "1"
for mode in ("device", "host"):
    if mode == "host":
        os.environ["MVBA_INDEX"] = "host"
    t0 = time.perf_counter()
    ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
    dt = time.perf_counter() - t0
    eng = ba._engine
    eng.cost(); eng.linearize(); E1 = eng.try_step(1e-3)
    print(f"{mode}-built index: {n} points x {m} cameras x {vis}: {sc.n_obs} observations, {eng.schur_info()}, from_observations {dt:.3f} s, "
          f"first trial cost {E1!r}", flush=True)
    del ba, eng
