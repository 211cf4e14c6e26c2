#!/usr/bin/env python3
"""it/s of the LM loop on the small configurations (BASELINE configs 1 and 2) + per-kernel ms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
import numpy as np
from lib.bundle_adjustment import BundleAdjuster, LevenbergMarquardt
from lib.synthetic import make_scene

CASES = ((200, 10, 1.0, 30), (10_000, 20, 1.0, 10), (100_000, 50, 0.2, 10), (125_000, 500, 0.05, 5),  # config-4 shape, 1/80 of its points
         (1_250_000, 500, 0.05, 3))  # config 4's per-GPU shard at 8 GPUs (31M observations); not in the default list
for n, m, p, steps in ([CASES[int(a)] for a in sys.argv[1:]] or CASES[:4]):  # optional case indices
    sc = make_scene(n, m, vis_p=p)
    ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
    eng = ba._engine
    lm = LevenbergMarquardt(eng, 2.0)
    for _ in range(2):
        E_, _ = lm.iterate(); lm.carry_on(E_)
    eng.set_profiling(True); eng.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        E_, _ = lm.iterate(); lm.carry_on(E_)
    dt = time.perf_counter() - t0
    st = eng.stats()
    print(f"{n}x{m} p={p}: {steps/dt:8.1f} it/s  ms/step {dt/steps*1e3:7.3f}  rmse {np.sqrt(E_/sc.n_obs):.3e}",
          {k: round(v['ms'] / steps, 3) for k, v in st.items() if k != 'counts'})
