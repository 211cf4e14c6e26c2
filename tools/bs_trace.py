#!/usr/bin/env python3
"""Where a chain step of the persistent back-substitution goes (timing-only build with -DMVBA_BS_TRACE, see tools/README.md):
usage: MVBA_LIBRARY=tools/ab/libmvba_bs.so python tools/bs_trace.py m"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
import numpy as np
from lib import _mvba
from lib.bundle_adjustment import BundleAdjuster
from lib.synthetic import make_scene
m = int(sys.argv[1]); n = 40000 if m >= 300 else 100000
sc = make_scene(n, m, vis_p=0.05 if m >= 300 else 0.1)
ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
eng = ba._engine
eng.cost(); eng.linearize()
for _ in range(3): eng.try_step(1e-4)
lib = ctypes.CDLL(_mvba.LIB_PATH)
buf = (ctypes.c_longlong * (8 * 256))()
assert lib.mvba_debug_bs_trace(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(256, 8)
S = (9 * m - 7 + 127) // 128
t = t[:S]; t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["start", "loaded", "flag seen", "acquired", "y ready", "tiles done", "posted"]
print("chain s: " + " ".join(f"{n:>10s}" for n in names))
for s in range(S - 1, -1, -1):
    print(f"{s:7d}: " + " ".join(f"{us[s, i]:10.2f}" for i in range(7)))
d = np.diff(us[::-1, 6])  # posted(s) - posted(s+1)
print("step (post to post) us: mean %.2f  median %.2f;  flag->acquired %.2f  acquired->y %.2f  y->tiles %.2f  tiles->posted %.2f  post(s+1)->flag seen(s) %.2f" % (
    d[1:-1].mean(), np.median(d[1:-1]), (us[1:-1, 3] - us[1:-1, 2]).mean(), (us[1:-1, 4] - us[1:-1, 3]).mean(),
    (us[1:-1, 5] - us[1:-1, 4]).mean(), (us[1:-1, 6] - us[1:-1, 5]).mean(), (us[1:-2, 2] - us[2:-1, 6]).mean()))
