"""Wall time of mvba_create's stages (MVBA_CREATE_TRACE=1) for a fully visible scene handed over as image planes, and of the whole
BundleAdjuster() around it.  python tools/time_create.py [points images]"""
import os, sys, time

os.environ["MVBA_CREATE_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-reconstruction-from-multi-view-exp_amd"))
import numpy as np
from lib import _mvba
from lib.bundle_adjustment import dense_to_observations
from lib.synthetic import make_scene

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 12)
sc = make_scene(n, m, vis_p=1.0, noise=1e-3)
xs = np.ascontiguousarray(sc.xy.reshape(n, m, 2).transpose(1, 0, 2)).transpose(1, 0, 2)  # the caller's np.stack(x_list).transpose(1, 0, 2)
for rep in range(3):
    t0 = time.perf_counter()
    pt_ptr, cam, xy = dense_to_observations(xs, None)
    t1 = time.perf_counter()
    e = _mvba.HipEngine(n, m, pt_ptr, cam, xy, 1.0, sc.axis)
    t2 = time.perf_counter()
    print(f"run {rep}: dense_to_observations {1e3 * (t1 - t0):.1f} ms, HipEngine() {1e3 * (t2 - t1):.1f} ms", file=sys.stderr)
    e.close()
