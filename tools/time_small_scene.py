"""The reference's own default scene (200 points x 10 cameras, tests/golden/euclid_default.npz) on the HIP engine: wall time of
optimize(2.0, 1e-8, 100), number of solves, and -- with the engine's per-kernel events on -- the device time inside it.
Answers whether the LM loop is launch-bound at this size.   python tools/time_small_scene.py"""
import contextlib, io, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"))
import numpy as np
from lib.bundle_adjustment import BundleAdjuster

d = np.load(os.path.join(ROOT, "tests", "golden", "euclid_default.npz"), allow_pickle=False)
args = (d["x"], d["init_X"], d["init_K"], d["init_R"], d["init_t"])
for prof in (False, True, False):
    ba = BundleAdjuster(*args, axis="x-up_z-forward")
    ba._engine.set_profiling(prof)
    ba._engine.reset_stats()
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        ba.optimize(2.0, 1e-8, max_iter=100)
        wall = time.perf_counter() - t0
    st = ba._engine.stats()
    ms = {k: v["ms"] for k, v in st.items() if k != "counts" and v["ms"]}
    print(f"profiling {prof}: optimize wall {1e3 * wall:.2f} ms, {ba._engine.n_solves} solves, {st['counts']}"
          + (f", device time of the timed kernels {sum(ms.values()):.3f} ms {({k: round(v, 3) for k, v in ms.items()})}" if prof else ""))
