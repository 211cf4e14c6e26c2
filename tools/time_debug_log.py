"""What optimize(is_debug=True) costs at config 3 (1M points x 100 cameras: 24 MB per log entry kept on the device):
best of three optimize(2.0, -1.0, max_iter=10) with and without the log.  A measurement (profiles/r0N_debug_log_cost.txt),
not a parity property: the parity suite checks the log's contents only."""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib.bundle_adjustment import BundleAdjuster  # noqa: E402
from lib.synthetic import make_scene  # noqa: E402

sc = make_scene(1_000_000, 100, vis_p=0.1)
ba = BundleAdjuster.from_observations(sc.n_points, 100, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
state0 = ba._engine.get_params()


def timed(debug):
    best = 1e9
    for _ in range(3):
        ba._engine.set_params(*state0)
        ba._engine.cost()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            ba.optimize(2.0, -1.0, max_iter=10, is_debug=debug)
        best = min(best, time.perf_counter() - t0)
    return best


timed(True)  # warm-up: the log's device memory is allocated once and kept
t_plain, t_debug = timed(False), timed(True)
line = (f"config 3, optimize(2.0, -1.0, max_iter=10): is_debug=False {t_plain * 1e3:.2f} ms, is_debug=True {t_debug * 1e3:.2f} ms "
        f"(+{(t_debug / t_plain - 1) * 100:.2f} %), 11 log entries of 24 MB kept on the device")
print(line)
out = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(out):
    with open(os.path.join(out, "debug_log_cost.txt"), "w") as fh:
        fh.write(line + "\n")
