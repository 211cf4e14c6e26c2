"""K3 (Schur) launch time of a scene shape, robust to timing-only builds whose numbers are wrong (knock-out / h-in-the-record
builds: the LM loop would fail on them): linearize once, then `try_step` n times, errors ignored, device time from the engine's
own hipEvents.   usage: python tools/time_schur.py [points cams vis [reps]]      (MVBA_LIBRARY picks the build)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib.bundle_adjustment import BundleAdjuster  # noqa: E402
from lib.synthetic import make_scene  # noqa: E402

n, m, vis = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1_000_000, 100, 0.10)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
sc = make_scene(n, m, vis_p=vis)
ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
eng = ba._engine
eng.linearize()
eng.set_profiling(True)
for phase in ("warm", "timed"):
    eng.reset_stats()
    for _ in range(3 if phase == "warm" else reps):
        try:
            eng.try_step(1e-4)
        except (np.linalg.LinAlgError, RuntimeError):
            pass
st = eng.stats()
info = eng.schur_info()
k = st["schur"]
print(f"{os.path.basename(os.environ.get('MVBA_LIBRARY', 'tree')):28s} {n}x{m}x{vis} {info['kernel']:6s} schur {k['ms'] / max(k['launches'], 1):.3f} ms/launch "
      f"({k['launches']} launches) rows {info['slot_rows']} items {info['items']}")
eng.close()
