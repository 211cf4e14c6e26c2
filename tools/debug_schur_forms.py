"""Reduced system of one LM trial on the GPU against the oracle, for a Schur kernel form / scene given on the command line:
python tools/debug_schur_forms.py n m p [form] [big]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT):
    sys.path.insert(0, p)
n, m, p = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
if len(sys.argv) > 4:
    os.environ["MVBA_SCHUR"] = sys.argv[4]
if len(sys.argv) > 5 and sys.argv[5] == "big":
    os.environ["MVBA_FORCE_BIG"] = "1"
import numpy as np  # noqa: E402

from lib.bundle_adjustment import BundleAdjuster  # noqa: E402
from lib.synthetic import make_scene  # noqa: E402
from oracle import ba_oracle as O  # noqa: E402

sc = make_scene(n, m, vis_p=p)
ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
g = O.OracleEngine(n, m, sc.pt_ptr, sc.cam_idx, sc.xy, 1.0, sc.axis)
X, R, t = O.normalize_scene(sc.init_X, sc.init_R, sc.init_t, sc.axis)
g.set_params(X, sc.init_K[:, 0, 0], sc.init_K[:, :2, 2], t, R)
eng = ba._engine
eng.linearize(); g.linearize()
E1 = eng.try_step(1e-2)
A, b = g.reduced_system(1e-2)
Ag = eng.debug_read("A_full").reshape(9 * m, 9 * m)
err = np.abs(Ag - A).reshape(m, 9, m, 9).max(axis=(1, 3)) / np.abs(A).max()
print(eng.schur_info(), "max err / max|A| = %.3e" % err.max(), " b err %.3e" % (np.abs(eng.debug_read("b_full") - b).max() / np.abs(b).max()))
bad = np.argwhere(err > 1e-10)
print("bad blocks:", len(bad), "of", m * m, " first:", bad[:12].tolist())
