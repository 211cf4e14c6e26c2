"""Upload per depth iteration at 5M rows x 8 images (fp32): the depths z only (mvsvd_run_scaled, 160 MB) against the
whole re-weighted matrix (mvsvd_load, 480 MB).  A measurement (profiles/), not a parity property."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib import _mvba  # noqa: E402

rng = np.random.default_rng(0)
n, m = 5_000_000, 8
x = rng.standard_normal((n, 3 * m), dtype=np.float32)
z = (1.0 + 0.1 * rng.random((n, m), dtype=np.float32)).astype(np.float32)
ws = _mvba.SvdWorkspace(n, 3 * m, np.float32)
ws.load_base(x)
ws.run_scaled(z, 3, 1, 4)
M, s, S, tm = ws.run_scaled(z, 3, 1, 4)
W = (x.reshape(n, m, 3) * z[..., None]).reshape(n, 3 * m)
W /= np.linalg.norm(W, axis=1, keepdims=True)
_, _, _, _, tm_full = ws.load(W).run(4)
ws.close()
line = (f"5,000,000 x 24 fp32 depth iteration: upload of z {tm['h2d_ms']:.2f} ms (160 MB) vs upload of W {tm_full['h2d_ms']:.2f} ms (480 MB); "
        f"device: gram {tm['gram_ms']:.3f} jacobi {tm['jacobi_ms']:.3f} project {tm['project_ms']:.3f} ms")
print(line)
out = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(out):
    with open(os.path.join(out, "svd_scaled_5m.txt"), "w") as fh:
        fh.write(line + "\n")
