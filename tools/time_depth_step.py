"""One projective-depth iteration at 5M points x 8 images (or: <points> <images>), fp64, entirely on the device (mvsvd_depth_step): wall time per
iteration and its device phases, both schemes -> a line for profiles/.  (The NumPy form of the same iteration took
~6 s at 1 M points on 8 cores: einsum 2.0 + batched 4 x 4 eigh 3.2 + reprojection 0.9.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib import _mvba  # noqa: E402

n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(0)
# a projective scene: points in front of m cameras, homogeneous image coordinates
X = rng.uniform(-1, 1, (n, 3))
lines = []
x = np.empty((n, m, 3))
for k in range(m):
    ph = 0.12 * k - 0.4  # cameras on an arc of radius 5 around the points, looking at the origin
    c = 5.0 * np.array([np.sin(ph), 0.0, -np.cos(ph)])
    R = np.array([[np.cos(ph), 0, -np.sin(ph)], [0, 1, 0], [np.sin(ph), 0, np.cos(ph)]])  # columns: right, up, forward
    Xc = (X - c) @ R
    x[:, k, 0], x[:, k, 1], x[:, k, 2] = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2], 1.0
x += 1e-3 * rng.standard_normal(x.shape) * np.array([1.0, 1.0, 0.0])
ws = _mvba.SvdWorkspace(n, 3 * m, np.float64)
t0 = time.perf_counter()
ws.load_base(x.reshape(n, 3 * m))
t_up = time.perf_counter() - t0
for method, name in ((1, "primary"), (2, "dual")):
    ws.depth_begin(3)
    ws.depth_step(method, 1.0)  # warm-up (allocations)
    walls, tms, Es = [], [], []
    for _ in range(5):
        t0 = time.perf_counter()
        E, tm = ws.depth_step(method, 1.0)
        walls.append(time.perf_counter() - t0)
        tms.append(tm); Es.append(E)
    tm = tms[int(np.argmin(walls))]
    t0 = time.perf_counter()
    z = ws.depth_read()
    t_dl = time.perf_counter() - t0
    lines.append(f"{n} x {m} fp64 {name} depth iteration on the device: wall {min(walls) * 1e3:.2f} ms (median {np.median(walls) * 1e3:.2f}); device: gram {tm['gram_ms']:.3f} "
                 f"jacobi {tm['jacobi_ms']:.3f} refine {tm['refine_ms']:.3f} project {tm['project_ms']:.3f} depth update {tm['depth_ms']:.3f} ms; "
                 f"PCIe per iteration: 8 bytes; E after 6 iterations {Es[-1]:.3e}; one-off: upload of x {t_up * 1e3:.0f} ms, download of z {t_dl * 1e3:.0f} ms")
ws.close()
for ln in lines:
    print(ln)
out = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(out):
    with open(os.path.join(out, "depth_step_5m.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
