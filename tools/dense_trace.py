"""Where the two roles of k_schur_dense spend a launch: work or the chunk barrier (timing-only build -DMVBA_DENSE_TRACE,
tools/ab/libmvba_dtrace.so).  python tools/dense_trace.py [points cams vis]"""
import ctypes as C, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MVBA_LIBRARY"] = os.path.join(ROOT, "tools", "ab", "libmvba_dtrace.so")
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
import numpy as np
from lib import _mvba
from lib.bundle_adjustment import BundleAdjuster
from lib.synthetic import make_scene

n, m, vis = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1_000_000, 12, 1.0)
sc = make_scene(n, m, vis_p=vis)
ba = BundleAdjuster.from_observations(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t, axis=sc.axis)
eng = ba._engine
eng.linearize()
eng.set_profiling(True)
for _ in range(4):
    eng.try_step(1e-4)
eng.reset_stats()
eng.try_step(1e-4)
ms = eng.stats()["schur"]["ms"]
lib = _mvba.load_library()
tr = np.zeros(1024 * 16 * 4, dtype=np.int64)
assert lib.mvba_dense_trace_read(tr.ctypes.data_as(C.c_void_p), tr.size) == 0
tr = tr.reshape(1024, 16, 4)
tr = tr[tr[:, 0, 2] > 0]  # the workgroups of this launch (256 or 512)
T = (9 * m + 15) // 16
nc, ch = (4 if T <= 8 else 8), (8 if T == 8 else 4)  # consumer / producer waves (dense_consumers, dense_ch in mvba.hip)
cons, prod = tr[:, :nc], tr[:, nc:nc + ch]
print(f"{n} x {m} x {vis}: K3 {ms:.3f} ms; {len(tr)} workgroups, chunks per workgroup {int(cons[:, 0, 3].mean())}")
for name, r in (("consumers", cons), ("producers", prod)):
    tot = r[..., 2].astype(float)
    print(f"  {name}: total ticks median {np.median(tot):.0f}; work {100 * np.median(r[..., 0] / tot):.1f} %  barrier wait {100 * np.median(r[..., 1] / tot):.1f} % "
          f"(p10 {100 * np.percentile(r[..., 1] / tot, 10):.1f}, p90 {100 * np.percentile(r[..., 1] / tot, 90):.1f}); ticks per chunk: work {np.median(r[..., 0] / np.maximum(r[..., 3], 1)):.0f}, wait {np.median(r[..., 1] / np.maximum(r[..., 3], 1)):.0f}")
eng.close()
