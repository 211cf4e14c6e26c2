"""The affine pipeline's self-calibration (ref affine_reconstruction.py:60-75 -> lib/affine_camera_calibration.py) at BASELINE config 5's
size through the public surface, with the wall time of its stages.  python tools/time_affine.py [points images [float32|float64]]"""
import os, sys, time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-reconstruction-from-multi-view-exp_amd"))
import numpy as np
from lib import _mvba
from lib import affine_camera_calibration as AC


def run(n=5_000_000, m=12, dtype=np.float32, reps=3):
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, (n, 3))
    data_list = []
    for k in range(m):  # orthographic views from an arc
        ph = 0.1 * k - 0.5
        R = np.array([[np.cos(ph), 0, -np.sin(ph)], [0, 1, 0]])
        data_list.append(np.ascontiguousarray((X @ R.T + rng.normal(0, 1e-3, (n, 2)) + [0.1 * k, -0.05 * k]).astype(dtype)))
    del X
    out = []
    for rep in range(reps):
        stages = {}

        def timed(mod, name):
            fn = getattr(mod, name)

            def w(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    stages[name] = stages.get(name, 0.0) + time.perf_counter() - t0
            setattr(mod, name, w)
            return fn

        saved = [(AC, "_svd_on_gpu", timed(AC, "_svd_on_gpu")), (AC, "_affine_core", timed(AC, "_affine_core")),
                 (AC, "_rotations", timed(AC, "_rotations")), (_mvba, "svd_factorize", timed(_mvba, "svd_factorize"))]
        try:
            t0 = time.perf_counter()
            Xr, R = AC.orthographic_self_calibration(data_list)
            stages["orthographic_self_calibration (total)"] = time.perf_counter() - t0
        finally:
            for mod, name, fn in saved:
                setattr(mod, name, fn)
        assert np.isfinite(Xr).all() and np.isfinite(R).all()
        out.append(stages)
        print(f"run {rep}: " + ", ".join(f"{k} {v:.3f} s" for k, v in stages.items()))
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]) if a else 5_000_000, int(a[1]) if len(a) > 1 else 12, np.dtype(a[2]).type if len(a) > 2 else np.float32)
