"""The Euclidean pipeline end to end at scale through the PUBLIC surfaces (ref euclidiean_reconstruction.py:36-57):
synthetic scene (N points x m images, full visibility) -> perspective_self_calibration(x_list, 1.0, tol, "dual") ->
BundleAdjuster(np.stack(x_list).transpose(1, 0, 2), X, K, R, t, axis="x-up_z-forward").optimize(2.0, 1e-8, max_iter), with the wall
time of every stage (the library's own functions wrapped in timers, nothing else changed) and property checks at the end.
usage: python tools/time_pipeline.py [points images [max_iter [tol]]]      (default 1,000,000 x 12)"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]


def run(n_points=1_000_000, n_images=12, max_iter=30, tol=1e-2, noise=1e-3, quiet=True):
    import lib.bundle_adjustment as BA
    import lib.perspective_camera_calibration as PC
    from lib.camera import calc_projected_points_gpu
    from lib.synthetic import make_scene

    stages = {}

    def timed(mod, name, label=None):
        fn = getattr(mod, name)

        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                stages[label or name] = stages.get(label or name, 0.0) + time.perf_counter() - t0

        setattr(mod, name, wrapper)
        return fn

    t_all = time.perf_counter()
    t0 = time.perf_counter()
    sc = make_scene(n_points, n_images, vis_p=1.0, noise=noise)
    x = sc.xy.reshape(n_points, n_images, 2)  # full visibility: the observation list IS the dense array
    x_list = [np.ascontiguousarray(x[:, k]) for k in range(n_images)]
    stages["scene generation (not part of the pipeline)"] = time.perf_counter() - t0

    saved = [(PC, n, timed(PC, n, lab)) for n, lab in (("_create_data_matrix", "self-calibration: _create_data_matrix"),
                                                        ("_depth_iterations", "self-calibration: projective depths (device loop)"),
                                                        ("_euclidean_upgrading", "self-calibration: _euclidean_upgrading"),
                                                        ("_reconstruct_3d", "self-calibration: _reconstruct_3d"),
                                                        ("correct_world_coordinates", "self-calibration: correct_world_coordinates"))]
    saved += [(PC._DeviceDepthLoop, "from_images", timed(PC._DeviceDepthLoop, "from_images", "self-calibration: upload of the images, x assembled on the device (depth loop workspace)")),
              (PC._DeviceDepthLoop, "factorize", timed(PC._DeviceDepthLoop, "factorize", "self-calibration: final factorisation of x o z (on the device, no upload)"))]
    saved += [(BA, "dense_to_observations", timed(BA, "dense_to_observations", "BundleAdjuster(): dense_to_observations")),
              (BA, "to_gauge_frame", timed(BA, "to_gauge_frame", "BundleAdjuster(): to_gauge_frame")),
              (BA, "lm_loop", timed(BA, "lm_loop", "optimize(): LM loop on the device"))]
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out if quiet else sys.stdout):
            t0 = time.perf_counter()
            X_, R_, t_, K_ = PC.perspective_self_calibration(x_list, 1.0, tol=tol, method="dual")
            stages["perspective_self_calibration (total)"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            xs = np.stack(x_list).transpose(1, 0, 2)
            stages["np.stack(x_list).transpose(1, 0, 2)"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            ba = BA.BundleAdjuster(xs, X_, K_, R_, t_, axis="x-up_z-forward")
            stages["BundleAdjuster() (total: observation list, gauge frame, mvba_create, upload)"] = time.perf_counter() - t0
            E0 = ba._engine.cost()
            t0 = time.perf_counter()
            X2, K2, R2, t2 = ba.optimize(2.0, 1e-8, max_iter=max_iter)
            stages["optimize() (total)"] = time.perf_counter() - t0
            E1 = ba._engine.cost() if False else None
    finally:
        for mod, name, fn in saved:
            setattr(mod, name, fn)
    depth_iters = out.getvalue().count("reprojection_error =")
    lm_iters = out.getvalue().count("reprojection_error_delta")
    wall = time.perf_counter() - t_all - stages["scene generation (not part of the pipeline)"]
    # properties: the reconstruction reprojects onto the observations at the noise floor (through the device projection)
    t0 = time.perf_counter()
    proj = np.stack(calc_projected_points_gpu(X2, K2, R2, t2), axis=1)
    rmse = float(np.sqrt(np.mean(np.sum((proj - x) ** 2, axis=2))))
    rmse0 = float(np.sqrt(np.mean(np.sum((np.stack(calc_projected_points_gpu(X_, K_, R_, t_), axis=1) - x) ** 2, axis=2))))
    t_check = time.perf_counter() - t0
    return {"workload": f"{n_points} points x {n_images} images, full visibility, noise {noise}; perspective_self_calibration(tol={tol}, 'dual') "
                        f"-> BundleAdjuster(...).optimize(2.0, 1e-8, max_iter={max_iter})",
            "pipeline_wall_s": wall, "stages_s": stages, "depth_iterations": depth_iters, "lm_iterations": lm_iters,
            "rmse_after_self_calibration": rmse0, "rmse_after_bundle_adjustment": rmse, "ba_cost_start": E0,
            "noise_floor_expected": noise * np.sqrt(2.0), "check_s": t_check,
            "finite": bool(np.isfinite(X2).all() and np.isfinite(R2).all() and np.isfinite(t2).all() and np.isfinite(K2).all())}


if __name__ == "__main__":
    a = sys.argv[1:]
    res = run(int(a[0]) if a else 1_000_000, int(a[1]) if len(a) > 1 else 12, int(a[2]) if len(a) > 2 else 30, float(a[3]) if len(a) > 3 else 1e-2)
    print(json.dumps(res, indent=1))
    tot = res["pipeline_wall_s"]
    print("\nstage                                                                              s      % of the pipeline wall", file=sys.stderr)
    for k, v in res["stages_s"].items():
        print(f"  {k:80s} {v:8.3f}  {100 * v / tot:5.1f}", file=sys.stderr)
