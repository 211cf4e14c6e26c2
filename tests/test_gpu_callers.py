"""The callers either side of the hot path, end to end on the GPU: affine and perspective
self-calibration over the GPU SVD, and the two driver scripts' main() (plots off)."""
import os
import sys

import numpy as np
import pytest

from lib import _mvba
from lib import affine_camera_calibration as A
from lib import perspective_camera_calibration as P
from lib.camera import calc_projected_points

pytestmark = pytest.mark.gpu
os.environ.setdefault("MPLBACKEND", "Agg")


def _rmse(x_list, X, K, R, t):
    r = np.stack(calc_projected_points(X, K, R, t)) - np.stack(x_list)
    return np.sqrt((r**2).sum(axis=2).mean())


def test_affine_self_calibration_gpu_svd_vs_reference(golden):
    """GPU SVD factors -> aligned to the reference's singular-vector signs -> identical X, R;
    and the public functions (own sign rule) give either that reconstruction or its mirror image."""
    d = golden("calibration")
    xs = [x.copy() for x in d["aff_x"]]
    U3, S3, t = A._svd_on_gpu(xs)
    np.testing.assert_allclose(t, d["aff_t"], atol=1e-13)
    sg = np.sign(np.sum(U3 * d["aff_U3"], axis=0))
    np.testing.assert_allclose(U3 * sg, d["aff_U3"], atol=1e-9)
    np.testing.assert_allclose(S3 * sg[:, None], np.diag(d["aff_sigma"][:3]) @ d["aff_Vt3"], atol=1e-8)
    for model, key, f in (("orthographic", "ortho", None), ("symmetric_affine", "symaff", None),
                          ("paraperspective", "para", np.ones(12))):
        X, R = A._affine_core(model, U3 * sg, S3 * sg[:, None], t, f)
        np.testing.assert_allclose(X, d[key + "_X"], atol=1e-7, err_msg=key)
        np.testing.assert_allclose(R, d[key + "_R"], atol=1e-7, err_msg=key)
    Xp, Rp = A.paraperspective_self_calibration(xs, np.ones(12))
    Xo, Ro = A.orthographic_self_calibration(xs)
    Xs, Rs = A.symmetric_affine_self_calibration(xs)
    for X, R, key in ((Xp, Rp, "para"), (Xo, Ro, "ortho"), (Xs, Rs, "symaff")):
        assert X.shape == (200, 3) and R.shape == (12, 3, 3)
        np.testing.assert_allclose(np.einsum("kij,kil->kjl", R, R), np.tile(np.eye(3), (12, 1, 1)), atol=1e-12)
        # same shape up to a rigid motion or a mirror: pairwise distances agree
        iu = np.triu_indices(200, 1)
        dist = lambda Y: np.linalg.norm(Y[:, None] - Y[None], axis=2)[iu]  # noqa: E731
        np.testing.assert_allclose(dist(X), dist(d[key + "_X"]), rtol=1e-6, atol=1e-8)
    with pytest.raises(ValueError):
        A.paraperspective_self_calibration(xs, np.ones(5))


@pytest.mark.parametrize("method", ["dual", "primary"])
def test_perspective_self_calibration_gpu(golden, method, capsys):
    d = golden("calibration")
    xs = [x.copy() for x in d["persp_x"]]
    X, R, t, K = P.perspective_self_calibration(xs, 1.0, tol=1e-2, method=method)
    out = capsys.readouterr().out
    assert out.splitlines()[0] == str(d[f"persp_{method}_stdout"]).strip().splitlines()[0]
    assert X.shape == (200, 3) and R.shape == (10, 3, 3) and t.shape == (10, 3) and K.shape == (10, 3, 3)
    ref = _rmse(xs, d[f"persp_{method}_X"], d[f"persp_{method}_K"], d[f"persp_{method}_R"], d[f"persp_{method}_t"])
    mine = _rmse(xs, X, K, R, t)
    if method == "dual":
        # the pipeline is invariant to the SVD's sign convention (SURVEY §7 hard part 4)
        assert abs(mine - ref) < 1e-6, (mine, ref)
        np.testing.assert_allclose(np.abs(K), np.abs(d["persp_dual_K"]), rtol=1e-5, atol=1e-7)
        iu = np.triu_indices(200, 1)
        dist = lambda Y: np.linalg.norm(Y[:, None] - Y[None], axis=2)[iu]  # noqa: E731
        np.testing.assert_allclose(dist(X), dist(d["persp_dual_X"]), rtol=1e-5, atol=1e-7)
    else:
        assert mine < 1.5 * ref + 1e-3, (mine, ref)  # (sanity only: the stages are pinned one by one below)
    with pytest.raises(ValueError):
        P.perspective_self_calibration(xs, method="bogus")


@pytest.mark.parametrize("method", ["primary", "dual"])
def test_perspective_self_calibration_stage_by_stage_on_the_gpu(golden, method, capsys):
    """Every stage of perspective_self_calibration (ref perspective_camera_calibration.py:513-540) on the GPU
    SVD against what the reference produced at that stage: converged depths (:61-144 / :147-235), the
    factorization of the re-weighted matrix (:533), one turn of the Omega <-> K loop on it (:238-411).
    The primary method's END result is chaotic on this scene (J = 1.9e9 after one turn: a 1e-15
    difference is O(1) two turns later), which is why the stages, not the end, carry the parity."""
    from lib.factorization import factorization_method

    d = golden("calibration")
    x = P._create_data_matrix([a.copy() for a in d["persp_x"]], 1.0)
    fn = getattr(P, f"_compute_projective_depth_{method}_method")
    z = fn(x, 1.0, 1e-2)
    capsys.readouterr()
    np.testing.assert_allclose(z, np.abs(d[f"persp_{method}_z"]), rtol=0, atol=1e-9)  # (per-image sign: see the CPU test)
    # the reference's own depths (sign included) -> the GPU factorization == the reference's M, S
    W = x * d[f"persp_{method}_z"][..., None]
    M, S = factorization_method(W.reshape(W.shape[0], -1).T)
    Mr, Sr = d[f"persp_{method}_M"], d[f"persp_{method}_S"]
    sg = np.sign(np.sum(M * Mr, axis=0))
    assert (np.abs(sg) == 1).all()
    np.testing.assert_allclose(M * sg, Mr, rtol=0, atol=1e-9)
    np.testing.assert_allclose(S * sg[:, None], Sr, rtol=0, atol=1e-8 * np.abs(Sr).max())
    np.testing.assert_allclose(M @ S, Mr @ Sr, rtol=0, atol=1e-9 * np.abs(Mr @ Sr).max())
    # one turn of the upgrade loop on the GPU factors
    Pm = (M * sg).reshape(-1, 3, 4)
    K0 = np.tile(np.eye(3), (10, 1, 1))
    Q0 = np.linalg.inv(K0) @ Pm
    Om, sig, _w = P._calc_omega(Q0)
    np.testing.assert_allclose(Om, d[f"persp_{method}_Omega1"], rtol=0, atol=1e-7 * np.abs(d[f"persp_{method}_Omega1"]).max())
    np.testing.assert_allclose(sig, d[f"persp_{method}_sigma1"], rtol=1e-6, atol=1e-9)
    K1, J1 = P._update_K(K0.copy(), Om, Q0)
    np.testing.assert_allclose(K1, d[f"persp_{method}_K1"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(J1, d[f"persp_{method}_J1"], rtol=1e-5)


def test_projective_depths_per_step_on_the_gpu_svd(golden, capsys):
    """Both depth iterations pinned PER STEP on the GPU SVD (resident workspace), as the CPU test
    pins them on NumPy's: three forced iterations against the reference's depths.  (The end-to-end
    primary-method upgrade is chaotic on this scene -- J = 1.9e9 after one step -- which is why its
    final error is only bounded above; the depths feeding it are exact.)"""
    d = golden("calibration")
    x = P._create_data_matrix([a.copy() for a in d["persp_x"]], 1.0)
    z = P._compute_projective_depth_primary_method(x, 1.0, 0.0, 3)
    np.testing.assert_allclose(z, d["persp_primary_z3"], rtol=0, atol=1e-9)
    z = P._compute_projective_depth_dual_method(x, 1.0, 0.0, 3)
    np.testing.assert_allclose(z, np.abs(d["persp_dual_z3"]), rtol=0, atol=1e-9)  # per-image sign: see the CPU test
    capsys.readouterr()


def test_euclidean_driver_reproduces_the_reference_run(golden, capsys):
    """Package-root driver with the reference's call sequence: same observations, then BA from a
    self-calibrated start converging to the reference's final reprojection error."""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-reconstruction-from-multi-view-exp_amd")
    sys.path.insert(0, pkg)
    import euclidiean_reconstruction as drv

    d = golden("euclid_default")
    x_list, (X, K, R, t), log = drv.main(show=False)
    out = capsys.readouterr().out
    assert "Bundle Adjustment" in out and "Iteration 1: reprojection_error_delta = " in out
    np.testing.assert_allclose(np.stack(x_list).transpose(1, 0, 2), d["x"], atol=1e-13)
    E = np.array([e["reprojection_error"] for e in log])
    rmse, rmse_ref = np.sqrt(E[-1] / 2000), np.sqrt(d["E_log"][-1] / 2000)
    assert abs(rmse - rmse_ref) < 1e-7, (rmse, rmse_ref)  # same minimum (start differs by a gauge only)
    # The START of BA is not pinned here: the reference's Omega <-> K upgrade loop (:383-411) ends on an update whose median
    # residual has jumped back to J ~ 1 (J_med = 2.4e-2, 1.2e-3, 1.02: it stops BECAUSE it got worse, and keeps that K), and
    # that last step amplifies a 1e-13 difference in the depths (device iteration vs NumPy: different rounding) to 2e-3 in
    # K -- measured: E0 = 66.508 with the device depths of THIS x_list, 66.319 with the oracle's or with the fixture's x
    # (equal to 1e-13).  Every stage is pinned on the reference's own inputs in the stage-by-stage tests; here: the same
    # minimum from whichever start, and a start in the same neighbourhood.
    assert abs(E[0] - d["E_log"][0]) < 2e-2 * d["E_log"][0]
    assert _rmse(x_list, X, K / K[:, 2:3, 2:3], R, t) < 0.01


def test_euclidean_driver_start_of_ba_is_pinned_through_the_oracle_depth_loop(golden, capsys, monkeypatch):
    """The tight check the device path cannot hold (see above), through a path that can: the same driver with the NumPy depth loop
    of oracle/depth_oracle.py in place of the device loop starts BA at the reference's E0 to 1e-5 (everything else -- GPU SVD,
    Euclidean upgrade, BA on the HIP engine -- unchanged); and the device loop itself is pinned where it is well conditioned: its
    depths after the driver's whole depth loop (dual scheme, tol 1e-2) equal the oracle's to 1e-10."""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-reconstruction-from-multi-view-exp_amd")
    sys.path.insert(0, pkg)
    import euclidiean_reconstruction as drv
    import lib.perspective_camera_calibration as P
    from oracle.depth_oracle import HostDepthLoop

    d = golden("euclid_default")
    x_list = [d["x"][:, k, :] for k in range(d["x"].shape[1])]
    x = P._create_data_matrix(x_list, 1.0)
    z_dev = P._compute_projective_depth_dual_method(x, 1.0, 1e-2)
    z_host = P._compute_projective_depth_dual_method(x, 1.0, 1e-2, loop=HostDepthLoop(x))
    np.testing.assert_allclose(z_dev, z_host, rtol=0, atol=1e-10)
    monkeypatch.setattr(P, "_DeviceDepthLoop", HostDepthLoop)
    _x, (X, K, R, t), log = drv.main(show=False)
    capsys.readouterr()
    E = np.array([e["reprojection_error"] for e in log])
    assert abs(E[0] - d["E_log"][0]) < 1e-5 * d["E_log"][0], (E[0], d["E_log"][0])
    assert abs(np.sqrt(E[-1] / 2000) - np.sqrt(d["E_log"][-1] / 2000)) < 1e-9


def test_euclidean_pipeline_end_to_end_at_a_million_points():
    """The reference's Euclidean pipeline (euclidiean_reconstruction.py:36-57) through the public surfaces at 1,000,000 points x 12
    images, full visibility: synthetic observations -> perspective_self_calibration(x_list, 1.0, tol=1e-2, "dual") ->
    BundleAdjuster(np.stack(x_list).transpose(1, 0, 2), ...).optimize(2.0, 1e-8, max_iter=30)  (tools/time_pipeline.py, which also
    times every stage).  Too large for the oracle in a test, so properties: everything finite, the self-calibrated start already
    reprojects to a few pixels' worth (1e-2 in normalised units), BA brings the reprojection RMSE -- evaluated independently through
    calc_projected_points_gpu -- to the noise floor."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import time_pipeline

    res = time_pipeline.run(1_000_000, 12, max_iter=30, tol=1e-2, noise=1e-3)
    _timing_line("pipeline 1M x 12 (full visibility): " + ", ".join(f"{k} {v:.3f} s" for k, v in res["stages_s"].items())
                 + f"; wall {res['pipeline_wall_s']:.3f} s, depth iterations {res['depth_iterations']}, LM iterations {res['lm_iterations']}")
    assert res["finite"] and res["depth_iterations"] >= 1 and res["lm_iterations"] >= 5
    assert res["rmse_after_self_calibration"] < 5e-2
    floor = res["noise_floor_expected"]  # sqrt(2) sigma; the fit absorbs (3 N + 9 m) of the 2 N m degrees of freedom
    assert 0.8 * floor < res["rmse_after_bundle_adjustment"] < 1.02 * floor, res
    # (the stage times go to the timing log only: with the wall at 0.3 s a single page-fault burst in one NumPy stage is a quarter of it,
    # and wall-clock comparisons do not belong in the parity suite -- tools/time_pipeline.py, profiles/r05_pipeline_1m_x12.txt)


def test_euclidean_pipeline_end_to_end_with_a_hundred_images():
    """The same pipeline at 20,000 points x 100 images, full visibility: W = 300 x N, so every factorisation inside the projective-depth
    loop is the wide path of the SVD (block power iteration; the Gram + Jacobi route needed ~0.4 s per 300-column eigenproblem, two
    per depth iteration), the depth updates run their 100-image variants, and BA solves a 893-unknown reduced system on 2 M
    observations.  Same properties as at a million points."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import time_pipeline

    res = time_pipeline.run(20_000, 100, max_iter=30, tol=1e-2, noise=1e-3)
    _timing_line("pipeline 20k x 100 (full visibility): " + ", ".join(f"{k} {v:.3f} s" for k, v in res["stages_s"].items())
                 + f"; wall {res['pipeline_wall_s']:.3f} s, depth iterations {res['depth_iterations']}, LM iterations {res['lm_iterations']}")
    assert res["finite"] and res["depth_iterations"] >= 1 and res["lm_iterations"] >= 5
    assert res["rmse_after_self_calibration"] < 5e-2
    floor = res["noise_floor_expected"]
    assert 0.8 * floor < res["rmse_after_bundle_adjustment"] < 1.02 * floor, res


def _timing_line(text):
    """Full-size runs leave their timing lines under gpurun_out/ (copied to profiles/ by hand)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "fullsize_timings.txt"), "a") as fh:
            fh.write(text + "\n")
    print(text)


def test_affine_driver_runs(golden, capsys):
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-reconstruction-from-multi-view-exp_amd")
    sys.path.insert(0, pkg)
    import affine_reconstruction as drv

    d = golden("affine_default")
    x_list, (X, K, R, t), log = drv.main(show=False)
    np.testing.assert_allclose(np.stack(x_list).transpose(1, 0, 2), d["x"], atol=1e-13)
    E = np.array([e["reprojection_error"] for e in log])
    # The reference's run: 100 outer iterations (max_iter) ending at E = 0.21790752620130377.  Which minimum BA
    # reaches depends on the PARITY of the three singular-vector signs (an odd number of flips mirrors the
    # affine reconstruction: BA then stops after 36 solves at 0.09933032816).  The repo's sign rule ("largest
    # entry of each column of U positive") has LAPACK's parity on this scene, so the driver reproduces the
    # reference's minimum -- tests/test_callers_cpu.py pins both minima on the oracle.
    assert len(E) == len(d["E_log"]) == 101
    assert E[-1] == pytest.approx(0.21790752620130377, rel=1e-6), E[-1]
    assert E[0] == pytest.approx(float(d["E_log"][0]), rel=1e-6)
    # and the opt-in mirror parity (lib.affine_camera_calibration._svd_on_gpu(..., mirror=True))
    U3, S3, t3 = A._svd_on_gpu([x.copy() for x in x_list], mirror=True)
    U3r, S3r, _ = A._svd_on_gpu([x.copy() for x in x_list])
    np.testing.assert_allclose(U3[:, :2], U3r[:, :2], atol=0)
    np.testing.assert_allclose(U3[:, 2], -U3r[:, 2], atol=0)
    np.testing.assert_allclose(S3[2], -S3r[2], atol=0)


def test_device_projection_matches_lib_camera(golden):
    """mvba_project (csrc/mvba.hip k_project_obs) vs the host camera model on the reference's own
    default scene: the fixture's noise-free projections x_clean were produced by the reference's
    calc_projected_points (ref lib/camera.py:74-81); dense grid and observation-list forms."""
    from lib import _mvba
    from lib.camera import calc_projected_points, calc_projected_points_gpu
    from lib.synthetic import make_scene

    d = golden("euclid_default")
    X, K, R, t = d["X_gt"], d["K_gt"], d["R_gt"], d["t_gt"]
    x_gpu = calc_projected_points_gpu(X, K, R, t)
    x_host = calc_projected_points(X, K, R, t)
    assert len(x_gpu) == len(x_host) == 10
    for k in range(10):
        np.testing.assert_allclose(x_gpu[k], d["x_clean"][k], rtol=0, atol=1e-13)
        np.testing.assert_allclose(x_gpu[k], x_host[k], rtol=0, atol=1e-13)
    # observation list: every third (point, camera) pair, non-trivial intrinsics
    rng = np.random.default_rng(5)
    K2 = K.copy()
    K2[:, 0, 0] = K2[:, 1, 1] = 1.0 + 0.1 * rng.uniform(size=10)
    K2[:, :2, 2] = 0.05 * rng.normal(size=(10, 2))
    vis = rng.uniform(size=(200, 10)) < 0.35
    pt, cam = np.nonzero(vis)
    pt_ptr = np.zeros(201, np.int64)
    np.cumsum(vis.sum(axis=1), out=pt_ptr[1:])
    xy = _mvba.project(X, K2, R, t, pt_ptr, cam)
    ref = np.stack(calc_projected_points(X, K2, R, t), axis=1)[pt, cam]
    np.testing.assert_allclose(xy, ref, rtol=0, atol=1e-13)
    with pytest.raises(ValueError):
        _mvba.project(X, K, R, t, pt_ptr, np.full_like(cam, 10))
    # the synthetic-scene generator on the device == on the host
    a = make_scene(70_000, 12, vis_p=0.2, project="gpu")
    b = make_scene(70_000, 12, vis_p=0.2, project="numpy")
    np.testing.assert_array_equal(a.cam_idx, b.cam_idx)
    np.testing.assert_allclose(a.xy, b.xy, rtol=0, atol=1e-13)
