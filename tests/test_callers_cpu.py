"""Scene helpers / camera model (the callers' side of the path) against reference vectors."""
import numpy as np

from lib.camera import Camera, calc_projected_points, get_camera_parames
from lib.utils import get_rotation_matrix, sample_hemisphere_points, set_points, unit_vec


def test_utils_against_reference(golden):
    d = golden("known_answers")
    for w, Rref in zip(d["omega"], d["rodrigues"]):
        np.testing.assert_allclose(get_rotation_matrix(w), Rref, rtol=0, atol=1e-15)
    assert (get_rotation_matrix(np.zeros(3)) == np.eye(3)).all()
    np.random.seed(5)
    np.testing.assert_allclose(sample_hemisphere_points(6, 5), d["hemi_seed5"], atol=1e-14)
    np.testing.assert_allclose(set_points(), d["set_points"], atol=1e-15)
    assert np.linalg.norm(unit_vec(np.array([3.0, 4.0, 0.0]))) == 1.0


def test_camera_against_reference(golden):
    d = golden("known_answers")
    cam = Camera.create((1.0, 2.0, -3.0), (0.1, -0.2, 0.3), f=1.3, f0=1.0)
    K, R, t = cam.get_parameters()
    np.testing.assert_allclose(K, d["cam_K"], atol=1e-15)
    np.testing.assert_allclose(R, d["cam_R"], atol=1e-15)
    np.testing.assert_allclose(t, d["cam_t"], atol=1e-15)
    np.testing.assert_allclose(cam.get_camera_matrix(), d["cam_P"], atol=1e-14)
    pts = set_points()[:15]
    np.testing.assert_allclose(cam.project_points(pts), d["cam_proj"], atol=1e-14)
    np.testing.assert_allclose(cam.project_points(pts, method="orthographic"), d["cam_ortho"], atol=1e-14)
    # the reference's own inline known answers (lib/camera.py:101-117)
    X = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    np.testing.assert_array_almost_equal(Camera.create((0, 0, -1), (0, 0, 1), f=1).project_points(X),
                                         np.array([[0, 0], [1, 0], [0, 1], [0, 0]]))
    np.testing.assert_array_almost_equal(Camera.create((0, -1, 0), (0, 1, 0), f=1).project_points(X),
                                         np.array([[0, 0], [1, 0], [0, 0], [0, -1]]))


def test_default_scene_generation_matches_the_scripts(golden):
    """euclidiean_reconstruction.py:14-40 with seed 123 -> the BA inputs' observations."""
    d = golden("euclid_default")
    np.random.seed(123)
    pos = sample_hemisphere_points(10, 5)
    targets = np.random.normal(0, 0.5, (10, 3))
    cams = [Camera.create(p, t, f=1.0, f0=1.0) for p, t in zip(pos, targets)]
    K, R, t = get_camera_parames(cams)
    np.testing.assert_allclose(R, d["R_gt"], atol=1e-14)
    x_list = calc_projected_points(set_points(), K, R, t)
    np.testing.assert_allclose(np.stack(x_list), d["x_clean"], atol=1e-13)
    for x in x_list:
        x += 0.005 * np.random.randn(*x.shape)
    np.testing.assert_allclose(np.stack(x_list).transpose(1, 0, 2), d["x"], atol=1e-13)


def test_affine_post_svd_pipeline_against_reference(golden):
    """Everything downstream of the SVD, fed with the reference's own SVD factors."""
    from lib import affine_camera_calibration as A

    d = golden("calibration")
    U3, t = d["aff_U3"], d["aff_t"]
    S3 = np.diag(d["aff_sigma"][:3]) @ d["aff_Vt3"]
    W, t_host = A._observation_matrix_host([x.copy() for x in d["aff_x"]])
    np.testing.assert_allclose(W, d["aff_W"], atol=1e-14)
    np.testing.assert_allclose(t_host, t, atol=1e-15)
    for model, key, f in (("orthographic", "ortho", None), ("symmetric_affine", "symaff", None),
                          ("paraperspective", "para", np.ones(12)), ("paraperspective", "para2", d["para2_f"])):
        X, R = A._affine_core(model, U3, S3, t, f)
        np.testing.assert_allclose(X, d[key + "_X"], rtol=0, atol=1e-9, err_msg=key)
        np.testing.assert_allclose(R, d[key + "_R"], rtol=0, atol=1e-9, err_msg=key)


def test_projective_depths_low_rank_restatement_equals_reference(golden, capsys):
    """The 4x4 / 12x12 companion-eigenproblem forms (oracle/depth_oracle.py: what the device kernels implement)
    reproduce the reference's depths: three forced iterations of each scheme and the converged loops, the product's
    loop control (print, stop rule) over the oracle's iteration -- NumPy SVD, runs without a GPU."""
    from lib import perspective_camera_calibration as P
    from oracle.depth_oracle import HostDepthLoop

    d = golden("calibration")
    x = P._create_data_matrix([a.copy() for a in d["persp_x"]], 1.0)
    z = P._compute_projective_depth_primary_method(x, 1.0, 0.0, 3, loop=HostDepthLoop(x))
    np.testing.assert_allclose(z, d["persp_primary_z3"], rtol=0, atol=1e-10)
    z = P._compute_projective_depth_dual_method(x, 1.0, 0.0, 3, loop=HostDepthLoop(x))
    # The sign of each image's depth vector is an eigenvector sign: LAPACK-dependent in the
    # reference (one image comes out negated in this vector), always positive here; the two are
    # projectively equivalent (P_k ~ -P_k).  Compare up to that per-image sign.
    ref = d["persp_dual_z3"]
    assert (np.abs(np.sign(ref).sum(axis=0)) == ref.shape[0]).all()  # whole columns share a sign
    np.testing.assert_allclose(z, np.abs(ref), rtol=0, atol=1e-10)
    out = capsys.readouterr().out
    assert out.count("Iteration 3: reprojection_error = ") == 2
    assert "Did not converge because the maximum number of iterations was reached." in out
    for m in ("primary", "dual"):
        fn = getattr(P, f"_compute_projective_depth_{m}_method")
        z1 = fn(x, 1.0, 1e-2, loop=HostDepthLoop(x))
        np.testing.assert_allclose(z1, np.abs(d[f"persp_{m}_z"]), rtol=0, atol=1e-10)
        first = capsys.readouterr().out.strip().splitlines()[0]
        assert first == str(d[f"persp_{m}_stdout"]).strip().splitlines()[0]


def _reproj_rmse(x_list, X, K, R, t):
    from lib.camera import calc_projected_points
    r = np.stack(calc_projected_points(X, K, R, t)) - np.stack(x_list)
    return np.sqrt((r**2).sum(axis=2).mean())


def test_euclidean_upgrade_and_reconstruction_against_reference(golden):
    from lib import perspective_camera_calibration as P

    d = golden("calibration")
    for m in ("primary", "dual"):
        M, S = d[f"persp_{m}_M"], d[f"persp_{m}_S"]
        Pm = M.reshape(-1, 3, 4)
        # one step of the Omega <-> K loop
        K0 = np.tile(np.eye(3), (10, 1, 1))
        Q0 = np.linalg.inv(K0) @ Pm
        Om, sig, w = P._calc_omega(Q0)
        np.testing.assert_allclose(Om, d[f"persp_{m}_Omega1"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(sig, d[f"persp_{m}_sigma1"], rtol=0, atol=1e-12)
        K1, J1 = P._update_K(K0.copy(), Om, Q0)
        np.testing.assert_allclose(K1, d[f"persp_{m}_K1"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(J1, d[f"persp_{m}_J1"], rtol=1e-9)
    # dual depths (what the driver script uses): the whole chain to 1e-8
    m = "dual"
    M, S = d[f"persp_{m}_M"], d[f"persp_{m}_S"]
    Pm = M.reshape(-1, 3, 4)
    H, K = P._euclidean_upgrading(Pm, 1.0)
    np.testing.assert_allclose(K, d[f"persp_{m}_Kup"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(H, d[f"persp_{m}_H"], rtol=0, atol=1e-8 * np.abs(d[f"persp_{m}_H"]).max())
    X, R, t = P._reconstruct_3d(Pm, S, K, H)
    np.testing.assert_allclose(X, d[f"persp_{m}_X3"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(R, d[f"persp_{m}_R3"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(t, d[f"persp_{m}_t3"], rtol=0, atol=1e-8)
    Xw, Rw, tw = P.correct_world_coordinates(X, R, t, method="predict")
    np.testing.assert_allclose(Xw, d[f"persp_{m}_X"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(Rw, d[f"persp_{m}_R"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(tw, d[f"persp_{m}_t"], rtol=0, atol=1e-8)
    Xc, Rc, tc = P.correct_world_coordinates(d["persp_dual_X3"], d["persp_dual_R3"], d["persp_dual_t3"])
    np.testing.assert_allclose(Xc, d["first_cam_X"], atol=1e-12)
    np.testing.assert_allclose(Rc, d["first_cam_R"], atol=1e-12)
    np.testing.assert_allclose(tc, d["first_cam_t"], atol=1e-12)
    # primary depths: the upgrade loop is chaotic on this scene (J_med 1.9e9 after one step, a
    # 1e-15 summation-order difference is O(1) two steps later), so the end result is judged by
    # what it is for: a metric reconstruction whose reprojection is as good as the reference's
    m = "primary"
    M, S = d[f"persp_{m}_M"], d[f"persp_{m}_S"]
    Pm = M.reshape(-1, 3, 4)
    H, K = P._euclidean_upgrading(Pm, 1.0)
    X, R, t = P._reconstruct_3d(Pm, S, K, H)
    mine = _reproj_rmse(d["persp_x"], X, K, R, t)
    ref = _reproj_rmse(d["persp_x"], d["persp_primary_X3"], d["persp_primary_Kup"], d["persp_primary_R3"], d["persp_primary_t3"])
    assert mine < 1.5 * ref + 1e-3, (mine, ref)


def test_affine_default_scene_which_minimum_bundle_adjustment_reaches(golden, capsys):
    """SURVEY 7 hard part 4, decided: the reference's default affine run ends at E = 0.21790752620130377 after
    100 outer iterations / 197 solves.  The parity of the three singular-vector signs decides the start: the
    repo's rule (largest entry of each column of U positive) applied to the reference's own SVD factors flips
    TWO of them -- LAPACK's parity, a rotated start, the same minimum; one more flip (the mirror parity) is a
    different start that converges in 36 solves to E = 0.0993303282.  (Oracle engine; the GPU driver test
    asserts the first number on the device.)"""
    from lib import affine_camera_calibration as A
    from lib.bundle_adjustment import dense_to_observations, lm_loop, to_gauge_frame
    from oracle import ba_oracle as O

    c, a = golden("calibration"), golden("affine_default")
    np.testing.assert_array_equal(c["aff_x"], a["x_noisy"])  # the calibration vectors ARE the default scene
    U = c["aff_U3"]
    S3 = np.diag(c["aff_sigma"][:3]) @ c["aff_Vt3"]
    own = np.sign(U[np.abs(U).argmax(axis=0), np.arange(3)])
    assert own.prod() == 1.0  # same parity as LAPACK's signs on this scene

    def run(sg):
        X0, R0 = A._affine_core("paraperspective", U * sg, S3 * sg[:, None], c["aff_t"], np.ones(12))
        t0, K0 = -3 * R0[:, :, 2], np.tile(np.eye(3), (12, 1, 1))  # affine_reconstruction.py:44-45
        pt_ptr, cam, xy = dense_to_observations(a["x"], None)
        g = O.OracleEngine(200, 12, pt_ptr, cam, xy, 1.0, "x-up_z-forward")
        Xn, Rn, tn = to_gauge_frame(X0, R0, t0, "x-up_z-forward")
        g.set_params(Xn, K0[:, 0, 0].copy(), K0[:, :2, 2].copy(), tn, Rn)
        E = lm_loop(g, 2.0, 1e-8, 100, verbose=False)
        return E, g.n_solves

    E, solves = run(own)
    assert solves == int(a["n_solves"]) == 197 and abs(E - 0.21790752620130377) < 1e-9
    E, solves = run(own * np.array([1.0, 1.0, -1.0]))
    assert solves == 36 and abs(E - 0.09933032816) < 1e-9
