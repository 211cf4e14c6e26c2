#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference (read-only).

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py``

Everything written is DATA (inputs and the reference's outputs on them) as
``.npz`` files next to this script; no reference source is copied.  The scene
generators below are this repo's own code; they only call reference functions
to obtain the expected outputs (SURVEY.md §8c lists the vectors).
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = os.environ.get("MVBA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)

from lib.bundle_adjustment import BundleAdjuster  # noqa: E402  (reference)
from lib.camera import Camera, calc_projected_points, get_camera_parames  # noqa: E402
from lib.factorization import factorization_method  # noqa: E402
from lib.utils import get_rotation_matrix, sample_hemisphere_points, set_points  # noqa: E402


def _quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def _count_inner_solves(ba, *args, **kw):
    """Run optimize() while counting np.linalg.solve calls (= inner LM solves)."""
    n = {"solve": 0}
    orig = np.linalg.solve

    def counting(a, b):
        n["solve"] += 1
        return orig(a, b)

    np.linalg.solve = counting
    try:
        out, txt = _quiet(ba.optimize, *args, **kw)
    finally:
        np.linalg.solve = orig
    return out, txt, n["solve"]


def _save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


# --------------------------------------------------------------------------
# (1) Euclidean default scene  (euclidiean_reconstruction.py:13-57 minus plots)
# --------------------------------------------------------------------------
def euclid_default():
    from lib.perspective_camera_calibration import perspective_self_calibration

    np.random.seed(123)
    n_images = 10
    camera_pos = sample_hemisphere_points(n_images, 5)
    targets = np.random.normal(0, 0.5, (n_images, 3))
    cameras = [Camera.create(p, t, f=1.0, f0=1.0) for p, t in zip(camera_pos, targets)]
    K_gt, R_gt, t_gt = get_camera_parames(cameras)
    X_gt = set_points()
    x_list = calc_projected_points(X_gt, K_gt, R_gt, t_gt)
    x_clean = np.stack(x_list).copy()
    for x in x_list:
        x += 0.005 * np.random.randn(*x.shape)
    (X_, R_, t_, K_), calib_txt = _quiet(
        perspective_self_calibration, x_list, 1.0, tol=1e-2, method="dual"
    )
    x = np.stack(x_list).transpose(1, 0, 2)
    ba = BundleAdjuster(x, X_, K_, R_, t_, axis="x-up_z-forward")
    (Xo, Ko, Ro, to), txt, solves = _count_inner_solves(ba, 2.0, 1e-8, max_iter=100, is_debug=True)
    log = ba.get_log()
    # W as handed to factorization_method at perspective_camera_calibration.py:533
    from lib.perspective_camera_calibration import (
        _compute_projective_depth_dual_method,
        _create_data_matrix,
    )

    xm = _create_data_matrix(x_list, 1.0)
    z, _ = _quiet(_compute_projective_depth_dual_method, xm, 1.0, 1e-2)
    Wm = xm * z[..., np.newaxis]
    W = Wm.reshape(Wm.shape[0], -1).T
    M, S = factorization_method(W)
    _save(
        "euclid_default",
        camera_pos=camera_pos, targets=targets, K_gt=K_gt, R_gt=R_gt, t_gt=t_gt, X_gt=X_gt,
        x_clean=x_clean, x=x, init_X=X_, init_K=K_, init_R=R_, init_t=t_,
        out_X=Xo, out_K=Ko, out_R=Ro, out_t=to,
        E_log=np.array([d["reprojection_error"] for d in log]),
        log0_points=log[0]["points"], log0_basis=log[0]["basis"], log0_pos=log[0]["pos"],
        logN_points=log[-1]["points"], logN_basis=log[-1]["basis"], logN_pos=log[-1]["pos"],
        n_outer=np.int64(len(log) - 1), n_solves=np.int64(solves),
        stdout=np.array(txt),
        fact_W=W, fact_M=M, fact_S=S, fact_sigma=np.linalg.svd(W, compute_uv=False),
    )
    return x_list


# --------------------------------------------------------------------------
# (2) Affine default scene  (affine_reconstruction.py:14-58 minus plots)
# --------------------------------------------------------------------------
def affine_default():
    from lib.affine_camera_calibration import (
        orthographic_self_calibration,
        paraperspective_self_calibration,
        symmetric_affine_self_calibration,
    )

    np.random.seed(123)
    f, n_images = 1.0, 12
    camera_pos = sample_hemisphere_points(n_images, 5)
    targets = np.random.normal(0, 0.5, (n_images, 3))
    cameras = [Camera.create(p, t, f) for p, t in zip(camera_pos, targets)]
    K_gt, R_gt, t_gt = get_camera_parames(cameras)
    X_gt = set_points()
    x_list = calc_projected_points(X_gt, K_gt, R_gt, t_gt)
    for x in x_list:
        x += 0.005 * np.random.randn(*x.shape)
    x_noisy = np.stack(x_list).copy()  # (m, N, 2) BEFORE the in-place centring below
    # NB: _get_observation_matrix centres W in place on an hstack copy -> x_list untouched
    X_, R_ = paraperspective_self_calibration([a.copy() for a in x_list], f * np.ones(n_images))
    Xo_, Ro_ = orthographic_self_calibration([a.copy() for a in x_list])
    Xs_, Rs_ = symmetric_affine_self_calibration([a.copy() for a in x_list])
    t_ = -3 * R_[:, :, 2]
    K_ = np.broadcast_to(np.eye(3), R_.shape)
    x = np.stack(x_list).transpose(1, 0, 2)
    ba = BundleAdjuster(x, X_, K_, R_, t_, axis="x-up_z-forward")
    (Xo, Ko, Ro, to), txt, solves = _count_inner_solves(ba, 2.0, 1e-8, max_iter=100, is_debug=True)
    log = ba.get_log()
    _save(
        "affine_default",
        x_noisy=x_noisy, x=x, init_X=X_, init_K=np.array(K_), init_R=R_, init_t=t_,
        out_X=Xo, out_K=Ko, out_R=Ro, out_t=to,
        E_log=np.array([d["reprojection_error"] for d in log]),
        n_outer=np.int64(len(log) - 1), n_solves=np.int64(solves),
        ortho_X=Xo_, ortho_R=Ro_, symaff_X=Xs_, symaff_R=Rs_,
        K_gt=K_gt, R_gt=R_gt, t_gt=t_gt,
    )


# --------------------------------------------------------------------------
# Own synthetic scene (NOT reference code): random cameras on a hemisphere
# looking at the origin, random points, optional partial visibility.
# --------------------------------------------------------------------------
def _lookat(pos, target):
    z = (target - pos) / np.linalg.norm(target - pos)
    up = np.array([1.0, 0.0, 0.0])
    y = np.cross(z, up)
    y /= np.linalg.norm(y)
    xax = np.cross(y, z)
    xax /= np.linalg.norm(xax)
    return np.stack([xax, y, z], axis=1)


def small_scene(rng, n_pts, n_img, vis_p, pp_sigma, noise, perturb):
    th = rng.uniform(0.1, np.pi / 2, n_img)
    ph = rng.uniform(0, 2 * np.pi, n_img)
    pos = 5.0 * np.stack([np.cos(th), np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph)], 1)
    tgt = rng.normal(0, 0.3, (n_img, 3))
    R = np.stack([_lookat(p, t) for p, t in zip(pos, tgt)])
    t = pos
    f = 1.0 + rng.normal(0, 0.05, n_img)
    u = rng.normal(0, pp_sigma, (n_img, 2))
    K = np.zeros((n_img, 3, 3))
    K[:, 0, 0] = f
    K[:, 1, 1] = f
    K[:, :2, 2] = u
    K[:, 2, 2] = 1.0
    X = rng.uniform(-1, 1, (n_pts, 3))
    d = X[:, None, :] - t[None]
    c = np.einsum("kji,akj->aki", R, d)
    x = np.stack(
        [(f * c[..., 0] + u[:, 0] * c[..., 2]) / c[..., 2],
         (f * c[..., 1] + u[:, 1] * c[..., 2]) / c[..., 2]], axis=2)
    x = x + rng.normal(0, noise, x.shape)
    vis = rng.uniform(size=(n_pts, n_img)) < vis_p
    for a in range(n_pts):  # every point at least 3 views
        while vis[a].sum() < 3:
            vis[a, rng.integers(n_img)] = True
    X0 = X + rng.normal(0, perturb, X.shape)
    t0 = t + rng.normal(0, perturb, t.shape)
    R0 = np.stack([get_rotation_matrix(w) @ Rk for w, Rk in zip(rng.normal(0, perturb, (n_img, 3)), R)])
    K0 = K.copy()
    K0[:, 0, 0] += rng.normal(0, perturb, n_img)
    K0[:, 1, 1] = K0[:, 0, 0]
    return x, vis, X0, K0, R0, t0


def linearization_dump(name, axis, seed):
    """(3)/(5): every intermediate of one LM linearisation + one trial at c=1e-4."""
    rng = np.random.default_rng(seed)
    x, vis, X0, K0, R0, t0 = small_scene(rng, 60, 7, 0.6, 0.05, 0.002, 0.02)
    ba = BundleAdjuster(x, X0, K0, R0, t0, f0=1.0, visibility_index=vis, axis=axis)
    nX, nR, nt = ba._X.copy(), ba._R.copy(), ba._t.copy()
    K = ba._get_K(ba._f, ba._u)
    P, p, q, r = ba._calc_pqr(ba._X, K, ba._R, ba._t)
    E0 = ba._calc_reprojection_error(p, q, r)
    dpdX, dqdX, drdX = ba._calc_X_diff_pqr(P)
    dpc, dqc, drc = ba._calc_camera_params_diff_pqr(p, q, r)
    d_P = ba._calc_d_P(p, q, r, dpdX, dqdX, drdX)
    d_F = ba._calc_d_F(p, q, r, dpc, dqc, drc)
    matE = ba._calc_matE(p, q, r, dpdX, dqdX, drdX)
    matF = ba._calc_matF(p, q, r, dpdX, dqdX, drdX, dpc, dqc, drc)
    matG = ba._calc_matG(p, q, r, dpc, dqc, drc)
    c = 1e-4
    n, m = x.shape[:2]
    matEc = matE.copy()
    i3 = np.arange(3)
    matEc[:, i3, i3] *= 1 + c
    matGc = matG.copy()
    iD = np.arange(9 * m - 7)
    matGc[iD, iD] *= 1 + c
    Einv = np.linalg.inv(matEc)
    FtEinv = matF.transpose(0, 2, 1) @ Einv
    A = matGc - (FtEinv @ matF).sum(axis=0)
    dXE = d_P.reshape(n, 3)[..., None]
    b = (FtEinv @ dXE).squeeze().sum(axis=0) - d_F
    dxi = np.linalg.solve(A, b)
    dX = -(Einv @ (matF @ dxi[:, None] + dXE)).squeeze()
    tX = ba._update_3d_points(dX)
    tf, tu, tt, tR = ba._update_camera_params(dxi)
    tK = ba._get_K(tf, tu)
    _, tp, tq, tr = ba._calc_pqr(tX, tK, tR, tt)
    E1 = ba._calc_reprojection_error(tp, tq, tr)
    # and a short full optimisation from the same start
    ba2 = BundleAdjuster(x, X0, K0, R0, t0, f0=1.0, visibility_index=vis, axis=axis)
    (Xo, Ko, Ro, to), txt, solves = _count_inner_solves(ba2, 10.0, 1e-8, max_iter=8, is_debug=True)
    _save(
        name,
        x=x, vis=vis, init_X=X0, init_K=K0, init_R=R0, init_t=t0, axis=np.array(axis),
        norm_X=nX, norm_R=nR, norm_t=nt, f=np.array(ba._f), u=np.array(ba._u),
        p=p, q=q, r=r, E0=np.float64(E0), d_P=d_P, d_F=d_F, matE=matE, matF=matF, matG=matG,
        c=np.float64(c), A=A, b=b, dxi=dxi, dX=dX, trial_X=tX, trial_f=tf, trial_u=tu,
        trial_t=tt, trial_R=tR, E1=np.float64(E1),
        out_X=Xo, out_K=Ko, out_R=Ro, out_t=to,
        E_log=np.array([d["reprojection_error"] for d in ba2.get_log()]),
        n_solves=np.int64(solves),
    )


def visibility_scene():
    """(4) 300x12, 30 % visible: 10 outer iterations with the scripts' schedule."""
    rng = np.random.default_rng(3)
    x, vis, X0, K0, R0, t0 = small_scene(rng, 300, 12, 0.3, 0.0, 0.001, 0.01)
    ba = BundleAdjuster(x, X0, K0, R0, t0, visibility_index=vis, axis="x-up_z-forward")
    (Xo, Ko, Ro, to), txt, solves = _count_inner_solves(ba, 2.0, -1.0, max_iter=10, is_debug=True)
    _save(
        "visibility_300x12",
        x=x, vis=vis, init_X=X0, init_K=K0, init_R=R0, init_t=t0,
        out_X=Xo, out_K=Ko, out_R=Ro, out_t=to,
        E_log=np.array([d["reprojection_error"] for d in ba.get_log()]),
        n_solves=np.int64(solves),
    )


def factorization_vectors():
    """(6) synthetic rank-3 + noise, 24 x 2000, fp64 and fp32."""
    rng = np.random.default_rng(0)
    A = rng.normal(size=(2000, 3))
    B = rng.normal(size=(3, 24))
    Wt = A @ B + 1e-3 * rng.normal(size=(2000, 24))  # (N, 2m) row-major, as the callers hold it
    out = {"Wt": Wt}
    for tag, dt in (("f64", np.float64), ("f32", np.float32)):
        W = Wt.astype(dt).T
        M, S = factorization_method(W, n_rank=3)
        out[f"M_{tag}"] = M
        out[f"S_{tag}"] = S
        out[f"sigma_{tag}"] = np.linalg.svd(W, compute_uv=False)
    M4, S4 = factorization_method(Wt.T)  # default n_rank=4
    out["M4_f64"] = M4
    out["S4_f64"] = S4
    _save("factorization_24x2000", **out)


def small_known_answers():
    """(7) Rodrigues + normalise/denormalise + helpers; (8) error behaviour."""
    rng = np.random.default_rng(7)
    om = np.concatenate(
        [np.zeros((1, 3)), np.array([[1e-12, 0, 0], [0, -2e-9, 1e-9], [np.pi, 0, 0], [0.3, -0.2, 0.9]]),
         rng.normal(0, 1.0, (11, 3))])
    rod = np.stack([get_rotation_matrix(w) for w in om])
    x, vis, X0, K0, R0, t0 = small_scene(rng, 9, 4, 1.0, 0.0, 0.0, 0.05)
    tr = {}
    for axis in ("x-right_z-forward", "x-up_z-forward"):
        nX, nR, nt = BundleAdjuster._transform_to_normalize_coodinates(X0, R0, t0, axis=axis)
        ba = BundleAdjuster(x, X0, K0, R0, t0, axis=axis)
        bX, bR, bt = BundleAdjuster._inverse_transform_to_global_coordinates(
            ba._init_camera0_params, nX, nR, nt)
        k = axis.split("_")[0].replace("-", "")
        tr.update({f"{k}_nX": nX, f"{k}_nR": nR, f"{k}_nt": nt, f"{k}_bX": bX, f"{k}_bR": bR, f"{k}_bt": bt,
                   f"{k}_c0c1": np.float64(ba._init_camera0_params["c0c1_len"])})
    errs = {}
    try:
        BundleAdjuster(x, X0, K0, R0, t0, axis="bogus")
        errs["bad_axis"] = "none"
    except Exception as e:  # noqa: BLE001
        errs["bad_axis"] = type(e).__name__
    vis0 = np.ones(x.shape[:2], bool)
    vis0[2] = False
    try:
        _quiet(BundleAdjuster(x, X0, K0, R0, t0, visibility_index=vis0).optimize, 10.0, 1e-8, 2)
        errs["zero_degree"] = "none"
    except Exception as e:  # noqa: BLE001
        errs["zero_degree"] = type(e).__name__ + ":" + str(e)
    # scene helpers (lib/utils.py, lib/camera.py)
    np.random.seed(5)
    hemi = sample_hemisphere_points(6, 5)
    pts = set_points()
    cam = Camera.create((1.0, 2.0, -3.0), (0.1, -0.2, 0.3), f=1.3, f0=1.0)
    Kc, Rc, tc = cam.get_parameters()
    proj = cam.project_points(pts[:15])
    ortho = cam.project_points(pts[:15], method="orthographic")
    _save(
        "known_answers",
        omega=om, rodrigues=rod, tr_X=X0, tr_R=R0, tr_t=t0, tr_K=K0, tr_x=x,
        err_bad_axis=np.array(errs["bad_axis"]), err_zero_degree=np.array(errs["zero_degree"]),
        hemi_seed5=hemi, set_points=pts, cam_K=Kc, cam_R=Rc, cam_t=tc, cam_P=cam.get_camera_matrix(),
        cam_proj=proj, cam_ortho=ortho, **tr,
    )


def calibration_vectors():
    """Callers either side of the hot path: affine self-calibration
    (three camera models) and perspective self-calibration (primary / dual projective depths),
    on the default scenes' observations."""
    from lib.affine_camera_calibration import (
        _get_observation_matrix,
        orthographic_self_calibration,
        paraperspective_self_calibration,
        symmetric_affine_self_calibration,
    )
    from lib import perspective_camera_calibration as pc

    d = np.load(os.path.join(HERE, "affine_default.npz"))
    x_list = [a.copy() for a in d["x_noisy"]]  # 12 x (200,2)
    W, t = _get_observation_matrix([a.copy() for a in x_list])
    U, Sig, Vt = np.linalg.svd(W)
    out = {"aff_x": np.stack(x_list), "aff_W": W, "aff_t": t, "aff_U3": U[:, :3], "aff_sigma": Sig,
           "aff_Vt3": Vt[:3]}
    Xo, Ro = orthographic_self_calibration([a.copy() for a in x_list])
    Xs, Rs = symmetric_affine_self_calibration([a.copy() for a in x_list])
    Xp, Rp = paraperspective_self_calibration([a.copy() for a in x_list], 1.0 * np.ones(12))
    Xp2, Rp2 = paraperspective_self_calibration([a.copy() for a in x_list], np.linspace(0.8, 1.3, 12))
    out.update(ortho_X=Xo, ortho_R=Ro, symaff_X=Xs, symaff_R=Rs, para_X=Xp, para_R=Rp, para2_X=Xp2, para2_R=Rp2,
               para2_f=np.linspace(0.8, 1.3, 12))

    e = np.load(os.path.join(HERE, "euclid_default.npz"))
    xe = [e["x"][:, k, :].copy() for k in range(e["x"].shape[1])]  # 10 x (200,2), noisy
    xm = pc._create_data_matrix(xe, 1.0)
    for method, fn in (("primary", pc._compute_projective_depth_primary_method),
                       ("dual", pc._compute_projective_depth_dual_method)):
        z, txt = _quiet(fn, xm, 1.0, 1e-2)
        (X, R, t_, K), txt2 = _quiet(pc.perspective_self_calibration, [a.copy() for a in xe], 1.0, tol=1e-2, method=method)
        out.update({f"persp_{method}_z": z, f"persp_{method}_stdout": np.array(txt),
                    f"persp_{method}_X": X, f"persp_{method}_R": R, f"persp_{method}_t": t_, f"persp_{method}_K": K})
        Wm = xm * z[..., np.newaxis]
        Wf = Wm.reshape(Wm.shape[0], -1).T
        M, S = factorization_method(Wf)
        P = M.reshape(-1, 3, 4)
        H, Kk = pc._euclidean_upgrading(P, 1.0)
        X3, R3, t3 = pc._reconstruct_3d(P, S, Kk, H)
        # one well-posed step of the upgrade loop (the loop itself is chaotic for the primary
        # depths on this scene: J_med = 1.9e9 after the first step)
        K0 = np.tile(np.eye(3), (P.shape[0], 1, 1))
        Q0 = np.linalg.inv(K0) @ P
        Om1, sig1, w1 = pc._calc_omega(Q0)
        K1, J1 = pc._update_K(K0.copy(), Om1, Q0)
        out.update({f"persp_{method}_Omega1": Om1, f"persp_{method}_sigma1": sig1, f"persp_{method}_K1": K1,
                    f"persp_{method}_J1": J1})
        out.update({f"persp_{method}_M": M, f"persp_{method}_S": S, f"persp_{method}_H": H, f"persp_{method}_Kup": Kk,
                    f"persp_{method}_X3": X3, f"persp_{method}_R3": R3, f"persp_{method}_t3": t3})
    out["persp_x"] = np.stack(xe)
    # three forced iterations of each projective-depth scheme (tolerance 0 never met)
    z3p, _ = _quiet(pc._compute_projective_depth_primary_method, xm, 1.0, 0.0, 3)
    z3d, _ = _quiet(pc._compute_projective_depth_dual_method, xm, 1.0, 0.0, 3)
    out.update(persp_primary_z3=z3p, persp_dual_z3=z3d)
    Xc, Rc, tc = pc.correct_world_coordinates(out["persp_dual_X3"], out["persp_dual_R3"], out["persp_dual_t3"],
                                              method="first_camera")
    out.update(first_cam_X=Xc, first_cam_R=Rc, first_cam_t=tc)
    _save("calibration", **out)


def trajectory_logs():
    """get_log() contents along the two default trajectories (ref :89-98, :175-183: copies of X, R, t in the
    NORMALISED frame per outer iteration): the reference's BundleAdjuster is run again on the BA inputs
    the fixtures above already hold, and a few log entries (first, two inside, last) are kept."""
    out = {}
    for name, picks in (("euclid_default", (0, 3, 17, -1)), ("affine_default", (0, 5, 50, -1)), ("visibility_300x12", (0, 4, -1))):
        d = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
        vis = d["vis"] if "vis" in d.files else None
        args = {"euclid_default": (2.0, 1e-8, 100), "affine_default": (2.0, 1e-8, 100), "visibility_300x12": (2.0, -1.0, 10)}[name]
        ba = BundleAdjuster(d["x"], d["init_X"], d["init_K"], d["init_R"], d["init_t"], visibility_index=vis, axis="x-up_z-forward")
        _quiet(ba.optimize, args[0], args[1], max_iter=args[2], is_debug=True)
        log = ba.get_log()
        assert np.allclose([e["reprojection_error"] for e in log], d["E_log"], rtol=1e-12, atol=0)
        out[name + "_len"] = np.int64(len(log))
        out[name + "_picks"] = np.array([p % len(log) for p in picks])
        for p in picks:
            i = p % len(log)
            out[f"{name}_{i}_points"], out[f"{name}_{i}_basis"], out[f"{name}_{i}_pos"] = log[i]["points"], log[i]["basis"], log[i]["pos"]
            out[f"{name}_{i}_E"] = np.float64(log[i]["reprojection_error"])
    _save("trajectory_logs", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. `make_golden.py trajectory_logs`: only the named vectors
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    euclid_default()
    affine_default()
    linearization_dump("linearize_60x7_xup", "x-up_z-forward", 11)
    linearization_dump("linearize_60x7_xright", "x-right_z-forward", 12)
    visibility_scene()
    factorization_vectors()
    small_known_answers()
    calibration_vectors()
    trajectory_logs()  # (reads the BA inputs out of the fixtures written above)
