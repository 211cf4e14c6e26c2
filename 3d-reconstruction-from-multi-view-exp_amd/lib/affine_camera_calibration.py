"""Affine self-calibration (orthographic / symmetric affine / paraperspective) with the
reference's call surface (lib/affine_camera_calibration.py:7-221).

Consumers of the factorization SVD.  What differs from the reference:
  * the SVD of the centred measurement matrix runs on the GPU (`mvsvd_factorize` with
    `center=1`: the centring of :224-240 is fused into the Gram pass; thin, no N x N factor);
  * the metric-constraint tensors B_cal are built with einsum over all images at once instead
    of 3^4 Python loops per image (:23-38, :75-115, :156-202);
everything after the SVD is a literal restatement, quirks included (see `_rotations`).

Singular-vector signs: the PARITY of the three triplet signs mirrors the reconstruction (SURVEY §7
hard part 4) -- an even number of flips is a rotation of the scene, which bundle adjustment does not
see; an odd number is its mirror image, a different start and a different minimum.  LAPACK's choice
cannot be restated; ours is "largest-magnitude entry of each column of U positive".  On the
reference's default affine scene (affine_reconstruction.py:14-58, seed 123) that rule has LAPACK's
parity: bundle adjustment from it ends where the reference's run ends (100 outer iterations, 197
solves, E = 0.2179075262); the mirror parity (`_svd_on_gpu(..., mirror=True)`) ends after 36 solves
at E = 0.0993303282 (tests/test_callers_cpu.py pins both on the oracle, tests/test_gpu_callers.py
the driver on the GPU).  `_affine_core` takes the factors explicitly so tests can pin everything
downstream of the SVD against the reference's own factors.
"""
from __future__ import annotations

import numpy as np
import numpy.typing as npt

_SQ2 = np.sqrt(2.0)
# symmetric 3x3 <-> 6-vector (T11, T22, T33, sqrt2 T23, sqrt2 T31, sqrt2 T12)   (:243-271)
_PAIRS = ((0, 0), (1, 1), (2, 2), (1, 2), (2, 0), (0, 1))
_WEIGHT = np.array([1.0, 1.0, 1.0, _SQ2, _SQ2, _SQ2])


def _observation_matrix_host(data_list):
    """(W centred (2m, N), t (m, 2)) on the host -- used only by the CPU-testable core path."""
    n = {len(x) for x in data_list}
    if len(n) != 1:
        raise ValueError()
    W = np.hstack(data_list).T.astype(np.float64)
    t = W.mean(axis=1, keepdims=True)
    return W - t, t.reshape(-1, 2)


def _quartic(a1, a2, a3, a4, w=None):
    """sum_n w_n a1[n,i] a2[n,j] a3[n,k] a4[n,l]"""
    if w is None:
        return np.einsum("ni,nj,nk,nl->ijkl", a1, a2, a3, a4)
    return np.einsum("n,ni,nj,nk,nl->ijkl", w, a1, a2, a3, a4)


def _constraint_tensor(model, a, b, t, f):
    """B_cal (3,3,3,3): a = first rows, b = second rows of the per-image 2x3 blocks of U[:, :3]."""
    q = _quartic
    if model == "orthographic":          # (a,Ta) = (b,Tb) = 1, (a,Tb) = 0            (:23-38)
        s = np.einsum("ni,nj->nij", a, b) + np.einsum("ni,nj->nij", b, a)
        return q(a, a, a, a) + q(b, b, b, b) + 0.25 * np.einsum("nij,nkl->ijkl", s, s)
    if model == "symmetric_affine":      # (:75-115)
        p = t[:, 0] * t[:, 1]
        c = t[:, 0] ** 2 - t[:, 1] ** 2
        return (q(a, a, a, a, p**2) + q(b, b, b, b, p**2) - q(a, a, b, b, p**2) - q(b, b, a, a, p**2)
                + 0.25 * (q(a, b, a, b, c**2) + q(b, a, a, b, c**2) + q(a, b, b, a, c**2) + q(b, a, b, a, c**2))
                - 0.5 * (q(a, a, a, b, p * c) + q(a, a, b, a, p * c) + q(a, b, a, a, p * c) + q(b, a, a, a, p * c)
                         - q(a, b, b, b, p * c) - q(b, a, b, b, p * c) - q(b, b, a, b, p * c) - q(b, b, b, a, p * c)))
    if model == "paraperspective":       # (:156-202)
        al = 1.0 / (1.0 + t[:, 0] ** 2 / f**2)
        be = 1.0 / (1.0 + t[:, 1] ** 2 / f**2)
        ga = t[:, 0] * t[:, 1] / f**2
        return (q(a, a, a, a, (ga**2 + 1) * al**2) + q(b, b, b, b, (ga**2 + 1) * be**2)
                + q(a, b, a, b) + q(a, b, b, a) + q(b, a, a, b) + q(b, a, b, a)
                - (q(a, a, a, b, al * ga) + q(a, a, b, a, al * ga) + q(a, b, a, a, al * ga) + q(b, a, a, a, al * ga))
                - (q(b, b, a, b, be * ga) + q(b, b, b, a, be * ga) + q(a, b, b, b, be * ga) + q(b, a, b, b, be * ga))
                + (q(a, a, b, b, (ga**2 - 1) * al * be) + q(b, b, a, a, (ga**2 - 1) * al * be)))
    raise ValueError()


def _pack_B(B_cal):
    """6x6 matrix of the quadratic form in tau (:243-258)."""
    B = np.empty((6, 6))
    for p_, (i, j) in enumerate(_PAIRS):
        for q_, (k, l) in enumerate(_PAIRS):
            B[p_, q_] = _WEIGHT[p_] * _WEIGHT[q_] * B_cal[i, j, k, l]
    return B


def _metric_from_tau(tau):
    """(:261-271)"""
    return np.array([[tau[0], tau[5] / _SQ2, tau[4] / _SQ2],
                     [tau[5] / _SQ2, tau[1], tau[3] / _SQ2],
                     [tau[4] / _SQ2, tau[3] / _SQ2, tau[2]]])


def _rotations(M, U3, T, t):
    """Camera rotations from the motion matrix (:274-341), restated literally, including two
    quirks that change numbers: (1) the 3 equations for (1/zeta^2, beta^2) pair the rows
    (1,tx^2), (1,ty^2), (0,tx ty) with (u1Tu1, u1Tu2, u2Tu2) in THAT order (:281-293);
    (2) the r3 denominator uses g.g of the FIRST image for every image (:326)."""
    m = t.shape[0]
    P = np.ones((m, 3, 2))
    P[:, :2, 1] = t**2
    P[:, 2, 0] = 0.0
    P[:, 2, 1] = t[:, 0] * t[:, 1]
    U1, U2 = U3[::2], U3[1::2]
    Q = np.stack([np.einsum("ni,ij,nj->n", U1, T, U1), np.einsum("ni,ij,nj->n", U1, T, U2),
                  np.einsum("ni,ij,nj->n", U2, T, U2)], axis=1)
    sol = (np.linalg.pinv(P) @ Q[..., None])[..., 0]
    zeta2_inv, beta2 = sol[:, 0].copy(), sol[:, 1].copy()
    beta2[beta2 < 0.0] = 0.0
    centred = (np.abs(t) < 1e-8).all(axis=1)
    beta2[centred] = 0.0
    zeta2_inv[centred] = ((Q[:, 0] + Q[:, 2]) / 2)[centred]
    zeta2_inv[zeta2_inv <= 0.0] = 1e8
    zeta, beta = np.sqrt(1 / zeta2_inv), np.sqrt(beta2)
    g = zeta[:, None] * t
    M1, M2 = M[::2], M[1::2]
    r3 = (zeta[:, None] * np.cross(M1, M2) - beta[:, None] * (g[:, :1] * M1 + g[:, 1:] * M2)) \
        / (1 + beta[:, None] ** 2 * (g[0] @ g[0]))
    r1 = zeta[:, None] * M1 + (beta * g[:, 0])[:, None] * r3
    r2 = zeta[:, None] * M2 + (beta * g[:, 1])[:, None] * r3
    R = np.stack([r1, r2, r3], axis=2)  # columns r1, r2, r3
    U, _, Vt = np.linalg.svd(R)         # nearest exact rotation
    return U @ Vt


def _affine_core(model, U3, S3, t, f=None):
    """Everything after the SVD.  U3 = U[:, :3] (2m x 3), S3 = diag(sigma[:3]) Vt[:3] (3 x N),
    t = image centroids (m x 2).  Returns (X (N,3), R (m,3,3))."""
    a, b = U3[::2], U3[1::2]
    B = _pack_B(_constraint_tensor(model, a, b, t, f))
    if model == "orthographic":
        tau = np.linalg.solve(B, np.array([1.0, 1, 1, 0, 0, 0]))
    else:
        lam, vec = np.linalg.eig(B)
        tau = vec[:, np.argmin(lam)]
    T = _metric_from_tau(tau)
    if np.linalg.det(T) < 0:
        T = -T
    A = np.linalg.cholesky(T)
    M = U3 @ A
    S = np.linalg.inv(A) @ S3
    return S.T, _rotations(M, U3, T, t)


def _svd_on_gpu(data_list, mirror=False):
    """Centred thin SVD of the measurement matrix through libmvba (rank 3).  Signs: largest-magnitude
    entry of each column of U positive; `mirror=True` flips the third triplet (the other sign parity:
    the mirror-image reconstruction, see the module docstring)."""
    from ._mvba import svd_factorize_images

    n = {len(x) for x in data_list}
    if len(n) != 1:
        raise ValueError()
    # W^T = np.hstack(data_list) (N, 2m), exactly the array the reference transposes (:224-240), is put together on the device from
    # the images' own arrays: the hstack on the host was 0.3 s at 5 M points x 12 images, the upload + factorisation 0.015
    U3, _sigma, S3, mu, _tm = svd_factorize_images(data_list, 3, center=True)
    U3, S3 = U3.astype(np.float64), S3.astype(np.float64, copy=False)
    if mirror:
        U3[:, 2] *= -1.0
        S3[2] *= -1.0
    return U3, S3, mu.astype(np.float64).reshape(-1, 2)


def orthographic_self_calibration(data_list: list[npt.NDArray[np.floating]]):
    U3, S3, t = _svd_on_gpu(data_list)
    return _affine_core("orthographic", U3, S3, t)


def symmetric_affine_self_calibration(data_list: list[npt.NDArray[np.floating]]):
    U3, S3, t = _svd_on_gpu(data_list)
    return _affine_core("symmetric_affine", U3, S3, t)


def paraperspective_self_calibration(data_list: list[npt.NDArray[np.floating]], f: npt.NDArray[np.floating]):
    if len(data_list) != len(f):
        raise ValueError()
    U3, S3, t = _svd_on_gpu(data_list)
    return _affine_core("paraperspective", U3, S3, t, np.asarray(f, dtype=np.float64))
