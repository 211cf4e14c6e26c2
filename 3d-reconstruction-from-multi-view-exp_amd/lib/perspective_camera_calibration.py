"""Perspective self-calibration with the reference's call surface
(lib/perspective_camera_calibration.py:453-540): projective reconstruction by the primary or
the dual projective-depth iteration, factorization, Euclidean upgrade through the absolute dual
quadric, metric reconstruction, world-axis normalisation.

Caller of the factorization SVD.  Differences from the reference:
  * every SVD of the 3m x N measurement matrix is the thin GPU one (`mvsvd_factorize`);
  * the projective-depth iterations run ON THE DEVICE (`mvsvd_depth_step`: nothing but the reprojection error
    crosses PCIe per iteration), with the per-point (primary, :99-113) and per-image (dual, :188-205)
    eigenproblems in their low-rank form: the reference's m x m matrix is C C^T with C (m x 4), and its N x N
    matrix is the Hadamard product of a rank-4 and a rank-3 Gram matrix, i.e. Z Z^T with
    Z = row-wise Kronecker product (N x 12).  The dominant eigenvector comes from the 4 x 4
    (12 x 12) companion problem, so the dual method no longer needs O(N^2) memory;
  * the quartic tensors of the absolute-quadric fit are einsum'd instead of 4^4 Python loops.
The tiny dense eigenproblems (10 x 10, 4 x 4) use the same NumPy routines as the reference.

Sign note: in the dual method the sign of an image's depth vector is the sign LAPACK happens to
give an eigenvector, so the reference can return whole images with negative depths; here they
are always positive.  Both are the same projective reconstruction (P_k ~ -P_k) and give the
same metric result.
"""
from __future__ import annotations

import numpy as np
import numpy.typing as npt

from .factorization import factorization_method
from .utils import unit_vec


class _DeviceDepthLoop:
    """One projective-depth loop on the device (`mvsvd_depth_*`, csrc/mvsvd.hip): the homogeneous observations x
    (N, m, 3) are uploaded ONCE, the depths z live on the device, and an iteration -- re-weight and normalise, rank-4
    factorisation, per-point (primary) or per-image (dual) eigenproblem, depth update, reprojection error -- moves
    8 bytes (the error) across PCIe.  The depths come back once, at the end."""

    def __init__(self, x: npt.NDArray):
        from ._mvba import SvdWorkspace

        n, m = x.shape[:2]
        self._ws = SvdWorkspace(n, 3 * m, np.float64)
        self._ws.load_base(np.ascontiguousarray(x.reshape(n, 3 * m), dtype=np.float64))
        self._ws.depth_begin(3)

    @classmethod
    def from_images(cls, x_list: list[npt.NDArray], f0: float):
        """The same loop from the images' own (N, 2) arrays: x = (x / f0, y / f0, 1) is assembled on the device
        (`mvsvd_load_base_images`) -- no (N, m, 3) array on the host, two thirds of the bytes over PCIe."""
        from ._mvba import SvdWorkspace

        self = cls.__new__(cls)
        self._ws = SvdWorkspace(len(x_list[0]), 3 * len(x_list), np.float64)
        try:
            self._ws.load_base_images(x_list, f0)
            self._ws.depth_begin(3)
        except Exception:
            self.close()
            raise
        return self

    def step(self, method: int, f0: float) -> float:
        return self._ws.depth_step(method, f0)[0]

    def depths(self) -> npt.NDArray:
        return self._ws.depth_read()

    def factorize(self, n_rank: int):
        """M, S of W = x o z (:531-533) from the observations and the depths already in the workspace: no W on the host, no upload."""
        M, _sigma, S, _tm = self._ws.run_scaled(None, 3, 0, n_rank)
        return M, S

    def close(self):
        if self._ws is not None:
            self._ws.close()
            self._ws = None


def _create_data_matrix(x_list: list[npt.NDArray], f0: float) -> npt.NDArray:
    """(N, m, 3) homogeneous observations (x/f0, y/f0, 1)  (:34-40)."""
    x = np.empty((len(x_list[0]), len(x_list), 3))  # filled in place, contiguous: the stack + column_stack + transposed view of
    for k, xi in enumerate(x_list):                 # the reference's form cost 0.08 s at 1 M points x 12 images, and a copy later
        x[:, k, :2] = np.asarray(xi) / f0
    x[:, :, 2] = 1.0
    return x


def _depth_iterations(x, f0, tolerance, max_iter, method, loop, close=True):
    """The reference's loop (:77-142 / :162-233) around one device iteration: print, stop rule, final depths
    (`close=False`: none, the loop stays open -- the caller factorises on the device and closes it)."""
    loop = loop or _DeviceDepthLoop(x)
    count = 0
    try:
        while True:
            E = loop.step(method, f0)
            count += 1
            print(f"Iteration {count}: reprojection_error = {E:.8}")
            if E < tolerance or count >= max_iter:
                break
        if count >= max_iter:
            print("Did not converge because the maximum number of iterations was reached.")
        return loop.depths() if close else None
    finally:
        if close:
            loop.close()


def _compute_projective_depth_primary_method(x, f0, tolerance, max_iter: int = 200, loop=None):
    """Primary method (:61-144): alternate a rank-4 fit of the column-normalised measurement matrix with per-point
    depth updates (the dominant eigenvector of the m x m matrix of :99-107, from its 4 x 4 companion).  `loop`: an
    object with the protocol of `_DeviceDepthLoop` (the tests inject the CPU oracle's)."""
    return _depth_iterations(x, f0, tolerance, max_iter, 1, loop)


def _compute_projective_depth_dual_method(x, f0, tolerance, max_iter: int = 50, loop=None):
    """Dual method (:147-235): rows (images) normalised by their squared norm, per-image depth updates (the dominant
    eigenvector of the N x N matrix of :188-205, from its 12 x 12 companion: O(N) memory instead of O(N^2))."""
    return _depth_iterations(x, f0, tolerance, max_iter, 2, loop)


# ---------------------------------------------------------------- Euclidean upgrade (:238-411)
_PAIRS4 = [(i, j) for i in range(4) for j in range(i + 1, 4)]
_IDX10 = [(i, i) for i in range(4)] + _PAIRS4
_W10 = np.array([1.0] * 4 + [np.sqrt(2.0)] * 6)


def _calc_omega(Q):
    """Absolute dual quadric from Q_k = K_k^-1 P_k: least squares over the conditions
    (q1,Oq1) = (q2,Oq2), (q1,Oq2) = (q2,Oq3) = (q3,Oq1) = 0, then the rank-3 projection (:238-349)."""
    q1, q2, q3 = Q[:, 0], Q[:, 1], Q[:, 2]
    outer = lambda a, b: np.einsum("ni,nj->nij", a, b)  # noqa: E731
    sym = lambda a, b: outer(a, b) + outer(b, a)        # noqa: E731
    d = outer(q1, q1) - outer(q2, q2)
    A_cal = np.einsum("nij,nkl->ijkl", d, d)
    for a, b in ((q1, q2), (q2, q3), (q3, q1)):
        s = sym(a, b)
        A_cal += 0.25 * np.einsum("nij,nkl->ijkl", s, s)
    A = np.empty((10, 10))
    for p_, (i, j) in enumerate(_IDX10):
        for q_, (k, l) in enumerate(_IDX10):
            A[p_, q_] = _W10[p_] * _W10[q_] * A_cal[i, j, k, l]
    lam, vec = np.linalg.eig(A)
    omega = vec[:, np.argmin(lam)]
    Omega = np.zeros((4, 4))
    for val, (i, j), w in zip(omega, _IDX10, _W10):
        Omega[i, j] = Omega[j, i] = val / w
    lam, vec = np.linalg.eig(Omega)
    order = np.argsort(lam)[::-1]
    sigma, w = lam[order], vec[:, order].T
    if sigma[2] > 0:
        Omega = (sigma[:3, None] * w[:3]).T @ w[:3]
    elif sigma[1] < 0:
        Omega = -((sigma[2:, None] * w[2:]).T @ w[2:])
    else:
        raise ValueError()
    return Omega, sigma, w


def _update_K(K, Omega, Q):
    """One intrinsic-parameter correction from C_k = Q_k Omega Q_k^T  (:352-394)."""
    C = Q @ Omega @ Q.transpose(0, 2, 1)
    c33 = C[:, 2, 2]
    F = (C[:, 0, 0] + C[:, 1, 1]) / c33 - (C[:, 0, 2] / c33) ** 2 - (C[:, 1, 2] / c33) ** 2
    J = np.full(F.shape, np.inf)
    ok = (c33 > 0) & (F > 0)
    if ok.any():
        du, dv = C[:, 0, 2] / c33, C[:, 1, 2] / c33
        with np.errstate(invalid="ignore"):
            df = np.sqrt(0.5 * ((C[:, 0, 0] + C[:, 1, 1]) / c33 - du**2 - dv**2))
        dK = np.zeros((len(F), 3, 3))
        dK[:, 0, 0] = dK[:, 1, 1] = df
        dK[:, 0, 2], dK[:, 1, 2], dK[:, 2, 2] = du, dv, 1.0
        K[ok] = (K @ dK)[ok]
        with np.errstate(invalid="ignore"):
            K[ok] = (np.sqrt(c33)[:, None, None] * K)[ok]
        J[ok] = ((C[:, 0, 0] / c33 - 1) ** 2 + (C[:, 1, 1] / c33 - 1) ** 2
                 + 2 * (C[:, 0, 1] ** 2 + C[:, 1, 2] ** 2 + C[:, 2, 0] ** 2) / c33**2)[ok]
    return K, J


def _euclidean_upgrading(P, f0):
    """Iterate Omega <-> K until the median residual stops improving  (:383-411)."""
    K = np.tile(np.eye(3) * f0, (P.shape[0], 1, 1))
    J_prev = np.inf
    while True:
        Q = np.linalg.inv(K) @ P
        Omega, lam, w = _calc_omega(Q)
        if lam[2] > 0:
            H = (np.append(np.sqrt(lam[:3]), 1.0)[:, None] * w).T
        elif lam[1] < 0:
            H = (np.append(1.0, np.sqrt(-lam[1:]))[:, None] * w)[::-1].T
        else:
            raise ValueError()
        K, J = _update_K(K, Omega, Q)
        J_med = np.median(J)
        if J_med < 1e-8 or J_med >= J_prev:
            break
        J_prev = J_med
    return H, K


def _reconstruct_3d(P, S, K, H):
    """Metric points and camera poses from the upgraded projective reconstruction (:414-450)."""
    Xh = (np.linalg.inv(H) @ S).T
    X = Xh[:, :3] / Xh[:, 3:]
    Ab = np.linalg.inv(K) @ (P @ H)
    s = np.cbrt(np.linalg.det(Ab[:, :, :3]))
    Ab = Ab / s[:, None, None]
    U, _, Vt = np.linalg.svd(Ab[:, :, :3])
    R = (U @ Vt).transpose(0, 2, 1)
    t = -(R @ Ab[:, :, 3:])[..., 0]
    if np.sign(((X - t[0]) @ R[0])[:, 2]).sum() <= 0:  # points must lie in front of camera 0
        X, t = -X, -t
    return X, R, t


def _predict_world_axis(X, R, t):
    """World x = mean camera x-axis, origin = mean camera centre  (:453-476)."""
    ex = unit_vec(R[:, :, 0].mean(axis=0))
    ey = unit_vec(np.cross(np.array([0.0, 0.0, 1.0]), ex))
    ez = unit_vec(np.cross(ex, ey))
    Rw = np.column_stack([ex, ey, ez])
    c = t.mean(axis=0)
    return (X - c) @ Rw, Rw.T @ R, (t - c) @ Rw


def _normalize_world_axis_with_first_camera(X, R, t):
    """Camera 0 at the origin, baseline y-component 1  (:479-497)."""
    s = np.array([0, 1, 0]) @ R[0].T @ (t[1] - t[0])[:, None]
    return ((X - t[0]) @ R[0]) / s, R[0].T @ R, ((t - t[0]) @ R[0]) / s


def correct_world_coordinates(X, R, t, method: str = "first_camera"):
    if method == "first_camera":
        return _normalize_world_axis_with_first_camera(X, R, t)
    if method == "predict":
        return _predict_world_axis(X, R, t)
    raise ValueError()


def perspective_self_calibration(x_list: list[npt.NDArray], f0=1.0, tol=0.01, method: str = "primary"):
    """-> (X (N,3), R (m,3,3), t (m,3), K (m,3,3))   (:513-540)"""
    if method not in ("primary", "dual"):
        raise ValueError()
    # the reference forms x (N, m, 3) and W = x * z on the host and factorises W (:34-40, :531-533); here the images' arrays go to
    # the device as they are, x is assembled there, and the depths are still there, next to x, when W is formed and factorised
    # (0.11 s of strided host writes + 0.13 s of host multiply and upload at 1 M points x 12 images)
    from_images = getattr(_DeviceDepthLoop, "from_images", None)
    x = None if from_images else _create_data_matrix(x_list, f0)  # (a loop class without the device workspace: the tests' oracle loop)
    loop = from_images(x_list, f0) if from_images else _DeviceDepthLoop(x)
    try:
        _depth_iterations(x, f0, tol, 200 if method == "primary" else 50, 1 if method == "primary" else 2, loop, close=False)
        if hasattr(loop, "factorize"):
            M, S = loop.factorize(4)
        else:  # (a loop object without the device workspace -- the tests inject the CPU oracle's: the reference's own form)
            W = x * loop.depths()[..., None]
            M, S = factorization_method(W.reshape(W.shape[0], -1).T)
    finally:
        loop.close()
    P = M.reshape(-1, 3, 4)
    H, K = _euclidean_upgrading(P, f0)
    X, R, t = _reconstruct_3d(P, S, K, H)
    X, R, t = correct_world_coordinates(X, R, t, method="predict")
    return X, R, t, K
