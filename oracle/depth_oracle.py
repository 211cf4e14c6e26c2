"""CPU oracle for the projective-depth iterations  --  TEST INFRASTRUCTURE ONLY.

NumPy restatement of ONE iteration of the reference's two depth schemes,
``lib/perspective_camera_calibration.py:79-129`` (primary: per-point eigenproblem) and ``:166-224`` (dual: per-image
eigenproblem), in the low-rank form the device kernels (``mvsvd_depth_step``, csrc/mvsvd.hip) use:

  * primary (:93-121): the reference's m x m matrix A_a = C_a C_a^T with C_a[k, i] = (x_ak . u_ki) / |x_ak| (m x 4);
    its dominant eigenvector is C_a v / |C_a v| with v the dominant eigenvector of the 4 x 4 companion C_a^T C_a;
  * dual (:182-215): the reference's N x N matrix B_k = (V4 V4^T) o (x_k x_k^T) / (|x_k||x_k|^T) is Z_k Z_k^T with
    Z_k[a] = V4[a] (x) x_ak / |x_ak| (N x 12); dominant eigenvector = Z_k w / |Z_k w|, w from the 12 x 12 companion.
    (The reference needs O(N^2) memory per image here; this form needs O(N).)
  * the reprojection error of :43-58 from the same M, S.

It is the checker of the device depth loop: only ``tests/`` may import it; the product
(``lib/perspective_camera_calibration.py``) runs the iteration on the GPU and has no host form of it.

Parity status: PINNED.  ``tests/test_callers_cpu.py`` runs three forced iterations and the converged loops of both
schemes over this module (NumPy SVD) against the depths captured from the reference
(``tests/golden/calibration.npz``: ``persp_primary_z3``, ``persp_dual_z3``, ``persp_*_z``, first stdout line).
Sign note: in the dual scheme the sign of an image's depth vector is the sign LAPACK gives an eigenvector in the
reference; here every image's vector is oriented to a non-negative sum (projectively equivalent, P_k ~ -P_k).
"""
from __future__ import annotations

import numpy as np


def numpy_svd4(Wt):
    """Wt (N, 3m) -> U[:, :4] (3m, 4), sigma, diag(sigma[:4]) Vt[:4] (4, N)  (what factorization_method returns, ref
    lib/factorization.py:10-13, thin)."""
    U, s, Vt = np.linalg.svd(Wt.T, full_matrices=False)
    return U[:, :4], s, np.diag(s[:4]) @ Vt[:4]


def reprojection_error(x, M, S, f0):
    """f0 * sqrt(mean |x - [M S normalised to third component 1]|^2)  (ref :43-58)."""
    PX = (M @ S).reshape(-1, 3, S.shape[1]).transpose(2, 0, 1)
    PX = PX / PX[..., 2:3]
    return float(f0 * np.sqrt(((x - PX) ** 2).sum(axis=2).mean()))


def dominant_left_vector(C):
    """Unit dominant left singular vector of each C[i] (.., p, q), q small: C v / |C v| with v the dominant
    eigenvector of the q x q companion C^T C."""
    G = np.einsum("...pi,...pj->...ij", C, C)
    _lam, vec = np.linalg.eigh(G)
    xi = np.einsum("...pq,...q->...p", C, vec[..., -1])
    return xi / np.linalg.norm(xi, axis=-1, keepdims=True)


class HostDepthLoop:
    """The protocol of the device depth loop (lib.perspective_camera_calibration._DeviceDepthLoop) on the host:
    step(method, f0) = one iteration (factorise x o z normalised, update z, return the reprojection error),
    depths() = the current z."""

    def __init__(self, x, svd4=numpy_svd4):
        self.x = np.asarray(x, dtype=np.float64)          # (N, m, 3) homogeneous observations (ref :34-40)
        self.n, self.m = self.x.shape[:2]
        self.z = np.ones((self.n, self.m))                # ref :75 / :160
        self.x_norm = np.linalg.norm(self.x, axis=2)
        self.svd4 = svd4

    def step(self, method, f0):
        x, z = self.x, self.z
        W = x * z[..., None]
        if method == 1:    # every point's 3m-vector to unit length (ref :81-85)
            W = W / np.linalg.norm(W, axis=(1, 2))[:, None, None]
        elif method == 2:  # every image's 3 x N block divided by its SQUARED Frobenius norm (ref :170-172)
            W = W / (W ** 2).sum(axis=(0, 2))[None, :, None]
        else:
            raise ValueError("method must be 1 (primary) or 2 (dual)")
        M, sigma, S = self.svd4(np.ascontiguousarray(W.reshape(self.n, -1)))
        if method == 1:
            U4 = M.reshape(self.m, 3, 4)
            C = np.einsum("akc,kci->aki", x, U4) / self.x_norm[..., None]   # ref :93-107
            xi = dominant_left_vector(C)                                    # ref :110-118
        else:
            if not (sigma[:4] > 0).all():
                raise np.linalg.LinAlgError("measurement matrix has rank < 4")
            V4 = (S / sigma[:4, None]).T                                    # ref :182 (N, 4)
            Z = np.einsum("ai,akc->kaic", V4, x / self.x_norm[..., None]).reshape(self.m, self.n, 12)
            xi = dominant_left_vector(Z).T                                  # ref :185-213 (N, m)
            xi = xi * np.where(xi.sum(axis=0) < 0, -1.0, 1.0)[None, :]      # every image's vector to a non-negative sum
        xi[xi.sum(axis=1) < 0] *= -1                                        # ref :121 / :217
        z[...] = xi / self.x_norm                                           # ref :124 / :220
        return reprojection_error(x, M, S, f0)                              # ref :129 / :224

    def depths(self):
        return self.z.copy()

    def close(self):
        pass
