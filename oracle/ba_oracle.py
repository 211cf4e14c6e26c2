"""CPU oracle for the bundle-adjustment hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement, in sparse per-observation form, of the
algorithm in the reference's ``lib/bundle_adjustment.py`` (class
``BundleAdjuster``, lines 10-677) and ``lib/utils.py:10-29`` (Rodrigues).  It is
the *checker* for the HIP engine: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``3d-reconstruction-from-multi-view-exp_amd/lib``) never does and fails loudly
without the HIP library.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against vectors captured from the reference itself
(``tests/golden/make_golden.py``): every intermediate of one linearisation
(p,q,r,dP,dF,E,F,G,A,b,dxi,dX,E'), the full LM trajectories of the two default
scenes (37 outer / 59 solves; 100 / 197), partial-visibility scenes, both gauge
axes, Rodrigues and the normalise / denormalise pair.

Notation: point a, camera k, X_a, f_k, (u_k,v_k), t_k,
R_k = [r1 r2 r3] (columns = camera axes in the world frame), constant f0.
Camera parameter order xi_k = [f, u, v, t1,t2,t3, w1,w2,w3]  (ref :423-425).
"""
from __future__ import annotations

import numpy as np

AXES = {"x-right_z-forward": 0, "x-up_z-forward": 1}


# ---------------------------------------------------------------- small helpers
def rodrigues(omega: np.ndarray) -> np.ndarray:
    """R(omega); exactly I when omega == 0  (ref lib/utils.py:10-29)."""
    omega = np.asarray(omega, dtype=np.float64)
    if (omega == 0.0).all():
        return np.eye(3)
    th = np.linalg.norm(omega)
    n = omega / th
    c, s = np.cos(th), np.sin(th)
    nx = np.array([[0.0, -n[2], n[1]], [n[2], 0.0, -n[0]], [-n[1], n[0], 0.0]])
    return (1.0 - c) * np.outer(n, n) + c * np.eye(3) + s * nx


def gauge_removed(axis: str) -> np.ndarray:
    """Fixed parameter indices: camera 0's t, omega and one t component of camera 1
    (ref :62-72): 12 (t_x) for x-right, 13 (t_y) for x-up."""
    if axis not in AXES:
        raise ValueError()
    return np.array([3, 4, 5, 6, 7, 8, 12 + AXES[axis]])


def normalize_scene(X, R, t, axis):
    """ref :208-240.  Camera 0 -> identity at the origin; baseline component -> +-1.
    The sign comes from the WORLD-frame component of t1-t0, the magnitude from the
    CAMERA-0-frame component (quirk)."""
    if axis not in AXES:
        raise ValueError()
    ax = AXES[axis]
    X_ = X - t[0]
    t_ = t - t[0]
    j = np.zeros(3)
    j[ax] = np.sign(t_[1, ax])
    s = j @ R[0].T @ t_[1][:, None]  # shape (1,)
    return (X_ @ R[0]) / s, R[0].T @ R, (t_ @ R[0]) / s


def baseline_length(R, t, axis):
    """ref :23-28 (abs of the camera-0-frame component)."""
    if axis not in AXES:
        raise ValueError()
    return np.abs(R[0, :, AXES[axis]] @ (t[1] - t[0]))


def denormalize_scene(R0, t0, scale, X, R, t):
    """ref :242-258."""
    return (scale * X) @ R0.T + t0, R0 @ R, (scale * t) @ R0.T + t0


def dense_to_observations(x, vis=None):
    """Dense (N,m,2) + bool mask -> CSR-by-point observation list."""
    n, m = x.shape[:2]
    vis = np.ones((n, m), bool) if vis is None else np.asarray(vis, bool)
    pt, cam = np.nonzero(vis)  # row-major: sorted by point then camera
    deg = vis.sum(axis=1)
    pt_ptr = np.zeros(n + 1, np.int64)
    np.cumsum(deg, out=pt_ptr[1:])
    return pt_ptr, cam.astype(np.int32), np.ascontiguousarray(x[pt, cam], dtype=np.float64)


def _segsum(idx, w, n):
    """sum of rows of w (n_obs, k...) grouped by idx -> (n, k...)."""
    flat = w.reshape(w.shape[0], -1)
    out = np.empty((n, flat.shape[1]))
    for j in range(flat.shape[1]):
        out[:, j] = np.bincount(idx, weights=flat[:, j], minlength=n)
    return out.reshape((n,) + w.shape[1:])


# ---------------------------------------------------------------- per-observation math
def project(X, f, u, t, R, f0, pt, cam):
    """(p,q,r) per observation  (ref :291-307)."""
    d = X[pt] - t[cam]
    Rk = R[cam]
    c = np.einsum("oji,oj->oi", Rk, d)  # camera-frame coordinates R^T d
    p = f[cam] * c[:, 0] + u[cam, 0] * c[:, 2]
    q = f[cam] * c[:, 1] + u[cam, 1] * c[:, 2]
    r = f0 * c[:, 2]
    return p, q, r, d


def residuals(X, f, u, t, R, f0, pt, cam, xy):
    p, q, r, _ = project(X, f, u, t, R, f0, pt, cam)
    return np.stack([p / r - xy[:, 0] / f0, q / r - xy[:, 1] / f0], axis=1)


def cost(X, f, u, t, R, f0, pt, cam, xy):
    """E = sum |e|^2 over visible observations  (ref :666-677) - a SUM, not a mean."""
    e = residuals(X, f, u, t, R, f0, pt, cam, xy)
    return float((e[:, 0] ** 2 + e[:, 1] ** 2).sum())


def jacobians(X, f, u, t, R, f0, pt, cam, xy):
    """e (n_obs,2), J_X (n_obs,2,3), J_C (n_obs,2,9)  (ref :309-427, :445-459).

    J rows are (r dp - p dr)/r^2 and (r dq - q dr)/r^2."""
    p, q, r, d = project(X, f, u, t, R, f0, pt, cam)
    fk, uk, vk = f[cam], u[cam, 0], u[cam, 1]
    Rk = R[cam]
    a_p = fk[:, None] * Rk[:, :, 0] + uk[:, None] * Rk[:, :, 2]  # = P[k,0,:3]  (:318)
    a_q = fk[:, None] * Rk[:, :, 1] + vk[:, None] * Rk[:, :, 2]
    a_r = f0 * Rk[:, :, 2]
    n = p.shape[0]
    dp = np.zeros((n, 9))
    dq = np.zeros((n, 9))
    dr = np.zeros((n, 9))
    dp[:, 0] = (p - uk / f0 * r) / fk  # ref :336
    dq[:, 0] = (q - vk / f0 * r) / fk  # ref :337
    dp[:, 1] = r / f0  # ref :350-355
    dq[:, 2] = r / f0
    dp[:, 3:6], dq[:, 3:6], dr[:, 3:6] = -a_p, -a_q, -a_r  # ref :368-376
    dp[:, 6:9] = np.cross(a_p, d)  # ref :391-396
    dq[:, 6:9] = np.cross(a_q, d)
    dr[:, 6:9] = np.cross(a_r, d)
    r2 = (r * r)[:, None]
    JX = np.stack([(r[:, None] * a_p - p[:, None] * a_r) / r2,
                   (r[:, None] * a_q - q[:, None] * a_r) / r2], axis=1)
    JC = np.stack([(r[:, None] * dp - p[:, None] * dr) / r2,
                   (r[:, None] * dq - q[:, None] * dr) / r2], axis=1)
    e = np.stack([p / r - xy[:, 0] / f0, q / r - xy[:, 1] / f0], axis=1)
    return e, JX, JC


# ---------------------------------------------------------------- the engine
class OracleEngine:
    """Same protocol as the HIP engine (lib/_mvba.py::HipEngine): set_params,
    get_params, cost, linearize, try_step, commit.  State is in the NORMALISED
    frame.  ``allreduce`` (optional) sums a float64 ndarray in place across
    point shards - the exchange step of SURVEY §8e."""

    def __init__(self, n_points, n_images, pt_ptr, cam_idx, xy, f0, axis, allreduce=None,
                 pair_chunk=200_000, sparse_schur=True):
        self.n, self.m = int(n_points), int(n_images)
        self.pt_ptr = np.asarray(pt_ptr, np.int64)
        self.cam = np.asarray(cam_idx, np.int64)
        self.xy = np.asarray(xy, np.float64).reshape(-1, 2)
        self.pt = np.repeat(np.arange(self.n), np.diff(self.pt_ptr))
        self.f0 = float(f0)
        self.removed = gauge_removed(axis)
        self.keep = np.setdiff1d(np.arange(9 * self.m), self.removed)
        self.allreduce = allreduce
        self.pair_chunk = pair_chunk
        self.sparse_schur = sparse_schur
        self.n_solves = 0

    # -- parameters
    def set_params(self, X, f, u, t, R):
        self.X, self.f, self.u = np.array(X, float), np.array(f, float), np.array(u, float)
        self.t, self.R = np.array(t, float), np.array(R, float)

    def get_params(self):
        return self.X.copy(), self.f.copy(), self.u.copy(), self.t.copy(), self.R.copy()

    def _global_sum(self, v):
        if self.allreduce is None:
            return v
        a = np.array([v], np.float64)
        self.allreduce(a)
        return float(a[0])

    def cost(self):
        return self._global_sum(cost(self.X, self.f, self.u, self.t, self.R, self.f0, self.pt, self.cam, self.xy))

    # -- linearisation at the committed state (ref :103-116)
    def linearize(self):
        e, JX, JC = jacobians(self.X, self.f, self.u, self.t, self.R, self.f0, self.pt, self.cam, self.xy)
        self.e, self.JX, self.JC = e, JX, JC
        self.dP = 2.0 * _segsum(self.pt, np.einsum("ori,or->oi", JX, e), self.n)  # (N,3)   ref :429-469
        self.dF = 2.0 * _segsum(self.cam, np.einsum("ori,or->oi", JC, e), self.m)  # (m,9)   ref :471-517
        self.E = 2.0 * _segsum(self.pt, np.einsum("ori,orj->oij", JX, JX), self.n)  # (N,3,3) ref :519-556
        self.F = 2.0 * np.einsum("ori,orj->oij", JX, JC)  # (n_obs,3,9)            ref :558-616
        self.G = 2.0 * _segsum(self.cam, np.einsum("ori,orj->oij", JC, JC), self.m)  # (m,9,9) ref :618-664


    def _schur_pairs(self, Y):
        """sum_a F_ak^T E^-1 F_al over all ordered pairs of observations of a point -> (m,m,9,9)."""
        m = self.m
        A4 = np.zeros((m, m, 9, 9))
        deg = np.diff(self.pt_ptr)
        start = 0
        while start < self.n:
            stop = start
            acc = 0
            while stop < self.n and (acc == 0 or acc + deg[stop] ** 2 <= self.pair_chunk):
                acc += int(deg[stop]) ** 2
                stop += 1
            d = deg[start:stop]
            o0 = self.pt_ptr[start:stop]
            rep = np.repeat(np.arange(stop - start), d * d)
            base = np.repeat(np.cumsum(d * d) - d * d, d * d)
            loc = np.arange(acc) - base
            dd = d[rep]
            oi = o0[rep] + loc // dd
            oj = o0[rep] + loc % dd
            blk = np.einsum("pji,pjk->pik", self.F[oi], Y[oj])  # F_ak^T E^-1 F_al
            np.add.at(A4, (self.cam[oi], self.cam[oj]), blk)
            start = stop
        return A4

    # -- reduced camera system for damping c (ref :118-143), FULL 9m x 9m before gauge removal
    def reduced_system(self, c):
        m = self.m
        Ec = self.E.copy()
        i3 = np.arange(3)
        Ec[:, i3, i3] *= 1.0 + c
        self.Einv = np.linalg.inv(Ec)  # LinAlgError("Singular matrix") on a zero-degree point, as ref :128
        Y = np.einsum("oij,ojk->oik", self.Einv[self.pt], self.F)  # E^-1 F_ak   (n_obs,3,9)
        if self.sparse_schur:
            # S = F^T (E^-1 F) as one block-sparse product (scipy BSR, 3x9 blocks): the
            # fast CPU form used for the timed cpu_baseline; equals the pair loop below.
            from scipy.sparse import bsr_matrix

            shape = (3 * self.n, 9 * m)
            Fs = bsr_matrix((self.F, self.cam, self.pt_ptr), shape=shape)
            Ys = bsr_matrix((Y, self.cam, self.pt_ptr), shape=shape)
            A = -np.asarray((Fs.T.tocsr() @ Ys.tocsr()).todense())
        else:
            A = -self._schur_pairs(Y).transpose(0, 2, 1, 3).reshape(9 * m, 9 * m)
        bvec = np.einsum("oji,oj->oi", Y, self.dP[self.pt])  # F^T E^-1 dP per obs (9)
        b = _segsum(self.cam, bvec, m) - self.dF
        for k in range(m):
            Gk = self.G[k].copy()
            Gk[np.arange(9), np.arange(9)] *= 1.0 + c
            A[9 * k:9 * k + 9, 9 * k:9 * k + 9] += Gk
        return A, b.reshape(-1)

    def try_step(self, c):
        """One LM trial (ref :118-162): returns the trial cost E'."""
        A, b = self.reduced_system(c)
        if self.allreduce is not None:  # C1: one all-reduce of [A | b] per solve
            packed = np.concatenate([A.reshape(-1), b])
            self.allreduce(packed)
            A, b = packed[:-b.size].reshape(A.shape), packed[-b.size:]
        dxi = self.solve_reduced(A, b)
        return self._global_sum(self.apply_step(dxi))

    def solve_reduced(self, A, b):
        """Gauge rows/columns out (ref :62-72), dense solve (ref :146); returns the full (9m,) increment."""
        self.A = A[np.ix_(self.keep, self.keep)]
        self.b = b[self.keep]
        dxi_red = np.linalg.solve(self.A, self.b)  # ref :146
        self.n_solves += 1
        dxi = np.zeros(9 * self.m)
        dxi[self.keep] = dxi_red
        self.dxi_red = dxi_red
        return dxi

    def apply_step(self, dxi):
        """Back-substitution, trial state and this engine's share of the trial cost (ref :152-162, :260-281)."""
        dxi = np.asarray(dxi).reshape(self.m, 9)
        Fd = np.einsum("oij,oj->oi", self.F, dxi[self.cam])
        self.dX = -np.einsum("aij,aj->ai", self.Einv, _segsum(self.pt, Fd, self.n) + self.dP)  # ref :152
        self.tX = self.X + self.dX  # ref :260-261
        self.tf = self.f + dxi[:, 0]  # ref :263-281
        self.tu = self.u + dxi[:, 1:3]
        self.tt = self.t + dxi[:, 3:6]
        self.tR = np.stack([rodrigues(w) for w in dxi[:, 6:9]]) @ self.R
        return cost(self.tX, self.tf, self.tu, self.tt, self.tR, self.f0, self.pt, self.cam, self.xy)

    def commit(self):
        self.X, self.f, self.u, self.t, self.R = self.tX, self.tf, self.tu, self.tt, self.tR


# ---------------------------------------------------------------- reference-shaped front end
class OracleBundleAdjuster:
    """Constructor / optimize / get_log with the reference's signatures
    (ref :11-21, :77-83, :204-206) on top of OracleEngine."""

    def __init__(self, x, init_X, init_K, init_R, init_t, f0=1.0, visibility_index=None,
                 axis="x-right_z-forward"):
        if axis not in AXES:
            raise ValueError()
        self._cam0 = (np.array(init_R[0]), np.array(init_t[0]), baseline_length(init_R, init_t, axis))
        X, R, t = normalize_scene(np.asarray(init_X, float), np.asarray(init_R, float),
                                  np.asarray(init_t, float), axis)
        n, m = x.shape[:2]
        pt_ptr, cam, xy = dense_to_observations(np.asarray(x), visibility_index)
        self.engine = OracleEngine(n, m, pt_ptr, cam, xy, f0, axis)
        self.engine.set_params(X, init_K[:, 0, 0], init_K[:, :2, 2], t, R)  # K[1,1], K[2,2] ignored (ref :45-48)
        self._f0 = f0
        self._log = []

    def _K(self, f, u):
        K = np.zeros((len(f), 3, 3))
        K[:, 0, 0] = K[:, 1, 1] = f
        K[:, :2, 2] = u
        K[:, 2, 2] = self._f0
        return K

    def optimize(self, scale_factor=10.0, delta_tol=1e-8, max_iter=100, is_debug=False, verbose=True):
        g = self.engine
        E = g.cost()

        def snap(err):
            X, f, u, t, R = g.get_params()
            return {"points": X, "basis": R, "pos": t, "reprojection_error": err}

        if is_debug:
            self._log.clear()
            self._log.append(snap(E))
        c, count = 0.0001, 0
        while True:
            g.linearize()
            while True:
                E_ = g.try_step(c)
                if E_ > E:
                    c *= scale_factor
                else:
                    break
            g.commit()
            if is_debug:
                self._log.append(snap(E_))
            count += 1
            delta = np.abs(E_ - E)
            if verbose:
                print(f"Iteration {count}: reprojection_error_delta = {delta}")
            if delta <= delta_tol or count >= max_iter:
                break
            E = E_
            c /= scale_factor
        X, f, u, t, R = g.get_params()
        R0, t0, scale = self._cam0
        Xo, Ro, to = denormalize_scene(R0, t0, scale, X, R, t)
        return Xo, self._K(f, u), Ro, to

    def get_log(self):
        return self._log
