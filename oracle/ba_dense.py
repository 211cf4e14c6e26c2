"""Dense-faithful CPU oracle  --  TEST / BASELINE INFRASTRUCTURE ONLY.

``oracle/ba_oracle.py`` restates the reference per observation (sparse).  This module restates it
in the reference's OWN algorithmic form -- every quantity a dense array over the full
(point, camera) grid with a multiplicative visibility mask, the cross block ``F`` laid out as
``(N, 3, 9m)`` with the gauge columns dropped, and the Schur complement through the batched
product ``(N, D, 3) @ (N, 3, D) -> (N, D, D)`` summed over points (ref
``lib/bundle_adjustment.py:103-162``, ``:291-677``) -- because that form, not the sparse one, is
what the reference costs on a CPU (76 % of its time is the ``(N, D, D)``
temporary).  It only fits configs 1-2 (2.4 GB at 10k x 20).

Engine protocol of ``ba_oracle.OracleEngine`` (set_params / cost / linearize / try_step / commit),
so the product's own LM loop drives it.  Parity: pinned through ``tests/test_oracle_golden.py``
(same golden trajectories as the sparse oracle).  Only ``tests/`` and ``bench.py``'s
``cpu_baseline`` leg may import it.
"""
from __future__ import annotations

import numpy as np

from . import ba_oracle as O


class DenseEngine:
    def __init__(self, x, visibility, f0, axis):
        self.x = np.asarray(x, float)                      # (N, m, 2)
        self.n, self.m = self.x.shape[:2]
        self.v = np.ones((self.n, self.m)) if visibility is None else np.asarray(visibility, float)
        self.f0 = float(f0)
        self.keep = np.setdiff1d(np.arange(9 * self.m), O.gauge_removed(axis))
        self.n_solves = 0

    def set_params(self, X, f, u, t, R):
        self.X, self.f, self.u = np.array(X, float), np.array(f, float), np.array(u, float)
        self.t, self.R = np.array(t, float), np.array(R, float)

    def get_params(self):
        return self.X.copy(), self.f.copy(), self.u.copy(), self.t.copy(), self.R.copy()

    # -- projection of every point into every camera (ref :291-307)
    def _pqr(self, X, f, u, t, R):
        K = np.zeros((self.m, 3, 3))
        K[:, 0, 0] = K[:, 1, 1] = f
        K[:, :2, 2] = u
        K[:, 2, 2] = self.f0
        Rt = R.transpose(0, 2, 1)
        P = K @ np.concatenate([Rt, -(Rt @ t[:, :, None])], axis=2)  # (m, 3, 4)
        Xh = np.concatenate([X, np.ones((self.n, 1))], axis=1)        # (N, 4)
        pqr = np.einsum("krc,ac->akr", P, Xh)                        # (N, m, 3)
        return P, pqr[..., 0], pqr[..., 1], pqr[..., 2]

    def _cost(self, p, q, r):
        return float((self.v * ((p / r - self.x[..., 0] / self.f0) ** 2 + (q / r - self.x[..., 1] / self.f0) ** 2)).sum())

    def cost(self):
        _, p, q, r = self._pqr(self.X, self.f, self.u, self.t, self.R)
        return self._cost(p, q, r)

    # -- dense first and second derivatives (ref :309-664)
    def linearize(self):
        n, m, f0 = self.n, self.m, self.f0
        P, p, q, r = self._pqr(self.X, self.f, self.u, self.t, self.R)
        self.p, self.q, self.r = p, q, r
        ones = np.ones((n, 1, 1))
        dX = [ones * P[None, :, row, :3] for row in range(3)]          # d(p|q|r)/dX  (N, m, 3)
        dC = [np.zeros((n, m, 9)) for _ in range(3)]                   # d(p|q|r)/d(f,u,v,t,omega)
        dC[0][..., 0] = (p - self.u[:, 0] / f0 * r) / self.f
        dC[1][..., 0] = (q - self.u[:, 1] / f0 * r) / self.f
        dC[0][..., 1] = r / f0
        dC[1][..., 2] = r / f0
        d = self.X[:, None, :] - self.t[None, :, :]                    # (N, m, 3)
        for row in range(3):
            dC[row][..., 3:6] = -dX[row]
            dC[row][..., 6:9] = np.cross(dX[row], d)
        r2 = (r * r)[..., None]
        gX = [(r[..., None] * dX[0] - p[..., None] * dX[2]), (r[..., None] * dX[1] - q[..., None] * dX[2])]
        gC = [(r[..., None] * dC[0] - p[..., None] * dC[2]), (r[..., None] * dC[1] - q[..., None] * dC[2])]
        e = [p / r - self.x[..., 0] / f0, q / r - self.x[..., 1] / f0]
        w = self.v[..., None]
        self.dP = 2.0 * (w * (e[0][..., None] * gX[0] + e[1][..., None] * gX[1]) / r2).sum(axis=1).reshape(-1)   # (3N,)
        dF = 2.0 * (w * (e[0][..., None] * gC[0] + e[1][..., None] * gC[1]) / r2).sum(axis=0).reshape(-1)        # (9m,)
        self.dF = dF[self.keep]
        r4 = (r2 * r2)[..., None]
        w2 = self.v[..., None, None]
        outer = lambda a, b: a[..., :, None] * b[..., None, :]  # noqa: E731
        self.E = 2.0 * (w2 * (outer(gX[0], gX[0]) + outer(gX[1], gX[1])) / r4).sum(axis=1)                      # (N, 3, 3)
        Fb = 2.0 * w2 * (outer(gX[0], gC[0]) + outer(gX[1], gC[1])) / r4                                         # (N, m, 3, 9)
        self.F = Fb.transpose(0, 2, 1, 3).reshape(n, 3, 9 * m)[:, :, self.keep]                                  # (N, 3, D) dense
        Gb = 2.0 * (w2 * (outer(gC[0], gC[0]) + outer(gC[1], gC[1])) / r4).sum(axis=0)                           # (m, 9, 9)
        G = np.zeros((9 * m, 9 * m))
        for k in range(m):
            G[9 * k:9 * k + 9, 9 * k:9 * k + 9] = Gb[k]
        self.G = G[np.ix_(self.keep, self.keep)]

    # -- one LM trial (ref :118-162)
    def try_step(self, c):
        Ec = self.E.copy()
        i3 = np.arange(3)
        Ec[:, i3, i3] *= 1.0 + c
        Gc = self.G.copy()
        iD = np.arange(Gc.shape[0])
        Gc[iD, iD] *= 1.0 + c
        Einv = np.linalg.inv(Ec)
        FtEinv = self.F.transpose(0, 2, 1) @ Einv                       # (N, D, 3)
        A = Gc - (FtEinv @ self.F).sum(axis=0)                          # the (N, D, D) temporary of ref :132-135
        dP = self.dP.reshape(self.n, 3, 1)
        b = (FtEinv @ dP)[..., 0].sum(axis=0) - self.dF
        dxi_red = np.linalg.solve(A, b)
        self.n_solves += 1
        dX = -(Einv @ (self.F @ dxi_red[:, None] + dP))[..., 0]
        dxi = np.zeros(9 * self.m)
        dxi[self.keep] = dxi_red
        dxi = dxi.reshape(self.m, 9)
        self.tX = self.X + dX
        self.tf, self.tu, self.tt = self.f + dxi[:, 0], self.u + dxi[:, 1:3], self.t + dxi[:, 3:6]
        self.tR = np.stack([O.rodrigues(wv) for wv in dxi[:, 6:9]]) @ self.R
        _, p, q, r = self._pqr(self.tX, self.tf, self.tu, self.tt, self.tR)
        return self._cost(p, q, r)

    def commit(self):
        self.X, self.f, self.u, self.t, self.R = self.tX, self.tf, self.tu, self.tt, self.tR


class DenseBundleAdjuster(O.OracleBundleAdjuster):
    """Reference-shaped front end (ctor / optimize / get_log, ref :11-21, :77-83, :204-206): the
    sparse oracle's LM driver over ``DenseEngine``."""

    def __init__(self, x, init_X, init_K, init_R, init_t, f0=1.0, visibility_index=None, axis="x-right_z-forward"):
        if axis not in O.AXES:
            raise ValueError()
        init_X, init_K = np.asarray(init_X, float), np.asarray(init_K, float)
        init_R, init_t = np.asarray(init_R, float), np.asarray(init_t, float)
        self._cam0 = (init_R[0].copy(), init_t[0].copy(), O.baseline_length(init_R, init_t, axis))
        X, R, t = O.normalize_scene(init_X, init_R, init_t, axis)
        self.engine = DenseEngine(x, visibility_index, f0, axis)
        self.engine.set_params(X, init_K[:, 0, 0], init_K[:, :2, 2], t, R)  # K[1,1], K[2,2] ignored (ref :45-48)
        self._f0 = f0
        self._log = []
