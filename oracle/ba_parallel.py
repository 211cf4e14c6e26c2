"""All-host-cores form of the CPU oracle  --  TEST / BASELINE INFRASTRUCTURE ONLY.

NumPy's element-wise and sparse kernels run on one thread, so a single
``OracleEngine`` leaves a many-core host idle.  Points are independent given the
cameras, so this module runs one ``OracleEngine`` per worker
process on a contiguous point shard -- the same split the GPUs use -- and the
parent sums the partial reduced systems ``[A | b]``, solves once, and hands the
camera increment back for the shards' back-substitution and trial cost.

It exposes the engine protocol of ``oracle/ba_oracle.py`` (cost / linearize /
try_step / commit), so ``lib.bundle_adjustment.lm_loop`` drives it unchanged.
Only ``bench.py``'s ``cpu_baseline`` leg and ``tests/`` may import it.
"""
from __future__ import annotations

import multiprocessing as mp
import os

import numpy as np

from . import ba_oracle as O

_G = {}  # parent-side scene, inherited by the forked workers (copy-on-write, nothing is pickled)


def _worker(rank, conn):
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)  # one BLAS thread per worker: the workers are the parallelism
    except Exception:  # noqa: BLE001
        limiter = None
    sc = _G["scene"]
    lo, hi = _G["parts"][rank]
    o0, o1 = int(sc["pt_ptr"][lo]), int(sc["pt_ptr"][hi])
    eng = O.OracleEngine(hi - lo, sc["m"], sc["pt_ptr"][lo:hi + 1] - o0, sc["cam"][o0:o1], sc["xy"][o0:o1],
                         sc["f0"], sc["axis"])
    eng.set_params(sc["X"][lo:hi], sc["f"], sc["u"], sc["t"], sc["R"])
    while True:
        name, payload = conn.recv()
        if name == "cost":
            out = O.cost(eng.X, eng.f, eng.u, eng.t, eng.R, eng.f0, eng.pt, eng.cam, eng.xy)
        elif name == "linearize":
            eng.linearize()
            out = None
        elif name == "reduced":
            A, b = eng.reduced_system(payload)
            out = np.concatenate([A.reshape(-1), b])
        elif name == "apply":
            out = eng.apply_step(payload)
        elif name == "commit":
            eng.commit()
            out = None
        elif name == "get_X":
            out = eng.X
        else:  # "stop"
            conn.send(None)
            break
        conn.send(out)
    del limiter


class ShardedOracle:
    """Engine protocol over ``n_workers`` forked processes, one point shard each (same contiguous,
    observation-balanced split as the GPUs: ``lib._distributed.partition_points``)."""

    def __init__(self, n_points, n_images, pt_ptr, cam_idx, xy, f0, axis, X, f, u, t, R, n_workers=None):
        from lib._distributed import partition_points

        self.m = int(n_images)
        self.n_workers = int(n_workers or min(os.cpu_count() or 1, 64))
        _G["scene"] = {"m": self.m, "pt_ptr": np.asarray(pt_ptr, np.int64), "cam": np.asarray(cam_idx),
                       "xy": np.asarray(xy), "f0": float(f0), "axis": axis, "X": np.asarray(X, float),
                       "f": np.asarray(f, float), "u": np.asarray(u, float), "t": np.asarray(t, float),
                       "R": np.asarray(R, float)}
        _G["parts"] = partition_points(pt_ptr, self.n_workers)
        self.keep = np.setdiff1d(np.arange(9 * self.m), O.gauge_removed(axis))
        ctx = mp.get_context("fork")
        self.conns, self.procs = [], []
        for r in range(self.n_workers):
            a, b = ctx.Pipe()
            pr = ctx.Process(target=_worker, args=(r, b), daemon=True)
            pr.start()
            self.conns.append(a)
            self.procs.append(pr)
        _G.pop("scene")
        self.n_solves = 0

    def _all(self, name, payload=None):
        for c in self.conns:
            c.send((name, payload))
        return [c.recv() for c in self.conns]  # rank order: sums are taken in a fixed order

    def cost(self):
        return float(sum(self._all("cost")))

    def linearize(self):
        self._all("linearize")

    def try_step(self, c):
        packed = sum(self._all("reduced", c))
        n9 = 9 * self.m
        A, b = packed[:-n9].reshape(n9, n9), packed[-n9:]
        dxi = np.zeros(n9)
        dxi[self.keep] = np.linalg.solve(A[np.ix_(self.keep, self.keep)], b[self.keep])
        self.n_solves += 1
        return float(sum(self._all("apply", dxi)))

    def commit(self):
        self._all("commit")

    def points(self):
        return np.concatenate(self._all("get_X"))

    def close(self):
        if self.conns:
            self._all("stop")
            for pr in self.procs:
                pr.join(timeout=10)
            self.conns, self.procs = [], []

    __del__ = close
