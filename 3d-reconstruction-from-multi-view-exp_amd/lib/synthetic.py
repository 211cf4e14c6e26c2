"""Synthetic multi-view scenes in sparse (observation-list) form.

The reference builds its demo scene densely (euclidiean_reconstruction.py:14-40:
hemisphere cameras via lib/utils.py:40-52, look-at via lib/camera.py:44-71, exact
projection + Gaussian noise).  This module follows the same recipe (SURVEY.md
§8d) but never forms an (N, m) grid, and is seeded per 65,536-point chunk so a
point shard can generate exactly its own slice of the global scene.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

CHUNK = 1 << 16


def rodrigues_batch(w: np.ndarray) -> np.ndarray:
    """R(omega) for rows of w (n,3); identity where omega == 0 (lib/utils.py:10-29)."""
    th = np.linalg.norm(w, axis=1)
    safe = np.where(th > 0, th, 1.0)
    n = w / safe[:, None]
    c, s = np.cos(th), np.sin(th)
    K = np.zeros((len(w), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -n[:, 2], n[:, 1]
    K[:, 1, 0], K[:, 1, 2] = n[:, 2], -n[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -n[:, 1], n[:, 0]
    R = (1 - c)[:, None, None] * n[:, :, None] * n[:, None, :] + c[:, None, None] * np.eye(3) + s[:, None, None] * K
    R[th == 0] = np.eye(3)
    return R


def look_at(pos: np.ndarray, target: np.ndarray) -> np.ndarray:
    """Columns = (camera up, camera right, optical axis), world-up = +x (lib/camera.py:44-56)."""
    z = target - pos
    z = z / np.linalg.norm(z, axis=1, keepdims=True)
    up = np.array([1.0, 0.0, 0.0])
    y = np.cross(z, up)
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    x = np.cross(y, z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.stack([x, y, z], axis=2)


def project_obs(X, f, u, t, R, f0, pt, cam):
    """Pinhole projection per observation: K [R^T | -R^T t] (lib/camera.py:13-14, 30-34)."""
    d = X[pt] - t[cam]
    Rk = R[cam]
    c = np.einsum("oji,oj->oi", Rk, d)
    return np.stack([(f[cam] * c[:, 0] + u[cam, 0] * c[:, 2]) / (f0 * c[:, 2]),
                     (f[cam] * c[:, 1] + u[cam, 1] * c[:, 2]) / (f0 * c[:, 2])], axis=1)


@dataclass
class Scene:
    n_points: int          # points in THIS slice
    n_images: int
    point_offset: int      # global id of local point 0
    pt_ptr: np.ndarray     # (n_points+1,) int64
    cam_idx: np.ndarray    # (n_obs,) int32
    xy: np.ndarray         # (n_obs,2)
    X_gt: np.ndarray
    K_gt: np.ndarray
    R_gt: np.ndarray
    t_gt: np.ndarray
    init_X: np.ndarray
    init_K: np.ndarray
    init_R: np.ndarray
    init_t: np.ndarray
    f0: float = 1.0
    axis: str = "x-up_z-forward"

    @property
    def n_obs(self):
        return int(self.cam_idx.shape[0])

    def dense(self):
        """(x (N,m,2), vis (N,m)) for the reference-shaped constructor (small scenes only)."""
        x = np.zeros((self.n_points, self.n_images, 2))
        vis = np.zeros((self.n_points, self.n_images), dtype=bool)
        pt = np.repeat(np.arange(self.n_points), np.diff(self.pt_ptr))
        x[pt, self.cam_idx] = self.xy
        vis[pt, self.cam_idx] = True
        return x, vis


def make_cameras(n_images, seed=0, perturb_seed=2, sigma=0.01):
    """Ground-truth cameras and their perturbed initial estimates (global, tiny)."""
    for attempt in range(64):
        rng = np.random.default_rng([seed, attempt])
        theta = rng.uniform(0, np.pi / 2, n_images)
        phi = rng.uniform(0, 2 * np.pi, n_images)
        pos = 5.0 * np.stack([np.cos(theta), np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi)], axis=1)
        tgt = rng.normal(0, 0.3, (n_images, 3))
        R = look_at(pos, tgt)
        prng = np.random.default_rng([perturb_seed, attempt])
        t0 = pos + prng.normal(0, sigma, pos.shape)
        R0 = rodrigues_batch(prng.normal(0, sigma, (n_images, 3))) @ R
        f0_ = 1.0 + prng.normal(0, sigma, n_images)
        # gauge baseline (x-up: camera-0-frame component 1) must be well away from 0 and its
        # sign must agree with the world-frame component so the output frame = input frame
        # (bundle_adjustment.py:23-26, :227-238)
        base_cam = (R0[0].T @ (t0[1] - t0[0]))[1]
        base_world = (t0[1] - t0[0])[1]
        if abs(base_cam) > 0.1 and np.sign(base_cam) == np.sign(base_world) and base_cam > 0:
            break
    else:
        raise RuntimeError("could not draw a camera set with a usable gauge baseline")
    K = np.tile(np.eye(3), (n_images, 1, 1))
    K0 = K.copy()
    K0[:, 0, 0] = K0[:, 1, 1] = f0_
    return K, R, pos, K0, R0, t0


def _project(X, K, R, t, pt, cam, how):
    """Exact projections of the chunk's observations: on the MI355X (``mvba_project``) when a device
    is there, else host NumPy.  The two agree to ~1e-15 (tests/test_gpu_callers.py)."""
    if how == "auto":
        from . import _mvba

        try:
            how = "gpu" if _mvba.device_count() > 0 else "numpy"
        except Exception:  # noqa: BLE001  (library not built: CPU-only test environments)
            how = "numpy"
    if how == "gpu":
        from . import _mvba

        pt_ptr = np.zeros(len(X) + 1, np.int64)
        np.cumsum(np.bincount(pt, minlength=len(X)), out=pt_ptr[1:])
        return _mvba.project(X, K, R, t, pt_ptr, cam)
    return project_obs(X, K[:, 0, 0], K[:, :2, 2], t, R, 1.0, pt, cam)


def _chunk_points(cid, n_images, vis_p, seed, vis_seed, perturb_seed, noise, sigma, cams, project="auto"):
    """Everything for global points [cid*CHUNK, (cid+1)*CHUNK)."""
    K, R, t = cams
    rng = np.random.default_rng([seed, 1, cid])
    X = rng.uniform(-1, 1, (CHUNK, 3))
    vr = np.random.default_rng([vis_seed, cid])
    if vis_p >= 1.0:
        pt = np.repeat(np.arange(CHUNK), n_images)
        cam = np.tile(np.arange(n_images), CHUNK)
    else:
        # iid Bernoulli(p) over the CHUNK x m grid by geometric skipping: O(n_obs)
        total = CHUNK * n_images
        n_draw = int(total * vis_p + 8 * np.sqrt(total * vis_p) + 64)
        flat = np.cumsum(vr.geometric(vis_p, n_draw)) - 1
        while flat[-1] < total:
            more = flat[-1] + np.cumsum(vr.geometric(vis_p, n_draw))
            flat = np.concatenate([flat, more])
        flat = flat[flat < total]
        pt, cam = flat // n_images, flat % n_images
        deg = np.bincount(pt, minlength=CHUNK)
        low = np.nonzero(deg < 3)[0]
        if len(low):  # repair: every point gets >= 3 views
            keep = ~np.isin(pt, low)
            add_pt, add_cam = [], []
            for a in low:
                cs = vr.choice(n_images, size=3, replace=False)
                add_pt.append(np.full(3, a))
                add_cam.append(np.sort(cs))
            pt = np.concatenate([pt[keep]] + add_pt)
            cam = np.concatenate([cam[keep]] + add_cam)
            order = np.lexsort((cam, pt))
            pt, cam = pt[order], cam[order]
    xy = _project(X, K, R, t, pt, cam.astype(np.int32), project) + vr.normal(0, noise, (len(pt), 2))
    X0 = X + np.random.default_rng([perturb_seed, 1, cid]).normal(0, sigma, X.shape)
    return X, X0, pt, cam.astype(np.int32), xy


def make_scene(n_points, n_images, vis_p=1.0, seed=0, vis_seed=1, perturb_seed=2, noise=1e-3, sigma=0.01,
               point_range=None, project="auto") -> Scene:
    """Scene slice for global points ``point_range = (lo, hi)`` (default: all).  ``project``:
    "gpu" (mvba_project), "numpy", or "auto" (the device when one is visible)."""
    lo, hi = (0, n_points) if point_range is None else point_range
    K, R, t, K0, R0, t0 = make_cameras(n_images, seed, perturb_seed, sigma)
    Xs, X0s, pts, cams, xys = [], [], [], [], []
    base = 0
    for cid in range(lo // CHUNK, (max(hi, lo + 1) - 1) // CHUNK + 1):
        X, X0, pt, cam, xy = _chunk_points(cid, n_images, vis_p, seed, vis_seed, perturb_seed, noise, sigma, (K, R, t), project)
        g0 = cid * CHUNK
        a, b = max(lo, g0) - g0, min(hi, g0 + CHUNK) - g0
        sel = (pt >= a) & (pt < b)
        Xs.append(X[a:b]); X0s.append(X0[a:b])
        pts.append(pt[sel] - a + base); cams.append(cam[sel]); xys.append(xy[sel])
        base += b - a
    n_loc = hi - lo
    pt = np.concatenate(pts) if pts else np.zeros(0, np.int64)
    pt_ptr = np.zeros(n_loc + 1, np.int64)
    np.cumsum(np.bincount(pt, minlength=n_loc), out=pt_ptr[1:])
    return Scene(n_loc, n_images, lo, pt_ptr, np.concatenate(cams), np.concatenate(xys),
                 np.concatenate(Xs), K, R, t, np.concatenate(X0s), K0, R0, t0)


def scene_shard(n_points, n_images, vis_p, rank, world):
    """Point range [lo, hi) of rank ``rank`` when the global scene is split ``world`` ways.
    Visibility is iid per (point, camera), so equal point counts are observation-balanced to
    ~1/sqrt(n_obs); ``lib._distributed.partition_points`` does the exact split when a global
    ``pt_ptr`` exists (it never does at config-4 size: 250 M observations)."""
    return n_points * rank // world, n_points * (rank + 1) // world
