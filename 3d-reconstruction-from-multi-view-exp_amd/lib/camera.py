"""Pinhole camera model with the reference's names (lib/camera.py:7-98): K [R^T | -R^T t],
columns of R = camera axes in the world frame, world-up = +x for the look-at constructor."""
from __future__ import annotations

import numpy as np
import numpy.typing as npt

from .utils import unit_vec


class Camera:
    def __init__(self, R: npt.NDArray, t: npt.NDArray, K: npt.NDArray = np.eye(3)):
        self._R, self._t, self._K = R, t, K

    def get_camera_matrix(self) -> npt.NDArray:
        Rt = self._R.T
        return self._K @ np.column_stack([Rt, -Rt @ self._t])

    def get_parameters(self) -> tuple[npt.NDArray, npt.NDArray, npt.NDArray]:
        return self._K, self._R, self._t

    def project_points(self, X: npt.NDArray, method: str = "perspective") -> npt.NDArray:
        Xh = np.column_stack([X, np.ones(len(X))])
        if method == "perspective":
            p = Xh @ self.get_camera_matrix().T
            return p[:, :2] / p[:, 2:]
        if method == "orthographic":
            Rt = self._R.T
            return (Xh @ np.column_stack([Rt, -Rt @ self._t]).T)[:, :2]
        raise ValueError()

    @staticmethod
    def create(origin=(0.0, 0.0, 0.0), target=(0.0, 0.0, 1.0), f: float = 1.0, f0: float = 1.0) -> "Camera":
        origin, target = np.asarray(origin), np.asarray(target)
        z = unit_vec(target - origin)                      # optical axis
        y = unit_vec(np.cross(z, np.array([1.0, 0.0, 0.0])))  # camera right
        x = unit_vec(np.cross(y, z))                       # camera up
        return Camera(np.column_stack([x, y, z]), origin, np.diag((f, f, f0)))


def calc_projected_points(X, K, R, t):
    """Perspective projections of X in every camera -> list of (N,2)."""
    return [Camera(Rk, tk, Kk).project_points(X, method="perspective") for Rk, tk, Kk in zip(R, t, K)]


def calc_projected_points_gpu(X, K, R, t, device=-1):
    """``calc_projected_points`` on the MI355X (``mvba_project``, csrc/mvba.hip): same list of (N,2)
    arrays, one launch for the whole (point, camera) grid."""
    from ._mvba import project

    x = project(X, K, R, t, device=device)  # (N, m, 2)
    return [np.ascontiguousarray(x[:, k]) for k in range(x.shape[1])]


def get_camera_parames(camera_list):
    K, R, t = zip(*(c.get_parameters() for c in camera_list))
    return np.stack(K), np.stack(R), np.stack(t)
