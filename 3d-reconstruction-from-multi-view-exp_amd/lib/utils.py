"""Small geometry / scene helpers with the names the reference's drivers import
(lib/utils.py:5-63).  Re-stated, not copied; pinned by tests/golden/known_answers.npz."""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray


def unit_vec(x: NDArray) -> NDArray:
    return x / np.linalg.norm(x)


def get_rotation_matrix(omega: NDArray) -> NDArray:
    """Rodrigues: rotation by |omega| about omega/|omega|; exactly I for omega == 0 (utils.py:14-15)."""
    assert omega.shape == (3,)
    if not omega.any():
        return np.eye(3)
    theta = np.linalg.norm(omega)
    n = omega / theta
    c, s = np.cos(theta), np.sin(theta)
    cross = np.array([[0.0, -n[2], n[1]], [n[2], 0.0, -n[0]], [-n[1], n[0], 0.0]])
    return (1.0 - c) * np.outer(n, n) + c * np.eye(3) + s * cross


def sample_hemisphere_points(num: int, r: float) -> NDArray:
    """Points on the x >= 0 hemisphere of radius r; draws (theta, phi) per point from the
    global NumPy RNG in that order, like the reference (utils.py:40-52), so seeded scenes agree."""
    out = np.empty((num, 3))
    for i in range(num):
        theta = np.random.uniform(0, np.pi / 2)
        phi = np.random.uniform(0, 2 * np.pi)
        out[i] = (r * np.cos(theta), r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi))
    return out


def set_points() -> NDArray:
    """The demo's 200-point bell surface: 10 x-slices times 20 angles (utils.py:55-63)."""
    xs = np.repeat(np.linspace(-1, 1, 10), 20)
    th = np.tile(np.linspace(np.pi / 2, 3 * np.pi / 2, 20), 10)
    r = 1 / (xs + 2)
    return np.stack([xs, r * np.cos(th), r * np.sin(th)], axis=1)
