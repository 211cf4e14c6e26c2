"""Plots with the names the reference's drivers import (lib/visualization.py:5-187):
show_3d_scene_data, show_2d_projection_data, animate.  UI only -- not on the hot path.
Headless-safe: on a non-interactive matplotlib backend (e.g. Agg) nothing blocks, and
`animate` renders its frames once and returns (the reference loops while the window exists,
which never ends without a window, :175)."""
from __future__ import annotations

import numpy as np


def _plt():
    import matplotlib.pyplot as plt

    return plt


def _interactive() -> bool:
    import matplotlib

    return matplotlib.get_backend().lower() not in ("agg", "pdf", "svg", "ps", "cairo", "template")


def _draw_scene(ax, X, R, t, axis_len=0.5):
    ax.scatter(X[:, 0], X[:, 1], X[:, 2], s=4, c="tab:blue")
    ax.scatter(t[:, 0], t[:, 1], t[:, 2], s=12, c="k")
    for Rk, tk in zip(R, t):
        for col, colour in zip(range(3), ("r", "g", "b")):
            tip = tk + axis_len * Rk[:, col]
            ax.plot([tk[0], tip[0]], [tk[1], tip[1]], [tk[2], tip[2]], c=colour, lw=1)
    pts = np.vstack([X, t])
    c, h = pts.mean(axis=0), np.ptp(pts, axis=0).max() / 2 + 1e-9
    ax.set_xlim(c[0] - h, c[0] + h)
    ax.set_ylim(c[1] - h, c[1] + h)
    ax.set_zlim(c[2] - h, c[2] + h)
    ax.set_xlabel("x"); ax.set_ylabel("y"); ax.set_zlabel("z")


def show_3d_scene_data(X, R, t):
    """3-D points with every camera's position and axes."""
    plt = _plt()
    fig = plt.figure()
    _draw_scene(fig.add_subplot(projection="3d"), np.asarray(X), np.asarray(R), np.asarray(t))
    if _interactive():
        plt.show()
    plt.close(fig)


def show_2d_projection_data(x_list, reproj_x_list=None, n_col=5):
    """One panel per camera: observations (blue) and, if given, re-projections (red)."""
    plt = _plt()
    n = len(x_list)
    n_row = (n + n_col - 1) // n_col
    fig, axes = plt.subplots(n_row, n_col, figsize=(3 * n_col, 3 * n_row), squeeze=False)
    for k, ax in enumerate(axes.ravel()):
        if k >= n:
            ax.axis("off")
            continue
        ax.scatter(x_list[k][:, 0], x_list[k][:, 1], s=4, c="tab:blue")
        if reproj_x_list is not None:
            ax.scatter(reproj_x_list[k][:, 0], reproj_x_list[k][:, 1], s=4, c="tab:red")
        ax.set_aspect("equal")
        ax.set_title(f"camera {k + 1}")
    if _interactive():
        plt.show()
    plt.close(fig)


def animate(data, interval=0.05):
    """Replay the LM trajectory returned by BundleAdjuster.get_log() (list of dicts with
    'points', 'basis', 'pos', 'reprojection_error')."""
    plt = _plt()
    fig = plt.figure()
    ax = fig.add_subplot(projection="3d")
    live = _interactive()
    while True:
        for i, frame in enumerate(data):
            ax.cla()
            _draw_scene(ax, frame["points"], frame["basis"], frame["pos"])
            ax.set_title(f"iteration {i}: E = {frame['reprojection_error']:.6g}")
            if live:
                if not plt.fignum_exists(fig.number):
                    return
                plt.pause(interval)
            else:
                fig.canvas.draw()
        if not live or not plt.fignum_exists(fig.number):
            break
    plt.close(fig)
