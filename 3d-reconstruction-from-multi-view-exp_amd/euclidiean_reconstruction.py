#!/usr/bin/env python3
"""Euclidean reconstruction demo on the MI355X backend: synthetic scene -> perspective
self-calibration (dual projective depths, GPU factorization SVD) -> bundle adjustment
(libmvba.so).  Same call sequence, seed and constants as the reference's
euclidiean_reconstruction.py:13-66; run it from this directory.  --no-show skips the plots."""
import sys

import numpy as np

from lib.bundle_adjustment import BundleAdjuster
from lib.camera import Camera, calc_projected_points, get_camera_parames
from lib.perspective_camera_calibration import perspective_self_calibration
from lib.utils import sample_hemisphere_points, set_points
from lib.visualization import animate, show_2d_projection_data, show_3d_scene_data


def main(show=True):
    np.random.seed(123)
    f, n_images = 1.0, 10
    camera_pos = sample_hemisphere_points(n_images, 5)
    targets = np.random.normal(0, 0.5, (n_images, 3))
    cameras = [Camera.create(pos, target, f=f, f0=1.0) for pos, target in zip(camera_pos, targets)]
    K_gt, R_gt, t_gt = get_camera_parames(cameras)
    X_gt = set_points()
    if show:
        show_3d_scene_data(X_gt, R_gt, t_gt)

    x_list = calc_projected_points(X_gt, K_gt, R_gt, t_gt)
    for x in x_list:
        x += 0.005 * np.random.randn(*x.shape)

    X_, R_, t_, K_ = perspective_self_calibration(x_list, 1.0, tol=1e-2, method="dual")
    if show:
        show_3d_scene_data(X_, R_, t_)
        show_2d_projection_data(x_list, calc_projected_points(X_, K_, R_, t_), n_col=5)

    print("Bundle Adjustment")
    bundle_adjuster = BundleAdjuster(np.stack(x_list).transpose(1, 0, 2), X_, K_, R_, t_, axis="x-up_z-forward")
    X_, K_, R_, t_ = bundle_adjuster.optimize(2.0, 1e-8, max_iter=100, is_debug=True)
    data = bundle_adjuster.get_log()
    if show:
        show_3d_scene_data(X_, R_, t_)
        show_2d_projection_data(x_list, calc_projected_points(X_, K_, R_, t_), n_col=5)
        animate(data)
    return x_list, (X_, K_, R_, t_), data


if __name__ == "__main__":
    main(show="--no-show" not in sys.argv)
