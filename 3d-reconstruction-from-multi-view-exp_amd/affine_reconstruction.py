#!/usr/bin/env python3
"""Affine reconstruction demo on the MI355X backend: synthetic scene -> paraperspective
self-calibration (GPU factorization SVD) -> bundle adjustment from t = -3 r3, K = I.  Same call
sequence, seed and constants as the reference's affine_reconstruction.py:14-65."""
import sys

import numpy as np

from lib.affine_camera_calibration import (  # noqa: F401  (all three are part of the surface)
    orthographic_self_calibration,
    paraperspective_self_calibration,
    symmetric_affine_self_calibration,
)
from lib.bundle_adjustment import BundleAdjuster
from lib.camera import Camera, calc_projected_points, get_camera_parames
from lib.utils import sample_hemisphere_points, set_points
from lib.visualization import show_2d_projection_data, show_3d_scene_data


def main(show=True):
    np.random.seed(123)
    f, n_images = 1.0, 12
    camera_pos = sample_hemisphere_points(n_images, 5)
    targets = np.random.normal(0, 0.5, (n_images, 3))
    cameras = [Camera.create(pos, target, f) for pos, target in zip(camera_pos, targets)]
    K_gt, R_gt, t_gt = get_camera_parames(cameras)
    X_gt = set_points()
    if show:
        show_3d_scene_data(X_gt, R_gt, t_gt)

    x_list = calc_projected_points(X_gt, K_gt, R_gt, t_gt)
    for x in x_list:
        x += 0.005 * np.random.randn(*x.shape)

    X_, R_ = paraperspective_self_calibration(x_list, f * np.ones(n_images))
    t_ = -3 * R_[:, :, 2]
    K_ = np.broadcast_to(np.eye(3), R_.shape)
    if show:
        show_3d_scene_data(X_, R_, t_)
        show_2d_projection_data(x_list, calc_projected_points(X_, K_, R_, t_), n_col=6)

    print("Bundle Adjustment")
    bundle_adjuster = BundleAdjuster(np.stack(x_list).transpose(1, 0, 2), X_, K_, R_, t_, axis="x-up_z-forward")
    X_, K_, R_, t_ = bundle_adjuster.optimize(2.0, 1e-8, max_iter=100, is_debug=True)
    if show:
        show_3d_scene_data(X_, R_, t_)
        show_2d_projection_data(x_list, calc_projected_points(X_, K_, R_, t_), n_col=6)
    return x_list, (X_, K_, R_, t_), bundle_adjuster.get_log()


if __name__ == "__main__":
    main(show="--no-show" not in sys.argv)
