"""``factorization_method`` with the reference's signature
(lib/factorization.py:5-15) on the MI355X tall-skinny SVD kernels.

The reference runs a FULL ``np.linalg.svd(W)`` of ``W = (3m | 2m) x N`` and keeps
``M = U[:, :r]`` and ``S = diag(sigma[:r]) @ Vt[:r]``; its N x N ``Vt`` is the
reason it cannot reach 5M points.  Here ``W.T`` (the N x n array the callers
actually hold: perspective_camera_calibration.py:533 passes a transposed view)
is streamed through ``mvsvd_factorize`` (csrc/mvsvd.hip): thin, never forms Vt.

Up to 64 rows of ``W`` (3m or 2m), or up to 256 with ``n_rank > 16``: Gram matrix + Jacobi, ``n_rank`` any
1 .. min(W.shape) like the reference's; float64 input gets a second, preconditioned pass so that small
singular values are as accurate as LAPACK's (the default ``n_rank = 4`` path of ``perspective_self_calibration``
uses the 4th triplet).  From 65 rows on (up to 12288 = three per image at the engine's 4096 cameras,
``n_rank <= 16``; ``ValueError`` beyond either): block power iteration with Rayleigh-Ritz on ``W W^T`` applied
implicitly -- two passes over ``W`` per iteration, 3-6 iterations for a measurement matrix, the same accuracy
(the products are formed from ``W`` itself).  Output dtype follows the input (B.10).

Signs: singular vectors are defined up to sign; LAPACK's choice is not a rule.
Ours: the largest-magnitude entry of every column of ``M`` is positive.  The
Euclidean pipeline is invariant to these signs (SURVEY §7 hard part 4).
"""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray


def factorization_method(
    W: NDArray[np.floating], n_rank: int = 4
) -> tuple[NDArray[np.floating], NDArray[np.floating]]:
    from ._mvba import svd_factorize

    W = np.asarray(W)
    # W arrives as a transposed view of an (N, n) row-major array: .T is then free
    Wt = W.T
    M, _sigma, S, _mu, _tm = svd_factorize(Wt, n_rank)
    return M, S
