"""Parity of the tall-skinny SVD kernels (mvsvd_factorize via lib.factorization.factorization_method)
with the reference's factorization_method outputs (tests/golden/factorization_24x2000.npz,
euclid_default.npz: fact_W/M/S captured from the reference).  Singular vectors are defined up to
sign, so sign-invariant quantities are compared: sigma, the product M @ S and |M^T M_ref|."""
import os

import numpy as np
import pytest

from lib import _mvba
from lib.factorization import factorization_method

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _check(M, S, M_ref, S_ref, sig, sig_ref, rtol_sigma, atol_prod):
    r = M.shape[1]
    np.testing.assert_allclose(sig[:r], sig_ref[:r], rtol=rtol_sigma)
    # M has orthonormal columns spanning the same subspace as the reference's
    np.testing.assert_allclose(M.T.astype(np.float64) @ M.astype(np.float64), np.eye(r), atol=50 * atol_prod)
    np.testing.assert_allclose(np.abs(M.T.astype(np.float64) @ M_ref.astype(np.float64)), np.eye(r), atol=50 * atol_prod)
    # S = diag(sigma) Vt[:r]  <=>  M @ S is the rank-r part of W, sign free
    scale = np.abs(M_ref @ S_ref).max()
    np.testing.assert_allclose(M.astype(np.float64) @ S.astype(np.float64),
                               M_ref.astype(np.float64) @ S_ref.astype(np.float64), rtol=0, atol=atol_prod * scale)


def test_factorization_vs_reference_f64(golden):
    d = golden("factorization_24x2000")
    W = d["Wt"].T  # (24, 2000) transposed view, as the callers pass it
    M, S = factorization_method(W, n_rank=3)
    assert M.shape == (24, 3) and S.shape == (3, 2000) and M.dtype == np.float64
    _, sig, _, _, _ = _mvba.svd_factorize(d["Wt"], 3)
    _check(M, S, d["M_f64"], d["S_f64"], sig, d["sigma_f64"], 1e-11, 1e-11)
    # default n_rank = 4 (ref factorization.py:6)
    M4, S4 = factorization_method(W)
    assert M4.shape == (24, 4) and S4.shape == (4, 2000)
    _check(M4, S4, d["M4_f64"], d["S4_f64"], sig, d["sigma_f64"], 1e-9, 1e-9)
    # every singular value, not only the leading ones (noise floor sigma ~ 1e-3 * sqrt(N))
    np.testing.assert_allclose(sig, d["sigma_f64"], rtol=1e-8)


def test_factorization_vs_reference_f32(golden):
    d = golden("factorization_24x2000")
    Wt32 = d["Wt"].astype(np.float32)
    M, S = factorization_method(Wt32.T, n_rank=3)
    assert M.dtype == np.float32 and S.dtype == np.float32  # output dtype follows the input (ref quirk B.10)
    _, sig, _, _, _ = _mvba.svd_factorize(Wt32, 3)
    # judged against the fp64 truth at the tolerance LAPACK fp32 itself achieves
    _check(M, S, d["M_f64"], d["S_f64"], sig.astype(np.float64), d["sigma_f64"], 1e-6, 1e-5)
    np.testing.assert_allclose(sig[:3], d["sigma_f32"][:3], rtol=1e-5)


def test_perspective_measurement_matrix_rank4(golden):
    """W (30 x 200) exactly as perspective_self_calibration hands it over (ref :533)."""
    d = golden("euclid_default")
    W = d["fact_W"]
    M, S = factorization_method(np.ascontiguousarray(W.T).T)
    _, sig, _, _, _ = _mvba.svd_factorize(np.ascontiguousarray(W.T), 4)
    _check(M, S, d["fact_M"], d["fact_S"], sig, d["fact_sigma"], 1e-10, 1e-10)
    np.testing.assert_allclose(sig, d["fact_sigma"], rtol=1e-6, atol=1e-12 * d["fact_sigma"][0])


@pytest.mark.parametrize("dtype,n_cols", [(np.float32, 24), (np.float64, 24), (np.float32, 200)])
def test_large_tall_skinny_properties(dtype, n_cols):
    """BASELINE config 5 shape (rows x 24, fp32) at 1M rows + the optional 200-column variant:
    size-independent properties against an fp64 LAPACK SVD of the same data."""
    rng = np.random.default_rng(0)
    n_rows = 1_000_000 if n_cols == 24 else 200_000
    A = rng.normal(size=(n_rows, 3))
    B = rng.normal(size=(3, n_cols))
    Wt = (A @ B + 1e-3 * rng.normal(size=(n_rows, n_cols))).astype(dtype)
    M, sig, S, mu, tm = _mvba.svd_factorize(Wt, 3)
    sig_ref = np.linalg.svd(Wt.astype(np.float64), compute_uv=False)
    tol = 1e-6 if dtype == np.float32 else 1e-11
    np.testing.assert_allclose(sig[:3].astype(np.float64), sig_ref[:3], rtol=tol)
    M64, S64 = M.astype(np.float64), S.astype(np.float64)
    np.testing.assert_allclose(M64.T @ M64, np.eye(3), atol=1e-5 if dtype == np.float32 else 1e-12)
    # S = M^T W exactly (that is how diag(sigma) Vt[:r] is produced), rows of S orthogonal with norms sigma
    np.testing.assert_allclose(S64, M64.T @ Wt.astype(np.float64).T, rtol=0,
                               atol=(1e-4 if dtype == np.float32 else 1e-10) * np.abs(S64).max())
    np.testing.assert_allclose(np.linalg.norm(S64, axis=1), sig_ref[:3], rtol=10 * tol)
    # rank-3 reconstruction error = the discarded singular values
    resid = Wt[:20000].astype(np.float64) - (M64 @ S64[:, :20000]).T
    assert np.abs(resid).max() < 1e-2
    assert tm["sweeps"] < 30
    # centring (affine callers, ref affine_camera_calibration.py:224-240)
    Mc, sigc, Sc, muc, _ = _mvba.svd_factorize(Wt, 3, center=True)
    W64 = Wt.astype(np.float64)
    np.testing.assert_allclose(muc.astype(np.float64), W64.mean(axis=0), rtol=0, atol=1e-5 if dtype == np.float32 else 1e-12)
    sig_c_ref = np.linalg.svd(W64 - W64.mean(axis=0), compute_uv=False)
    np.testing.assert_allclose(sigc[:3].astype(np.float64), sig_c_ref[:3], rtol=tol)


def test_config5_full_size_5m_rows_fp32():
    """BASELINE config 5 at its stated size: 5,000,000 x 24 fp32, rank 3 (ref factorization.py:10-13;
    centring as affine_camera_calibration.py:224-240).  Same property set as the 1M-row test, judged
    against an fp64 LAPACK SVD of the same data."""
    import os
    import time

    n_rows, n_cols = 5_000_000, 24
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n_rows, 3), dtype=np.float32)
    B = rng.standard_normal((3, n_cols), dtype=np.float32)
    Wt = A @ B + np.float32(1e-3) * rng.standard_normal((n_rows, n_cols), dtype=np.float32)
    del A
    t0 = time.perf_counter()
    M, sig, S, mu, tm = _mvba.svd_factorize(Wt, 3)
    wall = time.perf_counter() - t0
    W64 = Wt.astype(np.float64)
    sig_ref = np.linalg.svd(W64, compute_uv=False)
    np.testing.assert_allclose(sig[:3].astype(np.float64), sig_ref[:3], rtol=1e-6)
    M64, S64 = M.astype(np.float64), S.astype(np.float64)
    np.testing.assert_allclose(M64.T @ M64, np.eye(3), atol=1e-5)
    np.testing.assert_allclose(S64, M64.T @ W64.T, rtol=0, atol=1e-4 * np.abs(S64).max())
    np.testing.assert_allclose(np.linalg.norm(S64, axis=1), sig_ref[:3], rtol=1e-5)
    assert np.abs(W64[:20000] - (M64 @ S64[:, :20000]).T).max() < 1e-2
    Mc, sigc, Sc, muc, _ = _mvba.svd_factorize(Wt, 3, center=True)
    np.testing.assert_allclose(muc.astype(np.float64), W64.mean(axis=0), rtol=0, atol=1e-5)
    sig_c_ref = np.linalg.svd(W64 - W64.mean(axis=0), compute_uv=False)
    np.testing.assert_allclose(sigc[:3].astype(np.float64), sig_c_ref[:3], rtol=1e-6)
    line = (f"config 5 (5,000,000 x 24 fp32, rank 3): device {tm['gram_ms'] + tm['jacobi_ms'] + tm['project_ms']:.3f} ms "
            f"(gram {tm['gram_ms']:.3f}, jacobi {tm['jacobi_ms']:.3f}, project {tm['project_ms']:.3f}), H2D {tm['h2d_ms']:.1f} ms, "
            f"call wall {wall * 1e3:.1f} ms")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "fullsize_timings.txt"), "a") as fh:
            fh.write(line + "\n")
    print(line)


def _lapack_rank_r(Wt, r, center=False):
    W64 = Wt.astype(np.float64)
    if center:
        W64 = W64 - W64.mean(axis=0)
    U, s, Vt = np.linalg.svd(W64.T, full_matrices=False)
    return s, (U[:, :r] * s[:r]) @ Vt[:r]


@pytest.mark.parametrize("n_rows,n_cols,dtype,center", [(5000, 300, np.float64, False), (5000, 1000, np.float64, False), (4000, 3000, np.float64, True),
                                                        (5000, 1000, np.float32, False), (3000, 12288, np.float64, False)])
def test_wide_matrices_vs_lapack(n_rows, n_cols, dtype, center):
    """Many columns (more than 21 / 32 images in W = 3m x N / 2m x N; ref lib/factorization.py:10 takes any shape): the
    n x n eigenproblem of the Gram route runs in ONE workgroup (0.4 s at 256 columns, 3.3 s at 512) and its projection did not launch
    at all from ~900 columns on (round 5: tools/time_svd_wide.py).  Beyond 64 columns the leading triplets come from block power iteration with Rayleigh-Ritz
    on W^T W applied implicitly (csrc/mvsvd.hip, "wide matrices"): sigma, M @ S and M^T M against LAPACK on the same data, up to
    the engine's camera limit (12288 = 3 x 4096 columns)."""
    rng = np.random.default_rng(n_cols)
    Wt = (rng.standard_normal((n_rows, 4)) @ rng.standard_normal((4, n_cols)) + 1e-3 * rng.standard_normal((n_rows, n_cols)) + (40.0 if center else 0.0)).astype(dtype)
    M, sig, S, mu, tm = _mvba.svd_factorize(Wt, 4, center=center)
    s_ref, P_ref = _lapack_rank_r(Wt, 4, center)
    f32 = dtype == np.float32
    np.testing.assert_allclose(sig[:4].astype(np.float64), s_ref[:4], rtol=2e-6 if f32 else 1e-12)
    P = M.astype(np.float64) @ S.astype(np.float64)
    assert np.abs(P - P_ref).max() < (1e-5 if f32 else 1e-12) * s_ref[0]
    np.testing.assert_allclose(M.astype(np.float64).T @ M.astype(np.float64), np.eye(4), atol=2e-6 if f32 else 1e-12)
    assert M.shape == (n_cols, 4) and S.shape == (4, n_rows) and sig.shape == (n_cols,)
    # the block carries 32 Ritz values: the rest of sigma is not computed
    assert np.all(np.isfinite(sig[:32])) and np.all(np.isnan(sig[32:]))
    assert 1 <= tm["sweeps"] <= 12  # iterations of the block method (a measurement matrix converges in a handful)
    if center:
        np.testing.assert_allclose(mu, Wt.astype(np.float64).mean(axis=0), rtol=0, atol=1e-11)


def test_wide_matrices_hard_spectra():
    """The block iteration where it has to work for its result: a graded spectrum whose fourth singular value is 1e-7 of the first
    (the products are formed from W, not from an accumulated Gram matrix, and the basis is kept graded -- Ritz order, triangular
    orthogonalisation, the ROTATED product B Y fed to W^T --, so sigma_4 keeps ~eps sigma_1 / sigma_4 relative accuracy; with a
    symmetric orthogonalisation and a rotated Z the first version lost that direction to 2e-3);
    a slowly decaying spectrum; pure Gaussian noise (no gap: ~100 iterations); exact low rank in integers (columns of the block are
    exactly zero and are refilled); fewer rows than the block is wide; n_rank up to 16."""
    rng = np.random.default_rng(7)

    def check(Wt, r, tol_sigma, tol_prod):
        M, sig, S, _mu, tm = _mvba.svd_factorize(Wt, r)
        s_ref, P_ref = _lapack_rank_r(Wt, r)
        np.testing.assert_allclose(sig[:r], s_ref[:r], rtol=tol_sigma, atol=1e-14 * s_ref[0])
        assert np.abs(M @ S - P_ref).max() < tol_prod * s_ref[0]
        np.testing.assert_allclose(M.T @ M, np.eye(r), atol=1e-12)
        return tm

    check(_graded(20_000, 400, [1.0, 0.5, 1e-3, 1e-7], seed=3, noise=1e-13), 4, 1e-8, 1e-13)
    tm = check(_graded(20_000, 400, [1.0, 0.5, 1e-3, 1e-7], seed=3), 4, 1e-8, 1e-13)  # (LAPACK's own sigma_4 is good to eps sigma_1 / sigma_4 = 2e-9)
    assert tm["sweeps"] <= 4
    check(_graded(2000, 300, 0.9 ** np.arange(300), seed=4), 4, 1e-12, 1e-12)
    tm = check(rng.standard_normal((2000, 300)), 3, 1e-11, 1e-9)
    assert 20 < tm["sweeps"] < 1000
    check((rng.integers(-3, 4, (1000, 3)) @ rng.integers(-3, 4, (3, 300))).astype(np.float64), 3, 1e-12, 1e-12)
    check(rng.standard_normal((10, 300)), 3, 1e-12, 1e-12)
    check(rng.standard_normal((3000, 16)) @ rng.standard_normal((16, 500)) + 1e-4 * rng.standard_normal((3000, 500)), 16, 1e-12, 1e-12)


def test_wide_matrices_degenerate_inputs():
    """An all-zero matrix (every singular value 0, any orthonormal basis is a valid answer) and a matrix with a NaN in it (LAPACK raises
    LinAlgError("SVD did not converge"); so does the block iteration, at its first residual)."""
    M, sig, S, _mu, _tm = _mvba.svd_factorize(np.zeros((500, 300)), 3)
    assert np.all(sig[:32] == 0.0) and np.all(S == 0.0)
    np.testing.assert_allclose(M.T @ M, np.eye(3), atol=1e-12)
    W = np.random.default_rng(2).standard_normal((500, 300))
    W[17, 123] = np.nan
    with pytest.raises(np.linalg.LinAlgError, match="did not converge"):
        _mvba.svd_factorize(W, 3)
    with pytest.raises(np.linalg.LinAlgError, match="did not converge"):  # ... and the Gram + Jacobi route (it returned sigma = 0 until round 5)
        _mvba.svd_factorize(np.ascontiguousarray(W[:, 100:124]), 3)


def test_wide_matrix_limits_and_the_python_surface():
    """n_rank above half the block width and more than 12288 columns are refused with the reason (ValueError); the reference's
    surface, factorization_method(W, r) with W = (2m x N) for 200 images, goes through the same path."""
    from lib.factorization import factorization_method

    rng = np.random.default_rng(11)
    Wt = rng.standard_normal((500, 3)) @ rng.standard_normal((3, 400)) + 1e-4 * rng.standard_normal((500, 400))
    with pytest.raises(ValueError, match="n_rank <= 16"):
        _mvba.svd_factorize(Wt, 17)
    with pytest.raises(ValueError, match="12288"):
        _mvba.SvdWorkspace(10, 12289, np.float64)
    M, S = factorization_method(Wt.T, 3)
    s_ref, P_ref = _lapack_rank_r(Wt, 3)
    assert M.shape == (400, 3) and S.shape == (3, 500)
    assert np.abs(M @ S - P_ref).max() < 1e-12 * s_ref[0]
    # up to 256 columns more than 16 triplets are still served -- by the Gram + Jacobi route (every singular value, 0.1-0.4 s)
    W100 = rng.standard_normal((800, 100)) * np.logspace(0, -2, 100)[None, :]
    M20, sig20, S20, _mu, _tm = _mvba.svd_factorize(W100, 20)
    s_ref, P_ref = _lapack_rank_r(W100, 20)
    np.testing.assert_allclose(sig20, s_ref, rtol=1e-10)  # all 100 of them
    assert np.abs(M20 @ S20 - P_ref).max() < 1e-11 * s_ref[0]
    # ... and from 65 columns on, up to 16 triplets come from the block iteration (sigma: 32 Ritz values, the rest NaN)
    M4, sig4, S4, _mu, tm4 = _mvba.svd_factorize(W100, 4)
    s_ref, P_ref = _lapack_rank_r(W100, 4)
    np.testing.assert_allclose(sig4[:4], s_ref[:4], rtol=1e-12)
    assert np.abs(M4 @ S4 - P_ref).max() < 1e-11 * s_ref[0] and np.all(np.isnan(sig4[32:]))


def _graded(n_rows, n_cols, sigmas, seed=0, noise=0.0):
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.normal(size=(n_rows, len(sigmas))))
    V, _ = np.linalg.qr(rng.normal(size=(n_cols, len(sigmas))))
    W = (U * np.asarray(sigmas)) @ V.T
    if noise:
        W = W + noise * rng.normal(size=W.shape)
    return W


def test_fp64_graded_spectrum_small_singular_values_like_gesdd():
    """fp64 data with sigma_4 / sigma_1 = 1e-7 (the default n_rank = 4 path of
    perspective_self_calibration USES the 4th triplet, ref :533): one Gram pass would leave sigma_4
    good to ~1e-2 only; the preconditioned second pass must give it to ~eps * sigma_1 like LAPACK."""
    sig = [1.0, 0.3, 1e-3, 1e-7]
    Wt = _graded(200_000, 12, sig, noise=1e-13)
    M, s, S, _mu, tm = _mvba.svd_factorize(Wt, 4)
    U_ref, s_ref, Vt_ref = np.linalg.svd(Wt, full_matrices=False)
    np.testing.assert_allclose(s[:4], s_ref[:4], rtol=1e-7)
    assert abs(s[3] / s_ref[3] - 1.0) < 1e-7 and tm["refine_ms"] > 0
    # 4th left singular vector of W (= right singular vector of Wt) up to sign
    v4 = Vt_ref[3]
    assert min(np.abs(M[:, 3] - v4).max(), np.abs(M[:, 3] + v4).max()) < 1e-6
    np.testing.assert_allclose(M.T @ M, np.eye(4), atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(S, axis=1), s_ref[:4], rtol=1e-7)
    np.testing.assert_allclose(S, M.T @ Wt.T, rtol=0, atol=1e-13)


def test_fp64_centring_with_a_large_mean():
    """|mean| = 1e6 x spread: G - s s^T / N would cancel 12 digits; the rows are centred before
    they enter the Gram product (affine callers, ref affine_camera_calibration.py:224-240)."""
    rng = np.random.default_rng(3)
    Wt = _graded(100_000, 8, [50.0, 20.0, 5.0], seed=4, noise=1e-3) + 1e6 * (1.0 + rng.uniform(size=8))
    M, s, S, mu, _ = _mvba.svd_factorize(Wt, 3, center=True)
    Wc = Wt - Wt.mean(axis=0)
    s_ref = np.linalg.svd(Wc, compute_uv=False)
    np.testing.assert_allclose(mu, Wt.mean(axis=0), rtol=1e-13)
    np.testing.assert_allclose(s[:3], s_ref[:3], rtol=1e-9)
    # S = M^T (W - mean) with the means the call returned (a 2e-14 relative difference between two
    # fp64 means of 1e6-sized numbers is 3e-8 absolute: bigger than the tolerance on S itself)
    np.testing.assert_allclose(S, M.T @ (Wt - mu).T, rtol=0, atol=1e-9 * np.abs(S).max())


@pytest.mark.parametrize("n_cols", [1, 2, 3, 4, 5, 7, 16, 17, 23, 31, 32, 33, 48])
def test_every_order_of_the_small_eigen_solver(n_cols):
    """k_jacobi_small (up to 32 columns: fixed pairs with the data moving between two buffers, an odd order padded by a
    phantom index that is never rotated, one pair only at 2 columns) and the first orders past it: every singular
    value, the left basis orthonormal and M S = the rank-r truncation, against LAPACK in float64."""
    rng = np.random.default_rng(100 + n_cols)
    rows = 3000
    r = min(3, n_cols)
    Wt = rng.standard_normal((rows, n_cols)) * np.logspace(0, -3, n_cols)[None, :] + rng.standard_normal((rows, 1)) @ rng.standard_normal((1, n_cols))
    M, sig, S, _mu, tm = _mvba.svd_factorize(np.ascontiguousarray(Wt), r)
    U, s_ref, Vt = np.linalg.svd(Wt.T, full_matrices=False)  # W = Wt^T is (n_cols, rows)
    np.testing.assert_allclose(sig, s_ref, rtol=1e-9, atol=1e-12 * s_ref[0])
    np.testing.assert_allclose(M.T @ M, np.eye(r), atol=1e-12)
    np.testing.assert_allclose(M @ S, (U[:, :r] * s_ref[:r]) @ Vt[:r], atol=1e-8 * s_ref[0])
    assert tm["sweeps"] < 30


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_any_rank_and_workspace_reuse(dtype):
    """n_rank beyond 4 (the reference takes any rank, ref factorization.py:6,12-13) and the resident
    workspace: one load, several runs, identical to the one-shot call."""
    Wt = _graded(50_000, 30, [9, 8, 7, 6, 5, 4, 3], seed=7, noise=1e-4).astype(dtype)
    M, s, S, _mu, _ = _mvba.svd_factorize(Wt, 7)
    s_ref = np.linalg.svd(Wt.astype(np.float64), compute_uv=False)
    tol = 1e-5 if dtype == np.float32 else 1e-10
    np.testing.assert_allclose(s[:7].astype(np.float64), s_ref[:7], rtol=tol)
    M64, S64 = M.astype(np.float64), S.astype(np.float64)
    np.testing.assert_allclose(M64.T @ M64, np.eye(7), atol=10 * tol)
    np.testing.assert_allclose(S64, M64.T @ Wt.astype(np.float64).T, rtol=0, atol=10 * tol * np.abs(S64).max())
    Mf, sf, Sf, _, _ = _mvba.svd_factorize(Wt, 30)  # full rank: every column of the basis
    np.testing.assert_allclose(Mf.astype(np.float64).T @ Mf.astype(np.float64), np.eye(30), atol=10 * tol)
    ws = _mvba.SvdWorkspace(60_000, 30, dtype)
    ws.load(Wt)
    M2, s2, S2, _, _ = ws.run(7)
    Mc, sc_, Sc, muc, _ = ws.run(3, center=True)  # same resident matrix, no second upload
    # same numbers as the one-shot call up to the Gram partial sums' atomic order (a few ulp)
    np.testing.assert_allclose(M2, M, rtol=0, atol=(1e-6 if dtype == np.float32 else 1e-12))
    np.testing.assert_allclose(S2, S, rtol=0, atol=(1e-5 if dtype == np.float32 else 1e-11) * np.abs(S64).max())
    np.testing.assert_allclose(muc.astype(np.float64), Wt.astype(np.float64).mean(axis=0), atol=10 * tol)
    ws.load(Wt[:1000])  # a smaller matrix in the same workspace
    M3, s3, S3, _, _ = ws.run(3)
    assert S3.shape == (3, 1000)
    np.testing.assert_allclose(s3[:3].astype(np.float64), np.linalg.svd(Wt[:1000].astype(np.float64), compute_uv=False)[:3], rtol=tol)
    with pytest.raises(ValueError):
        ws.load(np.zeros((70_000, 30), dtype))
    ws.close()


def test_bad_arguments():
    with pytest.raises(ValueError):
        _mvba.svd_factorize(np.zeros((10, 4)), 5)


@pytest.mark.parametrize("dtype,norm,m", [(np.float64, 1, 8), (np.float64, 2, 8), (np.float32, 1, 8), (np.float64, 0, 8),
                                          (np.float64, 1, 30), (np.float32, 2, 30), (np.float32, 1, 70), (np.float64, 0, 70)])
def test_depth_weighted_factorisation_from_a_resident_base(dtype, norm, m):
    """mvsvd_load_base + mvsvd_run_scaled (the projective-depth loops, ref perspective_camera_calibration.py
    :81-87 and :170-179: W = x o z, rows to unit length / image blocks by their squared norm, SVD, 50-200 times):
    the base matrix is uploaded once, every call uploads the depths only and forms W on the device.  Equal to
    loading the host-formed W: the same kernels run on the same numbers (the scaling itself rounds once in the
    matrix dtype on both sides, in a different order of operations: a few ulp).  30 and 70 images: rows too long for the LDS
    tiles (k_scale_rows_wide: 32 / 64 lanes per row, and more groups than lanes) and the block iteration behind them, which starts
    the SECOND call on the same base from the vectors the first one ended with."""
    rng = np.random.default_rng(11)
    n = 40_000 if m == 8 else 6_000
    z = 1.0 + 0.2 * rng.uniform(size=(n, m))
    if m == 8:
        x = np.concatenate([rng.normal(size=(n, m, 2)), np.ones((n, m, 1))], axis=2)
    else:  # x o z of rank 4 (like a measurement matrix under its true depths): the leading vectors are well separated from the rest
        x = (rng.normal(size=(n, 4)) @ rng.normal(size=(4, 3 * m))).reshape(n, m, 3) / z[..., None] + 1e-4 * rng.normal(size=(n, m, 3))
    W = x * z[..., None]
    if norm == 1:
        W = W / np.linalg.norm(W, axis=(1, 2))[:, None, None]
    elif norm == 2:
        W = W / (W**2).sum(axis=(0, 2))[None, :, None]
    Wt = np.ascontiguousarray(W.reshape(n, 3 * m)).astype(dtype)
    ws = _mvba.SvdWorkspace(n, 3 * m, dtype)
    M0, s0, S0, _mu, _ = ws.load(Wt).run(4)
    ws.load_base(x.reshape(n, 3 * m).astype(dtype))
    M1, s1, S1, tm = ws.run_scaled(z.astype(dtype), 3, norm, 4)
    tol = 2e-5 if dtype == np.float32 else 1e-11
    np.testing.assert_allclose(s1[:4].astype(np.float64), s0[:4].astype(np.float64), rtol=tol)
    sg = np.sign(np.sum(M0.astype(np.float64) * M1.astype(np.float64), axis=0))
    np.testing.assert_allclose(M1 * sg, M0, rtol=0, atol=10 * tol)
    np.testing.assert_allclose(S1 * sg[:, None], S0, rtol=0, atol=10 * tol * np.abs(S0).max())
    if m > 8:  # the warm start: other depths on the same base, against the host-formed matrix loaded afresh
        z2 = z * (1.0 + 0.01 * rng.uniform(size=z.shape))
        W2 = x * z2[..., None]
        if norm == 1:
            W2 = W2 / np.linalg.norm(W2, axis=(1, 2))[:, None, None]
        elif norm == 2:
            W2 = W2 / (W2**2).sum(axis=(0, 2))[None, :, None]
        M2, s2, S2, tm2 = ws.run_scaled(z2.astype(dtype), 3, norm, 4)
        assert tm2["sweeps"] <= tm["sweeps"]
        ws2 = _mvba.SvdWorkspace(n, 3 * m, dtype)
        M3, s3, S3, _mu, _ = ws2.load(np.ascontiguousarray(W2.reshape(n, 3 * m)).astype(dtype)).run(4)
        np.testing.assert_allclose(s2[:4].astype(np.float64), s3[:4].astype(np.float64), rtol=tol)
        sg = np.sign(np.sum(M3.astype(np.float64) * M2.astype(np.float64), axis=0))
        np.testing.assert_allclose(M2 * sg, M3, rtol=0, atol=10 * tol)
        ws2.close()
    # and against LAPACK on the host-formed matrix
    s_ref = np.linalg.svd(Wt.astype(np.float64), compute_uv=False)
    np.testing.assert_allclose(s1[:4].astype(np.float64), s_ref[:4], rtol=1e-5 if dtype == np.float32 else 1e-9)
    # a second call with other depths on the same base: nothing but z is uploaded again
    z2 = z * (1.0 + 0.1 * rng.uniform(size=z.shape))
    M2, s2, S2, _ = ws.run_scaled(z2.astype(dtype), 3, norm, 4)
    assert not np.allclose(s2[:4], s1[:4], rtol=1e-6) or norm == 1
    with pytest.raises(ValueError):
        ws.run_scaled(z[:, :3], 3, norm, 4)
    ws.close()


def test_depth_loop_uploads_the_depths_only_at_5m_rows():
    """What mvsvd_run_scaled is for, at config 5's row count: per depth iteration 5M x 8 depths (160 MB fp32)
    cross PCIe instead of the 5M x 24 matrix (480 MB)."""
    rng = np.random.default_rng(0)
    n, m = 5_000_000, 8
    x = rng.standard_normal((n, 3 * m), dtype=np.float32)
    z = (1.0 + 0.1 * rng.random((n, m), dtype=np.float32)).astype(np.float32)
    ws = _mvba.SvdWorkspace(n, 3 * m, np.float32)
    ws.load_base(x)
    M, s, S, tm = ws.run_scaled(z, 3, 1, 4)
    M, s, S, tm = ws.run_scaled(z, 3, 1, 4)
    W = (x.reshape(n, m, 3) * z[..., None]).reshape(n, 3 * m)
    W /= np.linalg.norm(W, axis=1, keepdims=True)
    _, _, _, _, tm_full = ws.load(W).run(4)
    ws.close()
    # rows of unit length: sum of sigma^2 = n
    assert abs(float((s.astype(np.float64) ** 2).sum()) / n - 1.0) < 1e-4
    # (the upload times -- 3.0 ms of z against 8.7 ms of W -- are a measurement: tools/time_svd_scaled.py)
    assert tm["h2d_ms"] > 0 and tm_full["h2d_ms"] > 0


@pytest.mark.parametrize("method", [1, 2])
def test_depth_iteration_on_the_device_equals_the_oracle_per_step(method):
    """mvsvd_depth_step (one whole iteration of the reference's depth loops on the device: re-weighting, rank-4
    factorisation, the per-point 4 x 4 / per-image 12 x 12 companion eigenproblems, sign rules, depth update,
    reprojection error; ref perspective_camera_calibration.py:79-129, :166-224) against oracle/depth_oracle.py on a
    3000-point x 6-image scene: error and depths after each of four iterations, 1e-9."""
    from lib.perspective_camera_calibration import _create_data_matrix
    from lib.synthetic import make_scene
    from oracle.depth_oracle import HostDepthLoop

    sc = make_scene(3000, 6, vis_p=1.0)
    xd, _vis = sc.dense()
    x = _create_data_matrix([xd[:, k, :] for k in range(6)], 1.0)
    g = HostDepthLoop(x)
    ws = _mvba.SvdWorkspace(3000, 18, np.float64)
    ws.load_base(x.reshape(3000, 18))
    ws.depth_begin(3)
    for _ in range(4):
        E, tm = ws.depth_step(method, 1.0)
        Eo = g.step(method, 1.0)
        assert E == pytest.approx(Eo, rel=1e-9, abs=1e-14)
        np.testing.assert_allclose(ws.depth_read(), g.depths(), rtol=0, atol=1e-9)
        assert tm["depth_ms"] > 0
    with pytest.raises(ValueError):
        ws.depth_step(3, 1.0)
    # a caller's own depths (mvsvd_run_scaled) end the device loop: depth_step then asks for depth_begin again
    ws.run_scaled(np.ones((3000, 6)), 3, 1, 4)
    with pytest.raises(RuntimeError):
        ws.depth_step(method, 1.0)
    ws.close()


@pytest.mark.parametrize("n_points,m,method", [(1500, 3, 1), (1500, 3, 2), (1500, 2, 1), (1200, 16, 1), (1200, 16, 2), (800, 64, 1), (800, 64, 2),
                                                 (600, 100, 1), (600, 100, 2), (500, 128, 2), (1000, 30, 1), (1000, 30, 2), (400, 300, 1), (400, 300, 2),
                                                 (300, 768, 1)])
def test_depth_iteration_image_counts_vs_oracle(n_points, m, method):
    """Image counts at the edges of the device depth loop's kernel variants, against oracle/depth_oracle.py per step (1e-9): two and
    three views (the reference's loops only need 3 m >= 4 columns; the device loop refused fewer than four until round 5), 16 images
    in fp64 (the rows no longer fit the LDS tiles: the per-lane variants of k_depth_primary / k_dual_apply, which no test ran before),
    64 images (k_dual_gram stages fewer than 128 rows per pass there -- its launch used to fail from 48 images on -- and walks its
    14 m = 896 tasks in more than one batch per thread), 100 and 128 images (300 / 384 columns: the factorisation inside the step is
    the wide path's block iteration, `test_wide_matrices_vs_lapack`), 30 / 300 / 768 images (the depth updates with the lanes of a wave
    ACROSS a point's images -- 32 and 64 lanes per point, more images than lanes -- and the cap of the device loop: 768 images)."""
    from lib.perspective_camera_calibration import _create_data_matrix
    from lib.synthetic import make_scene
    from oracle.depth_oracle import HostDepthLoop

    sc = make_scene(n_points, m, vis_p=1.0)
    xd, _vis = sc.dense()
    x = _create_data_matrix([xd[:, k, :] for k in range(m)], 1.0)
    g = HostDepthLoop(x)
    ws = _mvba.SvdWorkspace(n_points, 3 * m, np.float64)
    ws.load_base(x.reshape(n_points, 3 * m))
    ws.depth_begin(3)
    for _ in range(3):
        E, _tm = ws.depth_step(method, 1.0)
        Eo = g.step(method, 1.0)
        assert E == pytest.approx(Eo, rel=1e-9, abs=1e-14)
        np.testing.assert_allclose(ws.depth_read(), g.depths(), rtol=0, atol=1e-9)
    ws.close()


def test_more_than_256_column_groups():
    """mvsvd_run_scaled with 300 / 900 column groups (the per-group sums of norm 2 were laid out for at most 256 groups per block and
    the call was refused beyond; the device depth loop ran into the same kernel from 257 images on)."""
    rng = np.random.default_rng(5)
    n, cols = 700, 900
    x = rng.normal(size=(n, 4)) @ rng.normal(size=(4, cols)) + 1e-3 * rng.normal(size=(n, cols))
    ws = _mvba.SvdWorkspace(n, cols, np.float64)
    ws.load_base(x)
    for group in (3, 1):
        ng = cols // group
        z = 1.0 + 0.1 * rng.uniform(size=(n, ng))
        W = (x.reshape(n, ng, group) * z[..., None])
        W = W / (W**2).sum(axis=(0, 2))[None, :, None]
        M, s, S, _tm = ws.run_scaled(z, group, 2, 4)
        s_ref, P_ref = _lapack_rank_r(W.reshape(n, cols), 4)
        np.testing.assert_allclose(s[:4], s_ref[:4], rtol=1e-11)
        assert np.abs(M @ S - P_ref).max() < 1e-11 * s_ref[0]
    ws.close()


def test_depth_loop_image_cap():
    ws = _mvba.SvdWorkspace(50, 3 * 769, np.float64)
    ws.load_base(np.ones((50, 3 * 769)))
    with pytest.raises(ValueError, match="768 images"):
        ws.depth_begin(3)
    ws.close()


def test_a_coarser_grouping_after_a_finer_one_fits_the_depth_buffer():
    """mvsvd_run_scaled with one depth per COLUMN after the depth loop (one per image) on the same handle, and back: the depth buffer
    is sized for the finest grouping from its first allocation (it used to keep the size of whichever caller came first)."""
    rng = np.random.default_rng(5)
    n, cols = 4000, 12
    X = rng.standard_normal((n, cols))
    ws = _mvba.SvdWorkspace(n, cols, np.float64)
    ws.load_base(X)
    ws.depth_begin(3)
    ws.depth_step(1, 1.0)
    z1 = rng.uniform(0.5, 1.5, (n, cols))
    M, sig, S, _ = ws.run_scaled(z1, 1, 0, 4)[:4]
    ref = np.linalg.svd((X * z1).T, full_matrices=False)[1]
    np.testing.assert_allclose(sig[:4], ref[:4], rtol=1e-10)
    z6 = rng.uniform(0.5, 1.5, (n, 2))
    sig6 = ws.run_scaled(z6, 6, 0, 4)[1]
    np.testing.assert_allclose(sig6[:4], np.linalg.svd((X * np.repeat(z6, 6, axis=1)).T, full_matrices=False)[1][:4], rtol=1e-10)
    ws.close()


def test_a_failed_regrow_does_not_leave_a_closed_workspace_in_the_cache(monkeypatch):
    """svd_factorize keeps one workspace per (dtype, columns, device) and grows it for a larger matrix: if that allocation fails the
    cache must not keep the CLOSED old workspace (every later, smaller call would then pass a null handle)."""
    rng = np.random.default_rng(2)
    _mvba.svd_cache_clear()
    small = rng.standard_normal((500, 9))
    _mvba.svd_factorize(small, 3)
    real = _mvba.SvdWorkspace

    class Boom(RuntimeError):
        pass

    def failing(*a, **k):
        raise Boom("out of memory (simulated)")

    monkeypatch.setattr(_mvba, "SvdWorkspace", failing)
    with pytest.raises(Boom):
        _mvba.svd_factorize(rng.standard_normal((900, 9)), 3)
    monkeypatch.setattr(_mvba, "SvdWorkspace", real)
    M, sig = _mvba.svd_factorize(small, 3)[:2]
    np.testing.assert_allclose(sig[:3], np.linalg.svd(small.T, full_matrices=False)[1][:3], rtol=1e-10)
    _mvba.svd_cache_clear()


def test_depth_iteration_in_float32():
    """The same iteration on a float32 workspace: depths good to float32."""
    from lib.perspective_camera_calibration import _create_data_matrix
    from lib.synthetic import make_scene
    from oracle.depth_oracle import HostDepthLoop

    sc = make_scene(2000, 5, vis_p=1.0)
    xd, _vis = sc.dense()
    x = _create_data_matrix([xd[:, k, :] for k in range(5)], 1.0)
    g = HostDepthLoop(x)
    ws = _mvba.SvdWorkspace(2000, 15, np.float32)
    ws.load_base(x.reshape(2000, 15).astype(np.float32))
    ws.depth_begin(3)
    for method in (1, 2):
        E, _ = ws.depth_step(method, 1.0)
        assert E == pytest.approx(g.step(method, 1.0), rel=1e-3)
        np.testing.assert_allclose(ws.depth_read(), g.depths(), rtol=0, atol=2e-4)
    ws.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_measurement_matrix_put_together_on_the_device_equals_the_host_hstack(dtype):
    """`svd_factorize_images(x_list, ...)` (`mvsvd_load_images`: the images' (N, 2) arrays as they are) against
    `svd_factorize(np.hstack(x_list), ...)` (ref lib/affine_camera_calibration.py:224-240): the same matrix on the device, so
    bitwise the same factors, centred or not; a non-contiguous image, mixed float32/float64 lists (promoted, as np.hstack does) and
    lists the device path does not take (integers, ragged) give what the host form gives."""
    rng = np.random.default_rng(41)
    n, m = 7001, 9
    x_list = [rng.uniform(-300, 300, (n, 2)).astype(dtype) for _ in range(m)]
    x_list[4] = np.asfortranarray(x_list[4])
    for center in (False, True):
        a = _mvba.svd_factorize(np.ascontiguousarray(np.hstack(x_list)), 3, center=center)
        b = _mvba.svd_factorize_images(x_list, 3, center=center)
        for u, v in zip(a[:4], b[:4]):
            assert u.dtype == v.dtype == dtype and np.array_equal(u, v, equal_nan=True)
    mixed = [x.astype(np.float32) if k % 2 else x.astype(np.float64) for k, x in enumerate(x_list)]
    a, b = _mvba.svd_factorize(np.ascontiguousarray(np.hstack(mixed)), 3), _mvba.svd_factorize_images(mixed, 3)
    assert b[0].dtype == np.float64 and all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a[:3], b[:3]))
    ints = [np.rint(x).astype(np.int64) for x in x_list]
    a, b = _mvba.svd_factorize(np.hstack(ints), 3), _mvba.svd_factorize_images(ints, 3)
    assert all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a[:3], b[:3]))
    with pytest.raises(ValueError):
        _mvba.svd_factorize_images(x_list[:-1] + [x_list[0][:-1]], 3)  # ragged: np.hstack's own error
    ws = _mvba.SvdWorkspace(n, 2 * m, dtype)
    with pytest.raises(ValueError):
        ws.load_images(x_list[:-1])
    ws.close()
    _mvba.svd_cache_clear()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_base_assembled_on_the_device_equals_the_host_data_matrix(dtype):
    """`mvsvd_load_base_images` (the images' (N, 2) arrays as they are; x / f0, y / f0, 1 formed on the device) against
    `mvsvd_load_base` of `_create_data_matrix`'s array (ref :34-40): the same base bit for bit -- an IEEE division either way --
    so the factors and a depth iteration that follow are bitwise the same.  Ragged and mis-sized inputs raise."""
    from lib.perspective_camera_calibration import _create_data_matrix

    rng = np.random.default_rng(31)
    n, m, f0 = 5003, 7, 612.5
    x_list = [rng.uniform(-400, 400, (n, 2)) for _ in range(m)]
    x_list[2] = np.asfortranarray(x_list[2])  # (a non-contiguous image: the binding makes it contiguous)
    x = _create_data_matrix(x_list, f0).reshape(n, 3 * m).astype(dtype)
    z = rng.uniform(0.5, 2.0, (n, m)).astype(dtype)
    a, b = _mvba.SvdWorkspace(n, 3 * m, dtype), _mvba.SvdWorkspace(n, 3 * m, dtype)
    a.load_base(x)
    b.load_base_images(x_list, f0)
    ra, rb = a.run_scaled(z, 3, 2, 4), b.run_scaled(z, 3, 2, 4)
    for u, v in zip(ra[:3], rb[:3]):
        assert np.array_equal(u, v, equal_nan=True)
    a.depth_begin(3), b.depth_begin(3)
    for method in (1, 2):
        assert a.depth_step(method, f0)[0] == b.depth_step(method, f0)[0]
    assert np.array_equal(a.depth_read(), b.depth_read())
    with pytest.raises(ValueError):
        b.load_base_images(x_list[:-1], f0)  # 3 columns per image
    with pytest.raises(ValueError):
        b.load_base_images(x_list[:-1] + [x_list[0][:-1]], f0)  # ragged
    with pytest.raises(ValueError):
        b.load_base_images(x_list, 0.0)
    a.close(), b.close()


@pytest.mark.parametrize("method", [1, 2])
def test_depth_iteration_at_a_million_points_properties(method):
    """The device depth loop at 1,000,000 points x 8 images (fp64) -- too large for the oracle's per-point eigh in a test, so
    size-independent properties: (i) the reprojection error a step returns equals an independent NumPy evaluation (ref :43-58) of
    the factors the GPU SVD gives for the depths the loop held BEFORE that step (`mvsvd_run_scaled` on a second workspace);
    (ii) depths stay positive with unit-length vectors (per point for the primary scheme, per image for the dual one, ref :118 /
    :213); (iii) on noise-free projections the error falls from iteration to iteration; (iv) two runs are
    bitwise identical (no atomics anywhere in the loop)."""
    from oracle.depth_oracle import reprojection_error

    n, m = 1_000_000, 8
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (n, 3))
    x = np.empty((n, m, 3))
    for k in range(m):
        ph = 0.12 * k - 0.4  # cameras on an arc of radius 5 around the points, looking at the origin
        c = 5.0 * np.array([np.sin(ph), 0.0, -np.cos(ph)])
        R = np.array([[np.cos(ph), 0, -np.sin(ph)], [0, 1, 0], [np.sin(ph), 0, np.cos(ph)]])  # columns: right, up, forward
        Xc = (X - c) @ R
        x[:, k, 0], x[:, k, 1], x[:, k, 2] = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2], 1.0
    xf = np.ascontiguousarray(x.reshape(n, 3 * m))
    x_norm = np.linalg.norm(x, axis=2)

    def run(iters):
        ws = _mvba.SvdWorkspace(n, 3 * m, np.float64)
        ws.load_base(xf)
        ws.depth_begin(3)
        Es, zs = [], [np.ones((n, m))]
        for _ in range(iters):
            Es.append(ws.depth_step(method, 1.0)[0])
            zs.append(ws.depth_read())
        ws.close()
        return Es, zs

    Es, zs = run(4)
    chk = _mvba.SvdWorkspace(n, 3 * m, np.float64)
    chk.load_base(xf)
    for i in (0, 3):  # (i) first and last step
        M, _sig, S, _tm = chk.run_scaled(zs[i], 3, method, 4)
        assert Es[i] == pytest.approx(reprojection_error(x, M, S, 1.0), rel=1e-9)
    chk.close()
    xi = zs[-1] * x_norm  # (ii)
    assert (zs[-1] > 0).all()
    np.testing.assert_allclose(np.linalg.norm(xi, axis=1 if method == 1 else 0), 1.0, rtol=0, atol=1e-12)
    assert all(b <= a * (1 + 1e-12) for a, b in zip(Es, Es[1:])) and Es[-1] < 0.97 * Es[0]  # (iii) exact projections: the fit improves step by step (1.7 % per primary step here, 2.4x per dual step)
    Es2, zs2 = run(2)  # (iv)
    assert Es2 == Es[:2]
    np.testing.assert_array_equal(zs2[2], zs[2])
