"""Pin the CPU oracle to vectors captured from the reference (tests/golden/*.npz)."""
import numpy as np
import pytest

from oracle import ba_oracle as O


def _engine_from(d, axis):
    n, m = d["x"].shape[:2]
    vis = d["vis"] if "vis" in d.files else None
    pt_ptr, cam, xy = O.dense_to_observations(d["x"], vis)
    g = O.OracleEngine(n, m, pt_ptr, cam, xy, 1.0, axis)
    X, R, t = O.normalize_scene(d["init_X"], d["init_R"], d["init_t"], axis)
    g.set_params(X, d["init_K"][:, 0, 0], d["init_K"][:, :2, 2], t, R)
    return g


@pytest.mark.parametrize("name,axis", [("linearize_60x7_xup", "x-up_z-forward"),
                                        ("linearize_60x7_xright", "x-right_z-forward")])
def test_one_linearisation_all_intermediates(golden, name, axis):
    d = golden(name)
    g = _engine_from(d, axis)
    np.testing.assert_allclose(g.X, d["norm_X"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(g.R, d["norm_R"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(g.t, d["norm_t"], rtol=0, atol=1e-13)
    vis = d["vis"]
    n, m = vis.shape
    p, q, r, _ = O.project(g.X, g.f, g.u, g.t, g.R, 1.0, g.pt, g.cam)
    for mine, ref in ((p, d["p"]), (q, d["q"]), (r, d["r"])):
        np.testing.assert_allclose(mine, ref[vis], rtol=1e-13, atol=1e-14)
    assert abs(g.cost() - float(d["E0"])) < 1e-14
    g.linearize()
    np.testing.assert_allclose(g.dP.ravel(), d["d_P"], rtol=1e-11, atol=1e-13)
    keep = g.keep
    np.testing.assert_allclose(g.dF.ravel()[keep], d["d_F"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(g.E, d["matE"], rtol=1e-11, atol=1e-13)
    # F: dense (N,3,9m-7) in the reference; sparse blocks here
    Fd = np.zeros((n, 3, 9 * m))
    for o, (a, k) in enumerate(zip(g.pt, g.cam)):
        Fd[a, :, 9 * k:9 * k + 9] = g.F[o]
    np.testing.assert_allclose(Fd[:, :, keep], d["matF"], rtol=1e-11, atol=1e-13)
    Gd = np.zeros((9 * m, 9 * m))
    for k in range(m):
        Gd[9 * k:9 * k + 9, 9 * k:9 * k + 9] = g.G[k]
    np.testing.assert_allclose(Gd[np.ix_(keep, keep)], d["matG"], rtol=1e-11, atol=1e-11)
    E1 = g.try_step(float(d["c"]))
    sc = np.abs(d["A"]).max()
    np.testing.assert_allclose(g.A, d["A"], rtol=0, atol=1e-12 * sc)
    np.testing.assert_allclose(g.b, d["b"], rtol=0, atol=1e-12 * np.abs(d["b"]).max())
    np.testing.assert_allclose(g.dxi_red, d["dxi"], rtol=0, atol=1e-10 * np.abs(d["dxi"]).max())
    np.testing.assert_allclose(g.dX, d["dX"], rtol=0, atol=1e-10 * np.abs(d["dX"]).max())
    np.testing.assert_allclose(g.tR, d["trial_R"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(g.tt, d["trial_t"], rtol=0, atol=1e-11)
    assert abs(E1 - float(d["E1"])) < 1e-12


@pytest.mark.parametrize("name,axis,args", [
    ("euclid_default", "x-up_z-forward", (2.0, 1e-8, 100)),
    ("affine_default", "x-up_z-forward", (2.0, 1e-8, 100)),
    ("linearize_60x7_xup", "x-up_z-forward", (10.0, 1e-8, 8)),
    ("linearize_60x7_xright", "x-right_z-forward", (10.0, 1e-8, 8)),
    ("visibility_300x12", "x-up_z-forward", (2.0, -1.0, 10)),
])
def test_full_trajectory(golden, name, axis, args):
    d = golden(name)
    vis = d["vis"] if "vis" in d.files else None
    ba = O.OracleBundleAdjuster(d["x"], d["init_X"], d["init_K"], d["init_R"], d["init_t"],
                                visibility_index=vis, axis=axis)
    X, K, R, t = ba.optimize(*args, is_debug=True, verbose=False)
    E = np.array([e["reprojection_error"] for e in ba.get_log()])
    assert len(E) == len(d["E_log"])
    assert ba.engine.n_solves == int(d["n_solves"])
    n_obs = ba.engine.xy.shape[0]
    rmse, rmse_ref = np.sqrt(E[-1] / n_obs), np.sqrt(d["E_log"][-1] / n_obs)
    assert abs(rmse - rmse_ref) < 1e-9  # BASELINE.json tolerance (fp64)
    np.testing.assert_allclose(E, d["E_log"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(X, d["out_X"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(K, d["out_K"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(R, d["out_R"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(t, d["out_t"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,axis,args", [
    ("euclid_default", "x-up_z-forward", (2.0, 1e-8, 100)),
    ("visibility_300x12", "x-up_z-forward", (2.0, -1.0, 10)),
    ("linearize_60x7_xright", "x-right_z-forward", (10.0, 1e-8, 8)),
])
def test_dense_faithful_oracle_full_trajectory(golden, name, axis, args):
    """oracle/ba_dense.py (the reference's dense broadcast form, timed as the CPU baseline at
    configs 1-2) reproduces the reference's trajectories: same iteration and solve counts, RMSE 1e-9."""
    from oracle import ba_dense as Dn

    d = golden(name)
    vis = d["vis"] if "vis" in d.files else None
    ba = Dn.DenseBundleAdjuster(d["x"], d["init_X"], d["init_K"], d["init_R"], d["init_t"],
                                visibility_index=vis, axis=axis)
    X, K, R, t = ba.optimize(*args, is_debug=True, verbose=False)
    E = np.array([e["reprojection_error"] for e in ba.get_log()])
    assert len(E) == len(d["E_log"]) and ba.engine.n_solves == int(d["n_solves"])
    n_obs = int(ba.engine.v.sum())
    assert abs(np.sqrt(E[-1] / n_obs) - np.sqrt(d["E_log"][-1] / n_obs)) < 1e-9
    np.testing.assert_allclose(E, d["E_log"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(X, d["out_X"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(R, d["out_R"], rtol=0, atol=1e-6)


def test_sharded_all_cores_oracle_equals_single_engine(golden):
    """oracle/ba_parallel.py (bench.py's all-cores cpu_baseline) == one OracleEngine on the same scene."""
    from lib.bundle_adjustment import lm_loop
    from oracle.ba_parallel import ShardedOracle

    d = golden("visibility_300x12")
    axis = "x-up_z-forward"
    g = _engine_from(d, axis)
    X, f, u, t, R = g.get_params()
    p = ShardedOracle(g.n, g.m, g.pt_ptr, g.cam, g.xy, 1.0, axis, X, f, u, t, R, n_workers=3)
    try:
        E1 = lm_loop(g, 2.0, -1.0, 4, verbose=False)
        E2 = lm_loop(p, 2.0, -1.0, 4, verbose=False)
        assert g.n_solves == p.n_solves
        assert abs(E1 - E2) <= 1e-12 * E1
        np.testing.assert_allclose(p.points(), g.X, rtol=0, atol=1e-11)
    finally:
        p.close()


def test_euclid_default_headline_numbers(golden):
    d = golden("euclid_default")
    assert int(d["n_outer"]) == 37 and int(d["n_solves"]) == 59
    assert d["E_log"][0] == pytest.approx(66.31926634440299, abs=1e-12)
    assert np.sqrt(d["E_log"][-1] / 2000) == pytest.approx(0.0063291001035384233, abs=1e-15)


def test_rodrigues_and_transforms(golden):
    d = golden("known_answers")
    for w, Rref in zip(d["omega"], d["rodrigues"]):
        np.testing.assert_allclose(O.rodrigues(w), Rref, rtol=0, atol=1e-15)
    assert (O.rodrigues(np.zeros(3)) == np.eye(3)).all()
    for axis, k in (("x-right_z-forward", "xright"), ("x-up_z-forward", "xup")):
        nX, nR, nt = O.normalize_scene(d["tr_X"], d["tr_R"], d["tr_t"], axis)
        np.testing.assert_allclose(nX, d[f"{k}_nX"], atol=1e-14)
        np.testing.assert_allclose(nR, d[f"{k}_nR"], atol=1e-14)
        np.testing.assert_allclose(nt, d[f"{k}_nt"], atol=1e-14)
        sc = O.baseline_length(d["tr_R"], d["tr_t"], axis)
        assert sc == pytest.approx(float(d[f"{k}_c0c1"]), abs=1e-15)
        bX, bR, bt = O.denormalize_scene(d["tr_R"][0], d["tr_t"][0], sc, nX, nR, nt)
        np.testing.assert_allclose(bX, d[f"{k}_bX"], atol=1e-13)
        np.testing.assert_allclose(bR, d[f"{k}_bR"], atol=1e-13)
        np.testing.assert_allclose(bt, d[f"{k}_bt"], atol=1e-13)


def test_error_behaviour(golden):
    d = golden("known_answers")
    assert str(d["err_bad_axis"]) == "ValueError"
    assert str(d["err_zero_degree"]).startswith("LinAlgError")
    with pytest.raises(ValueError):
        O.OracleBundleAdjuster(d["tr_x"], d["tr_X"], d["tr_K"], d["tr_R"], d["tr_t"], axis="bogus")
    vis0 = np.ones(d["tr_x"].shape[:2], bool)
    vis0[2] = False
    with pytest.raises(np.linalg.LinAlgError):
        O.OracleBundleAdjuster(d["tr_x"], d["tr_X"], d["tr_K"], d["tr_R"], d["tr_t"],
                               visibility_index=vis0).optimize(10.0, 1e-8, 2, verbose=False)
