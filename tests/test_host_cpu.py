"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol of
include/mvba.h; the per-observation math compiled into the library agrees with
the oracle; the product's LM control loop + normalisation reproduce the golden
trajectories when driven by the oracle engine; the product path refuses to run
without a GPU."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from lib import _mvba
from lib.bundle_adjustment import BundleAdjuster
from lib.synthetic import make_scene
from oracle import ba_oracle as O

from _engines import HostOracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mvba.h")).read()
    declared = set(re.findall(r"\b(mvba_[a-z_0-9]+|mvsvd_[a-z_0-9]+)\s*\(", hdr)) - {"mvba_handle", "mvba_problem", "mvba_stats"}
    assert declared == set(_mvba.SIGNATURES), declared ^ set(_mvba.SIGNATURES)
    lib = _mvba.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"mvba" in lib.mvba_version()
    assert lib.mvba_kernel_name(0) == b"resid_jac"


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_mvba.Problem) == 64
    assert ctypes.sizeof(_mvba.Stats) == 16 * 8 * 2 + 40


def test_host_obs_math_matches_oracle():
    rng = np.random.default_rng(0)
    sc = make_scene(40, 5, vis_p=1.0)
    f = sc.init_K[:, 0, 0]
    u = rng.normal(0, 0.05, (5, 2))
    pt = np.repeat(np.arange(40), 5)
    cam = np.tile(np.arange(5), 40)
    for f0 in (1.0, 1.7):
        e, JX, JC = O.jacobians(sc.init_X, f, u, sc.init_t, sc.init_R, f0, pt, cam, sc.xy)
        for o in range(0, 200, 7):
            k = cam[o]
            cam15 = np.concatenate([[f[k]], u[k], sc.init_t[k], sc.init_R[k].ravel()])
            e_, jx_, jc_ = _mvba.host_obs_math(sc.init_X[pt[o]], cam15, sc.xy[o], f0)
            np.testing.assert_allclose(e_, e[o], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(jx_, JX[o], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(jc_, JC[o], rtol=1e-12, atol=1e-13)


class _OracleBackedAdjuster(BundleAdjuster):
    """Product host logic (normalisation, LM loop, log, K assembly) over the CPU oracle engine."""

    def _make_engine(self, n_points, n_images, pt_ptr, cam_idx, xy, f0, axis, **kw):
        xy = np.asarray(xy)  # (image planes (m, N, 2) when the caller's x was a transposed stack: the oracle takes the list form)
        xy = xy.transpose(1, 0, 2).reshape(-1, 2) if xy.ndim == 3 else xy
        return HostOracleEngine(n_points, n_images, pt_ptr, cam_idx, xy, f0, axis)


@pytest.mark.parametrize("name,args", [("euclid_default", (2.0, 1e-8, 100)),
                                       ("visibility_300x12", (2.0, -1.0, 10))])
def test_product_lm_loop_over_oracle_engine(golden, name, args, capsys):
    d = golden(name)
    vis = d["vis"] if "vis" in d.files else None
    ba = _OracleBackedAdjuster(d["x"], d["init_X"], d["init_K"], d["init_R"], d["init_t"],
                               visibility_index=vis, axis="x-up_z-forward")
    X, K, R, t = ba.optimize(*args, is_debug=True)
    out = capsys.readouterr().out
    log = ba.get_log()
    E = np.array([e["reprojection_error"] for e in log])
    assert len(E) == len(d["E_log"]) and ba._engine.n_solves == int(d["n_solves"])
    np.testing.assert_allclose(E, d["E_log"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(X, d["out_X"], atol=1e-6)
    np.testing.assert_allclose(K, d["out_K"], atol=1e-6)
    np.testing.assert_allclose(R, d["out_R"], atol=1e-6)
    np.testing.assert_allclose(t, d["out_t"], atol=1e-6)
    assert set(log[0]) == {"points", "basis", "pos", "reprojection_error"}
    assert log[0]["points"].shape == d["x"].shape[:1] + (3,)
    if "stdout" in d.files:  # same line format as the reference's print (ref :188)
        ref_lines = str(d["stdout"]).strip().splitlines()
        got = out.strip().splitlines()
        assert len(got) == len(ref_lines)
        assert got[0].startswith("Iteration 1: reprojection_error_delta = ")
        np.testing.assert_allclose(float(got[0].split("= ")[1]), float(ref_lines[0].split("= ")[1]), rtol=1e-9)
    if name == "euclid_default":
        np.testing.assert_allclose(log[0]["points"], d["log0_points"], atol=1e-12)
        np.testing.assert_allclose(log[-1]["basis"], d["logN_basis"], atol=1e-6)
    tl = golden("trajectory_logs")  # the reference's own log entries (first, inside, last)
    if name + "_len" in tl.files:
        assert len(log) == int(tl[name + "_len"])
        for i in tl[name + "_picks"]:
            tol = 1e-9 if i < 10 else 1e-6
            for key in ("points", "basis", "pos"):
                np.testing.assert_allclose(log[i][key], tl[f"{name}_{i}_{key}"], rtol=0, atol=tol)


def test_bad_axis_raises_value_error(golden):
    d = golden("known_answers")
    with pytest.raises(ValueError):
        BundleAdjuster(d["tr_x"], d["tr_X"], d["tr_K"], d["tr_R"], d["tr_t"], axis="bogus")


@pytest.mark.skipif(_mvba.device_count() > 0, reason="a GPU is present")
def test_product_path_fails_loudly_without_gpu(golden):
    d = golden("known_answers")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BundleAdjuster(d["tr_x"], d["tr_X"], d["tr_K"], d["tr_R"], d["tr_t"])


def test_synthetic_scene_is_consistent_and_shardable():
    sc = make_scene(70000, 6, vis_p=0.5)
    assert sc.pt_ptr[-1] == sc.n_obs and (np.diff(sc.pt_ptr) >= 3).all()
    pt = np.repeat(np.arange(sc.n_points), np.diff(sc.pt_ptr))
    # ascending cameras within a point
    same = pt[1:] == pt[:-1]
    assert (np.diff(sc.cam_idx.astype(np.int64))[same] > 0).all()
    # observations are the GT projection + ~1e-3 noise
    e = O.residuals(sc.X_gt, sc.K_gt[:, 0, 0], sc.K_gt[:, :2, 2], sc.t_gt, sc.R_gt, 1.0, pt, sc.cam_idx, sc.xy)
    assert 0.5e-3 < e.std() < 2e-3
    # a slice generated on its own equals the slice of the whole
    part = make_scene(70000, 6, vis_p=0.5, point_range=(65000, 68000))
    o0, o1 = sc.pt_ptr[65000], sc.pt_ptr[68000]
    np.testing.assert_array_equal(part.cam_idx, sc.cam_idx[o0:o1])
    np.testing.assert_array_equal(part.xy, sc.xy[o0:o1])
    np.testing.assert_array_equal(part.init_X, sc.init_X[65000:68000])
    np.testing.assert_array_equal(part.pt_ptr, sc.pt_ptr[65000:68001] - o0)


def test_bench_cpu_baseline_and_pmc_helpers(monkeypatch):
    """bench.py's CPU-side pieces: the oracle-timed baseline returns the contract's fields on a
    tiny sample, and the PMC traffic helper only answers for the workload it was measured on."""
    import json

    import bench

    sc = make_scene(4000, 8, vis_p=0.5)
    cb = bench.cpu_baseline(sc, 8, iters=2, workers=2, config2=False)
    assert cb["kind"] == "port" and cb["cores"] == 2 and cb["unit"] == "it/s" and cb["value"] > 0
    assert "4000 points" in cb["sample"] and cb["host_cpus"] == os.cpu_count() and cb["rmse_end"] < cb["rmse_start"]
    d = json.load(open(bench.PMC_FILES[0]))
    k1 = d["kernels"]["k_resid_jac"]
    # the figures are quoted only for the kernel sources they were measured on (sha256 in the file) ...
    monkeypatch.setattr(bench, "csrc_sha256", lambda: d["csrc_sha256"])
    t, src = bench.pmc_traffic("k_resid_jac", d["n_obs"])
    assert t == pytest.approx((2 * k1["FETCH_SIZE_KiB"] + k1["WRITE_SIZE_KiB"]) * 1024) and "pmc_config3.json" in src
    assert 0.95 < t / (152 * d["n_obs"] + 96 * d["n_points"]) < 1.10  # HBM traffic ~ algorithmic bytes
    k3 = d["kernels"]["k_schur_slots"]
    t3, _ = bench.pmc_traffic("k_schur_slots", d["n_obs"])
    assert t3 == pytest.approx((2 * k3["FETCH_SIZE_KiB"] + k3["WRITE_SIZE_KiB"]) * 1024)
    assert 1.0 < t3 / (192 * d["n_obs"]) < 1.5  # the slot kernel reads a record once per launch, not 5 times
    assert bench.pmc_traffic("k_resid_jac", d["n_obs"] + 1) == (None, None)
    assert bench.pmc_traffic("no_such_kernel", d["n_obs"]) == (None, None)
    # ... and dropped, with a note, once the sources have changed
    monkeypatch.setattr(bench, "csrc_sha256", lambda: "0" * 64)
    t, src = bench.pmc_traffic("k_resid_jac", d["n_obs"])
    assert t is None and "STALE" in src
    monkeypatch.undo()
    assert len(bench.csrc_sha256()) == 64


def test_bench_round5_helpers_cores_allreduce_model_and_shard_roofline(monkeypatch):
    """bench.py's round-5 pieces on the CPU: the worker count follows what the box really grants (physical cores, affinity mask,
    cgroup quota -- the GPU boxes of this pool grant 16 CPUs of 128 physical cores), the all-reduce model prices C1's bytes on the
    xGMI links, and the Schur roofline object names what bounds each kernel form."""
    import bench

    assert bench.default_cpu_workers({"host_cpus": 256, "physical_cores": 128, "affinity_cpus": 256, "cgroup_quota_cpus": 16.0}) == 16
    assert bench.default_cpu_workers({"host_cpus": 256, "physical_cores": 128, "affinity_cpus": 256, "cgroup_quota_cpus": None}) == 128
    assert bench.default_cpu_workers({"host_cpus": 8, "physical_cores": None, "affinity_cpus": 4, "cgroup_quota_cpus": None}) == 4
    assert bench.default_cpu_workers({"host_cpus": 512, "physical_cores": 256, "affinity_cpus": 512, "cgroup_quota_cpus": None}) == 128
    info = bench.cpu_info()
    assert info["host_cpus"] == os.cpu_count() and info["affinity_cpus"] >= 1 and (info["physical_cores"] or 1) >= 1
    mdl = bench.allreduce_model(500, 8)
    assert mdl["bytes_per_solve"] == 8 * (81 * 500 * 501 // 2 + 9 * 500) == 81_198_000
    assert mdl["ring_one_link_ms"] == pytest.approx(2 * 7 / 8 * 81_198_000 / 153e9 * 1e3)
    assert mdl["direct_all_links_ms"] == pytest.approx(2 * (81_198_000 / 8) / 153e9 * 1e3)
    wk = bench.weak_scaling_model(100, 2.4)  # the weak series' N = 2, 4, 8 points as the all-reduce model prices them (3.28 MB per solve)
    ar8 = 2 * 7 / 8 * 8 * (81 * 100 * 101 // 2 + 9 * 100) / 153e9 * 1e3 + 0.040
    assert wk["8"]["allreduce_ms"] == pytest.approx(ar8) and wk["8"]["efficiency"] == pytest.approx(2.4 / (2.4 + ar8))
    assert wk["2"]["efficiency"] > wk["4"]["efficiency"] > wk["8"]["efficiency"] > 0.95 and "model" in wk["note"]
    monkeypatch.setattr(bench, "pmc_traffic", lambda kernel, n_obs: (None, None))
    unit = bench.schur_roofline({"kernel": "pairs", "items": 421_000_000, "offdiag_items": 390_000_000, "units": 1_000_000, "slot_rows": 0},
                                31_246_709, 13.7, 256)
    assert unit["bound"] == "fabric_line_fills" and unit["kernel"].startswith("k_schur_pairs")
    assert unit["achieved"] == pytest.approx(192 * 31_246_709 / 13.7e-3 / 1e9) and unit["frac"] == pytest.approx(unit["achieved"] / 8000.0)
    slot = bench.schur_roofline({"kernel": "slots", "items": 59_501_226, "offdiag_items": 49_499_384, "units": 47_600, "slot_rows": 66_863_118},
                                10_001_842, 1.58, 256)
    assert slot["bound"] == "l2_gather" and slot["gather"]["row_gathers_per_launch"] == 3 * 66_863_118
    # the dense-visibility form is priced against the f64 matrix cores: 0.75 MFMAs per tile pair and point + half a one per camera
    dense = bench.schur_roofline({"kernel": "dense", "items": 0, "offdiag_items": 0, "units": 0, "slot_rows": 0, "n_cams": 12, "n_points": 1_000_000},
                                 12_000_000, 1.31, 256)
    assert dense["bound"] == "mfma" and dense["unit"] == "TFLOP/s" and dense["traffic"] is None
    assert dense["achieved"] == pytest.approx(1e6 * (0.75 * 28 + 6) * 2048 / 1.31e-3 / 1e12) and dense["frac"] == pytest.approx(dense["achieved"] / 78.6)


def test_full_visibility_fast_paths_equal_the_general_ones():
    """Round 5's host-side shortcuts for the all-visible case give exactly what the general code gives: the observation list of a
    dense array without a mask (no np.nonzero / gather: 0.24 s at 1 M x 12) and the data matrix filled in place."""
    from lib.bundle_adjustment import dense_to_observations
    from lib.perspective_camera_calibration import _create_data_matrix

    rng = np.random.default_rng(3)
    x = rng.standard_normal((57, 6, 2))
    a, b = dense_to_observations(x, None), dense_to_observations(x, np.ones((57, 6), bool))
    for u, v in zip(a, b):
        assert u.dtype == v.dtype and np.array_equal(u, v)
    # the reference caller's array, np.stack(x_list).transpose(1, 0, 2): its memory (the image planes) is handed on as it is
    xs = np.stack([np.ascontiguousarray(x[:, k]) for k in range(6)]).transpose(1, 0, 2)
    p, c, planes = dense_to_observations(xs, None)
    assert planes.shape == (6, 57, 2) and np.shares_memory(planes, xs) and planes.flags.c_contiguous
    assert np.array_equal(p, a[0]) and np.array_equal(c, a[1]) and np.array_equal(planes.transpose(1, 0, 2).reshape(-1, 2), a[2])
    x_list = [x[:, k] for k in range(6)]
    ref = np.stack([np.column_stack([xi / 1.7, np.ones(len(xi))]) for xi in x_list]).transpose(1, 0, 2)  # the reference's form (:34-40)
    got = _create_data_matrix(x_list, 1.7)
    assert got.flags["C_CONTIGUOUS"] and np.array_equal(got, ref)


def test_bench_config1_leg_reproduces_the_reference_counts_on_the_cpu():
    """bench.py's config-1 leg: the dense-faithful oracle on the reference's default
    scene lands on 37 outer iterations / 59 solves / RMSE 0.0063291001035384233; without a GPU the
    HIP leg reports an error instead of taking the CPU leg down with it."""
    import bench

    r = bench.config1_default_scene()
    cpu = r["cpu_dense_faithful"]
    assert (cpu["outer_iterations"], cpu["solves"]) == (37, 59)
    assert abs(cpu["rmse"] - r["expected"]["rmse"]) < 1e-9
    assert "gpu" in r and ("error" in r["gpu"] or r["gpu"]["solves"] == 59)


def test_bench_self_launch_command_and_forwarding(monkeypatch, capsys):
    """`python bench.py --gpus N ...` without a launcher starts torch.distributed.run itself as a
    child (the driver's contract form) with the same arguments and forwards rank 0's JSON line."""
    import importlib
    import subprocess
    import types

    bench = importlib.import_module("bench")
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4, 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]

    seen = {}

    def fake_run(c, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = c, env
        return types.SimpleNamespace(returncode=0, stdout='RCCL banner\n{"metric": "m", "value": 1.0, "n_gpus": 2}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.self_launch(["--gpus", "2"], 2)
    out = capsys.readouterr()
    assert rc == 0 and out.out.strip() == '{"metric": "m", "value": 1.0, "n_gpus": 2}'
    assert "RCCL banner" in out.err  # library chatter goes to stderr, stdout carries the JSON line alone
    assert seen["cmd"][-2:] == ["--gpus", "2"] and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    # a failing child: its status is ours, nothing on stdout
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=3, stdout="boom\n"))
    assert bench.self_launch(["--gpus", "2"], 2) == 3
    assert capsys.readouterr().out == ""


def test_slot_kernel_loops_carry_no_vector_memory_operation_the_counted_waits_do_not_know():
    """k_schur_slots keeps two gathers in flight with COUNTED `s_waitcnt vmcnt(N)`; what that relies on in the generated
    ISA (exactly N vector-memory operations per iteration, no scratch access in the loops, pinned index registers, M0)
    is checked by csrc/check_isa.py -- a step of the BUILD (`make` fails when it fails; `__graft_entry__.build()` runs
    make).  Here: the check passes on the current sources, and it does catch the faults it exists for."""
    import subprocess

    csrc = os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "isa.ok"], check=True, capture_output=True)
    sys.path.insert(0, csrc)
    try:
        import check_isa
    finally:
        sys.path.remove(csrc)
    text = open(os.path.join(csrc, "mvba.s")).read()
    assert check_isa.check(text) == []
    # an uncounted operation in a loop (what a spill reload looks like), a register-returning load, a stray M0 write
    w = "s_waitcnt vmcnt(8)"
    m = re.search(r"^_ZN\d+_GLOBAL__N_113k_schur_slotsE\w*:.*?^\.Lfunc_end", text, re.S | re.M)
    body = m.group(0)
    assert body.count(w) == 1
    for inject, needle in (("scratch_load_dword v3, off, off offset:4", "scratch"),
                           ("global_load_dword v7, v2, s[4:5]", "does not count"),
                           ("global_load_lds_dwordx4 v2, s[4:5]", "LDS-DMA operations per iteration"),
                           ("s_add_u32 m0, m0, 4", "M0 written")):
        broken = text.replace(body, body.replace(w, w + "\n\t" + inject, 1), 1)
        errs = check_isa.check(broken)
        assert errs and any(needle in e for e in errs), (inject, errs)
    # the point-to-point back-substitution: a progress word must not overtake the data it announces, and the kernel must
    # stay free of device-wide fences
    m = re.search(r"^_ZN\d+_GLOBAL__N_1\d+k_chol_backsolve_allILb1EE\w*:.*?^\.Lfunc_end", text, re.S | re.M)
    body = m.group(0)
    assert "s_waitcnt vmcnt(0)" in body
    errs = check_isa.check(text.replace(body, re.sub(r"[ \t]*s_waitcnt vmcnt\(0\)[^\n]*\n", "", body), 1))
    assert errs and any("overtake" in e for e in errs), errs
    errs = check_isa.check(text.replace(body, body.replace("s_barrier", "buffer_wbl2 sc1\n\ts_barrier", 1), 1))
    assert errs and any("device-wide fence" in e for e in errs), errs
    errs = check_isa.check(text.replace(body, re.sub(r"(global_(?:load|store)_dwordx2[^\n]*?) sc1", r"\1", body), 1))
    assert errs and any("no sc1 load / store" in e for e in errs), errs
