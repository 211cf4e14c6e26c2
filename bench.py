#!/usr/bin/env python3
"""Headline benchmark: bundle-adjustment LM iterations/s (+ residual-Jacobian GObs/s), fp64.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE outer Levenberg-Marquardt iteration of the hot path over the whole
(synthetic, HBM-resident) observation list: K1 residual+Jacobian with K2 point blocks
fused, then per trial K3a/K3 Schur, (C1 all-reduce), K4 solve, K5/K6 back-substitution +
trial cost, commit; the LM control flow is the reference's own
(lib/bundle_adjustment.py:102-195) with optimize(2.0, -1.0, max_iter) semantics.

N = 1   BASELINE config 3: 1M points x 100 cameras, 10 % visibility (the configuration the
        metric is quoted on).
N > 1   (one process per GPU)  WEAK scaling of that same workload: every rank holds a config-3-sized
        point shard (1M points of an N x 1M-point scene; the 100 cameras are replicated), one RCCL
        all-reduce of the packed reduced camera system per LM solve.  `value` = the units all ranks
        processed / time = N x (LM iterations per second of the joint problem), one unit being an LM
        iteration over one config-3-sized shard -- so that the per-N values of one series measure one
        thing and `value(N) / (N value(1))` is the efficiency (`config.joint_it_per_s` is the plain rate).
        `--strong` instead runs BASELINE config 4 (10M points x 500 cameras, 5 % visibility) split by
        point id into N observation-balanced shards, `value` = plain it/s of the whole job -- a series
        of its own whose N = 1 point is `--config4` (rounds 2-5 made this the N > 1 default, which put
        two different workloads into the driver's N = 1, 2, 4, 8 series).
        Started either by a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py
        --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or
        plainly as `python bench.py --gpus N ...`: with no WORLD_SIZE in the environment the
        process starts that launcher itself as a CHILD (before anything touches the GPU), forwards
        rank 0's JSON line and exits with the child's status.
        `--transport host` moves the all-reduce through the host (gloo) and lets the ranks share
        GPUs (rank r on device r mod device count): the whole N > 1 bench path on a one-GPU box.

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import gc
import io
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC (RCCL needs it)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (f64 FMA issues 16 lanes/clk/SIMD)
PMC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("pmc_config3.json", "pmc_config4_shard.json")]
GATHER_CEILING_ROWS_PER_US_PER_CU = 740.0  # profiles/r03_microbench_gather_rows.txt: 717 (128-B rows) .. 759 (112-B rows) from L2, LDS-DMA


def csrc_sha256():
    """sha256 over the sources of the BA kernels (K1, K3: what `profiles/pmc_config3.json` was measured on -- the SVD
    file is not among them): the committed PMC figures are only quoted for the sources they were taken on."""
    import hashlib

    h = hashlib.sha256()
    for f in ("csrc/mvba.hip", "csrc/mvba_common.h", "csrc/Makefile"):
        with open(os.path.join(PKG, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(ROOT, "include", "mvba.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def pmc_traffic(kernel, n_obs):
    """HBM bytes per launch of `kernel` from the COMMITTED rocprofv3 --pmc passes of this workload
    (tools/final_profile.sh: FETCH_SIZE and WRITE_SIZE in separate passes, KiB; FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads on gfx950).  Not measured in
    this run: the second return value names the file and the commit it was taken at."""
    for path in PMC_FILES:
        try:
            d = json.load(open(path))
            k = d["kernels"][kernel]
            if int(d["n_obs"]) != int(n_obs):
                continue
            name = "profiles/" + os.path.basename(path)
            if d.get("csrc_sha256") != csrc_sha256():  # the kernels have changed since the counters were collected
                return None, f"{name} @ {d.get('commit', '?')} is STALE (kernel sources changed since): traffic omitted"
            return (2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0, f"{name} @ {d.get('commit', '?')}"
        except Exception:  # noqa: BLE001
            continue
    return None, None


def cpu_info():
    model = platform.processor() or ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    blas = None
    try:
        from threadpoolctl import threadpool_info
        blas = [{"api": i.get("internal_api"), "threads": i.get("num_threads")} for i in threadpool_info()]
    except Exception:  # noqa: BLE001
        pass
    cores = set()
    try:  # physical cores = distinct (socket, core id) pairs
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip() and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count()
    quota = None
    try:  # cgroup v2 CPU quota of this container, in CPUs
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    return {"host_cpus": os.cpu_count(), "physical_cores": len(cores) or None, "affinity_cpus": affinity, "cgroup_quota_cpus": quota,
            "cpu_model": model, "blas": blas, "numpy": np.__version__}


def default_cpu_workers(info):
    """One oracle worker per PHYSICAL core this process may use (NumPy's kernels are single-threaded; SMT siblings add nothing
    to an fp64 stream): min(physical cores, affinity mask, cgroup quota), at most 128 processes."""
    n = info.get("physical_cores") or info.get("host_cpus") or 1
    n = min(n, info.get("affinity_cpus") or n)
    if info.get("cgroup_quota_cpus"):
        n = min(n, max(1, int(info["cgroup_quota_cpus"])))
    return max(1, min(int(n), 128))


def schur_roofline(info, n_obs, k3_ms, n_cu):
    """`roofline` object of K3.  `achieved` / `frac` price SURVEY 8d's algorithmic 192 B/observation against the
    HBM peak, as the contract asks; `bound` says what actually bounds the kernel form that ran: the slot form is held back by
    what a CU can do per step beside its L2 -> LDS row gathers (see `gather`; round 5's knock-out builds: no single resource,
    profiles/r05_k3_knockouts.txt), the unit form by the fabric -- its L2 MISSES run at the line-fill rate of the chip
    (`traffic` = 12-13 x the algorithmic bytes at 500 cameras)."""
    k3_bytes = 192 * n_obs
    k3_ach = k3_bytes / (k3_ms * 1e-3) / 1e9
    if info["kernel"] == "dense":  # full visibility, up to 21 cameras: one rank-3N update of the reduced matrix on the f64 matrix cores
        m9 = 9 * info["n_cams"]
        tiles = (m9 + 15) // 16
        mfmas = info["n_points"] * (0.75 * tiles * (tiles + 1) / 2 + info["n_cams"] / 2)  # four rows of G per MFMA; two points per camera tile
        mfma_tflops = mfmas * 2048 / (k3_ms * 1e-3) / 1e12
        return {"kernel": "k_schur_dense (K3)", "bound": "mfma", "achieved": mfma_tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": mfma_tflops / FP64_VALU_PEAK_TFLOPS, "traffic": None,
                "bound_note": "v_mfma_f64_16x16x4 issued (padding of the 9 m columns to 16-column tiles included) x 2048 flop against the "
                              "78.6 TFLOP/s of the f64 matrix cores; the records (128 B/observation) are streamed once",
                "algorithmic_bytes_per_launch": 128 * n_obs, "hbm_GBs": 128 * n_obs / (k3_ms * 1e-3) / 1e9, "avg_launch_ms": k3_ms}
    k3_name = {"strip": "k_schur_strip", "pairs": "k_schur_pairs", "slots": "k_schur_slots"}[info["kernel"]]
    k3_traffic, k3_src = pmc_traffic(k3_name, n_obs)
    gather = info["items"] * (112 + 48 + 12) + info["offdiag_items"] * 112
    flops = info["items"] * 3 * 96 * 2
    gather_rows = 3 * (info["slot_rows"] or info["items"])
    rows_rate = gather_rows / (k3_ms * 1e3) / n_cu
    slot_form = info["kernel"] == "slots"
    roof = {"kernel": k3_name + " (K3)",
            "bound": "l2_gather" if slot_form else "fabric_line_fills",
            "bound_note": ("not HBM: the CU's L2 -> LDS row gathers beside LDS reads and fp64 issue (see `gather`); `achieved` / `frac` price "
                           "SURVEY 8d's algorithmic 192 B/observation against the HBM peak as the contract asks" if slot_form else
                           "not HBM bandwidth on algorithmic bytes: the kernel's L2 misses (l-side records, half of the point rows) are line "
                           "fills from the Infinity Cache / HBM at the chip's line-fill rate, `traffic` = 12-13 x the algorithmic bytes -- the "
                           "largest single named cost, yet only ~17 % of the launch (knock-out builds, profiles/r05_k3_knockouts.txt: 11.5 of "
                           "13.9 ms remain with every gather hitting one line; the rest is a step's serial chain at 12 waves per CU); "
                           "`achieved` / `frac` price SURVEY 8d's 192 B/observation against the HBM peak as the contract asks"),
            "achieved": k3_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": k3_ach / HBM_PEAK_GBS, "traffic": k3_traffic, "traffic_source": k3_src,
            "traffic_unit": "bytes per launch (PMC, committed profile -- not measured in this run)",
            "traffic_GBs": (k3_traffic / (k3_ms * 1e-3) / 1e9 if k3_traffic else None),
            "algorithmic_bytes_per_launch": k3_bytes, "algorithmic_bytes_per_obs": 192, "avg_launch_ms": k3_ms,
            "items": info["items"], "units": info["units"],
            "slot_rows_incl_padding": info["slot_rows"] or None,
            "gather": {"row_gathers_per_launch": gather_rows, "rows_per_us_per_cu": rows_rate,
                       "ceiling_rows_per_us_per_cu": GATHER_CEILING_ROWS_PER_US_PER_CU,
                       "frac": rows_rate / GATHER_CEILING_ROWS_PER_US_PER_CU, "compute_units": n_cu,
                       "definition": "3 row gathers per step row (padding rows included) / launch time / CUs; ceiling: "
                                     "profiles/r03_microbench_gather_rows.txt (717-759 rows/us per CU from L2)"},
            "gathered_bytes_per_launch": gather, "gather_GBs": gather / (k3_ms * 1e-3) / 1e9,
            "fp64_tflops": flops / (k3_ms * 1e-3) / 1e12,
            "frac_of_fp64_valu_peak": flops / (k3_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}
    return roof


def allreduce_model(n_cams, ranks=8):
    """C1 of an N-rank job priced on the xGMI fabric (7 links x ~153 GB/s per GPU, point to point; MI355X_MICROARCH.md): what the
    first real SCALE run's `allreduce.ms_per_solve` can be compared with.  ring = one ring over one link per neighbour (each rank
    sends 2 (N-1)/N of the buffer); direct = reduce-scatter + all-gather with every rank talking to its N-1 peers at once."""
    nbytes = 8 * (81 * n_cams * (n_cams + 1) // 2 + 9 * n_cams)
    link = 153e9
    return {"bytes_per_solve": nbytes, "ranks": ranks, "link_GBs": link / 1e9,
            "ring_one_link_ms": 2 * (ranks - 1) / ranks * nbytes / link * 1e3,
            "direct_all_links_ms": 2 * (nbytes / ranks) / link * 1e3,
            "note": "bandwidth terms only (no launch / latency terms: +10-30 us per collective); every rank then solves the reduced system redundantly"}


def weak_scaling_model(n_cams, ms_per_solve):
    """What the all-reduce model says the driver's N = 2, 4, 8 points of the weak series (`--gpus N`: a config-3-sized shard per GPU)
    will be, from THIS run's ms per solve: per-solve time + one all-reduce of the packed reduced system (bandwidth over one ring
    link + 30 us of launch / latency per collective) + the 16-byte cost gather (10 us).  A MODEL, labelled as such: no multi-GPU
    node has run this bench yet."""
    out = {"note": "model, not a measurement: ms_per_solve of this run + ring all-reduce over one xGMI link (153 GB/s) + 30 us + 10 us "
                   "for the cost gather; efficiency = value(N) / (N value(1)) of the weak series"}
    for n in (2, 4, 8):
        ar = allreduce_model(n_cams, n)["ring_one_link_ms"] + 0.030 + 0.010
        out[str(n)] = {"allreduce_ms": ar, "efficiency": ms_per_solve / (ms_per_solve + ar)}
    return out


def config4_shard_leg(device, steps=3):
    """BASELINE config 4's per-GPU shard (1.25 M points x 500 cameras x 5 % = 1/8 of the scene, no exchange) on the one GPU of an
    N = 1 run: the only hardware evidence for the multi-GPU configuration while no 8-GPU node runs the bench.  One warm-up LM
    iteration, then `steps` iterations with every phase timed (hipEvents on the engine's stream)."""
    import torch

    from lib.bundle_adjustment import BundleAdjuster, LevenbergMarquardt
    from lib.synthetic import make_scene

    n_pts, n_cams, vis = 1_250_000, 500, 0.05
    t0 = time.perf_counter()
    sc = make_scene(n_pts, n_cams, vis_p=vis)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    ba = BundleAdjuster.from_observations(sc.n_points, n_cams, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t,
                                          axis=sc.axis, device=device)
    t_create = time.perf_counter() - t0
    eng = ba._engine
    try:
        lm = LevenbergMarquardt(eng, 2.0)
        E0 = lm.E
        eng.set_profiling(True)
        lm.carry_on(lm.iterate()[0])  # warm-up (creates the events)
        eng.reset_stats()
        s0 = eng.n_solves
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            E_ = lm.iterate()[0]
            lm.carry_on(E_)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = eng.stats()
        solves = eng.n_solves - s0
        info = eng.schur_info()
        n_cu = int(torch.cuda.get_device_properties(device).multi_processor_count)
        per_solve = {k: v["ms"] / max(solves, 1) for k, v in st.items() if k != "counts"}
        k3_ms = st["schur"]["ms"] / max(st["schur"]["launches"], 1)
        return {"workload": f"BASELINE config 4, ONE of its 8 point shards: {n_pts} points x {n_cams} cameras x {vis:.0%} = {sc.n_obs} observations, "
                            f"D = {9 * n_cams - 7}, on one GPU (no exchange); {steps} LM iterations after one warm-up",
                "steps": steps, "inner_solves": solves, "ms_per_step": dt / steps * 1e3, "ms_per_inner_solve": dt / max(solves, 1) * 1e3,
                "it_per_s": steps / dt, "kernel_ms_per_solve": per_solve,
                "rmse_start": float(np.sqrt(E0 / sc.n_obs)), "rmse_end": float(np.sqrt(E_ / sc.n_obs)),
                "scene_generation_s": t_gen, "engine_create_s": t_create,
                "roofline": schur_roofline(info, sc.n_obs, k3_ms, n_cu),
                "counts": st["counts"],
                "allreduce_model_8_gpus": allreduce_model(n_cams, 8)}
    finally:
        eng.close()


def dense_visibility_leg(device, steps=10, n_pts=1_000_000, n_cams=12):
    """The reference's own scene shape at scale: every point seen by every camera, a dozen cameras (its demo scenes, BASELINE config 2,
    the pipeline test).  K3 is then one rank-3N update of the reduced matrix on the f64 matrix cores (k_schur_dense: no gathers, no
    index).  One warm-up LM iteration, then `steps` with every phase timed."""
    import torch

    from lib.bundle_adjustment import BundleAdjuster, LevenbergMarquardt
    from lib.synthetic import make_scene

    sc = make_scene(n_pts, n_cams, vis_p=1.0)
    t0 = time.perf_counter()
    ba = BundleAdjuster.from_observations(sc.n_points, n_cams, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K, sc.init_R, sc.init_t,
                                          axis=sc.axis, device=device)
    t_create = time.perf_counter() - t0
    eng = ba._engine
    try:
        lm = LevenbergMarquardt(eng, 2.0)
        E0 = lm.E
        eng.set_profiling(True)
        lm.carry_on(lm.iterate()[0])
        eng.reset_stats()
        s0 = eng.n_solves
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()  # (as in the main timed region: with torch imported a full collection is a 50 ms step now and then)
        step_s = []
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            E_ = lm.iterate()[0]
            lm.carry_on(E_)
            step_s.append(time.perf_counter() - ts)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        st = eng.stats()
        solves = eng.n_solves - s0
        info = eng.schur_info()
        info.update(n_cams=n_cams, n_points=sc.n_points)
        n_cu = int(torch.cuda.get_device_properties(device).multi_processor_count)
        k3_ms = st["schur"]["ms"] / max(st["schur"]["launches"], 1)
        return {"workload": f"{n_pts} points x {n_cams} cameras, full visibility = {sc.n_obs} observations, D = {9 * n_cams - 7}; {steps} LM iterations after one warm-up",
                "steps": steps, "inner_solves": solves, "ms_per_step": dt / steps * 1e3, "it_per_s": steps / dt,
                "step_ms": {"min": min(step_s) * 1e3, "median": float(np.median(step_s)) * 1e3, "max": max(step_s) * 1e3},
                "kernel_ms_per_solve": {k: v["ms"] / max(solves, 1) for k, v in st.items() if k != "counts"},
                "rmse_start": float(np.sqrt(E0 / sc.n_obs)), "rmse_end": float(np.sqrt(E_ / sc.n_obs)), "engine_create_s": t_create,
                "roofline": schur_roofline(info, sc.n_obs, k3_ms, n_cu)}
    finally:
        eng.close()


def svd_config5(rows, cols=24):
    """BASELINE config 5: rows x 24 fp32 measurement-matrix SVD (rank 3) on the GPU; device times
    from hipEvents inside mvsvd_factorize, HBM-resident (H2D excluded, reported separately).
    CPU leg: np.linalg.svd(full_matrices=False) on the SAME full matrix, all BLAS threads."""
    from lib import _mvba

    rng = np.random.default_rng(0)
    A = rng.standard_normal((rows, 3), dtype=np.float32)
    B = rng.standard_normal((3, cols), dtype=np.float32)
    Wt = A @ B + np.float32(1e-3) * rng.standard_normal((rows, cols), dtype=np.float32)
    _mvba.svd_factorize(Wt[:100000], 3)  # warm-up (module load)
    M, sig, S, mu, tm = _mvba.svd_factorize(Wt, 3)
    dev_ms = tm["gram_ms"] + tm["jacobi_ms"] + tm["project_ms"]
    alg = 2 * rows * cols * 4 + 3 * rows * 4
    t0 = time.perf_counter()
    sig_cpu = np.linalg.svd(Wt, full_matrices=False, compute_uv=True)[1]
    cpu_s = time.perf_counter() - t0
    return {"workload": f"{rows} x {cols} fp32, rank 3", "device_ms": dev_ms, "h2d_ms": tm["h2d_ms"],
            "gram_ms": tm["gram_ms"], "jacobi_ms": tm["jacobi_ms"], "project_ms": tm["project_ms"],
            "algorithmic_bytes": alg, "achieved_GBs": alg / (dev_ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "cpu_numpy_thin_svd_s": cpu_s, "cpu_rows": rows,
            "sigma": [float(x) for x in sig[:4]], "sigma_cpu": [float(x) for x in sig_cpu[:4]]}


def svd_wide(rows=200_000, cols=300):
    """A WIDE measurement matrix (100 images x 3 rows of W; the block iteration of csrc/mvsvd.hip, from 65 columns on): rows x cols
    fp64, rank 4, device time of the iteration + final pass from hipEvents, against LAPACK on the same matrix."""
    from lib import _mvba

    rng = np.random.default_rng(1)
    Wt = rng.standard_normal((rows, 4)) @ rng.standard_normal((4, cols)) + 1e-3 * rng.standard_normal((rows, cols))
    _mvba.svd_factorize(Wt[:20000], 4)  # warm-up
    M, sig, S, mu, tm = _mvba.svd_factorize(Wt, 4)
    dev_ms = tm["jacobi_ms"] + tm["refine_ms"]
    its = int(tm["sweeps"])
    passes = 2 * its + 1  # B = W Q and Z = W^T B per iteration, B once more at the end
    t0 = time.perf_counter()
    s_ref = np.linalg.svd(Wt, compute_uv=False)
    cpu_s = time.perf_counter() - t0
    return {"workload": f"{rows} x {cols} fp64, rank 4 (block power iteration, 32 vectors)", "device_ms": dev_ms, "h2d_ms": tm["h2d_ms"],
            "iterations": its, "passes_over_W": passes, "streamed_bytes": passes * Wt.nbytes,
            "achieved_GBs": passes * Wt.nbytes / (dev_ms * 1e-3) / 1e9, "frac_of_hbm_peak": passes * Wt.nbytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "sigma_rel_err_vs_lapack": float(np.max(np.abs(sig[:4] - s_ref[:4]) / s_ref[:4])), "cpu_numpy_singular_values_s": cpu_s}


def depth_iteration(rows, m=8):
    """The projective-depth iteration (ref lib/perspective_camera_calibration.py:79-129 primary, :166-224
    dual) entirely on the device: `rows` points seen by m images, fp64, one `mvsvd_depth_step` per iteration -- re-weighting,
    rank-4 factorisation, per-point 4 x 4 / per-image 12 x 12 eigenproblem, depth update, reprojection error; 8 bytes cross
    PCIe per iteration.  Wall time of the best of five iterations per scheme and its device phases."""
    from lib import _mvba

    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (rows, 3))
    x = np.empty((rows, m, 3))
    for k in range(m):  # cameras on an arc of radius 5 around the points, looking at the origin
        ph = 0.12 * k - 0.4
        c = 5.0 * np.array([np.sin(ph), 0.0, -np.cos(ph)])
        R = np.array([[np.cos(ph), 0, -np.sin(ph)], [0, 1, 0], [np.sin(ph), 0, np.cos(ph)]])
        Xc = (X - c) @ R
        x[:, k, 0], x[:, k, 1], x[:, k, 2] = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2], 1.0
    x += 1e-3 * rng.standard_normal(x.shape) * np.array([1.0, 1.0, 0.0])
    ws = _mvba.SvdWorkspace(rows, 3 * m, np.float64)
    out = {"workload": f"{rows} points x {m} images, fp64", "pcie_bytes_per_iteration": 8}
    try:
        ws.load_base(x.reshape(rows, 3 * m))
        for method, name in ((1, "primary"), (2, "dual")):
            ws.depth_begin(3)
            ws.depth_step(method, 1.0)  # warm-up (allocations)
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                E, tm = ws.depth_step(method, 1.0)
                wall = (time.perf_counter() - t0) * 1e3
                if best is None or wall < best[0]:
                    best = (wall, tm, E)
            # algorithmic bytes of one iteration: the factorisation needs the eigenvectors before any depth can be
            # updated, so X and z are read at least twice (Gram pass, update pass) and z is written once: (2 x 24 + 3 x 8) m = 72 m
            # bytes per point in fp64.  What this build streams: + the refinement pass of the fp64 factorisation (32 m) and, for the
            # dual scheme, the per-image companion pass (32 m): 104 m / 136 m bytes per point.
            alg = 72.0 * m * rows
            streamed = (104.0 if method == 1 else 136.0) * m * rows
            out[name] = {"wall_ms_per_iteration": best[0], "reprojection_error": best[2],
                         "device_ms": {k: best[1][k] for k in ("gram_ms", "jacobi_ms", "refine_ms", "project_ms", "depth_ms")},
                         "roofline": {"bound": "hbm", "achieved": alg / (best[0] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": alg / (best[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": alg,
                                      "streamed_bytes_per_iteration": streamed, "streamed_GBs": streamed / (best[0] * 1e-3) / 1e9,
                                      "note": "wall time of the whole iteration (five streaming passes, three small eigenproblems, one host "
                                              "round trip for the eigenvalue order); traffic not measured by PMC (null in the contract's sense)",
                                      "traffic": None}}
    finally:
        ws.close()
    return out


def cpu_baseline(sc, n_images, iters=3, workers=None, config2=True):
    """The oracle (NumPy/SciPy restatement of the reference, pinned by golden vectors) timed on
    this host.  (i) config 3 ITSELF, `iters` outer LM iterations of optimize(2.0, -1.0, iters), on
    all host cores: one oracle engine per worker process on a point shard (oracle/ba_parallel.py;
    NumPy's own kernels are single-threaded).  (ii) config 2 (10k x 20, full visibility) with the
    dense-faithful restatement of the reference's algorithm (oracle/ba_dense.py), 2 iterations."""
    from lib.bundle_adjustment import lm_loop
    from oracle import ba_oracle as O
    from oracle.ba_parallel import ShardedOracle

    info = cpu_info()
    workers = int(workers or default_cpu_workers(info))
    X, R, t = O.normalize_scene(sc.init_X, sc.init_R, sc.init_t, sc.axis)
    t0 = time.perf_counter()
    g = ShardedOracle(sc.n_points, n_images, sc.pt_ptr, sc.cam_idx, sc.xy, 1.0, sc.axis, X, sc.init_K[:, 0, 0],
                      sc.init_K[:, :2, 2], t, R, n_workers=workers)
    E0 = g.cost()
    t1 = time.perf_counter()
    E = lm_loop(g, 2.0, -1.0, iters, verbose=False)
    t2 = time.perf_counter()
    solves = g.n_solves
    g.close()
    out = {
        "value": iters / (t2 - t1), "unit": "it/s", "cores": workers, "kind": "port",
        "sample": f"oracle/ba_parallel.py: the WHOLE config-3 scene ({sc.n_points} points x {n_images} cameras, {sc.n_obs} "
                  f"observations), optimize(2.0, -1.0, {iters}) = {iters} outer LM iterations / {solves} solves in {t2 - t1:.1f} s on "
                  f"{workers} worker processes = {workers} of the host's {info.get('physical_cores')} physical cores ({info.get('host_cpus')} "
                  f"logical CPUs, affinity {info.get('affinity_cpus')}, cgroup quota {info.get('cgroup_quota_cpus')}); one point shard and 1 BLAS "
                  f"thread per worker; setup {t1 - t0:.1f} s not counted",
        "rmse_start": float(np.sqrt(E0 / sc.n_obs)), "rmse_end": float(np.sqrt(E / sc.n_obs)), **info,
    }
    if config2:
        try:
            from lib.synthetic import make_scene
            from oracle import ba_dense as Dn

            s2 = make_scene(10_000, 20, vis_p=1.0)
            x, vis = s2.dense()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                ba = Dn.DenseBundleAdjuster(x, s2.init_X, s2.init_K, s2.init_R, s2.init_t, axis=s2.axis)
                ba.optimize(2.0, -1.0, max_iter=2)
            dt = time.perf_counter() - t0
            out["dense_faithful_config2"] = {"it_per_s": 2 / dt, "mobs_per_s": 2 * s2.n_obs / dt / 1e6, "iterations": 2,
                                             "seconds": dt, "workload": "10k points x 20 cameras, full visibility",
                                             "kind": "oracle/ba_dense.py (dense broadcast algorithm of ref :103-162), BLAS threads as listed"}
        except Exception as exc:  # noqa: BLE001
            out["dense_faithful_config2"] = {"error": repr(exc)}
    out["config1_default_scene"] = config1_default_scene()
    return out


def config1_default_scene():
    """the reference's default scene (200 points x 10 cameras, the
    committed fixture tests/golden/euclid_default.npz captured from the reference), full
    optimize(2.0, 1e-8, max_iter=100) on the dense-faithful NumPy oracle and on the HIP engine; both
    must land on the reference's 37 outer iterations / 59 solves and RMSE 0.0063291001035384233."""
    try:
        from lib.bundle_adjustment import BundleAdjuster
        from oracle import ba_dense as Dn

        d = np.load(os.path.join(ROOT, "tests", "golden", "euclid_default.npz"), allow_pickle=False)
        args = (d["x"], d["init_X"], d["init_K"], d["init_R"], d["init_t"])
        res = {"workload": "200 points x 10 cameras (tests/golden/euclid_default.npz), optimize(2.0, 1e-8, max_iter=100)",
               "expected": {"outer_iterations": 37, "solves": 59, "rmse": 0.0063291001035384233}}
        for name, make in (("cpu_dense_faithful", lambda: Dn.DenseBundleAdjuster(*args, axis="x-up_z-forward")),
                           ("gpu", lambda: BundleAdjuster(*args, axis="x-up_z-forward"))):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    ba = make()
                    t0 = time.perf_counter()
                    ba.optimize(2.0, 1e-8, max_iter=100, is_debug=True)
                    dt = time.perf_counter() - t0
                log = ba.get_log()
                eng = getattr(ba, "_engine", None) or ba.engine
                n_outer = len(log) - 1
                res[name] = {"seconds": dt, "it_per_s": n_outer / dt, "outer_iterations": n_outer, "solves": int(eng.n_solves),
                             "rmse": float(np.sqrt(log[-1]["reprojection_error"] / 2000.0))}
            except Exception as exc:  # noqa: BLE001  (no GPU: the CPU leg still reports)
                res[name] = {"error": repr(exc)}
        return res
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def launcher_command(argv, n_gpus, port):
    """The command `python bench.py --gpus N ...` starts when no launcher started IT: one process
    per GPU under torch.distributed.run, same arguments (the driver's contract form)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv, n_gpus):
    """Run the N-rank job as a child process and forward its one JSON line.  Called before torch is
    imported: this process never initialises the GPU (a process that has must not exec another)."""
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(argv, n_gpus, free_port())
    print("bench.py: no WORLD_SIZE in the environment, starting", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        try:
            obj = json.loads(ln)
        except ValueError:
            obj = None
        if isinstance(obj, dict) and "metric" in obj:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=None, help="global points (default: 1M at N=1, 10M at N>1)")
    ap.add_argument("--cams", type=int, default=None)
    ap.add_argument("--vis", type=float, default=None)
    ap.add_argument("--weak", action="store_true", help="N>1: a config-3-sized shard per GPU (the default; kept for old command lines)")
    ap.add_argument("--strong", action="store_true", help="N>1: BASELINE config 4 (10M points x 500 cameras x 5 %%) split N ways instead")
    ap.add_argument("--config4", action="store_true",
                    help="N=1: run config 4 (10M points x 500 cameras x 5 %%, 250M observations) on the one GPU -- the N=1 point of "
                         "the config-4 scaling curve; needs ~80 GB of HBM and a few minutes of index building")
    ap.add_argument("--config4-shard", action="store_true",
                    help="N=1: the per-GPU shard of config 4 when it is split over 8 GPUs (1.25M points x 500 cameras x 5 %%, 31M "
                         "observations, D = 4493) on the one GPU: per-kernel ms per solve and the roofline of its dominant kernel")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-workers", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--svd-rows", type=int, default=5_000_000, help="config-5 SVD rows (0 = skip)")
    ap.add_argument("--depth-rows", type=int, default=5_000_000, help="points of the projective-depth iteration leg (0 = skip; 5 M x 8 images fp64 is the size DESIGN.md quotes)")
    ap.add_argument("--no-config4-shard-leg", action="store_true",
                    help="N=1 default run: skip the short leg on config 4's per-GPU shard (~6 s: scene 1.7 s, create 0.3 s, 4 LM iterations)")
    ap.add_argument("--transport", choices=("rccl", "host"), default="rccl",
                    help="N>1: rccl = one GPU per rank, ncclAllReduce inside libmvba (default); host = the reduced system is "
                         "staged through the host and summed over gloo, ranks may share a GPU (one-GPU rehearsal of the N>1 path)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    # Rank 0's stdout must carry ONE JSON line and nothing else, but libraries write there too (RCCL
    # prints a five-line version banner at communicator creation): everything this process and its
    # libraries print goes to stderr, the JSON line alone to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    host_transport = args.transport == "host"
    if not host_transport and local_rank >= max(torch.cuda.device_count(), 1):
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU of its own ({torch.cuda.device_count()} visible) and RCCL "
                         "refuses two ranks on one device; `--transport host` rehearses the N > 1 path with ranks sharing GPUs")
    device = local_rank % max(torch.cuda.device_count(), 1) if host_transport else local_rank
    torch.cuda.set_device(device)
    # MVBA_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL process group, communicator inside
    # libmvba, all-reduce per solve) even at world size 1 -- the rehearsal a one-GPU box allows.
    multi = world > 1 or os.environ.get("MVBA_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if host_transport:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    cdev = "cpu" if host_transport else "cuda"  # where the bench's own scalars are reduced

    from lib import _distributed, _mvba
    from lib.bundle_adjustment import BundleAdjuster, LevenbergMarquardt
    from lib.synthetic import make_scene, scene_shard

    config4 = (world > 1 and args.strong and not args.weak) or args.config4
    shard4 = args.config4_shard and world == 1 and not args.config4
    n_cams = args.cams or (500 if config4 or shard4 else 100)
    vis = args.vis or (0.05 if config4 or shard4 else 0.1)
    if shard4 and not args.points:
        args.points = 1_250_000
    if config4:
        n_total = args.points or 10_000_000
        lo, hi = scene_shard(n_total, n_cams, vis, rank, world)  # observation-balanced contiguous point ranges
        scaling, cfg_name = "strong", "BASELINE config 4"
    else:
        per = args.points or 1_000_000
        n_total, lo, hi = per * world, rank * per, (rank + 1) * per
        scaling, cfg_name = "weak", "BASELINE config 3" + (" shard per GPU" if world > 1 else "")
        if shard4:
            cfg_name = "BASELINE config 4, ONE of its 8 point shards on one GPU (no exchange)"
        elif args.points or args.cams or args.vis:
            cfg_name = "custom scene (not a BASELINE config)" + (", one shard per GPU" if world > 1 else "")
    t_gen = time.perf_counter()
    sc = make_scene(n_total, n_cams, vis_p=vis, point_range=(lo, hi))
    t_gen = time.perf_counter() - t_gen
    t_create = time.perf_counter()
    ba = BundleAdjuster.from_observations(sc.n_points, n_cams, sc.pt_ptr, sc.cam_idx, sc.xy, sc.init_X, sc.init_K,
                                          sc.init_R, sc.init_t, axis=sc.axis, device=device)
    t_create = time.perf_counter() - t_create
    eng = ba._engine
    if multi:
        (_distributed.attach_host_comm if host_transport else _distributed.attach_rccl)(eng)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # An "episode" is optimize(2.0, -1.0, max_iter) from the synthetic initial state: never stops on
    # tolerance.  On this scene the first 12 outer iterations accept their first trial; after that
    # the optimisation has converged to the noise floor and most iterations need two solves.  So that
    # `value` does not depend on where K falls, an episode is cut at EPISODE iterations and the next
    # one starts from the same initial state (mvba_snapshot_restore of the device-resident initial state +
    # one cost pass, inside the timed region).  The default W + K = 12 never restarts.
    EPISODE = 12
    state0 = eng.get_params()
    eng.snapshot_clear()
    eng.snapshot()  # the initial state, kept in device memory: an episode restart is a device-to-device copy
    lm = LevenbergMarquardt(eng, 2.0)
    E0 = lm.E
    n_restarts = 0

    def one_step():
        nonlocal lm, n_restarts
        if lm.count == EPISODE:
            eng.snapshot_restore(0)
            lm = LevenbergMarquardt(eng, 2.0)
            n_restarts += 1
        E_, _d = lm.iterate()
        lm.carry_on(E_)
        return E_

    # (as `timeit` does: no cyclic-GC pass inside the timed region -- with torch imported a full collection walks
    # ~10^6 objects and shows up as one 50-60 ms step in a run of 2.5 ms steps, about once in a hundred steps.  The
    # explicit collection comes BEFORE the warm-up: it leaves the CPU's caches full of everything but the launch path,
    # and the steps right after it were 0.1-0.3 ms slower on the host side)
    gc.collect()
    gc.disable()
    # Device timers (hipEvents on the engine's stream) in the timed region: the two kernels a roofline is quoted for
    # (K3, K1) only -- every timed phase is two marker packets on the stream, and with all eight phases timed a step
    # was ~0.07 ms (3 %) longer.  The full per-kernel table comes from a separate, untimed pass right after.
    eng.set_profiling(2)  # (before the warm-up: its steps also create the hipEvents the timers recycle)
    for _ in range(args.warmup):
        one_step()
    eng.reset_stats()
    solves0 = eng.n_solves
    fence()
    t0 = time.perf_counter()
    restarts0 = n_restarts
    step_end = []
    for _ in range(args.steps):
        E_ = one_step()  # (ends with the cost on the host: the accept / reject decision needs it)
        step_end.append(time.perf_counter())
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    step_ms = np.diff(np.array([t0] + step_end)) * 1e3  # host clock per step: shows a stalled step next to the mean
    if os.environ.get("MVBA_BENCH_STEP_DUMP") and rank == 0:
        print("step_ms:", " ".join(f"{v:.3f}" for v in step_ms), file=sys.stderr)
    st = eng.stats()
    n_solves = eng.n_solves - solves0
    # every phase timed, outside the timed region: TABLE_STEPS more steps of the same loop
    TABLE_STEPS = 6
    eng.set_profiling(True)
    one_step()  # (creates the additional events)
    eng.reset_stats()
    solves_t0 = eng.n_solves
    for _ in range(TABLE_STEPS):
        one_step()
    st_table = eng.stats()
    table_solves = eng.n_solves - solves_t0
    eng.set_profiling(False)

    # The same schedule through the PUBLIC surface: BundleAdjuster.optimize(2.0, -1.0, 10)
    # wall time -- includes the per-iteration print, the final get_params, the way back to the input
    # frame and the set_params the reference's optimize ends with (ref :198-202).
    eng.set_params(*state0)
    API_ITERS = 10
    fence()
    ta = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        ba.optimize(2.0, -1.0, max_iter=API_ITERS)
    fence()
    t_api = time.perf_counter() - ta

    if multi:
        tt = torch.tensor([dt, t_api], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, t_api = float(tt[0].item()), float(tt[1].item())
        no = torch.tensor([sc.n_obs], dtype=torch.float64, device=cdev)
        dist.all_reduce(no)
        n_obs_total = int(no.item())
    else:
        n_obs_total = sc.n_obs

    if rank == 0:
        per = {k: v["ms"] / max(v["launches"], 1) for k, v in st.items() if k != "counts"}
        k1_ms, k3_ms = per["resid_jac"], per["schur"]
        # K1 algorithmic bytes per launch on this rank: in xy 16 + cam 4 + point id 4,
        # out ONE 128-B record per observation (the 2x9 block's t and (u,v) columns are implied),
        # + per point 24 B in (X) and 72 B out (E_a, dP_a: K2 is fused into K1).
        # (SURVEY 8d's 232 B/obs assumed the 208-B materialised 2x9 form and a separate K2.)
        k1_bytes = 152 * sc.n_obs + 96 * sc.n_points
        k1_ach = k1_bytes / (k1_ms * 1e-3) / 1e9
        k1_traffic, k1_src = pmc_traffic("k_resid_jac", sc.n_obs)
        # K3 (the dominant kernel): SURVEY 8d's algorithmic read is 192 B/observation (J_X 48 + J_C 144);
        # what the pair-major kernel actually gathers is one 128-B record line per side of every
        # (point, camera pair) item + the point block, and its arithmetic is
        # ~100 fp64 FMA per lane-step of 21 items x 3 lanes.
        ms_solve = dt / max(n_solves, 1) * 1e3
        info = eng.schur_info()
        n_cu = 256
        try:
            n_cu = int(torch.cuda.get_device_properties(device).multi_processor_count)
        except Exception:  # noqa: BLE001
            pass
        info.update(n_cams=n_cams, n_points=sc.n_points)
        roof_k3 = schur_roofline(info, sc.n_obs, k3_ms, n_cu)
        roof_k1 = {"kernel": "k_resid_jac (K1+K2)", "bound": "hbm", "achieved": k1_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": k1_ach / HBM_PEAK_GBS, "traffic": k1_traffic, "traffic_source": k1_src,
                   "traffic_unit": "bytes per launch (PMC, committed profile -- not measured in this run)",
                   "algorithmic_bytes_per_launch": k1_bytes, "algorithmic_bytes_per_obs": 152, "avg_launch_ms": k1_ms}
        # One inner solve, algorithmic HBM bytes.  THIS build's kernels: K1 152 B/obs + 96 B/point (one 128-B record,
        # K2 fused), K3a 152 B/point, K3 192 B/obs, K5 4 B/obs + 176 B/point, K6 24 B/obs + 24 B/point.  SURVEY 8d's 824 B/obs
        # (K1 232 + K2 208 + K3 192 + K5 192: the materialised 2x9 Jacobian form, which this build never writes) kept beside it.
        own_bytes = (152 + 192 + 4 + 24) * sc.n_obs + (96 + 152 + 176 + 24) * sc.n_points
        step_bytes = 824 * sc.n_obs
        rmse = float(np.sqrt(E_ / n_obs_total))
        out = {
            "metric": "BA iterations/sec + residual-Jacobian GObs/s, 1M pts x 100 cams fp64",
            "value": (world if scaling == "weak" else 1) * args.steps / dt,
            "unit": "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{cfg_name}: {n_total} points x {n_cams} cameras, {vis:.0%} visibility, fp64, "
                            f"{n_obs_total} observations in total ({sc.n_obs} on rank 0), point-sharded over {world} GPU(s); "
                            f"LM optimize(2.0, -1.0, max_iter) schedule",
                "points_total": n_total, "points_rank0": sc.n_points, "cameras": n_cams, "visibility": vis,
                "observations_total": n_obs_total, "reduced_system_dim": 9 * n_cams - 7,
                "value_definition": "outer LM iterations per second of the whole job"
                                    + (" x point shards (weak scaling: every rank holds a config-3-sized shard; one unit = one LM iteration over one "
                                       "such shard, value = the units all ranks processed / time)" if scaling == "weak" and world > 1 else ""),
                "joint_it_per_s": args.steps / dt,
                "episode_iterations": EPISODE, "episode_restarts_in_timed_region": n_restarts - restarts0,
                "scene_generation_s": t_gen, "engine_create_s": t_create,
                "parallelism": f"point shards x{world}" + ("" if not multi else (", host-staged all-reduce over gloo" if host_transport
                                                                                 else ", RCCL all-reduce")),
                "transport": (args.transport if multi else None),
                "rccl": (eng.rccl_version() if multi and not host_transport else
                         ({"loaded": None, "compiled_against": None, "ranks": world} if multi else None)),
                "ranks_per_device": (world / max(torch.cuda.device_count(), 1) if host_transport and multi else 1),
            },
            # C1: one all-reduce of the packed [A|b] per inner solve (+ the 16-byte cost/status all-gather)
            "allreduce": ({"ms_per_solve": st_table["allreduce"]["ms"] / max(table_solves, 1),
                           "bytes_per_solve": 8 * (81 * n_cams * (n_cams + 1) // 2 + 9 * n_cams), "ranks": world,
                           "transport": args.transport} if multi else None),
            "allreduce_model": allreduce_model(n_cams, max(world, 2)) if (multi or config4 or shard4) else None,
            "weak_scaling_model": weak_scaling_model(n_cams, ms_solve) if (world == 1 and scaling == "weak" and not shard4) else None,
            "resid_jac_gobs_per_s": n_obs_total / (k1_ms * 1e-3) / 1e9,
            "inner_solves": n_solves,
            "ms_per_inner_solve": ms_solve,
            "rmse_start": float(np.sqrt(E0 / n_obs_total)), "rmse_end": rmse,
            "optimize_api": {"call": f"BundleAdjuster.optimize(2.0, -1.0, max_iter={API_ITERS})", "wall_s": t_api,
                             "it_per_s": API_ITERS / t_api},
            "roofline": roof_k3 if k3_ms >= k1_ms else roof_k1,
            "roofline_resid_jac": roof_k1,
            "roofline_schur": roof_k3,
            "step_roofline": {"bound": "hbm", "algorithmic_bytes_per_solve": own_bytes,
                              "bytes_definition": "this build's kernels: 372 B/observation + 448 B/point (K1 152+96/pt, K3a 152/pt, K3 192, K5 4+176/pt, K6 24+24/pt)",
                              "achieved": own_bytes / (ms_solve * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": own_bytes / (ms_solve * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "survey_8d_notional": {"bytes_per_obs": 824, "algorithmic_bytes_per_solve": step_bytes,
                                                     "achieved": step_bytes / (ms_solve * 1e-3) / 1e9,
                                                     "frac": step_bytes / (ms_solve * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                     "note": "SURVEY 8d's minimum-traffic figure for a build that materialises the 2x9 Jacobian "
                                                             "(K1 232 + K2 208 + K3 192 + K5 192 B/obs); this build moves fewer bytes, so this "
                                                             "fraction overstates its HBM use"}},
            "step_ms": {"min": float(step_ms.min()), "median": float(np.median(step_ms)), "max": float(step_ms.max()),
                        "argmax": int(step_ms.argmax())},
            "kernel_ms_per_step": {k: (st[k]["ms"] / args.steps if k in ("resid_jac", "schur") else v["ms"] / TABLE_STEPS)
                                   for k, v in st_table.items() if k != "counts"},
            "kernel_ms_per_solve": {k: (st[k]["ms"] / max(n_solves, 1) if k in ("resid_jac", "schur") else v["ms"] / max(table_solves, 1))
                                    for k, v in st_table.items() if k != "counts"},
            "kernel_ms_per_step_source": f"resid_jac and schur: hipEvents inside the timed region; the other phases: a separate pass of "
                                         f"{TABLE_STEPS} steps right after it with every phase timed (timing all of them costs ~3 % of a step)",
        }
        if not args.no_cpu_baseline and world == 1:  # CPU baseline and SVD leg: rank 0 at N = 1 only
            eng.close()
            out["cpu_baseline"] = cpu_baseline(sc, n_cams, iters=args.cpu_iters, workers=args.cpu_workers)
            if args.svd_rows > 0:
                out["factorization_svd_config5"] = svd_config5(args.svd_rows)
            if args.depth_rows > 0:
                out["depth_iteration"] = depth_iteration(args.depth_rows)
            if args.svd_rows > 0:
                out["factorization_svd_wide"] = svd_wide()
        if world == 1 and not args.no_config4_shard_leg and not (shard4 or config4 or args.points or args.cams or args.vis):
            eng.close()  # (idempotent) the config-3 engine's ~4 GB go back before the shard's ~11 GB are taken
            try:
                out["config4_shard"] = config4_shard_leg(device)
            except Exception as exc:  # noqa: BLE001
                out["config4_shard"] = {"error": repr(exc)}
            try:
                out["dense_visibility"] = dense_visibility_leg(device)
            except Exception as exc:  # noqa: BLE001
                out["dense_visibility"] = {"error": repr(exc)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
