"""Rehearsal of an N-rank point-sharded job on ONE GPU: N ranks as N threads of this process (lib._distributed.
InProcessGroup -- this pool admits six processes on a card, so eight ranks cannot be eight processes), each with its
own engine on its own shard, the packed reduced system [A|b] and the 16-byte cost/status record exchanged through the
library's host-staged transport (mvba_comm_init_host) and summed in rank order.  Everything an RCCL job runs except the
wire.  Checked against the SAME scene on one engine (N = 1): cost after every LM iteration to 1e-9 relative, equal solve
counts, cameras bitwise identical on all ranks and equal to the N = 1 cameras to 1e-9.

    python tools/rehearse_ranks.py --ranks 8 [--points 10000000 --cams 500 --vis 0.05 --iters 3] [--out file.json]

Default size = BASELINE config 4 in full (8 x ~10.6 GB + the N = 1 engine's ~85 GB, one after the other)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT]
from lib import _distributed as D  # noqa: E402
from lib import _mvba  # noqa: E402
from lib.bundle_adjustment import LevenbergMarquardt, to_gauge_frame  # noqa: E402
from lib.synthetic import make_scene, scene_shard  # noqa: E402


def lm_run(eng, iters, after=None):
    lm = LevenbergMarquardt(eng, 2.0)
    costs = [lm.E]
    t0 = time.perf_counter()
    for _ in range(iters):
        E_, _d = lm.iterate()
        lm.carry_on(E_)
        costs.append(E_)
        if after:
            after()
    return costs, time.perf_counter() - t0


def engine_for(sc, n_cams):
    X, R, t = to_gauge_frame(sc.init_X, sc.init_R, sc.init_t, sc.axis)
    eng = _mvba.HipEngine(sc.n_points, n_cams, sc.pt_ptr, sc.cam_idx, sc.xy, 1.0, sc.axis)
    eng.set_params(X, sc.init_K[:, 0, 0], sc.init_K[:, :2, 2], t, R)
    return eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--cams", type=int, default=500)
    ap.add_argument("--vis", type=float, default=0.05)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    m, W = a.cams, a.ranks
    # ---- N = 1: the whole scene on one engine
    t0 = time.perf_counter()
    sc = make_scene(a.points, m, vis_p=a.vis)
    one = engine_for(sc, m)
    n_obs_total = sc.n_obs
    del sc
    c1, t1 = lm_run(one, a.iters)
    _X1, f1, u1, t1c, R1 = one.get_params()
    solves1 = one.n_solves
    one.close()
    del one, _X1
    print(f"N=1: {a.iters} iterations / {solves1} solves in {t1:.2f} s ({time.perf_counter() - t0:.0f} s with scene + create); costs {c1}", flush=True)

    # ---- N = W ranks as threads
    grp = D.InProcessGroup(W)

    def body(rank, g):
        lo, hi = scene_shard(a.points, m, a.vis, rank, W)
        s = make_scene(a.points, m, vis_p=a.vis, point_range=(lo, hi))
        eng = engine_for(s, m)
        g.attach(eng, rank)
        n_obs = s.n_obs
        del s
        g.barrier()
        eng.set_profiling(True)
        eng.reset_stats()
        costs, dt = lm_run(eng, a.iters)
        st = eng.stats()
        _X, f, u, t, R = eng.get_params()
        res = {"rank": rank, "points": hi - lo, "n_obs": n_obs, "costs": costs, "seconds": dt, "solves": eng.n_solves,
               "cams": np.concatenate([f, u.ravel(), t.ravel(), R.ravel()]),
               "ms_per_solve": {k: v["ms"] / max(v["launches"], 1) for k, v in st.items() if k != "counts"},
               "counts": st["counts"], "schur": eng.schur_info()}
        eng.close()
        return res

    tw = time.perf_counter()
    res = grp.run(body)
    tw = time.perf_counter() - tw
    cw = res[0]["costs"]
    cams0 = res[0]["cams"]
    cams1 = np.concatenate([f1, u1.ravel(), t1c.ravel(), R1.ravel()])
    checks = {
        "costs_equal_on_all_ranks": all(r["costs"] == cw for r in res),
        "cost_rel_err_vs_n1": [abs(x - y) / abs(y) for x, y in zip(cw, c1)],
        "solve_counts_equal": all(r["solves"] == solves1 for r in res),
        "cameras_bitwise_identical_on_all_ranks": all(np.array_equal(r["cams"], cams0) for r in res),
        "cameras_max_abs_diff_vs_n1": float(np.abs(cams0 - cams1).max()),
        "observations_sum_to_the_scene": sum(r["n_obs"] for r in res) == n_obs_total,
    }
    nA = 81 * m * (m + 1) // 2 + 9 * m
    calls = grp.calls[0]
    out = {
        "what": f"{W} ranks as threads of one process on one GPU (host-staged transport, in-process rank-ordered sum); config: "
                f"{a.points} points x {m} cameras x {a.vis:.0%}, {n_obs_total} observations, {a.iters} LM iterations",
        "ranks": W, "checks": checks,
        "allreduce": {"bytes_per_solve": 8 * nA, "callback_calls_rank0": calls,
                      "bytes_through_the_callback_rank0": grp.bytes_reduced[0],
                      "expected_bytes_rank0": res[0]["solves"] * 8 * nA + (calls - res[0]["solves"]) * 16 * W},
        "n1": {"costs": c1, "solves": solves1, "seconds": t1},
        "nW": {"costs": cw, "solves": res[0]["solves"], "seconds_lm": max(r["seconds"] for r in res), "seconds_with_scene_and_create": tw,
               "ms_per_solve_rank0": res[0]["ms_per_solve"], "schur_rank0": res[0]["schur"],
               "points_per_rank": [r["points"] for r in res], "obs_per_rank": [r["n_obs"] for r in res],
               "lu_fallback": [r["counts"]["lu_fallback"] for r in res], "barrier_fallback": [r["counts"]["barrier_fallback"] for r in res]},
    }
    ok = (checks["costs_equal_on_all_ranks"] and checks["solve_counts_equal"] and checks["cameras_bitwise_identical_on_all_ranks"]
          and max(checks["cost_rel_err_vs_n1"]) < 1e-9 and checks["cameras_max_abs_diff_vs_n1"] < 1e-9 and checks["observations_sum_to_the_scene"])
    out["ok"] = bool(ok)
    line = json.dumps(out)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(line + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
