"""N > 1 path on CPU: world_size-2 gloo run of the point-sharding logic with the
oracle engine standing in for the GPU engine; the sharded LM run must equal the unsharded one."""
import os
import subprocess
import sys

import numpy as np

from lib import _distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_points_balances_observations():
    rng = np.random.default_rng(0)
    deg = rng.integers(3, 30, 10_000)
    pt_ptr = np.concatenate([[0], np.cumsum(deg)])
    for parts in (1, 2, 3, 8):
        ranges = D.partition_points(pt_ptr, parts)
        assert len(ranges) == parts and ranges[0][0] == 0 and ranges[-1][1] == 10_000
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        loads = [pt_ptr[hi] - pt_ptr[lo] for lo, hi in ranges]
        assert max(loads) - min(loads) <= 2 * deg.max()
    lo, hi = D.partition_points(pt_ptr, 4)[2]
    p, c, x = D.slice_observations(pt_ptr, np.arange(pt_ptr[-1]), np.zeros((pt_ptr[-1], 2)), lo, hi)
    assert p[0] == 0 and p[-1] == len(c) == pt_ptr[hi] - pt_ptr[lo] and c[0] == pt_ptr[lo]


def test_sharded_lm_over_gloo_world_size_2():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_OK" in out.stdout


def test_sharded_lm_eight_ranks_as_threads_over_the_oracle():
    """The 8-way split with the ranks as threads of one process (lib._distributed.InProcessGroup: what rehearses an
    8-rank job on a one-GPU box, tools/rehearse_ranks.py), the oracle engine standing in for the GPU engine: rank-ordered
    sums make every rank's cameras bitwise identical, and the run equals the unsharded one."""
    from lib.bundle_adjustment import lm_loop
    from lib.synthetic import make_scene
    from oracle import ba_oracle as O

    W = 8
    sc = make_scene(800, 7, vis_p=0.5)
    X, R, t = O.normalize_scene(sc.init_X, sc.init_R, sc.init_t, sc.axis)
    f, u = sc.init_K[:, 0, 0], sc.init_K[:, :2, 2]
    g1 = O.OracleEngine(sc.n_points, 7, sc.pt_ptr, sc.cam_idx, sc.xy, 1.0, sc.axis)
    g1.set_params(X, f, u, t, R)
    E1 = lm_loop(g1, 2.0, -1.0, 3, verbose=False)
    parts = D.partition_points(sc.pt_ptr, W)

    def body(rank, grp):
        lo, hi = parts[rank]
        pt_ptr, cam, xy = D.slice_observations(sc.pt_ptr, sc.cam_idx, sc.xy, lo, hi)
        g = O.OracleEngine(hi - lo, 7, pt_ptr, cam, xy, 1.0, sc.axis, allreduce=grp.allreduce(rank))
        g.set_params(X[lo:hi], f, u, t, R)
        E = lm_loop(g, 2.0, -1.0, 3, verbose=False)
        return E, np.concatenate([v.ravel() for v in g.get_params()[1:]]), g.n_solves

    res = D.InProcessGroup(W).run(body)
    for E, cams, solves in res:
        assert abs(E - E1) <= 1e-10 * E1 and solves == g1.n_solves
        np.testing.assert_array_equal(cams, res[0][1])
    # an exception on one rank reaches the caller instead of leaving the others in the barrier
    import pytest

    def boom(rank, grp):
        if rank == 3:
            raise ValueError("rank 3 fails")
        grp.barrier()

    with pytest.raises(ValueError, match="rank 3 fails"):
        D.InProcessGroup(W, timeout=30).run(boom)
