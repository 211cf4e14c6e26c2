"""Worker for tests/test_gpu_distributed.py: the HIP engine in a world-size-2 job on ONE GPU over
the host-staged transport (gloo all-reduce through mvba_comm_init_host; RCCL refuses two ranks on
one device).  Exercises what an RCCL job exercises except the wire: partial [A|b] summed across
ranks, rank-ordered cost sum, status flags OR-ed so that errors and the LU rescue are collective."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3d-reconstruction-from-multi-view-exp_amd"), ROOT):
    sys.path.insert(0, p)

import torch.distributed as dist  # noqa: E402

from lib import _distributed as D  # noqa: E402
from lib import _mvba  # noqa: E402
from lib.bundle_adjustment import lm_loop  # noqa: E402
from lib.synthetic import make_scene  # noqa: E402
from oracle import ba_oracle as O  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m = 12
    sc = make_scene(6000, m, vis_p=0.4, project="numpy")
    lo, hi = D.partition_points(sc.pt_ptr, world)[rank]
    pt_ptr, cam, xy = D.slice_observations(sc.pt_ptr, sc.cam_idx, sc.xy, lo, hi)
    X, R, t = O.normalize_scene(sc.init_X, sc.init_R, sc.init_t, sc.axis)
    f, u = sc.init_K[:, 0, 0], sc.init_K[:, :2, 2]
    eng = _mvba.HipEngine(hi - lo, m, pt_ptr, cam, xy, 1.0, sc.axis)
    D.attach_host_comm(eng)
    eng.set_params(X[lo:hi], f, u, t, R)
    E0 = eng.cost()
    E = lm_loop(eng, 2.0, -1.0, 4, verbose=False)
    Xs, fs, us, ts, Rs = eng.get_params()
    # the same problem unsharded, same GPU, no communicator
    one = _mvba.HipEngine(sc.n_points, m, sc.pt_ptr, sc.cam_idx, sc.xy, 1.0, sc.axis)
    one.set_params(X, f, u, t, R)
    assert abs(E0 - one.cost()) <= 1e-12 * E0
    E1 = lm_loop(one, 2.0, -1.0, 4, verbose=False)
    X1, f1, u1, t1, R1 = one.get_params()
    assert eng.n_solves == one.n_solves, (eng.n_solves, one.n_solves)
    assert abs(E - E1) <= 1e-9 * E1, (E, E1)
    np.testing.assert_allclose(Xs, X1[lo:hi], atol=1e-9)
    np.testing.assert_allclose(Rs, R1, atol=1e-10)
    np.testing.assert_allclose(ts, t1, atol=1e-10)
    # every rank holds bitwise-identical cameras (redundant deterministic solve, no broadcast)
    import torch
    mine = torch.from_numpy(np.concatenate([fs, us.ravel(), ts.ravel(), Rs.ravel()]))
    other = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    for o in other:
        assert torch.equal(o, mine)

    # collective errors: a zero-degree point on the LAST rank only -> LinAlgError on EVERY rank, nobody hangs
    deg = np.diff(pt_ptr)
    if rank == world - 1:
        keep = np.ones(len(cam), bool)
        keep[pt_ptr[5]:pt_ptr[6]] = False  # point 5 of this shard loses all its observations
        p2 = np.concatenate([[0], np.cumsum(np.where(np.arange(len(deg)) == 5, 0, deg))])
        bad = _mvba.HipEngine(hi - lo, m, p2, cam[keep], xy[keep], 1.0, sc.axis)
    else:
        bad = _mvba.HipEngine(hi - lo, m, pt_ptr, cam, xy, 1.0, sc.axis)
    D.attach_host_comm(bad)
    bad.set_params(X[lo:hi], f, u, t, R)
    bad.cost()
    bad.linearize()
    raised = False
    try:
        bad.try_step(1e-4)
    except np.linalg.LinAlgError:
        raised = True
    assert raised, f"rank {rank} did not see the other rank's singular point block"

    # collective LU rescue: a negative damping makes the reduced system indefinite on every rank
    eng.set_params(X[lo:hi], f, u, t, R)
    one.set_params(X, f, u, t, R)
    eng.linearize(); one.linearize()
    Ea, Eb = eng.try_step(-1.5), one.try_step(-1.5)
    assert eng.stats()["counts"]["lu_fallback"] == one.stats()["counts"]["lu_fallback"] == 1
    assert abs(Ea - Eb) <= 1e-6 * abs(Eb), (Ea, Eb)
    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK", eng.n_solves, E)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
