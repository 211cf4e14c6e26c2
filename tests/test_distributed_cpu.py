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
