"""N > 1 on the HIP engine: two processes share the one GPU of the box and exchange the reduced
camera system through the host-staged transport (mvba_comm_init_host over gloo).  The sharded LM
run must equal the unsharded one, errors and the LU rescue must be collective."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_engine_world_size_2_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_GPU_OK" in out.stdout
