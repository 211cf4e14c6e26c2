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


def test_bench_gpus_2_starts_its_own_ranks_over_the_host_transport():
    """The driver's command form, `python bench.py --gpus N ...` with no launcher: bench.py starts the
    N ranks itself.  On the one-GPU box the two ranks share the device (--transport host) at a
    reduced point count: the whole N > 1 bench path except the RCCL wire."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--points", "200000",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE JSON line on stdout, nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["transport"] == "host" and d["config"]["rccl"]["ranks"] == 2
    assert d["allreduce"]["ranks"] == 2 and d["allreduce"]["ms_per_solve"] > 0
    assert d["allreduce"]["bytes_per_solve"] == 8 * (81 * 500 * 501 // 2 + 9 * 500)
    assert d["config"]["points_total"] == 200000 and 0 < d["config"]["points_rank0"] < 200000
    assert d["rmse_end"] < d["rmse_start"]
