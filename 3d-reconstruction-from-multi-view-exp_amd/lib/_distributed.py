"""Point-sharded multi-GPU bundle adjustment: one process per GPU.

Points are independent given the cameras, so each rank owns a contiguous range
of point ids (balanced by observation count), a replica of the camera
parameters, and its slice of the observation list.  The only data-path exchange
is ONE all-reduce of the packed reduced camera system ``[A | b]`` per LM solve,
done inside ``libmvba.so`` with RCCL on the engine's own stream
(``mvba_comm_init``); trial costs are all-gathered (1 double per rank) and summed
in rank order so every rank takes the identical accept/reject decision.

``torch.distributed`` is used here only to bootstrap (ship RCCL's 128-byte
unique id) and by callers for barriers/timing - never on the data path.
"""
from __future__ import annotations

import numpy as np


def partition_points(pt_ptr: np.ndarray, n_parts: int) -> list[tuple[int, int]]:
    """Contiguous point ranges with ~equal observation counts (CSR prefix sums)."""
    pt_ptr = np.asarray(pt_ptr, dtype=np.int64)
    n = len(pt_ptr) - 1
    total = int(pt_ptr[-1])
    cuts = [0]
    for r in range(1, n_parts):
        target = total * r // n_parts
        a = int(np.searchsorted(pt_ptr, target, side="left"))
        cuts.append(min(max(a, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(n_parts)]


def slice_observations(pt_ptr, cam_idx, xy, lo, hi):
    """The CSR slice for points [lo, hi)."""
    o0, o1 = int(pt_ptr[lo]), int(pt_ptr[hi])
    return (np.asarray(pt_ptr[lo:hi + 1], dtype=np.int64) - o0, np.asarray(cam_idx[o0:o1]), np.asarray(xy[o0:o1]))


def broadcast_bytes(payload: bytes | None, n: int, src: int = 0, group=None) -> bytes:
    """Ship ``n`` bytes from ``src`` to every rank over torch.distributed (any backend)."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = torch.zeros(n, dtype=torch.uint8, device=dev)
    if dist.get_rank(group) == src:
        buf.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(buf, src=src, group=group)
    return bytes(buf.cpu().numpy().tobytes())


def attach_rccl(engine, group=None):
    """Give a HipEngine its RCCL communicator (collective: call on every rank)."""
    import torch.distributed as dist

    from . import _mvba

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = _mvba.comm_unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid, 128, 0, group)
    engine.comm_init(uid, rank, world)
    return rank, world


def attach_host_comm(engine, group=None):
    """Same job over the host-staged transport (``mvba_comm_init_host``): the packed reduced system
    goes through a torch.distributed all-reduce on the host (gloo).  For hosts without RCCL and for
    multi-process runs that share one GPU."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    engine.comm_init_host(rank, world, numpy_allreduce(group))
    return rank, world


def numpy_allreduce(group=None):
    """In-place float64 sum over ranks for host arrays (gloo); used to drive the
    CPU oracle engine through the same sharding logic in tests."""
    import torch
    import torch.distributed as dist

    def allreduce(a: np.ndarray):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    return allreduce


class InProcessGroup:
    """N ranks as N THREADS of one process sharing a GPU -- the rehearsal of an N-rank job a one-GPU box allows when
    N processes on one card are not (this pool admits six).  Every rank owns an engine (its own stream) and runs the
    same LM loop; `allreduce(rank)` is the callback for ``mvba_comm_init_host``: each rank deposits its array, all
    wait, every rank sums the N deposits IN RANK ORDER into its own array (so all ranks hold bitwise the same sum,
    as after an RCCL / gloo all-reduce), all wait again.  ctypes releases the GIL around library calls and the
    barrier waits release it too, so the ranks' kernels overlap on the device as separate processes' would."""

    def __init__(self, n_ranks: int, timeout: float = 600.0):
        import threading

        self.n = int(n_ranks)
        self._slots = [None] * self.n
        self._barrier = threading.Barrier(self.n, timeout=timeout)
        self.bytes_reduced = [0] * self.n
        self.calls = [0] * self.n

    def barrier(self):
        self._barrier.wait()

    def abort(self):
        self._barrier.abort()

    def allreduce(self, rank: int):
        def fn(a: np.ndarray):
            self._slots[rank] = a
            self._barrier.wait()
            total = self._slots[0].copy()
            for r in range(1, self.n):
                total += self._slots[r]
            self._barrier.wait()  # everybody has read every deposit before anybody overwrites its own
            a[...] = total
            self.bytes_reduced[rank] += a.nbytes
            self.calls[rank] += 1

        return fn

    def attach(self, engine, rank: int):
        engine.comm_init_host(rank, self.n, self.allreduce(rank))

    def run(self, body):
        """body(rank, group) on N threads; returns the N results in rank order.  An exception on one rank aborts the
        barrier (the others raise BrokenBarrierError instead of waiting for ever) and is re-raised here."""
        import threading

        out, err = [None] * self.n, [None] * self.n

        def wrap(r):
            try:
                out[r] = body(r, self)
            except BaseException as exc:  # noqa: BLE001
                err[r] = exc
                self._barrier.abort()

        th = [threading.Thread(target=wrap, args=(r,), name=f"rank{r}") for r in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        first = next((e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
        if first is None:
            first = next((e for e in err if e is not None), None)
        if first is not None:
            raise first
        return out
