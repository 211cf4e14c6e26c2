"""ctypes binding of libmvba.so (include/mvba.h) -- the only way into the HIP engine.

There is NO CPU fallback here: if the library is missing, or there is no GPU,
construction raises.  (The NumPy restatement lives in ``oracle/`` and is test
infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVBA_LIBRARY", os.path.join(os.path.dirname(_HERE), "libmvba.so"))

MVBA_OK, MVBA_ERR_BADARG, MVBA_ERR_SINGULAR, MVBA_ERR_HIP, MVBA_ERR_RCCL, MVBA_ERR_STATE = range(6)

KERNEL_IDS = ("resid_jac", "point_blocks", "point_inv", "schur", "allreduce", "solve", "backsub_cost", "cost")
BUF = {"residual": 0, "JX": 1, "JC": 2, "E": 3, "dP": 4, "A_full": 5, "b_full": 6, "dxi": 7, "dX": 8,
       "trial_X": 9, "trial_cam": 10, "index_k": 11, "index_l": 12, "index_a": 13, "index_seg": 14}

_dp = C.POINTER(C.c_double)


class Problem(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("n_obs", C.c_int64), ("n_images", C.c_int32),
                ("gauge_axis", C.c_int32), ("pt_ptr", C.POINTER(C.c_int64)),
                ("cam_idx", C.POINTER(C.c_int32)), ("xy", _dp), ("f0", C.c_double),
                ("device", C.c_int32), ("xy_layout", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("ms", C.c_double * 16), ("launches", C.c_int64 * 16),
                ("n_linearize", C.c_int64), ("n_try_step", C.c_int64), ("n_commit", C.c_int64),
                ("n_lu_fallback", C.c_int64), ("n_barrier_fallback", C.c_int64)]


# every symbol include/mvba.h declares: (restype, argtypes)
SIGNATURES = {
    "mvba_version": (C.c_char_p, []),
    "mvba_last_error": (C.c_char_p, []),
    "mvba_kernel_name": (C.c_char_p, [C.c_int32]),
    "mvba_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "mvba_create": (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_void_p)]),
    "mvba_destroy": (None, [C.c_void_p]),
    "mvba_set_params": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp]),
    "mvba_get_params": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp]),
    "mvba_apply_similarity": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double]),
    "mvba_snapshot": (C.c_int, [C.c_void_p]),
    "mvba_snapshot_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mvba_snapshot_read": (C.c_int, [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _dp]),
    "mvba_snapshot_clear": (C.c_int, [C.c_void_p]),
    "mvba_snapshot_restore": (C.c_int, [C.c_void_p, C.c_int64]),
    "mvba_cost": (C.c_int, [C.c_void_p, _dp]),
    "mvba_linearize": (C.c_int, [C.c_void_p]),
    "mvba_try_step": (C.c_int, [C.c_void_p, C.c_double, _dp]),
    "mvba_commit": (C.c_int, [C.c_void_p]),
    "mvba_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "mvba_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "mvba_reset_stats": (C.c_int, [C.c_void_p]),
    "mvba_get_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mvba_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mvba_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "mvba_comm_init_host": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mvba_debug_read": (C.c_int, [C.c_void_p, C.c_int32, _dp, C.c_int64, C.POINTER(C.c_int64)]),
    "mvba_host_obs_math": (C.c_int, [_dp, _dp, _dp, C.c_double, _dp]),
    "mvba_project": (C.c_int, [_dp, C.c_int64, _dp, _dp, _dp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                               C.c_int64, _dp, C.c_int32]),
    "mvsvd_factorize": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _dp, C.c_int32]),
    "mvsvd_create": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "mvsvd_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mvsvd_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _dp]),
    "mvsvd_load_images": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int32]),
    "mvsvd_load_base": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mvsvd_load_base_images": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_double]),
    "mvsvd_run_scaled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, _dp]),
    "mvsvd_depth_begin": (C.c_int, [C.c_void_p, C.c_int32]),
    "mvsvd_depth_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, _dp, _dp]),
    "mvsvd_depth_read": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mvsvd_destroy": (None, [C.c_void_p]),
}

_lib = None


def load_library():
    """dlopen libmvba.so (loudly) and attach the prototypes."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libmvba.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or make -C 3d-reconstruction-from-multi-view-exp_amd/csrc). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _as(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def raise_for(rc, lib=None):
    if rc == MVBA_OK:
        return
    msg = (lib or load_library()).mvba_last_error().decode()
    if rc == MVBA_ERR_SINGULAR:
        raise np.linalg.LinAlgError(msg)  # ref :128 / :146 raise numpy.linalg.LinAlgError
    if rc == MVBA_ERR_BADARG:
        raise ValueError(msg)  # ref :27-28
    raise RuntimeError(f"libmvba error {rc}: {msg}")


def device_count():
    lib = load_library()
    n = C.c_int32(0)
    rc = lib.mvba_device_count(C.byref(n))
    return n.value if rc == MVBA_OK else 0


class HipEngine:
    """Device-resident BA state + kernels.  Protocol (shared with the oracle's
    engine): set_params / get_params / cost / linearize / try_step / commit."""

    def __init__(self, n_points, n_images, pt_ptr, cam_idx, xy, f0, axis, device=-1):
        from .bundle_adjustment import AXES  # local import: avoid a cycle

        if axis not in AXES:
            raise ValueError()
        self.lib = load_library()
        if device_count() < 1:
            raise RuntimeError("libmvba: no HIP device visible; the BA engine has no CPU fallback")
        self.n, self.m = int(n_points), int(n_images)
        self._pt_ptr = _as(pt_ptr, np.int64)
        self._cam = _as(cam_idx, np.int32)
        self._xy = _as(xy, np.float64)
        planes = self._xy.ndim == 3  # (m, N, 2) image planes of a fully visible scene (xy_layout 1) instead of (n_obs, 2)
        if planes and self._xy.shape != (self.m, self.n, 2):
            raise ValueError("xy as image planes must be (n_images, n_points, 2)")
        if not planes:
            self._xy = self._xy.reshape(-1, 2)
        self.n_obs = int(self._cam.shape[0])
        prob = Problem(self.n, self.n_obs, self.m, AXES[axis],
                       self._pt_ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                       self._cam.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(self._xy), float(f0), int(device), int(planes))
        h = C.c_void_p()
        raise_for(self.lib.mvba_create(C.byref(prob), C.byref(h)), self.lib)
        self._h = h
        self.n_solves = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mvba_destroy(self._h)
            self._h = None

    __del__ = close

    def set_params(self, X, f, u, t, R):
        X, f, u, t, R = (_as(v, np.float64) for v in (X, f, u, t, R))
        assert X.shape == (self.n, 3) and f.shape == (self.m,) and u.shape == (self.m, 2)
        assert t.shape == (self.m, 3) and R.shape == (self.m, 3, 3)
        raise_for(self.lib.mvba_set_params(self._h, _ptr(X), _ptr(f), _ptr(u), _ptr(t), _ptr(R)), self.lib)

    def get_params(self):
        X = np.empty((self.n, 3)); f = np.empty(self.m); u = np.empty((self.m, 2))
        t = np.empty((self.m, 3)); R = np.empty((self.m, 3, 3))
        raise_for(self.lib.mvba_get_params(self._h, _ptr(X), _ptr(f), _ptr(u), _ptr(t), _ptr(R)), self.lib)
        return X, f, u, t, R

    # -- debug log (ref :89-98, :175-183): copies of the committed state kept on the device
    def snapshot(self):
        raise_for(self.lib.mvba_snapshot(self._h), self.lib)

    def snapshot_count(self):
        n = C.c_int64()
        raise_for(self.lib.mvba_snapshot_count(self._h, C.byref(n)), self.lib)
        return n.value

    def snapshot_read(self, i):
        X = np.empty((self.n, 3)); f = np.empty(self.m); u = np.empty((self.m, 2))
        t = np.empty((self.m, 3)); R = np.empty((self.m, 3, 3))
        raise_for(self.lib.mvba_snapshot_read(self._h, int(i), _ptr(X), _ptr(f), _ptr(u), _ptr(t), _ptr(R)), self.lib)
        return X, f, u, t, R

    def snapshot_clear(self):
        raise_for(self.lib.mvba_snapshot_clear(self._h), self.lib)

    def snapshot_restore(self, i):
        """Log entry i becomes the committed state again (set_params from device memory)."""
        raise_for(self.lib.mvba_snapshot_restore(self._h, int(i)), self.lib)

    def apply_similarity(self, R0, t0, scale):
        """Committed state -> scale * X R0^T + t0 (likewise t), R0 R, on the device (ref :242-258)."""
        R0, t0 = _as(R0, np.float64), _as(t0, np.float64)
        assert R0.shape == (3, 3) and t0.shape == (3,)
        raise_for(self.lib.mvba_apply_similarity(self._h, _ptr(R0), _ptr(t0), float(scale)), self.lib)

    def cost(self):
        E = C.c_double()
        raise_for(self.lib.mvba_cost(self._h, C.byref(E)), self.lib)
        return E.value

    def linearize(self):
        raise_for(self.lib.mvba_linearize(self._h), self.lib)

    def try_step(self, c):
        E = C.c_double()
        raise_for(self.lib.mvba_try_step(self._h, float(c), C.byref(E)), self.lib)
        self.n_solves += 1
        return E.value

    def commit(self):
        raise_for(self.lib.mvba_commit(self._h), self.lib)

    # -- measurement / multi-GPU / test hooks
    def set_profiling(self, on):
        """False / True, or 2: time the Schur and residual-Jacobian kernels only."""
        raise_for(self.lib.mvba_set_profiling(self._h, 2 if on == 2 and on is not True else int(bool(on))), self.lib)

    def reset_stats(self):
        raise_for(self.lib.mvba_reset_stats(self._h), self.lib)

    def stats(self):
        s = Stats()
        raise_for(self.lib.mvba_get_stats(self._h, C.byref(s)), self.lib)
        out = {k: {"ms": s.ms[i], "launches": s.launches[i]} for i, k in enumerate(KERNEL_IDS)}
        out["counts"] = {"linearize": s.n_linearize, "try_step": s.n_try_step, "commit": s.n_commit,
                         "lu_fallback": s.n_lu_fallback, "barrier_fallback": s.n_barrier_fallback}
        return out

    def _info(self):
        out = (C.c_int64 * 8)()
        raise_for(self.lib.mvba_get_info(self._h, out), self.lib)
        return list(out)

    def schur_info(self):
        i = self._info()
        return {"items": i[0], "offdiag_items": i[1], "units": i[2], "kernel": ("strip", "pairs", "slots", "dense")[i[3] & 0xff],
                "slot_rows": i[7],  # slot form: step-major rows incl. the padding rows of the bounded-skew merge
                "slot_rounds": (i[3] >> 8) & 0xffffff, "slot_groups": i[3] >> 32}

    def rccl_version(self):
        i = self._info()
        return {"loaded": i[4], "compiled_against": i[5], "ranks": i[6]}

    def comm_init(self, id128: bytes, rank: int, n_ranks: int):
        buf = C.create_string_buffer(bytes(id128), 128)
        raise_for(self.lib.mvba_comm_init(self._h, buf, int(rank), int(n_ranks)), self.lib)

    def comm_init_host(self, rank: int, n_ranks: int, allreduce):
        """Host-staged transport: ``allreduce(a)`` sums a float64 ndarray in place over the ranks."""
        HOSTFN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int64)

        def _cb(_user, buf, n):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(n,)))
                return 0
            except Exception:  # noqa: BLE001
                return 1

        self._host_cb = HOSTFN(_cb)  # keep the trampoline alive as long as the engine
        raise_for(self.lib.mvba_comm_init_host(self._h, int(rank), int(n_ranks), C.cast(self._host_cb, C.c_void_p), None),
                  self.lib)

    def debug_read(self, name):
        n = C.c_int64()
        raise_for(self.lib.mvba_debug_read(self._h, BUF[name], None, 0, C.byref(n)), self.lib)
        out = np.empty(n.value)
        raise_for(self.lib.mvba_debug_read(self._h, BUF[name], _ptr(out), n.value, C.byref(n)), self.lib)
        return out


def comm_unique_id() -> bytes:
    lib = load_library()
    buf = C.create_string_buffer(128)
    raise_for(lib.mvba_comm_unique_id(buf), lib)
    return buf.raw


def project(X, K, R, t, pt_ptr=None, cam_idx=None, device=-1):
    """Device pinhole projection (mvba_project).  With an observation list (pt_ptr, cam_idx):
    (n_obs, 2); without: the dense grid (N, m, 2).  No CPU fallback."""
    lib = load_library()
    if device_count() < 1:
        raise RuntimeError("libmvba: no HIP device visible; mvba_project has no CPU fallback")
    X, K, R, t = (_as(v, np.float64) for v in (X, K, R, t))
    n, m = X.shape[0], K.shape[0]
    assert X.shape == (n, 3) and K.shape == (m, 3, 3) and R.shape == (m, 3, 3) and t.shape == (m, 3)
    if pt_ptr is None:
        out = np.empty((n, m, 2))
        rc = lib.mvba_project(_ptr(X), n, _ptr(K), _ptr(R), _ptr(t), m, None, None, n * m, _ptr(out), int(device))
    else:
        pt_ptr, cam_idx = _as(pt_ptr, np.int64), _as(cam_idx, np.int32)
        out = np.empty((cam_idx.shape[0], 2))
        rc = lib.mvba_project(_ptr(X), n, _ptr(K), _ptr(R), _ptr(t), m, pt_ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                              cam_idx.ctypes.data_as(C.POINTER(C.c_int32)), cam_idx.shape[0], _ptr(out), int(device))
    raise_for(rc, lib)
    return out


def host_obs_math(X3, cam15, xy2, f0):
    lib = load_library()
    out = np.empty(26)
    X3, cam15, xy2 = _as(X3, np.float64), _as(cam15, np.float64), _as(xy2, np.float64)
    raise_for(lib.mvba_host_obs_math(_ptr(X3), _ptr(cam15), _ptr(xy2), float(f0), _ptr(out)), lib)
    return out[:2], out[2:8].reshape(2, 3), out[8:].reshape(2, 9)


def _tm(tm):
    return {"h2d_ms": tm[0], "gram_ms": tm[1], "jacobi_ms": tm[2], "project_ms": tm[3], "sweeps": int(tm[4]),
            "refine_ms": tm[5]}


class SvdWorkspace:
    """Device-resident factorization workspace (mvsvd_create / load / run / destroy): buffers,
    stream and events are made once; ``load`` is the only host-to-device copy; ``run`` may be
    called any number of times on the resident matrix."""

    def __init__(self, max_rows, n_cols, dtype, device=-1):
        self.lib = load_library()
        if device_count() < 1:
            raise RuntimeError("libmvba: no HIP device visible; the SVD kernel has no CPU fallback")
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.float32, np.float64):
            raise ValueError("dtype must be float32 or float64")
        self.max_rows, self.n_cols, self.n_rows, self.base_rows = int(max_rows), int(n_cols), 0, 0
        h = C.c_void_p()
        raise_for(self.lib.mvsvd_create(self.max_rows, self.n_cols, 0 if self.dtype == np.float32 else 1, int(device),
                                        C.byref(h)), self.lib)
        self._h = h

    def load(self, Wt):
        Wt = np.ascontiguousarray(Wt, dtype=self.dtype)
        if Wt.ndim != 2 or Wt.shape[1] != self.n_cols:
            raise ValueError("Wt must be (n_rows, n_cols) of the workspace")
        raise_for(self.lib.mvsvd_load(self._h, Wt.ctypes.data, Wt.shape[0]), self.lib)
        self.n_rows = Wt.shape[0]
        return self

    def load_images(self, x_list):
        """The matrix np.hstack(x_list) (n_rows, 2 m) put together on the device from the images' (n_rows, 2) arrays themselves
        (all float32 or all float64; converted to the workspace's dtype)."""
        src = np.result_type(*x_list)
        if src not in (np.float32, np.float64):
            raise ValueError("image arrays must be float32 or float64")
        arrs = [np.ascontiguousarray(a, dtype=src) for a in x_list]
        n_rows = arrs[0].shape[0]
        if 2 * len(arrs) != self.n_cols or any(a.shape != (n_rows, 2) for a in arrs):
            raise ValueError("x_list must hold n_cols / 2 arrays of shape (n_rows, 2)")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        raise_for(self.lib.mvsvd_load_images(self._h, ptrs, len(arrs), n_rows, 0 if src == np.float32 else 1), self.lib)
        self.n_rows = n_rows
        return self

    def load_base(self, X):
        """The resident base matrix of run_scaled (one upload for a whole depth loop)."""
        X = np.ascontiguousarray(X, dtype=self.dtype)
        if X.ndim != 2 or X.shape[1] != self.n_cols:
            raise ValueError("X must be (n_rows, n_cols) of the workspace")
        raise_for(self.lib.mvsvd_load_base(self._h, X.ctypes.data, X.shape[0]), self.lib)
        self.base_rows = X.shape[0]
        return self

    def load_base_images(self, x_list, f0):
        """The base (x / f0, y / f0, 1) per image, assembled on the device from the images' (n_rows, 2) arrays themselves."""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in x_list]
        n_rows = arrs[0].shape[0]
        if 3 * len(arrs) != self.n_cols or any(a.shape != (n_rows, 2) for a in arrs):
            raise ValueError("x_list must hold n_cols / 3 arrays of shape (n_rows, 2)")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        raise_for(self.lib.mvsvd_load_base_images(self._h, ptrs, len(arrs), n_rows, float(f0)), self.lib)
        self.base_rows = n_rows
        return self

    def run_scaled(self, z, group, norm, n_rank):
        """Factorise X o z (depths z (n_rows, n_cols / group) per column group, normalised: norm 1 = unit rows,
        2 = column groups by their squared norm) from the resident base: only z is uploaded.
        M (n_cols, r), sigma (n_cols,), S (r, n_rows), timings.  z = None: the depths a device depth loop left in the workspace."""
        if z is not None:
            z = np.ascontiguousarray(z, dtype=self.dtype)
            if z.shape != (self.base_rows, self.n_cols // int(group)):
                raise ValueError("z must be (rows of the base, n_cols / group)")
        self.n_rows = self.base_rows  # the workspace matrix becomes the re-weighted base
        M = np.empty((self.n_cols, n_rank), self.dtype)
        sigma = np.empty(self.n_cols, self.dtype)
        S = np.empty((n_rank, self.n_rows), self.dtype)
        tm = np.zeros(6)
        raise_for(self.lib.mvsvd_run_scaled(self._h, z.ctypes.data if z is not None else None, int(group), int(norm), int(n_rank), M.ctypes.data,
                                            sigma.ctypes.data, S.ctypes.data, _ptr(tm)), self.lib)
        return M, sigma, S, _tm(tm)

    # -- the projective-depth loops on the device (ref perspective_camera_calibration.py:61-144, :147-235)
    def depth_begin(self, group=3):
        """z <- 1 on the device for the resident base (homogeneous image coordinates: group = 3)."""
        raise_for(self.lib.mvsvd_depth_begin(self._h, int(group)), self.lib)

    def depth_step(self, method, f0):
        """One iteration (1 = primary, 2 = dual): re-weight, factorise, update the depths on the device.
        Returns (reprojection error, timings); nothing else crosses PCIe."""
        E = C.c_double()
        tm = np.zeros(6)
        raise_for(self.lib.mvsvd_depth_step(self._h, int(method), float(f0), C.byref(E), _ptr(tm)), self.lib)
        t = _tm(tm)
        t["depth_ms"] = t.pop("h2d_ms")  # (slot 0 of a depth step: the depth-update kernels)
        return E.value, t

    def depth_read(self):
        """The current depths (rows of the base, n_cols / 3), float64."""
        z = np.empty((self.base_rows, self.n_cols // 3), self.dtype)
        raise_for(self.lib.mvsvd_depth_read(self._h, z.ctypes.data), self.lib)
        return z.astype(np.float64, copy=False)

    def run(self, n_rank, center=False):
        """M (n_cols, r), sigma (n_cols,), S (r, n_rows), means (n_cols,), timings."""
        M = np.empty((self.n_cols, n_rank), self.dtype)
        sigma = np.empty(self.n_cols, self.dtype)
        S = np.empty((n_rank, self.n_rows), self.dtype)
        means = np.zeros(self.n_cols, self.dtype)
        tm = np.zeros(6)
        raise_for(self.lib.mvsvd_run(self._h, int(n_rank), int(bool(center)), M.ctypes.data, sigma.ctypes.data,
                                     S.ctypes.data, means.ctypes.data, _ptr(tm)), self.lib)
        return M, sigma, S, means, _tm(tm)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mvsvd_destroy(self._h)
            self._h = None

    __del__ = close


_svd_cache = {}  # (dtype, n_cols, device) -> SvdWorkspace kept between public factorization calls
_svd_cache_lock = threading.Lock()  # (ranks may be threads of one process: lib._distributed.InProcessGroup)
_svd_cache_atexit = False


def _svd_cache_clear():
    """Release the cached SVD workspaces (~2 GB of device memory at config 5); also registered at interpreter exit."""
    with _svd_cache_lock:
        for ws in _svd_cache.values():
            ws.close()
        _svd_cache.clear()


svd_cache_clear = _svd_cache_clear  # public name: call it to hand the cached workspaces' device memory back (e.g. before a large BundleAdjuster)


def _svd_cache_usable(n_rows, n_cols, n_rank):
    return os.environ.get("MVBA_SVD_CACHE", "1") != "0" and n_rows >= 1 and 1 <= n_rank <= n_cols <= 12288


def _svd_cached_run(dtype, n_rows, n_cols, device, load, n_rank, center):
    """`load(ws)` then `ws.run(n_rank, center)` on the cached workspace of (dtype, n_cols, device), made or grown as needed."""
    global _svd_cache_atexit
    dtype = np.dtype(dtype)
    key = (dtype.str, n_cols, int(device))
    with _svd_cache_lock:  # (held through the call: a workspace is one matrix and one stream)
        ws = _svd_cache.get(key)
        if ws is None or ws.max_rows < n_rows:
            old = _svd_cache.pop(key, None)  # out of the cache BEFORE it is closed: a failing allocation below must not leave a closed handle behind
            if old is not None:
                old.close()
            if not _svd_cache_atexit:
                import atexit

                atexit.register(_svd_cache_clear)
                _svd_cache_atexit = True
            ws = SvdWorkspace(n_rows, n_cols, dtype, device)
            _svd_cache[key] = ws
        load(ws)
        return ws.run(n_rank, center)


def svd_factorize_images(x_list, n_rank, center=False, device=-1):
    """`svd_factorize(np.hstack(x_list), ...)` without the hstack: the images' (n_rows, 2) arrays (all float32 or all float64) go to
    the device as they are and the (n_rows, 2 m) matrix is put together there (`mvsvd_load_images`).  Anything else -- other
    dtypes, ragged lists, the cache switched off -- takes the host's hstack."""
    if device_count() < 1:
        raise RuntimeError("libmvba: no HIP device visible; the SVD kernel has no CPU fallback")
    arrs = [np.asarray(a) for a in x_list]
    n_rows, n_cols = (arrs[0].shape[0] if arrs and arrs[0].ndim == 2 else 0), 2 * len(arrs)
    dt = np.result_type(*arrs) if arrs else np.dtype(np.float64)
    if (dt in (np.float32, np.float64) and all(a.shape == (n_rows, 2) for a in arrs) and _svd_cache_usable(n_rows, n_cols, n_rank)):
        return _svd_cached_run(dt, n_rows, n_cols, device, lambda ws: ws.load_images(arrs), n_rank, center)
    return svd_factorize(np.ascontiguousarray(np.hstack(arrs)), n_rank, center, device)


def svd_factorize(Wt, n_rank, center=False, device=-1):
    """Thin SVD of W = Wt^T.  Wt: (n_rows, n_cols) float32/float64, C-contiguous.
    Returns M (n_cols, r), sigma (n_cols,), S (r, n_rows), means (n_cols,), timings.
    A workspace (device buffers, stream, events: `mvsvd_create`) is KEPT between calls per (dtype, n_cols, device) and grown
    when a larger matrix arrives -- at config 5 allocating and freeing ~1 GB of device memory per call cost more than the
    PCIe copy of the matrix; `MVBA_SVD_CACHE=0` restores the one-shot `mvsvd_factorize` (create + load + run + destroy),
    `_svd_cache_clear()` (also at interpreter exit) releases the memory."""
    lib = load_library()
    if device_count() < 1:
        raise RuntimeError("libmvba: no HIP device visible; the SVD kernel has no CPU fallback")
    Wt = np.ascontiguousarray(Wt)
    if Wt.dtype not in (np.float32, np.float64):
        Wt = Wt.astype(np.float64)
    n_rows, n_cols = Wt.shape
    if _svd_cache_usable(n_rows, n_cols, n_rank):
        return _svd_cached_run(Wt.dtype, n_rows, n_cols, device, lambda ws: ws.load(Wt), n_rank, center)
    M = np.empty((n_cols, n_rank), Wt.dtype)
    sigma = np.empty(n_cols, Wt.dtype)
    S = np.empty((n_rank, n_rows), Wt.dtype)
    means = np.zeros(n_cols, Wt.dtype)
    tm = np.zeros(6)
    rc = lib.mvsvd_factorize(Wt.ctypes.data, n_rows, n_cols, 0 if Wt.dtype == np.float32 else 1, int(n_rank),
                             int(bool(center)), M.ctypes.data, sigma.ctypes.data, S.ctypes.data, means.ctypes.data,
                             _ptr(tm), int(device))
    raise_for(rc, lib)
    return M, sigma, S, means, _tm(tm)
