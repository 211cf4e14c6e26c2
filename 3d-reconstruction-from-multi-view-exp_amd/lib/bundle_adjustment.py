"""Bundle adjustment with the call surface of the reference's
``lib/bundle_adjustment.py`` (class ``BundleAdjuster``: constructor :11-21,
``optimize`` :77-83/:202, ``get_log`` :204-206) on the MI355X engine.

What stays in Python, exactly as the reference does it:
  * the scene normalisation / de-normalisation with its sign quirk (:208-258),
  * ``f <- K[:,0,0]``, ``u <- K[:,:2,2]`` (K[1,1], K[2,2] ignored, :45-48),
  * the Levenberg-Marquardt control flow: c0 = 1e-4, reject iff ``E_ > E``
    (strict), ``c *= s`` / ``c /= s``, stop on ``|dE| <= tol`` or ``max_iter``,
    the per-iteration print and the debug log (:100-195).
Everything numerical per observation / point / reduced system runs in
``libmvba.so`` (hand-written HIP, see csrc/mvba.hip) through ``_mvba.HipEngine``.
There is no CPU fallback.
"""
from __future__ import annotations

import os
from typing import Any

import numpy as np
import numpy.typing as npt

AXES = {"x-right_z-forward": 0, "x-up_z-forward": 1}


def dense_to_observations(x: npt.NDArray, visibility_index: npt.NDArray | None):
    """Dense ``x (N,m,2)`` + bool mask (ref :37, :56-60) -> CSR-by-point list (pt_ptr, cam_idx, xy (n_obs, 2)).
    Invisible entries are dropped instead of multiplied by 0.  Without a mask, an ``x`` that is the
    transposed view of a stack of image arrays comes back as that stack, xy (m, N, 2): see below."""
    n, m = x.shape[:2]
    if visibility_index is None:  # everything visible: the list is the array itself (np.nonzero + a gather took 0.24 s at 1 M x 12)
        x = np.asarray(x, dtype=np.float64)
        planes = x.transpose(1, 0, 2)
        if not x.flags.c_contiguous and planes.flags.c_contiguous:
            # the reference caller's np.stack(x_list).transpose(1, 0, 2) (euclidiean_reconstruction.py:50): the memory is the m image
            # planes.  They go to the engine as they are -- xy of shape (m, N, 2), mvba_problem.xy_layout 1 -- and the device puts
            # them into observation order (the strided host copy below: 0.10 s at 1 M x 12, a third of the whole pipeline)
            xy = planes
        else:
            xy = np.ascontiguousarray(x.reshape(n * m, 2))
        cam_idx = np.empty((n, m), dtype=np.int32)
        cam_idx[:] = np.arange(m, dtype=np.int32)  # (a broadcast store: np.tile of the same 12 M entries took 0.07 s)
        return np.arange(0, (n + 1) * m, m, dtype=np.int64), cam_idx.reshape(-1), xy
    vis = np.asarray(visibility_index, dtype=np.bool_)
    pt, cam = np.nonzero(vis)
    pt_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(vis.sum(axis=1), out=pt_ptr[1:])
    xy = np.ascontiguousarray(np.asarray(x)[pt, cam], dtype=np.float64)
    return pt_ptr, cam.astype(np.int32), xy


def to_gauge_frame(X, R, t, axis: str):
    """Scene -> the frame BA works in (ref :208-240): camera 0 at the origin with identity pose,
    camera 1's baseline component along the gauge axis of unit size.  The divisor keeps the
    reference's quirk: its SIGN comes from the world-frame component of t1 - t0, its
    MAGNITUDE from the camera-0-frame component."""
    if axis not in AXES:
        raise ValueError()
    g = AXES[axis]  # 0: x-right (index 0), 1: x-up (index 1)
    R0 = R[0]
    dX, dt = X - t[0], t - t[0]
    s = np.sign(dt[1, g]) * (R0[:, g] @ dt[1])
    s = np.array([s])  # shape (1,), as the reference's
    return (dX @ R0) / s, R0.T @ R, (dt @ R0) / s


def from_gauge_frame(camera0: dict[str, Any], X, R, t):
    """The way back (ref :242-258): scale by |baseline| (an abs, :23-26), rotate by camera 0's
    original pose, shift by its original centre."""
    R0, t0, length = camera0["R"], camera0["t"], camera0["c0c1_len"]
    return t0 + (length * X) @ R0.T, R0 @ R, t0 + (length * t) @ R0.T


def intrinsics_from(f, u, f0: float):
    """K_k = [[f,0,u0],[0,f,v0],[0,0,f0]] (ref :283-289): K[2,2] is forced to f0."""
    m = len(f)
    K = np.zeros((m, 3, 3))
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 0, 2], K[:, 1, 2] = u[:, 0], u[:, 1]
    K[:, 2, 2] = f0
    return K



class LevenbergMarquardt:
    """The reference's LM control state (:85-101, :118-195) over an engine that
    offers cost / linearize / try_step / commit; ``iterate()`` is one outer
    iteration: linearise once, retry with ``c *= s`` while the trial cost is
    strictly larger, commit."""

    def __init__(self, engine, scale_factor):
        self.engine, self.scale_factor = engine, scale_factor
        self.E = engine.cost()
        self.c = 0.0001
        self.count = 0

    def iterate(self):
        g = self.engine
        g.linearize()
        while True:  # no iteration cap, as the reference
            E_ = g.try_step(self.c)
            if E_ > self.E:
                self.c *= self.scale_factor
            else:
                break
        g.commit()
        self.count += 1
        delta = np.abs(E_ - self.E)
        return E_, delta

    def carry_on(self, E_):
        """ref :194-195"""
        self.E = E_
        self.c /= self.scale_factor


def lm_loop(engine, scale_factor, delta_tol, max_iter, on_state=None, verbose=True):
    """The reference's outer loop (:102-195).  Returns the final cost."""
    lm = LevenbergMarquardt(engine, scale_factor)
    if on_state is not None:
        on_state(lm.E)
    while True:
        E_, reprojection_error_delta = lm.iterate()
        if on_state is not None:
            on_state(E_)
        if verbose:
            print(f"Iteration {lm.count}: reprojection_error_delta = {reprojection_error_delta}")
        if reprojection_error_delta <= delta_tol or lm.count >= max_iter:
            break
        lm.carry_on(E_)
    return E_


class BundleAdjuster:
    def __init__(
        self,
        x: npt.NDArray,
        init_X: npt.NDArray,
        init_K: npt.NDArray,
        init_R: npt.NDArray,
        init_t: npt.NDArray,
        f0: float = 1.0,
        visibility_index: npt.NDArray | None = None,
        axis: str = "x-right_z-forward",
    ):
        x = np.asarray(x)
        pt_ptr, cam_idx, xy = dense_to_observations(x, visibility_index)
        self._setup(x.shape[0], x.shape[1], pt_ptr, cam_idx, xy, init_X, init_K, init_R, init_t, f0, axis)

    @classmethod
    def from_observations(cls, n_points, n_images, pt_ptr, cam_idx, xy, init_X, init_K, init_R, init_t,
                          f0: float = 1.0, axis: str = "x-right_z-forward", **engine_kw):
        """Extension for sizes where the dense (N,m,2) array cannot exist
        (SURVEY 8f rank 1): observation list in CSR-by-point form."""
        self = cls.__new__(cls)
        self._setup(n_points, n_images, pt_ptr, cam_idx, xy, init_X, init_K, init_R, init_t, f0, axis, **engine_kw)
        return self

    # -- construction ------------------------------------------------------
    def _make_engine(self, n_points, n_images, pt_ptr, cam_idx, xy, f0, axis, **kw):
        from ._mvba import HipEngine

        return HipEngine(n_points, n_images, pt_ptr, cam_idx, xy, f0, axis, **kw)

    def _setup(self, n_points, n_images, pt_ptr, cam_idx, xy, init_X, init_K, init_R, init_t, f0, axis, **engine_kw):
        init_X, init_K = np.asarray(init_X, dtype=np.float64), np.asarray(init_K, dtype=np.float64)
        init_R, init_t = np.asarray(init_R, dtype=np.float64), np.asarray(init_t, dtype=np.float64)
        # camera-0 pose and baseline length for the way back (ref :23-33)
        if axis == "x-right_z-forward":
            c0c1_len = np.abs(init_R[0, :, 0] @ (init_t[1] - init_t[0]))
        elif axis == "x-up_z-forward":
            c0c1_len = np.abs(init_R[0, :, 1] @ (init_t[1] - init_t[0]))
        else:
            raise ValueError()
        self._init_camera0_params = {"R": init_R[0], "t": init_t[0], "c0c1_len": c0c1_len}
        X, R, t = to_gauge_frame(init_X, init_R, init_t, axis)
        self._f0 = f0
        self._n_points, self._n_images = int(n_points), int(n_images)
        self._engine = self._make_engine(self._n_points, self._n_images, pt_ptr, cam_idx, xy, f0, axis, **engine_kw)
        self._engine.set_params(X, init_K[:, 0, 0], init_K[:, :2, 2], t, R)  # ref :45-48
        self._log: list[dict[str, npt.NDArray | float]] = []

    # -- the reference's public methods --------------------------------------
    def optimize(
        self,
        scale_factor: float = 10.0,
        delta_tol: float = 1e-8,
        max_iter: int = 100,
        is_debug: bool = False,
    ) -> tuple[npt.NDArray, npt.NDArray, npt.NDArray, npt.NDArray]:
        on_state = None
        if is_debug:
            self._log.clear()  # ref :90
            self._log_errors = []
            self._engine.snapshot_clear()
            # The log lives in device memory while optimize() runs: 24 N + 120 m bytes per outer iteration (24 MB at
            # 1 M points, 240 MB at 10 M), one device-to-device copy on the engine's stream per entry.  Above
            # MVBA_LOG_DEVICE_BYTES (default 16 GiB), or when the device cannot allocate the next slab, the entries
            # gathered so far are fetched to the host (as get_log() would) and the device log starts over.
            entry_bytes = 24 * self._n_points + 120 * self._n_images
            budget = int(os.environ.get("MVBA_LOG_DEVICE_BYTES", str(16 << 30)))

            def on_state(err):  # log entries are copies, normalised frame (ref :91-97, :175-183)
                if (len(self._log_errors) + 1) * entry_bytes > budget and self._log_errors:
                    self._fetch_log()
                try:
                    self._engine.snapshot()
                except RuntimeError:  # out of device memory for the next slab
                    if not self._log_errors:
                        raise
                    self._fetch_log()
                    self._engine.snapshot()
                self._log_errors.append(err)

        lm_loop(self._engine, scale_factor, delta_tol, max_iter, on_state)
        # the reference rebinds its state to the de-normalised values (:198-200): the engine applies
        # the way back (:242-258) to its committed state on the device, then hands it over
        cam0 = self._init_camera0_params
        self._engine.apply_similarity(cam0["R"], cam0["t"], cam0["c0c1_len"])
        X, f, u, t, R = self._engine.get_params()
        return X, intrinsics_from(f, u, self._f0), R, t

    def _fetch_log(self):
        """Device-resident log entries -> host dicts (in order), device log emptied."""
        for i, err in enumerate(self._log_errors):
            X, _, _, t, R = self._engine.snapshot_read(i)
            self._log.append({"points": X, "basis": R, "pos": t, "reprojection_error": err})
        self._log_errors = []
        self._engine.snapshot_clear()

    def get_log(self) -> list[dict[str, npt.NDArray | float]]:
        """ref :204-206.  The per-iteration states were kept in device memory while optimize(is_debug=True) ran
        (`mvba_snapshot`, 24 N + 120 m bytes each); they cross PCIe here, once, the first time the log is asked
        for.  (The device log belongs to this adjuster's engine: code that drives `_engine.snapshot*` itself between
        optimize() and get_log() -- bench.py's episode restarts do, without is_debug -- would replace its entries.)"""
        if getattr(self, "_log_errors", None):
            self._fetch_log()
        return self._log
