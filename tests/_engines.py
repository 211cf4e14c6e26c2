"""Test-side engines: the CPU oracle dressed in the HipEngine's full protocol, so that the product's host logic
(lib/bundle_adjustment.py: normalisation, LM loop, log, way back to the input frame) can be run over it without
the product carrying a host-engine branch.  Test infrastructure only."""
import numpy as np

from oracle import ba_oracle as O


class HostOracleEngine(O.OracleEngine):
    """OracleEngine + the device-log and similarity entry points of lib._mvba.HipEngine, kept on the host."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._snaps = []

    def snapshot(self):
        self._snaps.append(tuple(np.array(v, copy=True) for v in self.get_params()))

    def snapshot_count(self):
        return len(self._snaps)

    def snapshot_read(self, i):
        if not 0 <= i < len(self._snaps):
            raise ValueError("no such log entry")
        return tuple(v.copy() for v in self._snaps[i])

    def snapshot_clear(self):
        self._snaps = []

    def snapshot_restore(self, i):
        self.set_params(*self.snapshot_read(i))

    def apply_similarity(self, R0, t0, scale):
        """Committed state -> scale * X R0^T + t0 (likewise t), R0 R  (ref lib/bundle_adjustment.py:242-258)."""
        X, f, u, t, R = self.get_params()
        self.set_params(t0 + (scale * X) @ R0.T, f, u, t0 + (scale * t) @ R0.T, R0 @ R)
