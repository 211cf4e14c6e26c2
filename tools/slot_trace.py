#!/usr/bin/env python3
"""Summarise a per-wave trace of k_schur_slots (diagnostic build, `make trace`; MVBA_SLOT_TRACE=<file>)."""
import collections
import sys

import numpy as np

raw = np.loadtxt(sys.argv[1], comments="#")
raw = raw[raw[:, 8] > 0]
a = raw[:, :9]
ph = raw[:, 9:15] if raw.shape[1] >= 15 else None
blk = a[:, 0].astype(int)
t0, t1, wait, blocked, polls, hwid, xcc, nsteps = (a[:, i] for i in range(1, 9))
base = t0.min()
dur = (t1 - t0) / 100.0  # us
print(f"waves {len(a)}  launch span {(t1.max() - base) / 100:.1f} us  start spread {(t0.max() - base) / 100:.1f} us")
print(f"duration us: min {dur.min():.0f} p10 {np.percentile(dur, 10):.0f} median {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}")
print(f"wait us per wave: p10 {np.percentile(wait, 10) / 100:.0f} median {np.median(wait) / 100:.0f} p90 {np.percentile(wait, 90) / 100:.0f} max {wait.max() / 100:.0f};"
      f"  blocked crossings median {np.median(blocked):.0f};  polls median {np.median(polls):.0f}")
run = (dur - wait / 100.0) / nsteps
print(f"us per step outside the waits: p10 {np.percentile(run, 10):.3f} median {np.median(run):.3f} p90 {np.percentile(run, 90):.3f};  steps: median {np.median(nsteps):.0f} max {nsteps.max():.0f}")
print("block % 8 == XCC_ID for", float(np.mean((blk % 8) == xcc)) * 100, "% of the waves")
h = hwid.astype(int)
simd, cu, se, sh = (h >> 4) & 3, (h >> 8) & 15, (h >> 13) & 7, (h >> 12) & 1
key = xcc.astype(int) * 100000 + se * 1000 + sh * 100 + cu
percu = collections.Counter(key)
persimd = collections.Counter(zip(key, simd))
nsimd = np.array([persimd[(k, s)] for k, s in zip(key, simd)])
print("waves per CU:", dict(collections.Counter(percu.values())))
for n in sorted(set(nsimd)):
    m = nsimd == n
    print(f"  waves on a SIMD holding {n}: {m.sum():5d}  duration median {np.median(dur[m]):.0f}  wait median {np.median(wait[m]) / 100:.0f}  us/step outside waits {np.median(run[m]):.3f}")
for x in range(8):
    m = blk % 8 == x
    print(f"  range {x}: end median {np.median((t1[m] - base) / 100):7.0f} max {((t1[m] - base) / 100).max():7.0f}  wait median {np.median(wait[m]) / 100:6.0f}")

if ph is not None and ph[:, 5].max() > 0:  # phase sums of the loop in shader cycles (round 5)
    names = ("vmcnt wait", "pacing", "index reads + gather issue", "index DMA", "LDS reads + arithmetic")
    per = ph[:, :5] / nsteps[:, None]
    tot = ph[:, 5] / nsteps
    print(f"shader cycles per step (median over the waves; loop total {np.median(tot):.0f}, = {np.median(ph[:, 5] / (dur * 1e3)) * 1e3:.0f} MHz against the 100 MHz clock):")
    for i, nme in enumerate(names):
        print(f"  {nme:28s} {np.median(per[:, i]):7.0f}   p10 {np.percentile(per[:, i], 10):7.0f}  p90 {np.percentile(per[:, i], 90):7.0f}")
    for n in sorted(set(nsimd)):
        m = nsimd == n
        print(f"  waves on a SIMD holding {n}: " + "  ".join(f"{nme.split()[0]} {np.median(per[m, i]):.0f}" for i, nme in enumerate(names)))
