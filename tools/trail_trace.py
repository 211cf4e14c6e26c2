#!/usr/bin/env python3
"""Per-launch durations of the dense-solve kernels from a rocprofv3 kernel trace (one line per super-block of the LAST solve):
usage: tools/trail_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [(re.search(r"k_\w+", r["Kernel_Name"]) or [r["Kernel_Name"]])[0] for r in rows]
# the last k_compact starts the last solve
last = max(i for i, n in enumerate(names) if n.startswith("k_compact"))
out, t0 = [], int(rows[last]["Start_Timestamp"])
for r, n in zip(rows[last:], names[last:]):
    if not n.startswith(("k_compact", "k_chol")):
        break
    out.append((n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
tot = {}
for n, s, d in out:
    tot[n] = tot.get(n, 0.0) + d
print("solve span %.1f us; " % (out[-1][1] + out[-1][2]) + "  ".join(f"{n} {v:.1f}" for n, v in tot.items()))
print(" ".join(f"{n.replace('k_chol_', '')[:7]}:{d:.1f}" for n, s, d in out))
