#!/usr/bin/env python3
"""Pretty-print a rocprofv3 *_kernel_stats.csv (per-kernel calls / avg / total)."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{n:24s} calls {r['Calls']:>6} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.3f}"
          f" min_us {float(r['MinNs'])/1e3:8.1f} max_us {float(r['MaxNs'])/1e3:8.1f}")
