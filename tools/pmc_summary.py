#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel over dispatches (counter_collection.csv)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if k.startswith("__amd") or k.startswith("k_sum") or k.startswith("k_compact") or k.startswith("k_update"):
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:26s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
