#!/usr/bin/env python3
"""rocprofv3 --pmc passes of the default bench workload (tools/final_profile.sh) -> the JSON bench.py reads for
`roofline*.traffic`: FETCH_SIZE / WRITE_SIZE (KiB) and the TCC counters of K1 and K3, mean over the launches, with
the sha256 of the kernel sources they were taken on (bench.py drops the figure when the sources have changed).
usage: tools/pmc_to_json.py <n_obs> <n_points> <tag> <pmc dir>..."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha256  # noqa: E402

n_obs, n_points, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[4:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            if k.startswith("void "):  # a template instance: `void k_resid_jac<false>` -> the kernel's name (round 5: K1 / K5 / K6 are templates)
                k = k[5:].split("<")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB", "TCC_HIT_sum": "TCC_HIT", "TCC_MISS_sum": "TCC_MISS",
         "TCC_EA0_RDREQ_sum": "TCC_EA0_RDREQ"}
out = {"n_obs": n_obs, "n_points": n_points, "commit": tag, "csrc_sha256": csrc_sha256(), "kernels": {},
       "source": "rocprofv3 --kernel-trace --pmc <one group per pass>, bench.py --steps 3 --warmup 1 --no-cpu-baseline, mean over the "
                 "launches; tools/final_profile.sh " + tag,
       "note": "gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md): traffic = (2*FETCH + WRITE) KiB"}
for k in ("k_resid_jac", "k_schur_slots", "k_schur_pairs", "k_schur_strip"):
    if k in acc:
        out["kernels"][k] = {names[c]: sum(v) / len(v) for c, v in acc[k].items() if c in names}
print(json.dumps(out, indent=1))
