#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (one line per kernel)."""
import re
import sys

txt = open(sys.argv[1]).read()
for blk in txt.split("Function Name: ")[1:]:
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", blk.split()[0])[:24]

    def g(key):
        mm = re.search(re.escape(key) + r": (\d+)", blk)
        return mm.group(1) if mm else "?"

    print("%-26s SGPR %4s VGPR %4s AGPR %3s scratch %4s occ %2s LDS %6s" % (
        name, g("TotalSGPRs"), g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"),
        g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
