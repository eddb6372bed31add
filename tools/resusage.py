#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (one line per kernel)."""
import re
import sys

txt = open(sys.argv[1]).read()
for b in txt.split("Function Name: ")[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    m = re.search(
        r"fft_(pass|ptp_mid)_kernelI([a-z])Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)(?:ELi(-?\d+|n\d+))?",
        name,
    )
    tag = name[:70]
    if m:
        tag = "%-7s T=%s N=%s E=%s L=%s ax=%s blk=%sx%s split=%s dir=%s" % m.groups()
    vg, ag = g("VGPRs"), g("AGPRs")
    sc = g(r"ScratchSize \[bytes/lane\]")
    oc = g(r"Occupancy \[waves/SIMD\]")
    ld = g(r"LDS Size \[bytes/block\]")
    print(f"{tag:72s} VGPR={vg:4d} AGPR={ag:3d} scratch={sc:4d} occ={oc} LDS={ld}")
