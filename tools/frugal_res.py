#!/usr/bin/env python3
"""VGPR / scratch table of the frugal pass kernels from a -Rpass-analysis=kernel-resource-usage log."""
import re
import sys

txt = open(sys.argv[1]).read()
hot_only = len(sys.argv) > 2
rows = []
for b in txt.split("Function Name: ")[1:]:
    name = b.split("\n")[0].strip()
    m = re.search(r"frugal_pass_kernelI([df])Li(\d+)ELi16ELi(\d)ELi1ELi(\d)ELi4ELi2ELb1ELi(\d)ELi(\d)ELi(\d)E", name)
    if not m:
        continue
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    rows.append((m.group(1), int(m.group(2)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)),
                 g("VGPRs"), g(r"ScratchSize \[bytes/lane\]")))
for r in sorted(rows):
    if hot_only and not (r[3] == 0 and r[4] <= 1):
        continue
    print("%s N=%d axis=%d KPRE=%d KMID=%d NFFT=%d VGPR=%d scratch=%d" % r)
print("max scratch", max(r[-1] for r in rows))
