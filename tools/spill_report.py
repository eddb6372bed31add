#!/usr/bin/env python3
"""Scratch (spill) report of the frugal pass kernels from a build log made with
`make EXTRA=-Rpass-analysis=kernel-resource-usage > build/make.log 2>&1` (or `make spillcheck`): the pass kernels are
meant to fit their register budget without scratch; the known exceptions are the complex64 shapes with three phases
in one slot (<= 20 B per lane).  Exit status 1 if any other shape spills."""
import re
import sys

txt = open(sys.argv[1] if len(sys.argv) > 1 else "build/make.log").read()
rows = []
for b in txt.split("Function Name: ")[1:]:
    name = b.split("\n")[0].strip()
    m = re.search(r"frugal_pass_kernelI([df])Li(\d+)ELi16ELi(\d)ELi1ELi(\d)ELi(\d)ELi2ELb([01])ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", name)
    v, c = re.search(r"VGPRs: (\d+)", b), re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
    if m and v and c:
        rows.append((m.groups(), int(v.group(1)), int(c.group(1))))
if not rows:
    sys.exit("no frugal_pass_kernel resource remarks in the log")
bad = [r for r in rows if r[2] > 0 and not (r[0][0] == "f" and r[0][7] == "3" and r[2] <= 24)]
print(f"{len(rows)} frugal pass shapes, max VGPRs {max(r[1] for r in rows)}, max scratch {max(r[2] for r in rows)} B/lane")
for r in sorted(rows):
    if r[2] > 0:
        print("  scratch:", dict(zip(("type", "N", "lines", "axis", "BR", "split", "kpre", "kmid", "nfft", "store"), r[0])), r[1:], "" if r not in bad else "  <-- unexpected")
sys.exit(1 if bad else 0)
