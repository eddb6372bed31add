#!/usr/bin/env python3
"""Scratch (spill) report of the frugal pass kernels from a build log made with
`make EXTRA=-Rpass-analysis=kernel-resource-usage > build/make.log 2>&1` (or `make spillcheck`): the pass kernels are
meant to fit their register budget without scratch.  Known exceptions: the complex64 shapes with three phases in one
slot (<= 24 B per lane), since round 4 the LONG builds (a launch that runs two or three passes: <= 40 B per lane --
a few dwords spilled once per wave in a four- to six-transform kernel) and, since round 5, the OCC = 1 builds of 2048^2
(four workgroups per CU at 128 VGPRs: <= 80 B per lane).  Exit status 1 if any other shape spills: a
change to the kernel header that costs the ordinary shapes their allocation shows here, not only in the bench (round 4:
a generic lambda inside a discarded `if constexpr` branch did exactly that to every shape)."""
import re
import sys

txt = open(sys.argv[1] if len(sys.argv) > 1 else "build/make.log").read()
KEYS = ("type", "N", "lines", "axis", "BR", "split", "kpre", "kmid", "nfft", "store", "tab", "long", "occ")
rows = []
for b in txt.split("Function Name: ")[1:]:
    name = b.split("\n")[0].strip()
    m = re.search(r"frugal_pass_kernelI([df])Li(\d+)ELi\d+ELi(\d)ELi1ELi(\d)ELi(\d)ELi2ELb([01])ELi(\d)ELi(\d)ELi(\d)ELi(\d)E(?:Li(\d)ELi(\d)E)?(?:Li(\d)E)?", name)
    v, c = re.search(r"VGPRs: (\d+)", b), re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
    if m and v and c:
        g = tuple(x if x is not None else "0" for x in m.groups())
        rows.append((g, int(v.group(1)), int(c.group(1))))
if not rows:
    sys.exit("no frugal_pass_kernel resource remarks in the log")


def expected(r):
    g, _, scratch = r
    if g[0] == "f" and g[7] == "3" and scratch <= 24:
        return True
    if g[12] != "0":  # round 5: the four-per-CU shapes of 2048^2 (128 VGPRs where the ordinary shapes have 168)
        return scratch <= 80
    return g[11] != "0" and scratch <= 40


bad = [r for r in rows if r[2] > 0 and not expected(r)]
print(f"{len(rows)} frugal pass shapes, max VGPRs {max(r[1] for r in rows)}, max scratch {max(r[2] for r in rows)} B/lane, "
      f"{sum(r[2] > 0 for r in rows)} with scratch, {len(bad)} unexpected")
for r in sorted(rows):
    if r[2] > 0:
        print("  scratch:", dict(zip(KEYS, r[0])), r[1:], "" if r not in bad else "  <-- unexpected")
sys.exit(1 if bad else 0)
