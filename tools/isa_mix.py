#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S device listing: python tools/isa_mix.py lib.s <symbol-substring>"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
inside = False
mix = collections.Counter()
name = None
with open(path) as f:
    for line in f:
        if not inside:
            m = re.match(r"^(_Z\w+):", line)
            if m and key in m.group(1):
                inside, name = True, m.group(1)
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):  # a kernel may hold several s_endpgm (early exits)
            break
        m = re.match(r"^([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)", s)
        if m:
            mix[m.group(1)] += 1
total = sum(mix.values())
valu = sum(c for k, c in mix.items() if k.startswith("v_"))
print(name)
print("total", total, "VALU", valu)
for k, c in mix.most_common(28):
    print(f"{c:6d} {k}")
