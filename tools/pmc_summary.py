#!/usr/bin/env python3
"""Per-kernel average of rocprofv3 --pmc counters (counter_collection.csv), with the gfx950
corrections of MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports
half of the bytes of wide coalesced reads on gfx950 (doubled here)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"paos::(\w+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else name[:60]


def main(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        parts = []
        for cname, vals in cs.items():
            avg = sum(vals) / len(vals)
            if cname == "FETCH_SIZE":
                parts.append(f"FETCH_SIZE avg {avg:.0f} KiB -> x2 corrected {2 * avg * 1024 / 1e9:.3f} GB/launch")
            elif cname == "WRITE_SIZE":
                parts.append(f"WRITE_SIZE avg {avg:.0f} KiB -> {avg * 1024 / 1e9:.3f} GB/launch")
            else:
                parts.append(f"{cname} avg {avg:.4g}")
        print(f"{k:95s} n={len(vals):3d}  " + "; ".join(parts))


if __name__ == "__main__":
    main(sys.argv[1])
