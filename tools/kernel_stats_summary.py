#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 --kernel-trace output directory (…_kernel_trace.csv):
python tools/kernel_stats_summary.py <dir> ["header line"]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"paos::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:150]


def main(d, header=None):
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        sys.exit("no kernel_trace.csv under " + d)
    tot = defaultdict(float)
    cnt = defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            cnt[k] += 1
    total = sum(tot.values())
    if header:
        print("# " + header)
    print(f"# total kernel time {total:.1f} ms")
    for k in sorted(tot, key=tot.get, reverse=True):
        print(f"{tot[k]:9.2f} ms {100 * tot[k] / total:6.2f}% calls={cnt[k]:5d} avg={1e3 * tot[k] / cnt[k]:9.1f} us  {k}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
