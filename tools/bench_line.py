#!/usr/bin/env python3
"""Print the headline numbers of bench.py output files: value, ms/step, roofline fraction, mean pass launch."""
import json
import sys

for path in sys.argv[1:]:
    lines = [x for x in open(path) if x.startswith("{")]
    if not lines:
        print(path, "no JSON line")
        continue
    d = json.loads(lines[-1])
    r = d.get("roofline", {})
    dense = r.get("dense") or {}
    print(f"{path}: {d['value']:.1f} {d['unit']}, {d['ms_per_step']:.2f} ms/step, roofline {r.get('frac', 0):.3f} "
          f"({r.get('avg_launch_ms', 0):.4f} ms x {r.get('launches')} launches), dense pass {dense.get('frac', 0):.3f} "
          f"({dense.get('avg_launch_ms', 0):.3f} ms)")
