#!/usr/bin/env python3
"""Per class of pass launch: mean HIP-event time of bench.py output files -- python tools/classes_line.py a.json [b.json ...]"""
import json
import sys

T = "skips tiles of dead lines"
ORDER = ["full", T, "skips loads of dead positions", "skips stores nobody reads",
         "skips loads of dead positions + skips stores nobody reads", T + " + stores the PSF instead of the field",
         # the separable programs of round 4: every launch skips the tiles of dead / unwanted lines
         T + " + skips loads of dead positions", T + " + skips stores nobody reads",
         T + " + skips loads of dead positions + skips stores nobody reads",
         # ... and a launch may run two or three consecutive passes of a row / column chain
         T + " + skips loads of dead positions + skips stores nobody reads + runs two passes of a row / column chain",
         T + " + skips loads of dead positions + skips stores nobody reads + runs three passes of a row / column chain",
         T + " + skips loads of dead positions + stores the PSF instead of the field + runs three passes of a row / column chain"]
for path in sys.argv[1:]:
    d = json.load(open(path))
    cl = d["roofline"].get("classes", {})
    parts = [f"{d['value']:.1f} wf/s"] + [f"{cl[k]['avg_launch_ms']:.3f}" if k in cl else "-" for k in ORDER]
    print(f"{path}: " + "  ".join(parts))
print("# columns: value | " + " | ".join(ORDER) + "  (ms per launch)")
