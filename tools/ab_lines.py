#!/usr/bin/env python3
"""One line per bench_detail.json: rate, dense / ptp fractions and the mean time of every class of pass launch."""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
short = {"skips tiles of dead lines": "T", "skips loads of dead positions": "L", "skips stores nobody reads": "S",
         "stores the PSF instead of the field": "P", "runs two passes of a row / column chain": "x2",
         "runs three passes of a row / column chain": "x3", "full": "full"}
cls = "  ".join("+".join(short.get(p, p) for p in k.split(" + ")) + f" {v['avg_launch_ms']:.4f}" for k, v in r.get("classes", {}).items())
print(f"{sys.argv[2]:12s} round {sys.argv[3]}: {d['value']:7.1f} wf/s  dense {r.get('dense', {}).get('frac', 0):.4f}  "
      f"ptp {d['ptp_step']['frac_bytes_moved']:.4f}  pass ms/step {r['all_launches']['ms']:.3f} | {cls}", flush=True)
