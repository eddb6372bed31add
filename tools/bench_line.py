#!/usr/bin/env python3
"""Print the headline numbers of bench.py records -- the contract line (a *.json holding the one line) or the detail record
(bench_detail.json): value, ms/step, the dominant launch's roofline fractions, dense / ptp fractions."""
import json
import sys

for path in sys.argv[1:]:
    text = open(path).read().strip()
    try:
        d = json.loads(text)
    except ValueError:
        lines = [x for x in text.splitlines() if x.startswith("{")]
        if not lines:
            print(path, "no JSON")
            continue
        d = json.loads(lines[-1])
    r = d.get("roofline", {})
    dom = d.get("dominant_launch") or {}
    dense = r.get("dense") or {}
    frac = r.get("flop_frac", dom.get("flop_frac"))
    print(f"{path}: {d['value']:.1f} {d['unit']}, {d['ms_per_step']:.2f} ms/step | dominant launch {r.get('avg_launch_ms', dom.get('avg_launch_ms', 0)):.4f} ms: "
          f"flop {frac or 0:.3f} issue {(r.get('issue_frac') or dom.get('issue_frac') or 0):.3f} hbm {(r.get('hbm_frac') or dom.get('hbm_frac') or 0):.3f} | "
          f"dense {(r.get('dense_frac') or dense.get('frac') or 0):.3f} ptp {(r.get('ptp_step_frac') or (d.get('ptp_step') or {}).get('frac_bytes_moved') or 0):.3f}")
