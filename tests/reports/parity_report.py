#!/usr/bin/env python3
"""Print the measured GPU-vs-oracle errors (fields, PSF) for DESIGN.md / profiles.

    python tests/reports/parity_report.py [--sizes 256 512 1024] > profiles/rNN_parity.txt
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.run_np import run as oracle_run  # noqa: E402
from paos_amd.chains import syn20_chain  # noqa: E402
from paos_amd.parse_config import parse_config  # noqa: E402
from paos_amd.run import run  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(np.ravel(b)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[256, 512, 1024])
    ap.add_argument("--chains", nargs="+", default=["SYN20", "Hubble_simple", "Excite_TEL", "Ariel_AIRS-CH0", "Ariel_FGS-FGS1"])
    args = ap.parse_args()
    print("chain            N     surf  field_max   field_L2    PSF_max     PSF_L2      fp32_PSF_max  t_gpu[s] t_cpu[s]")
    for name in args.chains:
        if name == "SYN20":
            spec = dict(pup=1.0, wl=1.0e-6, zoom=4, field={"us": 0.0, "ut": 0.0}, chain=syn20_chain())
        else:
            pup, par, wls, fields, chains = parse_config(os.path.join(ROOT, "data", "lens", name + ".ini"))
            spec = dict(pup=pup, wl=1.0e-6 * wls[0], zoom=par["zoom"], field=fields[0], chain=chains[0])
        for n in args.sizes:
            a = (spec["pup"], spec["wl"], n, spec["zoom"], spec["field"], spec["chain"])
            t0 = time.perf_counter()
            got = run(*a)
            t1 = time.perf_counter()
            ref = oracle_run(*a, light=True)
            t2 = time.perf_counter()
            g32 = run(*a, precision="fp32")
            k = max(ref)
            psf_g, psf_r = got[k]["amplitude"] ** 2, ref[k]["amplitude"] ** 2
            print(f"{name:16s} {n:5d} S{k:02d}   {rel(got[k]['wfo'], ref[k]['wfo']):.2e}   "
                  f"{l2(got[k]['wfo'], ref[k]['wfo']):.2e}   {rel(psf_g, psf_r):.2e}   {l2(psf_g, psf_r):.2e}   "
                  f"{rel(g32[k]['amplitude'] ** 2, psf_r):.2e}      {t1 - t0:7.3f}  {t2 - t1:7.2f}", flush=True)


if __name__ == "__main__":
    main()
