#!/usr/bin/env python3
"""Run the five BASELINE.json configurations on one MI355X and print what was measured.

    python tests/reports/run_configs.py > profiles/rNN_baseline_configs.txt

Multi-GPU configs (4, 5) are run here as their single-GPU shard (the driver's scaling run
covers N > 1); parity is checked against the NumPy oracle on one item per config at a size
the oracle finishes in seconds.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.run_np import run as oracle_run  # noqa: E402
from paos_amd import _lib  # noqa: E402
from paos_amd.chains import inject_wfe, parse_config_variant, read_wfe_table  # noqa: E402
from paos_amd.parse_config import parse_config  # noqa: E402
from paos_amd.run import run, run_batch  # noqa: E402

LENS = os.path.join(ROOT, "data", "lens")
WFE = os.path.join(ROOT, "data", "wfe", "wfe_realization_SN20210914.csv")


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def timed_batch(pup, wls, n, zoom, field, chains, precision="fp64", reps=3, outputs=()):
    dev = _lib.DeviceFields(n, len(chains), precision)
    stats = {}
    run_batch(pup, wls, n, zoom, field, chains, precision=precision, outputs=(), dev=dev, sync=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        res = run_batch(pup, wls, n, zoom, field, chains, precision=precision, outputs=outputs, dev=dev,
                        sync=False, stats=stats)
    dev.sync()
    dt = (time.perf_counter() - t0) / reps
    dev.close()
    return dt, res, stats


def main():
    print("# BASELINE.json configs on 1 x MI355X (see tests/reports/run_configs.py)")

    # 1. Hubble_simple, 1 wavelength, 512^2: the reference's own CPU-runnable case
    pup, par, wls, fields, chains = parse_config(os.path.join(LENS, "Hubble_simple.ini"))
    a = (pup, 1e-6 * wls[0], 512, par["zoom"], fields[0], chains[0])
    t0 = time.perf_counter(); ref = oracle_run(*a); t_cpu = time.perf_counter() - t0
    run(*a)
    t0 = time.perf_counter(); got = run(*a); t_gpu = time.perf_counter() - t0
    k = max(ref)
    print(f"1 Hubble_simple 512^2 fp64 1 wl : CPU oracle {t_cpu:.2f} s | GPU run() {t_gpu * 1e3:.1f} ms incl. downloads | "
          f"PSF err {rel(got[k]['amplitude'] ** 2, ref[k]['amplitude'] ** 2):.1e}")

    # 2. Ariel_AIRS-CH0, 1 wavelength, 1024^2
    pup, par, wls, fields, chains = parse_config(os.path.join(LENS, "Ariel_AIRS-CH0.ini"))
    a = (pup, 1e-6 * wls[0], 1024, par["zoom"], fields[0], chains[0])
    t0 = time.perf_counter(); ref = oracle_run(*a, light=True); t_cpu = time.perf_counter() - t0
    run(*a)
    t0 = time.perf_counter(); got = run(*a); t_gpu = time.perf_counter() - t0
    k = max(ref)
    print(f"2 Ariel_AIRS-CH0 1024^2 fp64 1 wl: CPU oracle {t_cpu:.2f} s | GPU run() {t_gpu * 1e3:.1f} ms incl. downloads | "
          f"PSF err {rel(got[k]['amplitude'] ** 2, ref[k]['amplitude'] ** 2):.1e} | dx,dy {got[k]['dx']:.6e},{got[k]['dy']:.6e}")

    # 3. Ariel_AIRS-CH0, 64-wavelength batch, 2048^2
    sweep = np.linspace(1.95, 3.9, 64)
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Ariel_AIRS-CH0.ini"), sweep)
    dt, res, stats = timed_batch(pup, [1e-6 * w for w in wls], 2048, par["zoom"], fields[0], chains)
    print(f"3 Ariel_AIRS-CH0 2048^2 fp64 64 wl batch: {dt * 1e3:.1f} ms per batch = {64 / dt:.1f} wavefronts/s, "
          f"{stats.get('fused_passes')} fused passes per wavefront")

    light = [{key: dict(item, save=item["name"] == "IMAGE_PLANE") for key, item in c.items()} for c in chains]
    dtl, _, statsl = timed_batch(pup, [1e-6 * w for w in wls], 2048, par["zoom"], fields[0], light)
    print(f"3b same, light_output (image plane only): {dtl * 1e3:.1f} ms per batch = {64 / dtl:.1f} wavefronts/s, "
          f"{statsl.get('fused_passes')} fused passes per wavefront")

    # 4. Ariel_FGS-FGS1 + WFE table, 256 Monte-Carlo draws, 2048^2 (one GPU: 8 batches of 32)
    _, _, _, table = read_wfe_table(WFE)
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Ariel_FGS-FGS1.ini"), unignore=("Z1",))
    base, wl = chains[0], 1e-6 * wls[0]
    t0 = time.perf_counter()
    powers, ree90 = [], []
    radii = np.geomspace(2.0, 256.0, 16)  # pixels; rEE90 interpolated from the on-device encircled energies
    dev = _lib.DeviceFields(2048, 32)
    for lo in range(0, 256, 32):
        mc = [inject_wfe(base, table[:, k]) for k in range(lo, lo + 32)]
        res = run_batch(pup, [wl] * 32, 2048, par["zoom"], fields[0], mc, outputs=(), dev=dev, sync=True,
                        metrics_radii_px=radii)
        for r in res:
            rec = r[max(r)]
            powers.append(rec["power"])
            ee = rec["metrics"]["encircled"] / rec["metrics"]["power"]
            ree90.append(float(np.interp(0.9, ee, radii)) * rec["dx"])
    dt = time.perf_counter() - t0
    dev.close()
    mc0 = inject_wfe(base, table[:, 0])
    ref = oracle_run(pup, wl, 512, par["zoom"], fields[0], mc0, light=True)
    got = run(pup, wl, 512, par["zoom"], fields[0], mc0)
    k = max(ref)
    print(f"4 Ariel_FGS-FGS1 2048^2 fp64 256 WFE draws: {dt:.2f} s = {256 / dt:.1f} wavefronts/s on 1 GPU incl. on-device "
          f"PSF metrics; image-plane power {min(powers):.6f}..{max(powers):.6f}; rEE90 {1e6 * min(ree90):.2f}..{1e6 * max(ree90):.2f} um "
          f"(median {1e6 * float(np.median(ree90)):.2f}); draw 0 @512^2 PSF err vs oracle {rel(got[k]['amplitude'] ** 2, ref[k]['amplitude'] ** 2):.1e}")

    # 4b. the same study the way the reference's CLI runs it (--light_output: only the image plane is saved): no
    # saved surface interrupts the pass program, so consecutive operators fuse and the compiler's identities apply
    light = {key: dict(item, save=item["name"] == "IMAGE_PLANE") for key, item in base.items()}
    stats = {}
    t0 = time.perf_counter()
    ree90b = []
    dev = _lib.DeviceFields(2048, 32)
    for lo in range(0, 256, 32):
        mc = [inject_wfe(light, table[:, k]) for k in range(lo, lo + 32)]
        res = run_batch(pup, [wl] * 32, 2048, par["zoom"], fields[0], mc, outputs=(), dev=dev, sync=True,
                        metrics_radii_px=radii, stats=stats)
        for r in res:
            rec = r[max(r)]
            ee = rec["metrics"]["encircled"] / rec["metrics"]["power"]
            ree90b.append(float(np.interp(0.9, ee, radii)) * rec["dx"])
    dtb = time.perf_counter() - t0
    dev.close()
    print(f"4b same, light_output (image plane only): {dtb:.2f} s = {256 / dtb:.1f} wavefronts/s, {stats['fused_passes']} fused passes "
          f"per wavefront; rEE90 agrees with 4 to {max(abs(a - b) for a, b in zip(ree90, ree90b)) / max(ree90):.1e}")

    # 5. Excite_TEL, wavelength sweep, 4096^2, fp32 vs fp64 (one GPU: 8 of the 512 wavelengths)
    sweep = np.linspace(1.0, 4.0, 512)[::64]
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Excite_TEL.ini"), sweep)
    w = [1e-6 * x for x in wls]
    dt64, r64, _ = timed_batch(pup, w, 4096, par["zoom"], fields[0], chains, "fp64", reps=2, outputs=("psf",))
    dt32, r32, _ = timed_batch(pup, w, 4096, par["zoom"], fields[0], chains, "fp32", reps=2, outputs=("psf",))
    errs = [rel(a[max(a)]["psf"], b[max(b)]["psf"]) for a, b in zip(r32, r64)]
    print(f"5 Excite_TEL 4096^2, 8 of 512 wavelengths: fp64 {8 / dt64:.1f} wavefronts/s, fp32 {8 / dt32:.1f} wavefronts/s "
          f"(both incl. PSF download); fp32-vs-fp64 PSF max-norm error {min(errs):.1e}..{max(errs):.1e}")
    # 5b. the same sweep with the PSFs left in HBM (32 of the 512 wavelengths per batch): what the two precisions cost
    # on the GPU -- 5 is bound by 128 MiB of PSF crossing PCIe per wavefront in either mode
    sweep = np.linspace(1.0, 4.0, 512)[::16]
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Excite_TEL.ini"), sweep)
    w = [1e-6 * x for x in wls]
    d64, _, st64 = timed_batch(pup, w, 4096, par["zoom"], fields[0], chains, "fp64", reps=3)
    d32, _, _ = timed_batch(pup, w, 4096, par["zoom"], fields[0], chains, "fp32", reps=3)
    print(f"5b same chain, 32 wavelengths per batch, PSFs stay on the GPU: fp64 {32 / d64:.1f} wavefronts/s, fp32 {32 / d32:.1f} "
          f"wavefronts/s ({d64 / d32:.2f}x), {st64.get('fused_passes')} fused passes per wavefront")


if __name__ == "__main__":
    main()
