#!/usr/bin/env python3
"""BASELINE.json configs[2]: Ariel_AIRS-CH0.ini, 64-wavelength batch, 2048^2 fp64 on one MI355X,
results kept in HBM (final PSFs written on the device; `POWER=1` also reduces sum |u|^2 at each of
the 12 saved surfaces).  Meant to run under rocprofv3 --kernel-trace
(tools/profile_round.sh style); prints wavefronts/s and the mean time of a pass launch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from paos_amd import _lib  # noqa: E402
from paos_amd.chains import parse_config_variant  # noqa: E402
from paos_amd.run import run_batch  # noqa: E402

n, nb, reps = 2048, 64, 3
sweep = np.linspace(1.95, 3.9, nb)
pup, par, wls, fields, chains = parse_config_variant(os.path.join(ROOT, "data", "lens", "Ariel_AIRS-CH0.ini"), sweep)
w = [1e-6 * x for x in wls]
dev = _lib.DeviceFields(n, nb)
stats = {}


def step():
    return run_batch(pup, w, n, par["zoom"], fields[0], chains, outputs=(), dev=dev, sync=False, stats=stats,
                     keep_psf=True, power=os.environ.get("POWER") == "1")


step()
dev.sync()
dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=4096)
t0 = time.perf_counter()
for _ in range(reps):
    res = step()
dev.sync()
dt = (time.perf_counter() - t0) / reps
launches, ms = dev.profile_end()
pass_bytes = 2 * 16 * n * n * nb
print(f"Ariel_AIRS-CH0 {n}^2 fp64, {nb} wavelengths 1.95-3.9 um: {dt * 1e3:.1f} ms per batch = {nb / dt:.1f} wavefronts/s; "
      f"{stats.get('fused_passes')} fused passes per wavefront; pass launches {launches // reps} per batch, mean "
      f"{ms / launches:.3f} ms = {pass_bytes / (ms / launches * 1e-3) / 1e9:.0f} GB/s algorithmic "
      f"({pass_bytes / (ms / launches * 1e-3) / 8e12 * 100:.0f} % of 8 TB/s)")
dev.close()
