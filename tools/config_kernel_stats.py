#!/usr/bin/env python3
"""One BASELINE configuration as a batched sweep, for rocprofv3 --kernel-trace --stats:
   python tools/config_kernel_stats.py Excite_TEL 4096 32 [light]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from paos_amd import _lib
from paos_amd.chains import parse_config_variant
from paos_amd.run import run_batch

name, n, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
light = len(sys.argv) > 4 and sys.argv[4] == "light"
sweep = {"Excite_TEL": np.linspace(1.0, 4.0, 512)[:: 512 // nb][:nb], "Ariel_AIRS-CH0": np.linspace(1.95, 3.9, nb)}.get(name)
pup, par, wls, fields, chains = parse_config_variant(os.path.join("data", "lens", name + ".ini"), sweep, unignore=("Z1",) if "FGS" in name else ())
if len(chains) < nb:
    chains, wls = [chains[0]] * nb, [wls[0]] * nb
if light:
    chains = [{k: dict(it, save=it["name"] == "IMAGE_PLANE") for k, it in c.items()} for c in chains]
w = [1e-6 * x for x in wls]
dev = _lib.DeviceFields(n, nb)
stats = {}
for _ in range(4):
    run_batch(pup, w, n, par["zoom"], fields[0], chains, outputs=(), dev=dev, sync=True, keep_psf=True, stats=stats)
print(name, n, nb, "light" if light else "all saved", stats)
dev.close()
