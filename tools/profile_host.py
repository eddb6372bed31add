#!/usr/bin/env python3
"""cProfile of the host side of one run_batch step (where does Python time go?)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paos_amd import _lib  # noqa: E402
from paos_amd.chains import syn20_chain, syn20_wavelength  # noqa: E402
from paos_amd.run import run_batch  # noqa: E402

n, nb = int(sys.argv[1]), int(sys.argv[2])
dev = _lib.DeviceFields(n, nb)
wls = [syn20_wavelength(k) for k in range(nb)]
chains = [syn20_chain() for _ in range(nb)]
field = {"us": 0.0, "ut": 0.0}


def step():
    return run_batch(1.0, wls, n, 4, field, chains, outputs=(), dev=dev, sync=False)


step(); dev.sync()
t0 = time.perf_counter(); step(); t1 = time.perf_counter(); dev.sync(); t2 = time.perf_counter()
print(f"host time to enqueue one step: {1e3 * (t1 - t0):.1f} ms; until GPU done: {1e3 * (t2 - t0):.1f} ms")
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); dev.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
