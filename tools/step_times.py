#!/usr/bin/env python3
"""Wall time of each of the first ten walked SYN20 steps on a fresh context (4096^2 x 32, one sync per step): the first step pays the
lazy allocations, the next five the clock ramp (round 4, one box: 37.2 21.3 21.3 20.7 20.1 20.1 19.8 19.9 19.9 19.9 ms) -- why bench.py
defaults to --warmup 3."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from paos_amd import _lib
from paos_amd.chains import syn20_chain, syn20_wavelength
from paos_amd.run import run_batch
n, nb = 4096, 32
dev = _lib.DeviceFields(n, nb)
chains = [syn20_chain() for _ in range(nb)]
ts = []
for g in range(10):
    wls = [syn20_wavelength((g * nb + i) % 512) for i in range(nb)]
    t0 = time.perf_counter()
    run_batch(1.0, wls, n, 4, {"us": 0.0, "ut": 0.0}, chains, outputs=(), dev=dev, sync=False, keep_psf=True)
    dev.sync()
    ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{t:.1f}" for t in ts))
