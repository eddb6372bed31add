#!/usr/bin/env python3
"""Where the host's time goes in one run_batch step (python tools/profile_host.py GRID BATCH): wall time to enqueue a step
against the time until the GPU is done, the library calls (ctypes: C++ planning, staging copies, launches) timed one by
one, and a cProfile of the Python side."""
import cProfile
import collections
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
chains = [syn20_chain() for _ in range(nb)]
field = {"us": 0.0, "ut": 0.0}
g = [0]


def step():
    wls = [syn20_wavelength((g[0] * nb + i) % 512) for i in range(nb)]  # a walked sweep, like bench.py
    g[0] += 1
    res = run_batch(1.0, wls, n, 4, field, chains, outputs=(), dev=dev, sync=False, keep_psf=True)
    for t in {rec["power_ticket"] for r in res for rec in r.values() if "power_ticket" in rec}:
        t.release()


for _ in range(3):
    step()
dev.sync()
ts, tg = [], []
for _ in range(8):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); dev.sync(); t2 = time.perf_counter()
    ts.append(1e3 * (t1 - t0)); tg.append(1e3 * (t2 - t0))
print(f"{n}^2 x {nb}: host time to enqueue one step: min {min(ts):.1f} median {sorted(ts)[4]:.1f} ms; until GPU done: median {sorted(tg)[4]:.1f} ms")

# the library calls, one by one
acc, cnt = collections.Counter(), collections.Counter()
for name in dir(_lib.DeviceFields):
    f = getattr(_lib.DeviceFields, name)
    if name.startswith("_") or not callable(f):
        continue

    def wrap(f=f, name=name):
        def timed(*a, **k):
            t0 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[name] += time.perf_counter() - t0
                cnt[name] += 1
        return timed
    setattr(_lib.DeviceFields, name, wrap())
t0 = time.perf_counter()
for _ in range(5):
    step()
tot = (time.perf_counter() - t0) / 5 * 1e3
dev.sync()
print(f"library calls per step (of {tot:.1f} ms to enqueue):")
for k, v in acc.most_common(12):
    print(f"  {k:24s} {v / 5 * 1e3:7.3f} ms  ({cnt[k] / 5:.0f} calls)")
print(f"  {'all library calls':24s} {sum(acc.values()) / 5 * 1e3:7.3f} ms")
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); dev.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
