#!/usr/bin/env python3
"""Experiment (round 3): do two half-batches on two contexts (= two HIP streams) overlap each other's passes?
A pass launch fills every CU with two workgroups of ONE kind (bound by the fp64 pipe when it skips loads / stores,
by HBM otherwise); two streams let a CU host one workgroup of each.  Compares wavefronts/s of one context of B
wavefronts per step with K contexts of B / K driven round-robin (no synchronisation inside the timed region)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paos_amd import _lib  # noqa: E402
from paos_amd.chains import syn20_chain, syn20_wavelength  # noqa: E402
from paos_amd.run import run_batch  # noqa: E402

ON_AXIS = {"us": 0.0, "ut": 0.0}


def measure(n, total, k, steps=20, warmup=5, stagger=0, pads=None):
    nb = total // k
    devs = []
    for i in range(k):  # PAOS_LDS_PAD is read when a context is created (round 4: one context's workgroups padded so
        os.environ["PAOS_LDS_PAD"] = str(pads[i] if pads else 0)  # that a CU holds ONE of them + one of the other's)
        devs.append(_lib.DeviceFields(n, nb))
    os.environ["PAOS_LDS_PAD"] = "0"
    wls = [[syn20_wavelength(i * nb + j) for j in range(nb)] for i in range(k)]
    chains = [[syn20_chain() for _ in range(nb)] for _ in range(k)]
    pending = [None] * k

    def release(i):
        if pending[i] is not None:
            for t in {rec["power_ticket"] for r in pending[i] for rec in r.values() if "power_ticket" in rec}:
                devs[i].norm2_release(t)

    def step():
        for i in range(k):
            release(i)
            pending[i] = run_batch(1.0, wls[i], n, 4, ON_AXIS, chains[i], outputs=(), dev=devs[i], sync=False, keep_psf=True)

    for _ in range(warmup):
        step()
    for d in devs:
        d.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for d in devs:
        d.sync()
    dt = time.perf_counter() - t0
    for d in devs:
        d.close()
    return total * steps / dt


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    for k, pads in ((1, None), (2, None), (2, (6144, 0)), (1, None), (2, None), (2, (6144, 0)), (2, (6144, 6144))):
        print(f"{n}^2, {total} wavefronts per step on {k} context(s) / stream(s), LDS padding {pads}: "
              f"{measure(n, total, k, pads=pads):.1f} wavefronts/s", flush=True)
