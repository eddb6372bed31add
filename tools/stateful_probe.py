#!/usr/bin/env python3
"""Diagnostic: the second walked step on a long-lived context against the same step on a fresh one (PSF of item 0 and 31),
with the library's state-carrying features switched off one at a time (environment switches, read per call)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paos_amd import _lib  # noqa: E402
from paos_amd.chains import syn20_chain, syn20_wavelength  # noqa: E402
import paos_amd.run as prun  # noqa: E402
from paos_amd.run import run_batch  # noqa: E402

n, nb = int(sys.argv[1]), int(sys.argv[2])
ON_AXIS = {"us": 0.0, "ut": 0.0}
chains = [syn20_chain() for _ in range(nb)]
wl = lambda g: [syn20_wavelength((g * nb + i) % 512) for i in range(nb)]  # noqa: E731


def two_steps():
    dev = _lib.DeviceFields(n, nb)
    try:
        for g in (0, 1):
            res = run_batch(1.0, wl(g), n, 4, ON_AXIS, chains, outputs=(), dev=dev, sync=False, keep_psf=True)
            for t in {rec["power_ticket"] for r in res for rec in r.values() if "power_ticket" in rec}:
                t.release()
        return [dev.psf_fetch(i) for i in (0, nb - 1)]
    finally:
        dev.close()


def fresh():
    dev = _lib.DeviceFields(n, nb)
    try:
        run_batch(1.0, wl(1), n, 4, ON_AXIS, chains, outputs=(), dev=dev, sync=True, keep_psf=True)
        return [dev.psf_fetch(i) for i in (0, nb - 1)]
    finally:
        dev.close()


def report(tag):
    a, b = two_steps(), fresh()
    for k, (x, y) in enumerate(zip(a, b)):
        d = np.abs(x - y)
        where = np.unravel_index(np.argmax(d), d.shape)
        print(f"{tag:28s} item {(0, nb - 1)[k]:3d}: max |stateful - fresh| / max = {d.max() / y.max():.3e} at {where}, pixels that differ: {int((d > 0).sum())}", flush=True)


report("default")
prun.START_BOX = False
report("START_BOX off")
prun.START_BOX = True
for name in ("PAOS_FUSE_PAIRS", "PAOS_SHARE_START", "PAOS_SHARE_WFE", "PAOS_BATCHED_RECORDS", "PAOS_COMPACT_GRID"):
    os.environ[name] = "0"
    report(name + "=0")
    del os.environ[name]
