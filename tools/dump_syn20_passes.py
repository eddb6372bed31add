#!/usr/bin/env python3
"""PAOS_DUMP_PASSES=1 python tools/dump_syn20_passes.py [grid]: every pass launch of one lean SYN20 step (two wavelengths of the
sweep) with what its slots carry and the planner's ranges -- "+pass" lines ride in the launch of the pass above them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PAOS_DUMP_PASSES", "1")
from paos_amd import _lib  # noqa: E402
from paos_amd.chains import syn20_chain, syn20_wavelength  # noqa: E402
from paos_amd.run import run_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wls = [syn20_wavelength(k) for k in (0, 31)]
dev = _lib.DeviceFields(n, len(wls))
run_batch(1.0, wls, n, 4, {"us": 0.0, "ut": 0.0}, [syn20_chain() for _ in wls], outputs=(), dev=dev, keep_psf=True)
dev.sync()
dev.close()
