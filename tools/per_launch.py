"""Mean duration of every pass launch of one SYN20 step (4096^2 c128, batch 32), in launch order, with the class tag
(bit 0 tiles skipped, 1 loads skipped, 2 stores skipped, 3 PSF stored): [PAOS_NO_PRUNE=1 ...] python tools/per_launch.py [grid [fp64|fp32]]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from paos_amd import _lib
from paos_amd.chains import syn20_chain, syn20_wavelength
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
precision = sys.argv[2] if len(sys.argv) > 2 else "fp64"
nb = max(32, 32 * (4096 // n) ** 2)
chains = [syn20_chain() for _ in range(nb)]
dev = _lib.DeviceFields(n, nb, precision)
# (the same 32 wavelengths every step: per-launch times of ONE program; bench.py itself walks the sweep)
m = bench.measure(dev, n, precision, lambda g: [syn20_wavelength(k) for k in range(nb)], chains, 4, 1)
ms, tags = m["launch_ms"], m["launch_tags"]
per = len(ms) // 4
ms = ms.reshape(4, per).mean(axis=0); tags = tags[:per]
os.environ["PAOS_DUMP_PASSES"] = "0"
print("launches per step", per)
for i, (t, g) in enumerate(zip(ms, tags)):
    print(f"  pass {i:2d}  tag {int(g):2d}  {t:.3f} ms")
dev.close()
