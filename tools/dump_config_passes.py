import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from paos_amd.chains import parse_config_variant, syn20_chain, syn20_wavelength
from paos_amd.run import run_batch
LENS = os.path.join(os.getcwd(), "data", "lens")
for name, sweep, n in (("Ariel_AIRS-CH0", np.linspace(1.95, 3.9, 4), 2048), ("Ariel_FGS-FGS1", None, 2048), ("Excite_TEL", np.linspace(1.0, 4.0, 4), 4096)):
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, name + ".ini"), sweep, unignore=("Z1",) if "FGS" in name else ())
    w = [1e-6 * x for x in wls][:4]; chains = chains[:4]
    for light in (False, True):
        ch = [{k: dict(it, save=(it["name"] == "IMAGE_PLANE") if light else it["save"]) for k, it in c.items()} for c in chains]
        print(f"#### {name} light={light}", file=sys.stderr, flush=True)
        run_batch(pup, w, n, par["zoom"], fields[0], ch, outputs=(), keep_psf=True)
