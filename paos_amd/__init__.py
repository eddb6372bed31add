"""paos_amd -- MI355X-native Fresnel propagation core behind the PAOS API.

Drop-in for the wavefront-propagation hot path of arielmission-space/PAOS
(``paos.core.run.run`` driving ``paos.classes.wfo.WFO``): the same
``paos_amd.run.run(pupil_diameter, wavelength, gridsize, zoom, field, opt_chain)`` and
``WFO(beam_diameter, wl, grid_size, zoom)`` surfaces, with the N x N complex
field living in HBM and every field operator a hand-written HIP kernel for
gfx950 reached through a ctypes C-ABI (``libpaoship.so``, include/paos_hip.h).

Importing the package does not need a GPU; creating a ``WFO`` or calling
``run`` loads the HIP library and fails loudly when it (or a GPU) is missing --
there is no CPU fallback in the product path.
"""
from .abcd import ABCD
from .coordinate_break import coordinate_break
from .parse_config import parse_config
from .zernike import Zernike

from .raytrace import raytrace

__all__ = ["ABCD", "WFO", "Zernike", "coordinate_break", "parse_config", "raytrace", "run_batch", "run_sharded"]
__version__ = "0.1.0"


def __getattr__(name):
    # device-backed entry points are imported on first use so that host-only helpers (parser,
    # ABCD, chain builders) work on machines without a GPU.  ``paos_amd.run`` is the MODULE
    # (like paos.core.run); its function is ``paos_amd.run.run``.
    import importlib

    if name == "WFO":
        return importlib.import_module(".wfo", __name__).WFO
    if name == "run_batch":
        return importlib.import_module(".run", __name__).run_batch
    if name == "run_sharded":
        return importlib.import_module(".dist", __name__).run_sharded
    raise AttributeError(name)
