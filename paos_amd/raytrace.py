"""Diagnostic paraxial ray trace of an optical chain (host scalars only).

Drop-in for ``paos.core.raytrace.raytrace`` (reference paos/core/raytrace.py:7-60), the check a caller runs on a
parsed lens file before a propagation (pipeline.py:92-98): one ray, given by its position and slopes in the
tangential (y) and sagittal (x) planes, is carried through every surface -- a coordinate break first re-expresses it
in the tilted / decentred frame, then the surface's two ray-transfer matrices act on it -- and reported per surface
in the reference's text format, which downstream logs and notebooks show verbatim.
"""
import numpy as np

from .coordinate_break import coordinate_break

# "S03 - M1              y:  0.000mm ut: 0.000e+00 rad x:  0.000mm us: 0.000e+00 rad" (raytrace.py:53-55)
_LINE = "S{:02d} - {:15s} y:{:7.3f}mm ut:{:10.3e} rad x:{:7.3f}mm us:{:10.3e} rad"


def trace(field, opt_chain, x=0.0, y=0.0):
    """Yield ``(key, name, vt, vs)`` behind every surface: ``vt = [y, ut]`` and ``vs = [x, us]`` in metres and
    radians, the ray vectors ``run`` centres its apertures on (run.py:97-108 uses the same recurrence)."""
    vt = np.array([y, field["ut"]])
    vs = np.array([x, field["us"]])
    for key, item in opt_chain.items():
        if item["type"] == "Coordinate Break":
            vt, vs = coordinate_break(vt, vs, item["xdec"], item["ydec"], item["xrot"], item["yrot"], 0.0)
        vt = item["ABCDt"]() @ vt
        vs = item["ABCDs"]() @ vs
        yield key, item["name"], vt, vs


def raytrace(field, opt_chain, x=0.0, y=0.0):
    """``field = {'ut': slope_y, 'us': slope_x}``, ``opt_chain`` as returned by ``parse_config``, ``(x, y)`` the
    ray's starting point in metres.  Returns one string per surface (positions in mm)."""
    return [_LINE.format(key, name, 1000 * vt[0], vt[1], 1000 * vs[0], vs[1])
            for key, name, vt, vs in trace(field, opt_chain, x, y)]
