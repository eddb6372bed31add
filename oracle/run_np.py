"""CPU restatement of the PAOS propagation loop.  TEST INFRASTRUCTURE ONLY.

Follows paos/core/run.py:30-228 surface by surface: coordinate break
(run.py:81-91), aperture (run.py:96-122), stop (run.py:125-127), Zernike
(run.py:129-152), push_results for EVERY surface (run.py:12-27,179), scalar
gating of Magnification / ChangeMedium / lens / propagate (run.py:181-207),
chief-ray and ABCD bookkeeping (run.py:209-219) and the saved-surface dict
(run.py:221-223).  Grid Sag / PSD surfaces are out of scope.

``trace`` (optional list) receives one tuple per executed WFO call so tests and
DESIGN.md can count FFTs exactly as the reference executes them.
"""
from copy import deepcopy

import numpy as np

from .paraxial_np import RayMatrix, tilt_decentre
from .pop_numpy import RefWFO


def snapshot(wfo):
    """run.py:12-27."""
    return {
        "amplitude": wfo.amplitude,
        "wz": wfo.wz,
        "distancetofocus": wfo.distancetofocus,
        "fratio": wfo.fratio,
        "phase": wfo.phase,
        "dx": wfo.dx,
        "dy": wfo.dy,
        "wfo": wfo.wfo,
        "wl": wfo.wl,
        "extent": wfo.extent,
        "propagator": wfo.propagator,
    }


def run(pupil_diameter, wavelength, gridsize, zoom, field, opt_chain, trace=None,
        light=False):
    """``light=True`` skips the per-surface amplitude/phase/copy of unsaved
    surfaces (the reference always computes them, run.py:179); results are
    identical, it only bounds the CPU-baseline cost honestly when asked."""
    assert isinstance(opt_chain, dict), "opt_chain must be a dict"
    out = {}
    vt = np.array([0.0, field["ut"]])
    vs = np.array([0.0, field["us"]])
    acc_t = RayMatrix()
    acc_s = RayMatrix()
    wfo = RefWFO(pupil_diameter, wavelength, gridsize, zoom)

    def note(*rec):
        if trace is not None:
            trace.append(rec)

    for _, item in opt_chain.items():
        if item["type"] == "Coordinate Break":
            vt, vs = tilt_decentre(vt, vs, item["xdec"], item["ydec"], item["xrot"],
                                   item["yrot"], 0.0)
        rec = {"aperture": None}
        if "aperture" in item:
            ap = item["aperture"]
            xdec = ap["xc"] if np.isfinite(ap["xc"]) else vs[0]
            ydec = ap["yc"] if np.isfinite(ap["yc"]) else vt[0]
            xrad = ap["xrad"]
            yrad = ap["yrad"]
            xrad *= np.sqrt(1 / (vs[1] ** 2 + 1))
            yrad *= np.sqrt(1 / (vt[1] ** 2 + 1))
            xaper = xdec - vs[0]
            yaper = ydec - vt[0]
            obsc = ap["type"] != "aperture"
            if np.all(np.isfinite([xrad, yrad])):
                note("aperture", ap["shape"], xaper, yaper, xrad, yrad, obsc)
                rec["aperture"] = wfo.aperture(xaper, yaper, hx=xrad, hy=yrad,
                                               shape=ap["shape"], obscuration=obsc)
        if item["is_stop"]:
            note("make_stop")
            wfo.make_stop()
        if item["type"] == "Zernike":
            radius = item["Zradius"] if np.isfinite(item["Zradius"]) else wfo.wz
            note("zernikes", radius)
            zmask = False
            if item["Zorthonorm"]:  # run.py:133-141: the pupil of THIS surface's aperture
                assert "aperture" in item, "Zorthonorm requires aperture"
                zmask = ~rec["aperture"].to_mask(method="exact").to_image(wfo._wfo.shape).astype(bool)
            rec["wfe"] = wfo.zernikes(item["Zindex"], item["Z"], item["Zordering"],
                                      item["Znormalize"], radius, origin=item["Zorigin"],
                                      orthonorm=item["Zorthonorm"], mask=zmask)
        if item["type"] in ("Grid Sag", "PSD"):
            raise NotImplementedError(f"surface type {item['type']} is out of scope")

        if item["save"] or not light:
            rec.update(snapshot(wfo))

        Ms = item["ABCDs"].M
        Mt = item["ABCDt"].M
        fl = np.inf if (item["ABCDt"].power == 0) else item["ABCDt"].cout / item["ABCDt"].power
        T = item["ABCDt"].cout * item["ABCDt"].thickness
        n1n2 = item["ABCDt"].n1n2
        if Mt != 1.0 or Ms != 1.0:
            note("Magnification", Mt, Ms)
            wfo.Magnification(Mt, Ms)
        if np.abs(n1n2) != 1.0:
            note("ChangeMedium", n1n2)
            wfo.ChangeMedium(n1n2)
        if np.isfinite(fl):
            note("lens", fl)
            wfo.lens(fl)
        if np.isfinite(T) and np.abs(T) > 1e-10:
            z0, c0 = wfo.z, wfo.C
            wfo.propagate(T)
            note("propagate", T, wfo.propagator)
        vt = item["ABCDt"]() @ vt
        vs = item["ABCDs"]() @ vs
        acc_t = item["ABCDt"] * acc_t
        acc_s = item["ABCDs"] * acc_s
        rec["ABCDt"] = acc_t
        rec["ABCDs"] = acc_s
        if item["save"]:
            out[item["num"]] = deepcopy(rec)
        del rec
    return out
