#!/usr/bin/env python3
"""Generate golden vectors by running the PAOS reference itself (build container only).

    python tests/golden_tests/golden_tools/make_golden.py   # writes tests/golden/*.npz

The reference tree (/root/reference, read-only) is imported UNMODIFIED through
tests/golden_tests/golden_tools/ref_import.py (stub modules only for absent third-party packages).  The
only non-reference arithmetic in these vectors is the aperture-mask VALUES,
which come from oracle/aperture_np.py through the photutils stub classes
(photutils is absent; SURVEY.md 8c, "parity unpinned" at that boundary).
Everything committed under tests/golden/ is data: inputs and expected outputs.

Vector families:
  zernike_index.npz   j -> (m, n) and back, 4 orderings x 400 indices
  zernike_maps.npz    WFO.zernikes wfe maps, N=64, orderings x normalize x origin
  primitives.npz      lens / ptp / stw / wts / make_stop / zernikes / Magnification /
                      ChangeMedium on a seeded random field, N=64, incl. dx != dy
  scalars_<chain>.npz pilot-beam scalars after EVERY surface of a chain (N=64)
  run_<chain>.npz     end-to-end run(): complex field, amplitude, phase, wfe of the
                      saved surfaces (N=128 SYN20/Hubble, N=64 others)
  run_more_chains.npz the other eight runnable shipped lens files, first and last wavelength:
                      scalars + propagators of the saved surfaces, field of the last one (N=64)
  run_offaxis.npz     Hubble_simple, Ariel_AIRS-CH0 and SYN20 for an off-axis field point (the ray
                      vectors move the apertures, run.py:96-121): scalars + last field (N=64)
  orthonorm.npz       PolyOrthoNorm (covariance, M, polynomials) on an elliptical annulus,
                      WFO.zernikes(orthonorm=True), and run() of SYN20 with Zorthonorm (8f-3)
  kat.npz             the reference's own recorded known answers
"""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from oracle import aperture_np  # noqa: E402
import ref_import  # noqa: E402

ref_import.install(aperture_np.EllipticalAperture, aperture_np.RectangularAperture)

from paos.classes.abcd import ABCD as RefABCD  # noqa: E402
from paos.classes.wfo import WFO as RefWFO  # noqa: E402
from paos.classes.zernike import PolyOrthoNorm as RefPolyOrthoNorm  # noqa: E402
from paos.classes.zernike import Zernike as RefZernike  # noqa: E402
from paos.core.parseConfig import parse_config as ref_parse  # noqa: E402
from paos.core.run import run as ref_run  # noqa: E402
from paos.core.coordinateBreak import coordinate_break as ref_cb  # noqa: E402

from paos_amd.chains import (  # noqa: E402
    inject_wfe,
    read_wfe_table,
    syn20_chain,
    syn20_orthonorm_chain,
)

OUT = os.path.join(ROOT, "tests", "golden")
LENS = "/root/reference/lens data"
WFE = "/root/reference/wfe data/wfe_realization_SN20210914.csv"
ORDERINGS = ("ansi", "noll", "fringe", "standard")


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB  {len(arrays)} arrays")


def beam_scalars(w):
    return np.array(
        [w.wl, w.z, w.w0, w.zw0, w.zr, w.dx, w.dy, w.C, w.fratio, w.wz, w.distancetofocus],
        dtype=np.float64,
    )


def seeded_field(n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))


def gen_zernike_index():
    out = {}
    for o in ORDERINGS:
        m, n = RefZernike.j2mn(400, o)
        out[f"{o}_m"] = m
        out[f"{o}_n"] = n
        out[f"{o}_j"] = np.asarray(RefZernike.mn2j(m, n, o))
    save("zernike_index.npz", **out)


def gen_zernike_maps():
    out = {}
    rng = np.random.default_rng(7)
    coef = rng.normal(0, 30e-9, 36)
    out["coef"] = coef
    for o in ORDERINGS:
        for norm in (True, False):
            for origin in ("x", "y"):
                w = RefWFO(1.0, 1.0e-6, 64, 2)
                wfe = w.zernikes(np.arange(36), coef, o, norm, 0.5, origin=origin)
                key = f"{o}_{int(norm)}_{origin}"
                out[key + "_wfe"] = wfe.filled(0.0)
                out[key + "_mask"] = np.ma.getmaskarray(wfe)
                out[key + "_u"] = w.wfo
    # anamorphic sampling + offset radius (dx != dy), 15 terms
    w = RefWFO(1.0, 2.0e-6, 64, 2)
    w.Magnification(1.5, 0.75)
    wfe = w.zernikes(np.arange(15), coef[:15], "noll", True, 0.62, origin="x")
    out["anam_wfe"] = wfe.filled(0.0)
    out["anam_mask"] = np.ma.getmaskarray(wfe)
    out["anam_u"] = w.wfo
    out["anam_dxdy"] = np.array([w.dx, w.dy])
    save("zernike_maps.npz", **out)


def gen_primitives():
    """Each case: set up a reference WFO, overwrite the field with a seeded random
    one, apply one primitive, store scalars before/after and the output field."""
    out = {}
    n = 64
    u0 = seeded_field(n, 11)
    out["u0"] = u0

    def fresh(wl=3.0e-6, anam=False):
        w = RefWFO(1.0, wl, n, 4)
        if anam:
            w.Magnification(1.25, 0.8)  # (My, Mx) -> dy *= 1.25, dx *= 0.8
        w._wfo = u0.copy()
        return w

    def record(tag, w, before):
        out[tag + "_before"] = before
        out[tag + "_after"] = beam_scalars(w)
        out[tag + "_u"] = w.wfo

    for anam in (False, True):
        sfx = "_anam" if anam else ""
        w = fresh(anam=anam)
        b = beam_scalars(w)
        w.make_stop()
        record("make_stop" + sfx, w, b)

        for fl in (10.0, -3.0, 0.4):
            w = fresh(anam=anam)
            b = beam_scalars(w)
            w.lens(fl)
            record(f"lens_{fl}{sfx}", w, b)

        for dz in (0.5, -0.25, 1.0e-8):
            w = fresh(anam=anam)
            b = beam_scalars(w)
            w.ptp(dz)
            record(f"ptp_{dz}{sfx}", w, b)

        # stw needs a curved reference surface: lens first, then stw to the waist
        for fl in (10.0, -7.0):
            w = fresh(anam=anam)
            w.lens(fl)
            w._wfo = u0.copy()
            b = beam_scalars(w)
            dz = w.zw0 - w.z
            w.stw(dz)
            record(f"stw_{fl}{sfx}", w, b)
            out[f"stw_{fl}{sfx}_dz"] = np.float64(dz)

        for dz in (2.0, -1.5):
            w = fresh(anam=anam)
            b = beam_scalars(w)
            w.wts(dz)
            record(f"wts_{dz}{sfx}", w, b)

    # propagate(): all four regimes, from the start of a beam and after a lens
    cases = {
        "II": (lambda w: None, 1.0),
        "OI": (lambda w: w.lens(10.0), 10.0),
        "IO": (lambda w: None, 4.0e6),
        "OO": (lambda w: w.lens(10.0), 20.0),
    }
    for tag, (prep, dist) in cases.items():
        w = RefWFO(1.0, 3.0e-6, n, 4)
        prep(w)
        w._wfo = u0.copy()
        b = beam_scalars(w)
        w.propagate(dist)
        assert w.propagator == tag, (tag, w.propagator)
        record("propagate_" + tag, w, b)
        out["propagate_" + tag + "_dz"] = np.float64(dist)

    # scalar-only updates
    w = fresh()
    w.lens(5.0)
    b = beam_scalars(w)
    w.Magnification(1.3, 0.7)
    out["magnification_before"] = b
    out["magnification_after"] = beam_scalars(w)
    w = fresh()
    w.lens(5.0)
    b = beam_scalars(w)
    w.ChangeMedium(0.66)
    out["changemedium_before"] = b
    out["changemedium_after"] = beam_scalars(w)

    # aperture products (mask values from oracle/aperture_np.py -- unpinned boundary)
    w = fresh()
    w.aperture(0.1, -0.2, hx=0.9, hy=0.6, shape="elliptical")
    out["aperture_ell_u"] = w.wfo
    w = fresh()
    w.aperture(0.0, 0.0, hx=0.7, hy=0.3, shape="rectangular", obscuration=True)
    out["aperture_rect_obsc_u"] = w.wfo
    save("primitives.npz", **out)


def all_saved(chain):
    chain = copy.deepcopy(chain)
    for item in chain.values():
        item["save"] = True
    return chain


def chain_specs():
    specs = {}
    for name in ("Hubble_simple", "Excite_TEL", "Ariel_AIRS-CH0", "Ariel_FGS-FGS1"):
        pup, par, wls, fields, chains = ref_parse(os.path.join(LENS, name + ".ini"))
        specs[name] = dict(pup=pup, wl=1.0e-6 * wls[0], zoom=par["zoom"], field=fields[0],
                           chain=chains[0])
    specs["SYN20"] = dict(pup=1.0, wl=1.0e-6, zoom=4, field={"us": 0.0, "ut": 0.0},
                          chain=syn20_chain(abcd_cls=RefABCD))
    return specs


def gen_chain_scalars(specs):
    for name, s in specs.items():
        ret = ref_run(s["pup"], s["wl"], 64, s["zoom"], s["field"], all_saved(s["chain"]))
        nums = np.array(sorted(ret.keys()))
        tab = np.array(
            [
                [ret[k]["wl"], ret[k]["dx"], ret[k]["dy"], ret[k]["wz"], ret[k]["distancetofocus"],
                 ret[k]["fratio"]]
                for k in nums
            ]
        )
        props = np.array([ret[k]["propagator"] for k in nums])
        abcdt = np.array([ret[k]["ABCDt"]() for k in nums])
        abcds = np.array([ret[k]["ABCDs"]() for k in nums])
        extent = np.array([ret[k]["extent"] for k in nums])
        save(f"scalars_{name}.npz", nums=nums, table=tab, propagator=props, ABCDt=abcdt,
             ABCDs=abcds, extent=extent)


def gen_chain_runs(specs):
    sizes = {"SYN20": 128, "Hubble_simple": 128}
    for name, s in specs.items():
        n = sizes.get(name, 64)
        ret = ref_run(s["pup"], s["wl"], n, s["zoom"], s["field"], s["chain"])
        out = {"nums": np.array(sorted(ret.keys())), "gridsize": np.int64(n)}
        for k in sorted(ret.keys()):
            r = ret[k]
            out[f"S{k:02d}_wfo"] = r["wfo"]
            out[f"S{k:02d}_scal"] = np.array([r["wl"], r["dx"], r["dy"], r["wz"],
                                              r["distancetofocus"], r["fratio"]])
            out[f"S{k:02d}_prop"] = np.array(r["propagator"])
            if "wfe" in r:
                out[f"S{k:02d}_wfe"] = r["wfe"].filled(0.0)
        save(f"run_{name}.npz", **out)

    # Monte-Carlo injection (pipeline.py:116-129): FGS1 with Z1 un-ignored is not
    # expressible without editing the .ini, so the golden uses SYN20's Z1 surface
    # with draws 0 and 1 of the shipped WFE table.
    _, _, _, table = read_wfe_table(WFE)
    s = specs["SYN20"]
    for col in (0, 1):
        chain = inject_wfe(s["chain"], table[:, col])
        ret = ref_run(s["pup"], s["wl"], 64, s["zoom"], s["field"], chain)
        save(f"run_SYN20_wfe{col}.npz", wfo=ret[20]["wfo"], draw_nm=table[:, col],
             scal=np.array([ret[20]["dx"], ret[20]["dy"], ret[20]["fratio"]]))


MORE_CHAINS = ("Ariel_AIRS-CH1", "Ariel_FGS-FGS2", "Ariel_FGS-NIRSpec", "Ariel_FGS-VISPhot",
               "lens_file_TA_Ground", "lens_file_TA_OGSE_Ground", "lens_file_template", "periscope")


def gen_more_chains():
    out = {}
    for name in MORE_CHAINS:
        pup, par, wls, fields, chains = ref_parse(os.path.join(LENS, name + ".ini"))
        for tag, iw in (("first", 0), ("last", len(wls) - 1)):
            ret = ref_run(pup, 1.0e-6 * wls[iw], 64, par["zoom"], fields[0], chains[iw])
            nums = np.array(sorted(ret.keys()))
            key = f"{name}_{tag}"
            out[key + "_wl_um"] = np.float64(wls[iw])
            out[key + "_nums"] = nums
            out[key + "_table"] = np.array([[ret[k]["wl"], ret[k]["dx"], ret[k]["dy"], ret[k]["wz"],
                                             ret[k]["distancetofocus"], ret[k]["fratio"]] for k in nums])
            out[key + "_propagator"] = np.array([ret[k]["propagator"] for k in nums])
            out[key + "_wfo"] = ret[nums[-1]]["wfo"]
    save("run_more_chains.npz", **out)


OFFAXIS = {"us": 5.0e-5, "ut": -2.0e-5}


def gen_offaxis(specs):
    out = {"us": np.float64(OFFAXIS["us"]), "ut": np.float64(OFFAXIS["ut"])}
    for name in ("Hubble_simple", "Ariel_AIRS-CH0", "SYN20"):
        s = specs[name]
        ret = ref_run(s["pup"], s["wl"], 64, s["zoom"], dict(OFFAXIS), s["chain"])
        nums = np.array(sorted(ret.keys()))
        out[name + "_nums"] = nums
        out[name + "_table"] = np.array([[ret[k]["wl"], ret[k]["dx"], ret[k]["dy"], ret[k]["wz"],
                                          ret[k]["distancetofocus"], ret[k]["fratio"]] for k in nums])
        out[name + "_wfo"] = ret[nums[-1]]["wfo"]
        out[name + "_first_wfo"] = ret[nums[0]]["wfo"]
    save("run_offaxis.npz", **out)


def gen_orthonorm():
    """PolyOrthoNorm (zernike.py:320-402) and the Zorthonorm path of run() (run.py:133-141)."""
    out = {}
    n = 64
    x = np.linspace(-1.0, 1.0, n)
    xx, yy = np.meshgrid(x, x)
    mask = (xx**2 + (yy / 0.6) ** 2 > 1.0) | (xx**2 + yy**2 < 0.2**2)  # elliptical annulus
    phi = np.arctan2(yy, xx)
    out["poly_mask"] = mask
    out["poly_rho"] = np.sqrt(xx**2 + yy**2)
    out["poly_phi"] = phi
    for ordering in ("noll", "ansi"):
        rho = np.ma.MaskedArray(data=np.sqrt(xx**2 + yy**2), mask=mask.copy(), fill_value=0.0)
        plain = RefZernike(15, rho.copy(), phi, ordering=ordering, normalize=True)
        out[f"poly_{ordering}_cov"] = plain.cov()
        poly = RefPolyOrthoNorm(15, rho, phi, ordering=ordering, normalize=True)
        out[f"poly_{ordering}_M"] = poly.M
        out[f"poly_{ordering}_U"] = poly().filled(0.0)
        out[f"poly_{ordering}_Umask"] = np.ma.getmaskarray(poly())
        coeff = np.linspace(-1.0, 1.0, 15)
        out[f"poly_{ordering}_toZernike"] = poly.toZernike(coeff)

    # WFO.zernikes with orthonorm=True and an explicit mask, seeded field
    w = RefWFO(1.0, 1.2e-6, n, 2)
    w._wfo = seeded_field(n, 77)
    xs = (np.arange(n) - n // 2) * w.dx
    gx, gy = np.meshgrid(xs, xs)
    zmask = (gx / 0.45) ** 2 + (gy / 0.3) ** 2 > 1.0
    coeff = 1.0e-9 * np.array([0.0, 30.0, -20.0, 50.0, 10.0, -15.0, 25.0, 5.0, -8.0, 12.0])
    out["wfo_in"] = w.wfo.copy()
    out["wfo_zmask"] = zmask
    out["wfo_coeff"] = coeff
    wfe = w.zernikes(np.arange(10), coeff, "noll", True, 0.5, origin="x", orthonorm=True, mask=zmask)
    out["wfo_wfe"] = wfe.filled(0.0)
    out["wfo_wfe_mask"] = np.ma.getmaskarray(wfe)
    out["wfo_out"] = w.wfo

    # run(): SYN20 with an elliptical pupil on the Zernike surface and Zorthonorm
    for size in (64, 128):
        chain = syn20_orthonorm_chain(abcd_cls=RefABCD)
        chain[2]["save"] = True
        ret = ref_run(1.0, 1.0e-6, size, 4, {"us": 0.0, "ut": 0.0}, chain)
        out[f"run{size}_S02_wfe"] = ret[2]["wfe"].filled(0.0)
        out[f"run{size}_S02_wfe_mask"] = np.ma.getmaskarray(ret[2]["wfe"])
        out[f"run{size}_S02_wfo"] = ret[2]["wfo"]
        out[f"run{size}_S20_wfo"] = ret[20]["wfo"]
    save("orthonorm.npz", **out)


def gen_kat():
    w = RefWFO(1.0, 3e-6, 256, 4)
    w.lens(10.0)
    lens_kat = beam_scalars(w)
    w.make_stop()
    w.propagate(10.0)
    prop_kat = beam_scalars(w)
    w2 = RefWFO(1.1, 0.55e-6, 1024, 4)
    wfe = w2.zernikes(np.arange(6), np.array([0, 10, 0, -30, 20, 0]) * 1e-9, "noll", True, 0.55)
    # coordinate break: tilts as shipped files use them plus a generic case
    cb = []
    for args in ((0.0, 0.0, 12.0, 0.0), (0.01, -0.02, 5.0, 0.0), (0.0, 0.003, -20.0, 7.5)):
        vt, vs = ref_cb(np.array([0.001, 0.01]), np.array([-0.002, 0.02]), *args, 0.0)
        cb.append(np.concatenate([args, vt, vs]))
    save("kat.npz", lens=lens_kat, propagate=prop_kat, zernike_std=np.float64(np.std(wfe)),
         zernike_pv=np.float64(wfe.max() - wfe.min()), coordinate_break=np.array(cb))


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_zernike_index()
    gen_zernike_maps()
    gen_primitives()
    specs = chain_specs()
    gen_chain_scalars(specs)
    gen_chain_runs(specs)
    gen_more_chains()
    gen_offaxis(specs)
    gen_orthonorm()
    gen_kat()


if __name__ == "__main__":
    main()
