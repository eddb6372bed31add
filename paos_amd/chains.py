"""Optical-chain builders for benchmarks, tests and Monte-Carlo batches.

* :func:`syn20_chain` -- the synthetic 20-surface relay SURVEY.md 8d defines
  for the headline metric (none of the shipped lens files has 20 surfaces).  It
  is a plain ``opt_chain`` dict, built the way the reference's
  notebook/ValidateThicklens.ipynb cell 2 builds chains by hand.
* :func:`inject_wfe` -- the Monte-Carlo wavefront-error injection of the
  reference pipeline (paos/core/pipeline.py:116-129): column ``k`` of the WFE
  realisation table (nm) replaces the coefficients of the surface named ``Z1``.
* :func:`read_wfe_table` -- reader for ``wfe data/wfe_realization_*.csv``
  (3 comment lines, then rows J, N, M, WFE000..WFE999).
"""
import configparser
import copy
import os
import tempfile

import numpy as np

from .abcd import ABCD

SYN20_SEED = 20210914


def syn20_coefficients(rng_seed=SYN20_SEED, sigma=20.0e-9):
    """[0, 0, 0] + 33 draws of N(0, 20 nm) from default_rng(20210914)."""
    rng = np.random.default_rng(rng_seed)
    return np.concatenate([np.zeros(3), rng.normal(0.0, sigma, 33)])


def syn20_chain(coefficients=None, focal=10.0, gap=0.1, abcd_cls=ABCD):
    """20 surfaces: stop + Zernike, five relays (lens f | free space f | lens f
    with a 0.5 m stop-sized aperture and a short gap), a final lens, a 2 mm field
    stop and the image plane.  ``abcd_cls`` lets the golden-vector generator
    build the same chain out of the reference's own ABCD class."""
    if coefficients is None:
        coefficients = syn20_coefficients()
    coefficients = np.asarray(coefficients, dtype=np.float64)

    def pair(thickness=0.0, curvature=0.0):
        return (
            abcd_cls(thickness=thickness, curvature=curvature),
            abcd_cls(thickness=thickness, curvature=curvature),
        )

    def pupil(kind="aperture", shape="elliptical", rad=0.5):
        return {"shape": shape, "type": kind, "xrad": rad, "yrad": rad, "xc": 0.0, "yc": 0.0}

    chain = {}

    def add(kind, name, thickness=0.0, curvature=0.0, **extra):
        num = len(chain) + 1
        t, s = pair(thickness, curvature)
        item = {"num": num, "type": kind, "name": name, "is_stop": False, "save": False,
                "ABCDt": t, "ABCDs": s}
        item.update(extra)
        chain[num] = item

    add("Standard", "STOP", is_stop=True, save=True, aperture=pupil())
    add("Zernike", "Z1", Zindex=np.arange(coefficients.size, dtype=np.int64), Z=coefficients,
        Zordering="standard", Znormalize=True, Zradius=0.5, Zorigin="x", Zorthonorm=False)
    for k in range(5):
        add("Paraxial Lens", f"R{k}a", thickness=focal, curvature=1.0 / focal)
        add("Standard", f"R{k}b", thickness=focal)
        add("Paraxial Lens", f"R{k}c", thickness=gap, curvature=1.0 / focal, aperture=pupil())
    add("Paraxial Lens", "L18", thickness=focal, curvature=1.0 / focal)
    add("Standard", "FIELD_STOP", aperture=pupil(shape="rectangular", rad=2.0e-3))
    add("Standard", "IMAGE_PLANE", save=True)
    assert len(chain) == 20
    return chain


def syn20_orthonorm_chain(coefficients=None, yrad=0.35, abcd_cls=ABCD):
    """SYN20 whose Zernike surface carries an elliptical pupil (0.5 x ``yrad`` m) and expands the
    WFE in polynomials orthonormalised over that pupil (``Zorthonorm``, run.py:133-141)."""
    chain = syn20_chain(coefficients, abcd_cls=abcd_cls)
    z1 = chain[2]
    assert z1["type"] == "Zernike"
    z1["Zorthonorm"] = True
    z1["aperture"] = {"shape": "elliptical", "type": "aperture", "xrad": 0.5, "yrad": yrad,
                      "xc": 0.0, "yc": 0.0}
    return chain


def syn20_wavelength(k, base=1.0e-6, steps=512):
    """Wavelength sweep of SURVEY.md 8d: lambda_k = 1 um * (1 + k/512)."""
    return base * (1.0 + k / float(steps))


def read_wfe_table(path):
    """Return (J, N, M, coefficients[nterms, ndraws]) in nanometres."""
    rows = []
    with open(path) as fh:
        for line in fh:
            if line.startswith("#") or not line.strip():
                continue
            rows.append([float(t) for t in line.strip().rstrip(",").split(",")])
    table = np.array(rows, dtype=np.float64)
    return table[:, 0].astype(int), table[:, 1].astype(int), table[:, 2].astype(int), table[:, 3:]


def inject_wfe(opt_chain, draw_nm, surface_name="Z1"):
    """Copy of ``opt_chain`` whose ``Z1`` surface carries one WFE realisation:
    ordering 'standard', normalisation truthy, origin 'x', coefficients
    [0, 0, 0] + draw * 1e-9 (pipeline.py:121-128)."""
    out = {}
    hit = False
    for key, item in opt_chain.items():
        item = copy.copy(item)
        if item.get("name") == surface_name:
            item["Zordering"] = "standard"
            item["Znormalize"] = "True"
            item["Zorigin"] = "x"
            item["Z"] = np.append(np.zeros(3), np.asarray(draw_nm, dtype=np.float64) * 1.0e-9)
            hit = True
        out[key] = item
    if not hit:
        raise KeyError(f"no surface named {surface_name!r} in the chain (is it 'Ignore = True'?)")
    return out


def parse_config_variant(path, wavelengths_um=None, unignore=()):
    """``parse_config`` on a lens file with (a) the ``[wavelengths]`` section replaced by
    ``wavelengths_um`` (wavelength sweeps beyond the handful a shipped file lists) and (b)
    ``Ignore`` cleared on the surfaces whose ``Comment`` is in ``unignore`` (the shipped
    Ariel_FGS-FGS1.ini keeps the Monte-Carlo surface ``Z1`` ignored).  The file
    on disk is not touched: a temporary copy is parsed."""
    from .parse_config import parse_config

    cfg = configparser.ConfigParser()
    cfg.read(os.path.expanduser(path))
    if wavelengths_um is not None:
        cfg.remove_section("wavelengths")
        cfg.add_section("wavelengths")
        for k, wl in enumerate(wavelengths_um, start=1):
            cfg["wavelengths"][f"w{k}"] = repr(float(wl))
    for name in cfg.sections():
        if name.startswith("lens_") and cfg[name].get("Comment", "") in unignore:
            cfg[name]["Ignore"] = "False"
    with tempfile.NamedTemporaryFile("w", suffix=".ini", delete=False) as fh:
        cfg.write(fh)
        tmp = fh.name
    try:
        return parse_config(tmp)
    finally:
        os.unlink(tmp)
