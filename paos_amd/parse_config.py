"""Lens-file (.ini) reader for the host planner.

Drop-in for ``paos.core.parseConfig.parse_config`` (reference
paos/core/parseConfig.py:35-401): returns
``(pup_diameter, parameters, wavelengths, fields, opt_chain_list)`` with one
``opt_chain`` dict per wavelength whose items carry the keys ``run`` consumes
(reference paos/core/run.py:77-223).  SURVEY.md 8f-1 / 9.7 list the quirks kept
on purpose:

* INIT pupil diameter is read from comma fields 2 and 3 of the aperture string
  (parseConfig.py:166-171);
* the wavelength / field lists stop at the first missing or zero entry
  (:114-121, :127-136);
* ``Ignore = True`` surfaces are skipped but keep their file index as ``num``
  (:147-153);
* blank / "infinity" radii become curvature 0 (:362-363), MIRROR flips the sign
  of n, known glasses use the index at ambient temperature (:368-373);
* an ``ABCD`` surface right-multiplies the user matrix onto a thickness matrix
  (:349-354).

``Grid Sag`` and ``PSD`` surfaces are parsed like the reference does (:218-284); the unit of a PSD
surface is kept as its name instead of an astropy object.
"""
import configparser
import os

import numpy as np

from .abcd import ABCD
from .material import Material

GRID_SIZES = (64, 128, 256, 512, 1024, 2048, 4096)
ZOOMS = (1, 2, 4, 8, 16)


def _num(text):
    try:
        return np.float64(text)
    except (TypeError, ValueError):
        return np.nan


def _aperture_dict(text):
    parts = text.split(",")
    shape, kind = parts[0].split()
    return {
        "shape": shape,
        "type": kind,
        "xrad": _num(parts[1]),
        "yrad": _num(parts[2]),
        "xc": _num(parts[3]),
        "yc": _num(parts[4]),
    }


def _numbered(section, prefix, getter):
    k = 1
    while True:
        val = getter(section, f"{prefix}{k:d}")
        if not val:
            return
        yield val
        k += 1


def _plain_surface(thickness, curvature, n1, n2):
    return (
        ABCD(thickness=thickness, curvature=curvature, n1=n1, n2=n2, M=1.0),
        ABCD(thickness=thickness, curvature=curvature, n1=n1, n2=n2, M=1.0),
    )


def _surface(el, item, n1, glass):
    """Fill the type-specific keys of ``item`` and return the outgoing index."""
    kind = item["type"]
    finite_t = item["T"] if np.isfinite(item["T"]) else 0.0
    aperture = el.get("aperture", "")

    if kind == "Zernike":
        wave = 1.0e-6 * _num(el.get("Par1", ""))
        item["Zordering"] = el.get("Par2", "").lower()
        item["Znormalize"] = el.getboolean("Par3")
        item["Zradius"] = _num(el.get("Par4", ""))
        item["Zorigin"] = el.get("Par5", "x")
        item["Zorthonorm"] = el.get("Par6", "False").lower() == "true"
        item["Zindex"] = np.array(
            [int(float(t)) for t in el.get("Zindex", "").split(",") if t.strip()], dtype=np.int64
        )
        item["Z"] = (
            np.array([float(t) for t in el.get("Z", "").split(",") if t.strip()], dtype=np.float64)
            * wave
        )
        if aperture:
            item["aperture"] = _aperture_dict(aperture)
        item["ABCDt"], item["ABCDs"] = _plain_surface(0.0, 0.0, n1, n1)
        return n1

    if kind == "Grid Sag":  # parseConfig.py:218-258
        wave = 1.0e-6 * _num(el.get("Par1", ""))
        for key, par in (("nx", "Par2"), ("ny", "Par3"), ("delx", "Par4"), ("dely", "Par5"), ("xdec", "Par6"),
                         ("ydec", "Par7")):
            item[key] = _num(el.get(par, ""))
        path = el.get("Par8", "")
        if not os.path.exists(path):
            raise ValueError(f"Grid sag file does not exist: {path}")
        with open(path, "rb") as fh:
            grid_sag = np.load(fh, allow_pickle=True).item()
        assert "data" in grid_sag.keys(), "The .npy file must contain a dictionary with a 'data' key"
        for key in ("nx", "ny", "delx", "dely", "xdec", "ydec"):
            if key in grid_sag.keys():
                item[key] = grid_sag[key]
        item["grid_sag"] = grid_sag["data"] * wave
        item["ABCDt"], item["ABCDs"] = _plain_surface(0.0, 0.0, n1, n1)
        return n1

    if kind == "PSD":  # parseConfig.py:274-284; ``units`` stays the unit NAME (astropy is not needed:
        # psd.py:160 only asks it for the factor to metres, paos_amd/phase_maps.py)
        for key, par in (("A", "Par1"), ("B", "Par2"), ("C", "Par3"), ("fknee", "Par4"), ("fmin", "Par5"),
                         ("fmax", "Par6"), ("SR", "Par7")):
            item[key] = _num(el.get(par, ""))
        item["units"] = el.get("Par8", "")
        item["ABCDt"], item["ABCDs"] = _plain_surface(0.0, 0.0, n1, n1)
        return n1

    if kind == "Coordinate Break":
        for key, par in (("xdec", "Par1"), ("ydec", "Par2"), ("xrot", "Par3"), ("yrot", "Par4")):
            item[key] = _num(el.get(par, ""))
        item["ABCDt"], item["ABCDs"] = _plain_surface(finite_t, 0.0, n1, n1)
        return n1

    if kind == "Paraxial Lens":
        focal = _num(el.get("Par1", ""))
        curvature = 1 / focal if np.isfinite(focal) else 0.0
        if aperture:
            item["aperture"] = _aperture_dict(aperture)
        item["ABCDt"], item["ABCDs"] = _plain_surface(finite_t, curvature, n1, n1)
        return n1

    if kind == "ABCD":
        vals = [_num(el.get(f"Par{k}", "")) for k in range(1, 9)]
        sag, tan = _plain_surface(finite_t, 0.0, n1, n1)[0], _plain_surface(finite_t, 0.0, n1, n1)[1]
        sag.ABCD = sag() @ np.array([[vals[0], vals[1]], [vals[2], vals[3]]])
        tan.ABCD = tan() @ np.array([[vals[4], vals[5]], [vals[6], vals[7]]])
        if aperture:
            item["aperture"] = _aperture_dict(aperture)
        item["ABCDt"], item["ABCDs"] = tan, sag
        return n1

    if kind == "Standard":
        curvature = 1 / item["R"] if np.isfinite(item["R"]) else 0.0
        if aperture:
            item["aperture"] = _aperture_dict(aperture)
        if item["material"] == "MIRROR":
            n2 = -n1
        elif item["material"] in glass.materials.keys():
            n2 = glass.nmat(item["material"])[1] * np.sign(n1)
        else:
            n2 = 1.0 * np.sign(n1)
        item["ABCDt"], item["ABCDs"] = _plain_surface(finite_t, curvature, n1, n2)
        return n2

    raise ValueError(f"Surface Type not recognised: {str(kind):s}")


def parse_config(filename):
    filename = os.path.expanduser(filename)
    if not os.path.isfile(filename):
        # the reference logs and calls sys.exit() (parseConfig.py:67-71)
        raise SystemExit(f"Input file {filename} does not exist or is not a file. Quitting...")
    cfg = configparser.ConfigParser()
    cfg.read(filename)

    general = cfg["general"]
    parameters = {"project": general["project"], "version": general["version"]}
    grid = general.getint("grid_size")
    if grid not in GRID_SIZES:
        raise ValueError(f"Grid size not allowed. Allowed values are {list(GRID_SIZES)}")
    parameters["grid_size"] = grid
    zoom = general.getint("zoom")
    if zoom not in ZOOMS:
        raise ValueError(f"Zoom value not allowed. Allowed values are {list(ZOOMS)}")
    parameters["zoom"] = zoom
    if general.get("lens_unit", "") != "m":
        raise ValueError("Verify lens_unit=m in ini file")
    t_amb = general.getfloat("Tambient")
    p_amb = general.getfloat("Pambient")
    parameters["Tambient"] = t_amb
    parameters["Pambient"] = p_amb

    wavelengths = list(_numbered(cfg["wavelengths"], "w", lambda s, k: s.getfloat(k)))
    fields = []
    for text in _numbered(cfg["fields"], "f", lambda s, k: s.get(k)):
        slopes = np.tan(np.deg2rad(np.array([float(t) for t in text.split(",")])))
        fields.append({"us": slopes[0], "ut": slopes[1]})

    chains = []
    pup_diameter = None
    for wl in wavelengths:
        n1 = None
        glass = Material(wl, Tambient=t_amb, Pambient=p_amb)
        chain = {}
        k = 1
        while f"lens_{k:02d}" in cfg:
            el = cfg[f"lens_{k:02d}"]
            item = {"num": k}
            k += 1
            if el.getboolean("Ignore"):
                continue
            item["type"] = el.get("SurfaceType", None)
            item["R"] = _num(el.get("Radius", ""))
            item["T"] = _num(el.get("Thickness", ""))
            item["material"] = el.get("Material", None)
            item["is_stop"] = el.getboolean("Stop", False)
            item["save"] = el.getboolean("Save", False)
            item["name"] = el.get("Comment", "")

            if item["type"] == "INIT":
                n1 = 1.0
                parts = el.get("aperture", "").split(",")
                shape, kind = parts[0].split()
                if shape == "elliptical" and kind == "aperture":
                    pup_diameter = 2.0 * max(_num(parts[2]), _num(parts[3]))
                continue
            if n1 is None or pup_diameter is None:
                raise ValueError("INIT is not the first surface in Lens Data.")
            n1 = _surface(el, item, n1, glass)
            chain[item["num"]] = item
        chains.append(chain)
    return pup_diameter, parameters, wavelengths, fields, chains
