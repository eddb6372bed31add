"""Import the PAOS reference (read-only, /root/reference) in THIS container only.

Used exclusively by tests/golden_tools/make_golden.py to generate golden fixtures; never
shipped to the GPU box (the reference does not travel).  Recipe from SURVEY.md
section 8c: a fake ``paos`` package whose __path__ points at the reference tree
(bypassing paos/__init__.py, which needs loguru + package metadata), a no-op
``logger``, and empty stub modules for third-party packages that are absent
here (astropy.units, photutils.aperture, skimage.transform).  The reference
source files themselves are imported UNMODIFIED.
"""
import sys
import types

REF_ROOT = "/root/reference"


class _NoLog:
    def __getattr__(self, name):
        return lambda *a, **k: None


def install(ellipse_cls=None, rect_cls=None):
    """Install stubs and return the fake ``paos`` package module."""
    if "paos" in sys.modules and getattr(sys.modules["paos"], "_is_ref_stub", False):
        pkg = sys.modules["paos"]
    else:
        pkg = types.ModuleType("paos")
        pkg.__path__ = [REF_ROOT + "/paos"]
        pkg.logger = _NoLog()
        pkg._is_ref_stub = True
        sys.modules["paos"] = pkg

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Unit:
        def __init__(self, *a, **k):
            pass

    if "astropy" not in sys.modules:
        ap = stub("astropy")
        ap.units = stub("astropy.units", m=_Unit(), Unit=_Unit)

    class _Missing:
        def __init__(self, *a, **k):
            raise RuntimeError("photutils is absent: supply aperture classes")

    ph = stub("photutils")
    ph.aperture = stub(
        "photutils.aperture",
        EllipticalAperture=ellipse_cls or _Missing,
        RectangularAperture=rect_cls or _Missing,
    )
    if "skimage" not in sys.modules:
        sk = stub("skimage")
        sk.transform = stub("skimage.transform", rescale=None, resize=None)
    return pkg
