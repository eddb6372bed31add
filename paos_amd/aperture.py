"""Aperture handles returned by ``WFO.aperture`` / ``run``.

The reference returns photutils ``EllipticalAperture`` / ``RectangularAperture``
objects (paos/classes/wfo.py:246,264,278) and its callers read ``.positions``,
``.a``/``.b``/``.theta`` or ``.w``/``.h``/``.theta`` and call
``.to_mask(method=...).to_image(shape)`` (paos/core/run.py:136-141,
paos/core/plot.py:164-184, paos/core/saveOutput.py:148-149).  These classes offer
exactly that surface; the mask itself is rendered by the GPU kernel that also
applies it (csrc/pointwise.h: aperture_kernel), so there is one implementation of
the mask arithmetic in the product.
"""
import numpy as np

from . import _lib


class _MaskImage:
    def __init__(self, shape_code, block):
        self._shape_code = shape_code
        self._block = block

    def to_image(self, shape):
        ny, nx = int(shape[0]), int(shape[1])
        if ny != nx or nx < 64 or nx > 4096 or nx & (nx - 1):
            raise NotImplementedError("mask rendering supports square 2**k grids, 64..4096")
        dev = _lib.DeviceFields(nx, 1)
        try:
            return dev.aperture_mask(self._shape_code, self._block)
        finally:
            dev.close()


class _Centred:
    """``positions`` (the centre, pixels) as the ndarray photutils' apertures carry, made when somebody reads it: a batch
    of 256 wavefronts builds ~2000 handles per walk and reads the attribute of a handful (round 5)."""

    __slots__ = ()

    @property
    def positions(self):
        return np.array(self._xy, dtype=np.float64)

    @positions.setter
    def positions(self, value):
        x, y = np.asarray(value, dtype=np.float64)
        self._xy = (float(x), float(y))


class EllipticalAperture(_Centred):
    __slots__ = ("_xy", "a", "b", "theta")

    def __init__(self, positions, a, b, theta=0.0):
        x, y = positions
        self._xy = (float(x), float(y))
        self.a, self.b, self.theta = float(a), float(b), float(theta)

    def block(self, obscuration=False, enable=True):
        xc, yc = self._xy
        return [1.0 if enable else 0.0, xc, yc, self.a, self.b, self.theta,
                1.0 if obscuration else 0.0, 0.0]

    def to_mask(self, method="exact", subpixels=5):
        if method != "exact":
            raise NotImplementedError("only method='exact' is used by PAOS for ellipses")
        return _MaskImage(_lib.SHAPE_ELLIPSE, self.block())

    def __repr__(self):
        return f"<EllipticalAperture({list(self.positions)}, a={self.a}, b={self.b}, theta={self.theta})>"


class RectangularAperture(_Centred):
    __slots__ = ("_xy", "w", "h", "theta")

    def __init__(self, positions, w, h, theta=0.0):
        x, y = positions
        self._xy = (float(x), float(y))
        self.w, self.h, self.theta = float(w), float(h), float(theta)

    def block(self, obscuration=False, enable=True, subpixels=32):
        xc, yc = self._xy
        return [1.0 if enable else 0.0, xc, yc, self.w, self.h, self.theta,
                1.0 if obscuration else 0.0, float(subpixels)]

    def to_mask(self, method="subpixel", subpixels=32):
        if method == "exact":  # photutils serves "exact" rectangles with the 32 x 32 sub-pixel rule
            method, subpixels = "subpixel", 32
        if method != "subpixel":
            raise NotImplementedError("only method='subpixel' (or 'exact' = subpixel 32) is used by PAOS for rectangles")
        return _MaskImage(_lib.SHAPE_RECT, self.block(subpixels=subpixels))

    def __repr__(self):
        return f"<RectangularAperture({list(self.positions)}, w={self.w}, h={self.h}, theta={self.theta})>"


def make_aperture(n, dx, dy, xc, yc, hx=None, hy=None, r=None, shape="elliptical", tilt=None):
    """Pixel-unit aperture object for a physical aperture -- wfo.py:236-271."""
    ixc = xc / dx + n / 2
    iyc = yc / dy + n / 2
    if shape == "elliptical":
        if hx is None or hy is None:
            raise AssertionError("Semi major/minor axes not defined")
        theta = 0.0 if tilt is None else np.deg2rad(tilt)
        return EllipticalAperture((ixc, iyc), hx / dx, hy / dy, theta=theta)
    if shape == "circular":
        if r is None:
            raise AssertionError("Radius not defined")
        return EllipticalAperture((ixc, iyc), r / dx, r / dy, theta=0.0)
    if shape == "rectangular":
        if hx is None or hy is None:
            raise AssertionError("Semi major/minor axes not defined")
        theta = 0.0 if tilt is None else np.deg2rad(tilt)
        return RectangularAperture((ixc, iyc), hx / dx, hy / dy, theta=theta)
    raise ValueError(f"Aperture {shape:s} not defined yet.")


def bbox_misses_grid(ap, n):
    """photutils' ``to_image`` returns None when the mask's bounding box does not
    overlap the image; the reference then fails on ``u *= None``."""
    import math

    ct, st = math.cos(ap.theta), math.sin(ap.theta)
    if isinstance(ap, EllipticalAperture):
        xe = math.sqrt((ap.a * ct) ** 2 + (ap.b * st) ** 2)
        ye = math.sqrt((ap.a * st) ** 2 + (ap.b * ct) ** 2)
    else:
        hw, hh = ap.w / 2.0, ap.h / 2.0
        xe = max(abs(hw * ct - hh * st), abs(hw * ct + hh * st))
        ye = max(abs(hw * st + hh * ct), abs(hw * st - hh * ct))
    xc, yc = ap._xy
    if not (math.isfinite(xc) and math.isfinite(yc) and math.isfinite(xe) and math.isfinite(ye)):
        return True
    x0, x1 = math.floor(xc - xe + 0.5), math.ceil(xc + xe + 0.5)
    y0, y1 = math.floor(yc - ye + 0.5), math.ceil(yc + ye + 0.5)
    return max(x0, 0) >= min(x1, n) or max(y0, 0) >= min(y1, n)
