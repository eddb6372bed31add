"""Aperture masks on the CPU (NumPy).  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED at this boundary.  The reference obtains mask values from a
third-party dependency that is NOT under /root/reference and not installed in
this image: photutils (pinned 1.11.0 in the reference's poetry.lock:2360-2361),
called at paos/classes/wfo.py:246-247 (EllipticalAperture, method="exact"),
wfo.py:255-256 (circular = ellipse with a == b) and wfo.py:264-268
(RectangularAperture, method="subpixel", subpixels=32).  No reference test or
fixture holds mask values.  This file restates photutils' *published*
behaviour:

  * pixel k covers [k-0.5, k+0.5]; the aperture centre is given in pixel units;
  * the mask lives on the bounding box  ixmin = floor(xc - ext + 0.5),
    ixmax = ceil(xc + ext + 0.5) (exclusive), ext = sqrt((a cos t)^2+(b sin t)^2)
    for the ellipse, the rotated half-sizes for the rectangle; outside it is 0;
    ``to_image(shape)`` pastes the overlapping part and returns None when there
    is no overlap at all;
  * "exact": value = area(pixel INTERSECT ellipse) / area(pixel), evaluated in the
    frame where the ellipse is the unit circle;
  * "subpixel": each pixel is sampled at subpixels^2 sub-pixel centres, the
    sample coordinate accumulated by repeated ``+= 1/subpixels``; a sample counts
    if |x_rot| < w/2 and |y_rot| < h/2 (strict); value = count / subpixels^2.

and is anchored by analytic properties in tests/test_aperture_oracle.py
(sum(mask) == pi a b to 1e-12, interior == 1.0, exterior == 0.0, symmetry,
rectangle values in {k/1024}).

The ellipse area is computed edge-wise (Green's theorem over the pixel's
image, a parallelogram, clipped by the unit circle): for every directed edge
p -> p+d the part inside the circle contributes 1/2 cross and the parts
outside contribute 1/2 of their signed subtended angle.  All cross products are
taken against the SHORT edge vector d so the result is conditioned like
photutils' own triangle formula (relative error ~ 1e-16 * a).
The HIP kernels (paos_amd/csrc/pointwise.h: ellipse_pixel / rect_pixel / ApertureEval) perform the same operations in the
same order, so the {0, partial, 1} classification is bit-identical by
construction.
"""
import math

import numpy as np

SUBPIX_DEFAULT = 32


def _bbox(xc, yc, x_ext, y_ext):
    ixmin = int(math.floor(xc - x_ext + 0.5))
    ixmax = int(math.ceil(xc + x_ext + 0.5))
    iymin = int(math.floor(yc - y_ext + 0.5))
    iymax = int(math.ceil(yc + y_ext + 0.5))
    return ixmin, ixmax, iymin, iymax


def _clip_bbox(shape, box):
    ny, nx = int(shape[0]), int(shape[1])
    ixmin, ixmax, iymin, iymax = box
    jx0, jx1 = max(ixmin, 0), min(ixmax, nx)
    jy0, jy1 = max(iymin, 0), min(iymax, ny)
    if jx0 >= jx1 or jy0 >= jy1:
        return None
    return jx0, jx1, jy0, jy1


def _edge_term(px, py, dx, dy):
    """Signed area of (triangle O,p,p+d) INTERSECT (unit disk), and whether a
    piece of the edge lies strictly inside the disk."""
    cr = px * dy - py * dx
    pp = px * px + py * py
    pd = px * dx + py * dy
    dd = dx * dx + dy * dy
    cc = pp - 1.0
    disc = pd * pd - dd * cc  # (B/2)^2 - A C with B/2 = pd
    has = disc > 0.0
    sq = np.sqrt(np.where(has, disc, 0.0))
    # stable roots of dd t^2 + 2 pd t + cc = 0
    qq = -(pd + np.where(pd >= 0.0, sq, -sq))
    with np.errstate(divide="ignore", invalid="ignore"):
        ta = qq / dd
        tb = np.where(qq != 0.0, cc / qq, ta)
    t1 = np.minimum(ta, tb)
    t2 = np.maximum(ta, tb)
    t1c = np.minimum(np.maximum(t1, 0.0), 1.0)
    t2c = np.minimum(np.maximum(t2, 0.0), 1.0)
    part = has & (t1c < t2c)
    whole = 0.5 * np.arctan2(cr, pp + pd)
    a_in = np.arctan2(t1c * cr, pp + t1c * pd)
    a_out = np.arctan2((1.0 - t2c) * cr, pp + (1.0 + t2c) * pd + t2c * dd)
    inside = 0.5 * (a_in + (t2c - t1c) * cr + a_out)
    return np.where(part, inside, whole), part


def ellipse_mask(shape, xc, yc, a, b, theta=0.0):
    """Exact pixel/ellipse overlap fractions on an image of ``shape`` (ny, nx).

    Restates photutils EllipticalAperture(...).to_mask("exact").to_image(shape)
    as used at paos/classes/wfo.py:246-247,255-256.  Returns None if the
    bounding box misses the image (photutils' to_image does)."""
    ct, st = math.cos(theta), math.sin(theta)
    x_ext = math.sqrt((a * ct) ** 2 + (b * st) ** 2)
    y_ext = math.sqrt((a * st) ** 2 + (b * ct) ** 2)
    clip = _clip_bbox(shape, _bbox(xc, yc, x_ext, y_ext))
    if clip is None:
        return None
    jx0, jx1, jy0, jy1 = clip
    out = np.zeros((int(shape[0]), int(shape[1])), dtype=np.float64)
    kx = np.arange(jx0, jx1, dtype=np.float64)
    ky = np.arange(jy0, jy1, dtype=np.float64)
    x0 = ((kx - 0.5) - xc)[None, :]
    x1 = ((kx + 0.5) - xc)[None, :]
    y0 = ((ky - 0.5) - yc)[:, None]
    y1 = ((ky + 0.5) - yc)[:, None]

    def to_unit(x, y):
        return (x * ct + y * st) / a, (y * ct - x * st) / b

    # corners, counter-clockwise
    c0x, c0y = to_unit(x0, y0)
    c1x, c1y = to_unit(x1, y0)
    c2x, c2y = to_unit(x1, y1)
    c3x, c3y = to_unit(x0, y1)
    in0 = (c0x * c0x + c0y * c0y) <= 1.0
    in1 = (c1x * c1x + c1y * c1y) <= 1.0
    in2 = (c2x * c2x + c2y * c2y) <= 1.0
    in3 = (c3x * c3x + c3y * c3y) <= 1.0
    all_in = in0 & in1 & in2 & in3
    any_in = in0 | in1 | in2 | in3

    e0, h0 = _edge_term(c0x, c0y, c1x - c0x, c1y - c0y)
    e1, h1 = _edge_term(c1x, c1y, c2x - c1x, c2y - c1y)
    e2, h2 = _edge_term(c2x, c2y, c3x - c2x, c3y - c2y)
    e3, h3 = _edge_term(c3x, c3y, c0x - c3x, c0y - c3y)
    touched = any_in | h0 | h1 | h2 | h3
    area = ((e0 + e1) + (e2 + e3)) * (a * b)
    area = np.minimum(np.maximum(area, 0.0), 1.0)
    # pixel that swallows the whole ellipse (a, b << 1): full disk area
    holds_centre = (x0 <= 0.0) & (x1 >= 0.0) & (y0 <= 0.0) & (y1 >= 0.0)
    untouched = np.where(holds_centre, min(math.pi * a * b, 1.0), 0.0)
    vals = np.where(all_in, 1.0, np.where(touched, area, untouched))
    out[jy0:jy1, jx0:jx1] = vals
    return out


def _subpixel_counts_1d(k, centre, half, subpixels):
    """#sub-samples of pixel k (1-D) with |x| < half, sample positions built by
    repeated addition exactly as the reference's dependency does."""
    step = 1.0 / subpixels
    x = ((k - 0.5) - centre) - 0.5 * step
    cnt = np.zeros(k.shape, dtype=np.int64)
    for _ in range(subpixels):
        x = x + step
        cnt += np.abs(x) < half
    return cnt


def rectangle_mask(shape, xc, yc, w, h, theta=0.0, subpixels=SUBPIX_DEFAULT):
    """Sub-pixel sampled rectangle mask (values k / subpixels^2).

    Restates photutils RectangularAperture(...).to_mask("subpixel",
    subpixels=32).to_image(shape) as used at paos/classes/wfo.py:264-268;
    ``w`` and ``h`` are FULL sizes in pixels (wfo.py:224,261-264)."""
    ct, st = math.cos(theta), math.sin(theta)
    hw, hh = w / 2.0, h / 2.0
    x_ext = max(abs(hw * ct - hh * st), abs(hw * ct + hh * st))
    y_ext = max(abs(hw * st + hh * ct), abs(hw * st - hh * ct))
    clip = _clip_bbox(shape, _bbox(xc, yc, x_ext, y_ext))
    if clip is None:
        return None
    jx0, jx1, jy0, jy1 = clip
    out = np.zeros((int(shape[0]), int(shape[1])), dtype=np.float64)
    kx = np.arange(jx0, jx1, dtype=np.float64)
    ky = np.arange(jy0, jy1, dtype=np.float64)
    if theta == 0.0:
        # x_tr = y*0 + x*1 and y_tr = y*1 - x*0 are exact, so the 2-D count
        # factorises into a product of two 1-D counts.
        cx = _subpixel_counts_1d(kx, xc, hw, subpixels)
        cy = _subpixel_counts_1d(ky, yc, hh, subpixels)
        cnt = cy[:, None] * cx[None, :]
    else:
        step = 1.0 / subpixels
        cnt = np.zeros((ky.size, kx.size), dtype=np.int64)
        x = (((kx - 0.5) - xc) - 0.5 * step)[None, :]
        for _ in range(subpixels):
            x = x + step
            y = (((ky - 0.5) - yc) - 0.5 * step)[:, None]
            for _ in range(subpixels):
                y = y + step
                x_tr = y * st + x * ct
                y_tr = y * ct - x * st
                cnt += (np.abs(x_tr) < hw) & (np.abs(y_tr) < hh)
    out[jy0:jy1, jx0:jx1] = cnt / float(subpixels * subpixels)
    return out


class _MaskImage:
    def __init__(self, fn):
        self._fn = fn

    def to_image(self, shape):
        return self._fn(shape)


class EllipticalAperture:
    """Duck-type of the photutils object returned by WFO.aperture
    (wfo.py:246,278): .positions, .a, .b, .theta and .to_mask(...).to_image()."""

    def __init__(self, positions, a, b, theta=0.0):
        self.positions = np.asarray(positions, dtype=np.float64)
        self.a, self.b, self.theta = float(a), float(b), float(theta)

    def to_mask(self, method="exact", subpixels=5):
        if method != "exact":
            raise NotImplementedError("oracle restates method='exact' only")
        xc, yc = self.positions
        return _MaskImage(
            lambda shape: ellipse_mask(shape, xc, yc, self.a, self.b, self.theta)
        )


class RectangularAperture:
    """Duck-type of photutils RectangularAperture (wfo.py:264): .positions,
    .w, .h, .theta and .to_mask("subpixel", subpixels=32).to_image()."""

    def __init__(self, positions, w, h, theta=0.0):
        self.positions = np.asarray(positions, dtype=np.float64)
        self.w, self.h, self.theta = float(w), float(h), float(theta)

    def to_mask(self, method="subpixel", subpixels=SUBPIX_DEFAULT):
        if method == "exact":
            # photutils has no exact rectangle overlap: Aperture._translate_mask_mode turns
            # ("exact", rectangle) into ("subpixel", subpixels=32) -- the request run.py:137 makes for
            # a Zorthonorm surface with a rectangular aperture (restated from memory, unpinned)
            method, subpixels = "subpixel", 32
        if method != "subpixel":
            raise NotImplementedError("oracle restates method='subpixel' (and 'exact' -> subpixel 32) only")
        xc, yc = self.positions
        return _MaskImage(
            lambda shape: rectangle_mask(
                shape, xc, yc, self.w, self.h, self.theta, subpixels
            )
        )
