"""A SECOND, independent restatement of the two aperture-mask rules.  TEST INFRASTRUCTURE ONLY.

photutils (the dependency that produces the reference's mask values, paos/classes/wfo.py:246-268) is
neither under /root/reference nor installed, so mask values cannot be pinned to it ("parity unpinned",
oracle/aperture_np.py).  What CAN be done is to compute the same two published rules a second time by
different mathematics and different code, and demand agreement where a slip in either would show:

* ``ellipse_mask_alt`` -- area(pixel INTERSECT ellipse) / area(pixel) WITHOUT the edge-wise Green's
  theorem of aperture_np.py / the HIP kernel: for an untilted ellipse the overlap is integrated in
  closed form strip by strip (``integral of clip(h(x), y0, y1) - clip(-h(x), y0, y1) dx`` with the
  antiderivative of the half-chord ``h``); for a tilted one the pixel is mapped to the unit-circle
  frame and the chord-length integrand is integrated by Gauss-Legendre quadrature in the angle variable
  between its breakpoints (vertex abscissae, edge/circle crossings).  No bounding box is used at all:
  every pixel of the image is evaluated, so agreement also checks aperture_np's box arithmetic.
* ``rectangle_mask_alt`` -- the literal 32 x 32 sub-sample loop, pixel by pixel in plain Python floats,
  with the sample coordinate accumulated by repeated ``+= 1/subpixels`` (no separable shortcut, no
  vectorisation, no bounding box).
"""
import math

import numpy as np


def _chord_primitive(x, a, b):
    """Antiderivative of h(x) = b sqrt(1 - x^2/a^2) on [-a, a]."""
    t = min(1.0, max(-1.0, x / a))
    return 0.5 * b * (x * math.sqrt(max(0.0, 1.0 - t * t)) + a * math.asin(t))


def _strip_overlap(x0, x1, y0, y1, a, b):
    """Area of [x0,x1] x [y0,y1] inside the untilted ellipse x^2/a^2 + y^2/b^2 <= 1 (closed form)."""
    lo, hi = max(x0, -a), min(x1, a)
    if lo >= hi:
        return 0.0

    def inv(y):  # |x| at which the half-chord equals |y|
        t = abs(y) / b
        return a * math.sqrt(max(0.0, 1.0 - t * t)) if t < 1.0 else 0.0

    cuts = {lo, hi}
    for y in (y0, y1):
        if abs(y) < b:
            for s in (-1.0, 1.0):
                c = s * inv(y)
                if lo < c < hi:
                    cuts.add(c)
    pts = sorted(cuts)
    area = 0.0
    for p, q in zip(pts[:-1], pts[1:]):
        mid = p + 0.381966011250105 * (q - p)  # off-centre: a chord tangent to a pixel edge touches at the centre
        h = b * math.sqrt(max(0.0, 1.0 - (mid / a) ** 2))
        top_is_chord = h < y1       # upper bound of the overlap: min(y1, h)
        bot_is_chord = -h > y0      # lower bound: max(y0, -h)
        upper = min(y1, h)
        lower = max(y0, -h)
        if upper <= lower:
            continue
        piece = 0.0
        piece += (_chord_primitive(q, a, b) - _chord_primitive(p, a, b)) if top_is_chord else y1 * (q - p)
        piece -= -(_chord_primitive(q, a, b) - _chord_primitive(p, a, b)) if bot_is_chord else y0 * (q - p)
        area += piece
    return area


_GL_X, _GL_W = np.polynomial.legendre.leggauss(48)


def _quad_overlap(poly):
    """Area of a convex polygon (list of (u, v), unit-circle frame) inside the unit disk, by quadrature
    of the vertical chord overlap in u = sin(phi)."""
    us = [p[0] for p in poly]
    lo, hi = max(min(us), -1.0), min(max(us), 1.0)
    if lo >= hi:
        return 0.0
    n = len(poly)
    edges = [(poly[i], poly[(i + 1) % n]) for i in range(n)]

    def span(u):  # [vmin, vmax] of the polygon on the vertical line at u
        vs = []
        for (u0, v0), (u1, v1) in edges:
            if u0 == u1:
                if u == u0:
                    vs += [v0, v1]
                continue
            t = (u - u0) / (u1 - u0)
            if -1e-15 <= t <= 1.0 + 1e-15:
                vs.append(v0 + t * (v1 - v0))
        return (min(vs), max(vs)) if vs else (0.0, 0.0)

    cuts = {lo, hi}
    for u in us:
        if lo < u < hi:
            cuts.add(u)
    for (u0, v0), (u1, v1) in edges:  # edge / circle crossings
        du, dv = u1 - u0, v1 - v0
        aa, bb, cc = du * du + dv * dv, 2.0 * (u0 * du + v0 * dv), u0 * u0 + v0 * v0 - 1.0
        disc = bb * bb - 4.0 * aa * cc
        if aa > 0.0 and disc > 0.0:
            for s in (-1.0, 1.0):
                t = (-bb + s * math.sqrt(disc)) / (2.0 * aa)
                if 0.0 < t < 1.0:
                    u = u0 + t * du
                    if lo < u < hi:
                        cuts.add(u)
    pts = sorted(cuts)
    area = 0.0
    for p, q in zip(pts[:-1], pts[1:]):
        fp, fq = math.asin(p), math.asin(q)
        phi = 0.5 * (fq - fp) * _GL_X + 0.5 * (fq + fp)
        acc = 0.0
        for ph, w in zip(phi, _GL_W):
            u, c = math.sin(ph), math.cos(ph)
            vmin, vmax = span(u)
            acc += w * max(0.0, min(vmax, c) - max(vmin, -c)) * c
        area += 0.5 * (fq - fp) * acc
    return area


def ellipse_mask_alt(shape, xc, yc, a, b, theta=0.0):
    ny, nx = int(shape[0]), int(shape[1])
    out = np.zeros((ny, nx), dtype=np.float64)
    ct, st = math.cos(theta), math.sin(theta)
    reach = max(a, b) + 1.5
    for ky in range(max(0, int(math.floor(yc - reach))), min(ny, int(math.ceil(yc + reach)) + 1)):
        for kx in range(max(0, int(math.floor(xc - reach))), min(nx, int(math.ceil(xc + reach)) + 1)):
            x0, x1 = kx - 0.5 - xc, kx + 0.5 - xc
            y0, y1 = ky - 0.5 - yc, ky + 0.5 - yc
            if theta == 0.0:
                val = _strip_overlap(x0, x1, y0, y1, a, b)
            else:
                corners = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
                poly = [((x * ct + y * st) / a, (y * ct - x * st) / b) for x, y in corners]
                val = _quad_overlap(poly) * a * b
            out[ky, kx] = min(1.0, max(0.0, val))
    return out


def rectangle_mask_alt(shape, xc, yc, w, h, theta=0.0, subpixels=32):
    ny, nx = int(shape[0]), int(shape[1])
    out = np.zeros((ny, nx), dtype=np.float64)
    ct, st = math.cos(theta), math.sin(theta)
    hw, hh = w / 2.0, h / 2.0
    reach = math.hypot(hw, hh) + 1.5
    step = 1.0 / subpixels
    for ky in range(max(0, int(math.floor(yc - reach))), min(ny, int(math.ceil(yc + reach)) + 1)):
        for kx in range(max(0, int(math.floor(xc - reach))), min(nx, int(math.ceil(xc + reach)) + 1)):
            count = 0
            x = (kx - 0.5 - xc) - 0.5 * step
            for _ in range(subpixels):
                x += step
                y = (ky - 0.5 - yc) - 0.5 * step
                for _ in range(subpixels):
                    y += step
                    if abs(y * st + x * ct) < hw and abs(y * ct - x * st) < hh:
                        count += 1
            out[ky, kx] = count / float(subpixels * subpixels)
    return out
