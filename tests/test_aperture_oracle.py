"""Analytic anchors for the aperture-mask restatement (parity UNPINNED against photutils,
which is absent: see oracle/aperture_np.py).  These are the properties SURVEY.md 8c lists."""
import math

import numpy as np
import pytest

from oracle.aperture_np import EllipticalAperture, RectangularAperture, ellipse_mask, rectangle_mask


@pytest.mark.parametrize("xc,yc,a,b,th", [(128, 128, 32, 32, 0), (128.3, 127.1, 40.7, 25.2, 0),
                                          (120.3, 131.1, 40.7, 25.2, 0.6), (64.5, 64.5, 3.3, 2.1, 1.1),
                                          (100.2, 99.9, 0.3, 0.2, 0.3)])
def test_ellipse_area_and_range(xc, yc, a, b, th):
    m = ellipse_mask((256, 256), xc, yc, a, b, th)
    assert abs(m.sum() - math.pi * a * b) < 1e-12 * math.pi * a * b
    assert m.min() == 0.0 and m.max() <= 1.0
    if a > 2 and b > 2:
        inner = ellipse_mask((256, 256), xc, yc, a - 1.5, b - 1.5, th) > 0
        assert np.all(m[inner] == 1.0)  # interior pixels are exactly one
        outer = ellipse_mask((256, 256), xc, yc, a + 1.5, b + 1.5, th) == 0
        assert np.all(m[outer] == 0.0)  # exterior pixels exactly zero


def test_ellipse_symmetry_and_supersampling():
    m = ellipse_mask((128, 128), 64.0, 64.0, 20.0, 13.0, 0.0)  # centre on a pixel centre
    assert np.array_equal(m[1:, 1:], m[1:, 1:][::-1, ::-1])
    assert np.max(np.abs(m[1:, 1:] - m[1:, 1:][::-1, :])) < 1e-14  # mirror: same area, other edge order
    xc, yc, a, b, th = 31.7, 32.4, 10.3, 6.1, 0.4
    m = ellipse_mask((64, 64), xc, yc, a, b, th)
    sub = (np.arange(64) + 0.5) / 64 - 0.5
    ct, st = math.cos(th), math.sin(th)
    for iy, ix in zip(*np.nonzero((m > 0) & (m < 1))):
        X = (ix + sub)[None, :] - xc
        Y = (iy + sub)[:, None] - yc
        frac = ((((X * ct + Y * st) / a) ** 2 + ((Y * ct - X * st) / b) ** 2) <= 1).mean()
        assert abs(frac - m[iy, ix]) < 6e-3  # 64x64 point sampling: error ~ perimeter / 64


def test_rectangle_subpixel_rule():
    r = rectangle_mask((64, 64), 32.0, 32.0, 10.5, 6.25, 0.0)
    assert np.all((r * 1024) % 1 == 0)  # values are k/1024
    assert r.sum() == 10.5 * 6.25  # edges on sub-pixel boundaries: exact area
    full = rectangle_mask((64, 64), 32.1, 31.7, 10.5, 6.25, 1e-300)  # forces the 2-D sampling loop
    sep = rectangle_mask((64, 64), 32.1, 31.7, 10.5, 6.25, 0.0)
    assert np.array_equal(full, sep)  # the separable count equals the 32x32 loop at theta = 0
    rot = rectangle_mask((64, 64), 32.0, 32.0, 20.0, 8.0, math.pi / 2)
    assert abs(rot.sum() - 160.0) < 1.0 and rot[32, 32] == 1.0 and rot[32, 45] == 0.0


def test_bounding_box_and_objects():
    assert ellipse_mask((64, 64), 200.0, 10.0, 5.0, 5.0) is None  # photutils' to_image -> None
    e = EllipticalAperture((10.0, 12.0), 4.0, 3.0, theta=0.1)
    assert e.to_mask(method="exact").to_image((32, 32)).shape == (32, 32)
    assert (e.a, e.b, e.theta) == (4.0, 3.0, 0.1) and list(e.positions) == [10.0, 12.0]
    q = RectangularAperture((10.0, 12.0), 4.0, 3.0)
    assert q.to_mask(method="subpixel", subpixels=32).to_image((32, 32)).sum() == 12.0
    cut = ellipse_mask((32, 32), 2.0, 30.0, 6.0, 6.0)  # partly outside the grid
    assert 0 < cut.sum() < math.pi * 36
