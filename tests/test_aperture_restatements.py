"""Two independent restatements of the aperture-mask rules must agree (photutils itself is absent, so mask
values stay "parity unpinned"; this is the strongest check the environment allows).  oracle/aperture_np.py
(edge-wise Green's theorem on the bounding box; the algorithm the HIP kernel mirrors) against
oracle/aperture_alt.py (closed-form strip integrals / quadrature, literal Python sub-sample loops, no
bounding box) on the cases most likely to expose a slip: centres on exact half-integers, apertures that
touch or straddle the grid edge, sub-pixel apertures, tilted apertures, rectangle edges that fall exactly
on sub-sample positions (strict '<')."""
import math

import numpy as np
import pytest

from oracle.aperture_alt import ellipse_mask_alt, rectangle_mask_alt
from oracle.aperture_np import ellipse_mask, rectangle_mask

ELLIPSES = [
    # (shape, xc, yc, a, b, theta)
    (40, 20.0, 20.0, 9.0, 6.0, 0.0),        # centre on a pixel centre
    (40, 19.5, 20.5, 9.0, 6.0, 0.0),        # centre on pixel corners (exact half-integers)
    (40, 19.5, 19.5, 7.5, 7.5, 0.0),        # circle, extent lands exactly on pixel boundaries
    (40, 20.3, 19.1, 11.7, 4.2, 0.0),       # generic
    (40, 2.0, 37.5, 6.0, 5.0, 0.0),         # straddles two grid edges
    (40, -0.5, 20.0, 4.0, 8.0, 0.0),        # centre ON the grid edge
    (40, 39.5, 39.5, 3.0, 3.0, 0.0),        # touches the far corner
    (32, 16.2, 15.7, 0.4, 0.3, 0.0),        # sub-pixel ellipse inside one pixel
    (32, 16.5, 16.0, 0.45, 0.2, 0.0),       # sub-pixel ellipse across a pixel boundary
    (32, 16.0, 16.0, 0.9, 0.6, 0.0),        # a, b < 1 px but wider than one pixel
    (32, 15.0, 17.0, 1.0, 1.0, 0.0),        # unit circle on a pixel centre
    (40, 20.3, 19.1, 11.7, 4.2, 0.6),       # tilted
    (40, 19.5, 20.5, 8.0, 3.0, math.pi / 2),  # quarter turn on a half-integer centre
    (32, 16.2, 15.7, 0.4, 0.3, 1.1),        # tilted sub-pixel
]


@pytest.mark.parametrize("n,xc,yc,a,b,th", ELLIPSES)
def test_ellipse_restatements_agree(n, xc, yc, a, b, th):
    m1 = ellipse_mask((n, n), xc, yc, a, b, th)
    m2 = ellipse_mask_alt((n, n), xc, yc, a, b, th)
    assert m1 is not None
    tol = 2e-14 if th == 0.0 else 5e-13  # closed form vs 48-point quadrature
    assert np.max(np.abs(m1 - m2)) < tol, np.max(np.abs(m1 - m2))
    # the {0, partial, 1} classification ("bit-exact aperture index masks") agrees wherever the second
    # method is not within rounding of 0 or 1
    cls1 = np.where(m1 == 0.0, 0, np.where(m1 == 1.0, 2, 1))
    sure = (np.abs(m2) > 1e-12) & (np.abs(m2 - 1.0) > 1e-12)
    assert np.all(cls1[sure] == 1)
    assert np.all(m1[m2 < -1e-12 + 0.0] == 0.0)
    assert np.all(m1[np.abs(m2 - 1.0) < 1e-15] >= 1.0 - 1e-13)
    # nothing of the ellipse falls outside aperture_np's bounding box
    inside = m2 > 1e-13
    assert np.all(m1[inside] > 0.0)


RECTS = [
    (40, 20.0, 20.0, 10.5, 6.25, 0.0),       # edges ON sub-sample boundaries: strict '<' decides
    (40, 20.0, 20.0, 10.0 + 1.0 / 32, 6.0 - 1.0 / 32, 0.0),  # edges ON sub-sample CENTRES
    (40, 19.5, 20.5, 9.0, 5.0, 0.0),         # half-integer centre
    (40, 20.3, 19.1, 11.7, 4.2, 0.0),        # generic
    (40, 1.0, 38.0, 6.0, 5.0, 0.0),          # straddles the grid edge
    (32, 16.2, 15.7, 0.5, 0.25, 0.0),        # sub-pixel rectangle
    (32, 16.0, 16.0, 1.0 / 32, 1.0 / 32, 0.0),  # narrower than one sub-sample
    (40, 20.3, 19.1, 11.7, 4.2, 0.6),        # tilted
    (40, 20.0, 20.0, 12.0, 4.0, math.pi / 2),  # quarter turn
]


@pytest.mark.parametrize("n,xc,yc,w,h,th", RECTS)
def test_rectangle_restatements_agree(n, xc, yc, w, h, th):
    m1 = rectangle_mask((n, n), xc, yc, w, h, th, 32)
    m2 = rectangle_mask_alt((n, n), xc, yc, w, h, th, 32)
    assert m1 is not None
    assert np.array_equal(m1, m2)  # counts of the same samples: exactly equal
    assert np.all((m1 * 1024) % 1 == 0)


def test_bounding_box_conventions_at_half_integers():
    """ixmin = floor(xc - ext + 0.5), ixmax = ceil(xc + ext + 0.5) exclusive: an extent ending exactly on a
    pixel boundary leaves the neighbouring pixel out of the box -- and that pixel's true overlap is zero."""
    n = 24
    for xc, a in ((12.0, 4.5), (11.5, 4.0), (12.5, 3.0)):
        m1 = ellipse_mask((n, n), xc, 12.0, a, 3.0, 0.0)
        m2 = ellipse_mask_alt((n, n), xc, 12.0, a, 3.0, 0.0)
        right = int(round(xc + a + 0.5))  # first pixel beyond the extent
        assert np.all(m1[:, right:] == 0.0) and np.all(m2[:, right:] < 1e-15)
        assert m1[12, right - 1] > 0.0
    # an aperture wholly off the grid has no image (the reference then fails on u *= None)
    assert ellipse_mask((n, n), 40.0, 12.0, 3.0, 3.0) is None
    assert rectangle_mask((n, n), -20.0, 12.0, 3.0, 3.0) is None
    # one whose box just overlaps column 0
    edge = ellipse_mask((n, n), -2.6, 12.0, 3.0, 3.0)
    assert edge is not None and edge[:, 0].sum() > 0.0 and edge[:, 1:].sum() == 0.0
    assert np.max(np.abs(edge - ellipse_mask_alt((n, n), -2.6, 12.0, 3.0, 3.0))) < 2e-14
