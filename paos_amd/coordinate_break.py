"""Chief-ray decentre + tilt (host scalars).

Drop-in for ``paos.core.coordinateBreak.coordinate_break`` (reference
paos/core/coordinateBreak.py:7-72): decentre first, then an "xyz" Euler
rotation evaluated with SciPy exactly as the reference calls it (lower-case
axes = extrinsic in SciPy).  Decides where ``run`` centres the
aperture masks (reference paos/core/run.py:97-108); no GPU work.
"""
import numpy as np
from scipy.spatial.transform import Rotation


def _finite_or_zero(v):
    return v if np.isfinite(v) else 0.0


def coordinate_break(vt, vs, xdec, ydec, xrot, yrot, zrot, order=0):
    if order != 0:
        raise ValueError("Coordinate break orders other than 0 not implemented yet")
    xdec, ydec = _finite_or_zero(xdec), _finite_or_zero(ydec)
    angles = [_finite_or_zero(xrot), _finite_or_zero(yrot), _finite_or_zero(zrot)]
    back = Rotation.from_euler("xyz", angles, degrees=True).inv()

    direction = back.apply([vs[1], vt[1], 1])
    direction /= direction[2]
    point = back.apply([vs[0] - xdec, vt[0] - ydec, 0.0])
    # slide along the ray to the new z = 0 plane
    point = point - direction * point[2] / direction[2]
    return np.array([point[1], direction[1]]), np.array([point[0], direction[0]])
