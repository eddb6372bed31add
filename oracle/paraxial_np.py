"""Paraxial ray-transfer helpers on the CPU.  TEST INFRASTRUCTURE ONLY.

Restates paos/classes/abcd.py (2x2 ray matrix and its T.D.M factorisation,
abcd.py:44-116,157-164) and paos/core/coordinateBreak.py:7-72 (decentre + tilt of
the chief ray).  Pinned by tests/golden/scalars_*.npz and the reference
notebook numbers quoted in SURVEY.md section 9.9.
"""
import numpy as np
from scipy.spatial.transform import Rotation


class RayMatrix:
    """ABCD matrix with the derived quantities run() reads (abcd.py:98-116)."""

    def __init__(self, thickness=0.0, curvature=0.0, n1=1.0, n2=1.0, M=1.0):
        if n1 == 0 or n2 == 0 or M == 0:
            raise ValueError("Refractive index and magnification shall not be zero")
        shift = np.array([[1.0, thickness], [0, 1.0]])
        if n1 == n2:  # thin lens, abcd.py:81-83
            power = np.array([[1.0, 0.0], [-curvature, 1.0]])
        else:  # dioptre or mirror, abcd.py:84-86
            power = np.array([[1.0, 0.0], [-(1 - n1 / n2) * curvature, n1 / n2]])
        mag = np.array([[M, 0.0], [0.0, 1.0 / M]])
        self._m = shift @ power @ mag
        self.cin = np.sign(n1)
        self.cout = np.sign(n2)

    @property
    def ABCD(self):
        return self._m

    @ABCD.setter
    def ABCD(self, value):
        self._m = value.copy()

    def __call__(self):
        return self._m

    @property
    def thickness(self):
        (_, b), (_, d) = self._m
        return b / d

    @property
    def M(self):
        (a, b), (c, d) = self._m
        return (a * d - b * c) / d

    @property
    def n1n2(self):
        (_, _), (_, d) = self._m
        return d * self.M

    @property
    def power(self):
        (_, _), (c, _) = self._m
        return -c / self.M

    @property
    def f_eff(self):
        return 1 / (self.power * self.M)

    def __mul__(self, other):
        out = RayMatrix()
        out.ABCD = self._m @ other()
        out.cin = other.cin
        out.cout = other.cout
        return out


def tilt_decentre(vt, vs, xdec, ydec, xrot, yrot, zrot, order=0):
    """New (y, uy) and (x, ux) after a coordinate break -- coordinateBreak.py:7-72
    (SciPy "xyz" = extrinsic, as the reference actually calls it)."""
    if order != 0:
        raise ValueError("Coordinate break orders other than 0 not implemented yet")
    xdec, ydec, xrot, yrot, zrot = [
        v if np.isfinite(v) else 0.0 for v in (xdec, ydec, xrot, yrot, zrot)
    ]
    rot = Rotation.from_euler("xyz", [xrot, yrot, zrot], degrees=True)
    r0 = [vs[0] - xdec, vt[0] - ydec, 0.0]
    n0 = [vs[1], vt[1], 1]
    n1 = rot.inv().apply(n0)
    n1 /= n1[2]
    r1_ln1 = rot.inv().apply(r0)
    r1 = r1_ln1 - n1 * r1_ln1[2] / n1[2]
    return np.array([r1[1], n1[1]]), np.array([r1[0], n1[0]])
