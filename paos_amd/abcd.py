"""Paraxial ray-transfer matrix used by the host planner.

Drop-in for ``paos.classes.abcd.ABCD`` (reference paos/classes/abcd.py:6-164):
same constructor, same derived properties (``thickness``, ``M``, ``n1n2``,
``power``, ``cin``, ``cout``, ``f_eff``, ``ABCD``), ``__call__`` returning the
2x2 array and ``__mul__`` composing two surfaces.  Host-side scalar code only:
``run`` reads ``fl = cout/power``, ``T = cout*thickness``, ``n1n2`` and ``M``
from it (reference paos/core/run.py:181-190); no GPU kernel is involved.
"""
import numpy as np


# Bumped by every in-place change of a matrix or a direction flag (the ``ABCD`` / ``cin`` / ``cout`` setters): callers that
# keep derived values of MANY matrices (run.py: the per-surface gate arrays of a batch) compare it instead of asking
# each object again.
EPOCH = [0]


def _compose(thickness, curvature, n1, n2, mag):
    """[[1,t],[0,1]] @ D @ diag(M, 1/M) with D a thin lens when n1 == n2 and a
    dioptre/mirror otherwise (abcd.py:79-90).  Plain matmul keeps the
    reference's floating-point results."""
    ratio = n1 / n2
    t_mat = np.array([[1.0, thickness], [0, 1.0]])
    if n1 == n2:
        d_mat = np.array([[1.0, 0.0], [-curvature, 1.0]])
    else:
        d_mat = np.array([[1.0, 0.0], [-(1 - ratio) * curvature, ratio]])
    m_mat = np.array([[mag, 0.0], [0.0, 1.0 / mag]])
    return t_mat @ d_mat @ m_mat


class ABCD:
    def __init__(self, thickness=0.0, curvature=0.0, n1=1.0, n2=1.0, M=1.0):
        if n1 == 0 or n2 == 0 or M == 0:
            raise ValueError("Refractive index and magnification shall not be zero")
        self._mat = _compose(thickness, curvature, n1, n2, M)
        self._cin = np.sign(n1)
        self._cout = np.sign(n2)
        self._cache = {}  # read-outs of the factorisation, filled on first use

    # matrix access ---------------------------------------------------------
    def __call__(self):
        return self._mat

    @property
    def ABCD(self):
        return self._mat

    @ABCD.setter
    def ABCD(self, value):
        self._mat = value.copy()
        self._cache = {}
        EPOCH[0] += 1

    # direction of travel (+1 left-to-right, -1 right-to-left) ---------------
    @property
    def cin(self):
        return self._cin

    @cin.setter
    def cin(self, value):
        self._cin = value
        EPOCH[0] += 1

    @property
    def cout(self):
        return self._cout

    @cout.setter
    def cout(self, value):
        self._cout = value
        self._cache.pop("gates", None)
        EPOCH[0] += 1

    def gates(self):
        """(M, fl, T, n1n2) the way the propagation loop derives them from a surface's matrix
        (run.py:181-190): fl = cout / power (inf for a powerless surface), T = cout * thickness."""
        g = self._cache.get("gates")
        if g is None:
            power = self.power
            g = (self.M, np.inf if power == 0 else self.cout / power, self.cout * self.thickness, self.n1n2)
            self._cache["gates"] = g
        return g

    # factorisation read-outs (abcd.py:98-116,142-144) ------------------------
    def _readout(self, name):
        val = self._cache.get(name)
        if val is None:
            m = self._mat
            if name == "thickness":
                val = m[0, 1] / m[1, 1]
            elif name == "M":
                val = (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) / m[1, 1]
            elif name == "n1n2":
                val = m[1, 1] * self._readout("M")
            else:  # power
                val = -m[1, 0] / self._readout("M")
            self._cache[name] = val
        return val

    @property
    def thickness(self):
        return self._readout("thickness")

    @property
    def M(self):
        return self._readout("M")

    @property
    def n1n2(self):
        return self._readout("n1n2")

    @property
    def power(self):
        return self._readout("power")

    @property
    def f_eff(self):
        return 1 / (self.power * self.M)

    def __mul__(self, other):
        """self after other: matrix product, direction flags taken from
        ``other`` (abcd.py:157-164)."""
        res = ABCD.__new__(ABCD)  # skip the constructor's three 2x2 products
        res._mat = self._mat @ other()
        res._cin = other.cin
        res._cout = other.cout
        res._cache = {}
        return res

    def __repr__(self):
        (a, b), (c, d) = self._mat
        return f"ABCD([[{a!r}, {b!r}], [{c!r}, {d!r}]])"
