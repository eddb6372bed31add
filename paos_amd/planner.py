"""Host planner: the scalar half of every WFO operator.

The reference mixes, inside each ``WFO`` method, a few dozen floating-point
operations on the pilot Gaussian beam (paos/classes/wfo.py:318-357, 386-416,
434-443, 454-460, 483-508, 520-544, 556-572) with O(N^2) array work.  The array
work lives on the GPU; this module keeps the scalar algebra on the host, in IEEE
double with the reference's operation order, because it *decides* which
propagator runs (``|z - zw0| < 2 zr``, ``|dz| < wl/1000``) and those branches
must agree with the reference bit for bit.

Each method returns what the device needs: a 5-double block
``[enable, sx, sy, coef, sgn]`` (include/paos_hip.h, PAOS_PHASE_STRIDE) or
``None`` when the reference would return without touching the field.
"""
import math

import numpy as np


def _sqrt(x):
    # math.sqrt == np.sqrt bit for bit (both correctly rounded); this one skips NumPy's
    # scalar dispatch.  Negative / nan inputs keep NumPy's nan instead of raising.
    return math.sqrt(x) if x >= 0.0 else np.sqrt(x)


class PilotBeam:
    """Scalar state of one wavefront -- the attributes of wfo.py:105-120."""

    def __init__(self, beam_diameter, wl, grid_size, zoom):
        assert np.log2(grid_size).is_integer(), "Grid size should be 2**n"
        assert zoom > 0, "zoom factor should be positive"
        assert beam_diameter > 0, "beam diameter should be positive"
        assert wl > 0, "a wavelength should be positive"
        self.n = int(grid_size)
        self.wl = wl
        self.z = 0.0
        self.w0 = beam_diameter / 2.0
        self.zw0 = 0.0
        self.zr = np.pi * self.w0**2 / wl
        self.rayleigh_factor = 2.0
        self.dx = beam_diameter * zoom / grid_size
        self.dy = beam_diameter * zoom / grid_size
        self.C = 0.0
        self.fratio = np.inf
        self.propagator = ""

    # ---- read-outs ---------------------------------------------------------------
    @property
    def wz(self):
        return self.w0 * _sqrt(1.0 + ((self.z - self.zw0) / self.zr) ** 2)

    @property
    def distancetofocus(self):
        return self.zw0 - self.z

    @property
    def extent(self):
        n = self.n
        return (-n // 2 * self.dx, (n // 2 - 1) * self.dx, -n // 2 * self.dy, (n // 2 - 1) * self.dy)

    def region(self, z=None):
        """'I' within rayleigh_factor * zr of the waist, else 'O' (wfo.py:280-302)."""
        gap = (self.z if z is None else z) - self.zw0
        return "I" if abs(gap) < self.rayleigh_factor * self.zr else "O"

    # ---- operators -----------------------------------------------------------------
    def lens(self, lens_fl):
        """wfo.py:318-366.  Returns the phase block of
        exp(2 pi i * (-(x^2+y^2) * (0.5 lens_phase / wl)))."""
        wz = self.w0 * _sqrt(1.0 + ((self.z - self.zw0) / self.zr) ** 2)
        gap = self.z - self.zw0
        regime = self.region()
        curv_in = gap / (gap**2 + self.zr**2)
        curv_out = curv_in - 1.0 / lens_fl
        self.w0 = wz / _sqrt(1.0 + (np.pi * wz**2 * curv_out / self.wl) ** 2)
        self.zw0 = -curv_out / (curv_out**2 + (self.wl / (np.pi * wz**2)) ** 2) + self.z
        self.zr = np.pi * self.w0**2 / self.wl
        regime += self.region()

        ref_in = 0.0 if (regime[0] == "I" or self.C == 0.0) else 1 / gap
        gap = self.z - self.zw0
        ref_out = 0.0 if regime[1] == "I" else 1 / gap
        self.C = ref_out
        if regime == "II":
            power = 1.0 / lens_fl
        elif regime == "IO":
            power = 1 / lens_fl + ref_out
        elif regime == "OI":
            power = 1.0 / lens_fl - ref_in
        else:
            power = 1.0 / lens_fl - ref_in + ref_out
        self.fratio = abs(gap) / (2 * wz)
        return [1.0, self.dx, self.dy, 0.5 * power / self.wl, -1.0]

    def magnification(self, My, Mx=None):
        """wfo.py:386-416 (note run() passes (Mt, Ms), run.py:195)."""
        if Mx is None:
            Mx = My
        assert Mx > 0.0, "Negative magnification not implemented yet."
        assert My > 0.0, "Negative magnification not implemented yet."
        self.dx *= Mx
        self.dy *= My
        if np.abs(Mx - 1.0) < 1.0e-8:
            return
        gap = self.z - self.zw0
        wz = self.w0 * np.sqrt(1.0 + ((self.z - self.zw0) / self.zr) ** 2)
        gap *= Mx**2
        wz *= Mx
        self.w0 *= Mx
        self.zr *= Mx**2
        self.zw0 = self.z - gap
        self.fratio = np.abs(gap) / (2 * wz)

    def change_medium(self, n1n2):
        """wfo.py:434-443."""
        gap = self.z - self.zw0
        gap /= n1n2
        self.zr /= n1n2
        self.wl *= n1n2
        self.zw0 = self.z - gap
        self.fratio /= n1n2

    def _freq_steps(self):
        # np.fft.fftfreq(n, d): val = 1.0 / (n * d), results = integers * val
        return 1.0 / (self.n * self.dx), 1.0 / (self.n * self.dy)

    def ptp(self, dz):
        """wfo.py:454-472: block for H = exp(-i (pi wl dz)(fx^2 + fy^2))."""
        if abs(dz) < 0.001 * self.wl:
            return None
        if self.C != 0:
            raise ValueError("PTP wavefront should be planar")
        fsx, fsy = self._freq_steps()
        block = [1.0, fsx, fsy, (np.pi * self.wl * dz), -1.0]
        self.z = self.z + dz
        return block

    def stw(self, dz):
        """wfo.py:483-509: (block for Q = exp(+i (pi wl dz) f^2), inverse flag)."""
        if abs(dz) < 0.001 * self.wl:
            return None
        if self.C == 0.0:
            raise ValueError("STW wavefront should not be planar")
        fsx, fsy = self._freq_steps()
        block = [1.0, fsx, fsy, (np.pi * self.wl * dz), 1.0]
        self.z = self.z + dz
        self.C = 0.0
        # (fx[1] - fx[0]) * wl * |dz| with fx[1] = 1 * val, fx[0] = 0 * val
        self.dx = (fsx - 0.0) * self.wl * abs(dz)
        self.dy = (fsy - 0.0) * self.wl * abs(dz)
        return block, not (dz >= 0)

    def wts(self, dz):
        """wfo.py:520-545: (block for P = exp(+i (pi/(dz wl)) (x^2+y^2)), inverse flag)."""
        if abs(dz) < 0.001 * self.wl:
            return None
        if self.C != 0.0:
            raise ValueError("WTS wavefront should be planar")
        block = [1.0, self.dx, self.dy, (np.pi / (dz * self.wl)), 1.0]
        self.z = self.z + dz
        self.C = 1 / (self.z - self.zw0)
        self.dx = self.wl * abs(dz) / (self.n * self.dx)
        self.dy = self.wl * abs(dz) / (self.n * self.dy)
        return block, not (dz >= 0)

    def propagate(self, dz):
        """wfo.py:556-572: list of ('stw'|'ptp'|'wts', block[, inverse]) steps."""
        regime = self.region() + self.region(self.z + dz)
        z1 = self.z
        z2 = self.z + dz
        steps = []

        def add(kind, res):
            if res is None:
                return
            if kind == "ptp":
                steps.append((kind, res, False))
            else:
                steps.append((kind, res[0], res[1]))

        if regime == "II":
            add("ptp", self.ptp(dz))
        elif regime == "OI":
            add("stw", self.stw(self.zw0 - z1))
            add("ptp", self.ptp(z2 - self.zw0))
        elif regime == "IO":
            add("ptp", self.ptp(self.zw0 - z1))
            add("wts", self.wts(z2 - self.zw0))
        elif regime == "OO":
            add("stw", self.stw(self.zw0 - z1))
            add("wts", self.wts(z2 - self.zw0))
        self.propagator = regime
        return steps


# ---- the same scalars for a whole batch, in C (include/paos_plan.h) ------------------------------------
_PROP_NAMES = ("", "II", "IO", "OI", "OO")
_PLAN_ERRORS = {
    1: (AssertionError, "Negative magnification not implemented yet."),
    2: (ValueError, "PTP wavefront should be planar"),
    3: (ValueError, "STW wavefront should not be planar"),
    4: (ValueError, "WTS wavefront should be planar"),
}
_plan_lib = None


def _plan():
    global _plan_lib
    if _plan_lib is None:
        import ctypes

        from . import _lib

        lib = _lib.load()
        dp = ctypes.POINTER(ctypes.c_double)
        lib.paos_plan_init.restype = ctypes.c_int
        lib.paos_plan_init.argtypes = [ctypes.c_int, ctypes.c_double, dp, ctypes.c_int, ctypes.c_double, dp]
        lib.paos_plan_readout.restype = ctypes.c_int
        lib.paos_plan_readout.argtypes = [ctypes.c_int, dp, dp, dp]
        lib.paos_plan_surface.restype = ctypes.c_int
        lib.paos_plan_surface.argtypes = [ctypes.c_int, ctypes.c_int, dp] + [dp] * 5 + [dp] * 6 + [ctypes.POINTER(ctypes.c_int)]
        _plan_lib = (lib, dp, ctypes)
    return _plan_lib


class BeamBatch:
    """``PilotBeam`` for B wavefronts at once: the state lives in one [B][10] array, one C call per
    surface does what B x (magnification, change_medium, lens, propagate) calls do in Python
    (include/paos_plan.h; bit-identical to ``PilotBeam``, tests/test_r2_host.py)."""
    WL, Z, W0, ZW0, ZR, DX, DY, C, FRATIO, PROP = range(10)

    def __init__(self, beam_diameter, wavelengths, grid_size, zoom):
        assert np.log2(grid_size).is_integer(), "Grid size should be 2**n"
        assert zoom > 0, "zoom factor should be positive"
        assert beam_diameter > 0, "beam diameter should be positive"
        wl = np.ascontiguousarray(wavelengths, dtype=np.float64).reshape(-1)
        assert np.all(wl > 0), "a wavelength should be positive"
        self.n, self.batch = int(grid_size), int(wl.size)
        self.state = np.empty((self.batch, 10), dtype=np.float64)
        lib, dp, _ = _plan()
        if lib.paos_plan_init(self.batch, float(beam_diameter), wl.ctypes.data_as(dp), self.n, float(zoom),
                              self.state.ctypes.data_as(dp)) != 0:
            raise ValueError("paos_plan_init rejected its arguments")
        b = self.batch
        self._lens, self._stw, self._ptp, self._wts = (np.empty((b, 5)) for _ in range(4))
        self._inv_stw, self._inv_wts = np.empty(b), np.empty(b)
        self._status = np.empty(b, dtype=np.int32)

    def column(self, k):
        return self.state[:, k]

    def readout(self):
        """(wz, distancetofocus) arrays -- wfo.py:142-150."""
        lib, dp, _ = _plan()
        wz, dtf = np.empty(self.batch), np.empty(self.batch)
        lib.paos_plan_readout(self.batch, self.state.ctypes.data_as(dp), wz.ctypes.data_as(dp), dtf.ctypes.data_as(dp))
        return wz, dtf

    def propagators(self):
        return [_PROP_NAMES[int(c)] for c in self.state[:, self.PROP]]

    def extents(self):
        n = self.n
        return [(-n // 2 * dx, (n // 2 - 1) * dx, -n // 2 * dy, (n // 2 - 1) * dy)
                for dx, dy in zip(self.state[:, self.DX].tolist(), self.state[:, self.DY].tolist())]

    def surface(self, Mt, Ms, fl, T, n1n2):
        """run.py:181-207 for every item.  Returns fresh arrays (lens, stw, ptp, wts blocks [B][5] with
        enable = 0 where the step does not run; inverse flags of stw and wts [B])."""
        lib, dp, ctypes = _plan()
        args = [np.ascontiguousarray(a, dtype=np.float64) for a in (Mt, Ms, fl, T, n1n2)]
        if any(a.shape != (self.batch,) for a in args):
            raise ValueError("one value per wavefront is required")
        lens, stw, ptp, wts = (np.empty((self.batch, 5)) for _ in range(4))
        inv_stw, inv_wts = np.empty(self.batch), np.empty(self.batch)
        bad = lib.paos_plan_surface(self.batch, self.n, self.state.ctypes.data_as(dp), *[a.ctypes.data_as(dp) for a in args],
                                    lens.ctypes.data_as(dp), stw.ctypes.data_as(dp), ptp.ctypes.data_as(dp),
                                    wts.ctypes.data_as(dp), inv_stw.ctypes.data_as(dp), inv_wts.ctypes.data_as(dp),
                                    self._status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        if bad:
            code = int(self._status[np.nonzero(self._status)[0][0]])
            exc, msg = _PLAN_ERRORS[code]
            raise exc(msg)
        return lens, stw, ptp, wts, inv_stw, inv_wts


# ---- Zernike tables -------------------------------------------------------------------
def jacobi_recurrence(nmax):
    """Constants of P_k^{(a,0)}(x) = (A x + B) P_{k-1} - C P_{k-2} for a = 0..nmax,
    k = 0..nmax//2, shape (nmax+1, kdim, 3).  The GPU evaluates the radial
    polynomial of the reference, (-1)^k rho^a P_k^{(a,0)}(1 - 2 rho^2)
    (zernike.py:245-247), with this stable three-term recurrence."""
    kdim = nmax // 2 + 1
    tab = np.zeros((nmax + 1, kdim, 3), dtype=np.float64)
    for a in range(nmax + 1):
        for k in range(1, kdim):
            if k == 1:
                tab[a, k] = ((a + 2) / 2.0, a / 2.0, 0.0)
                continue
            den = 2.0 * k * (k + a) * (2 * k + a - 2)
            tab[a, k, 0] = (2 * k + a - 1) * (2 * k + a) * (2 * k + a - 2) / den
            tab[a, k, 1] = (2 * k + a - 1) * (a * a) / den
            tab[a, k, 2] = 2.0 * (k + a - 1) * (k - 1) * (2 * k + a) / den
    return tab


_BLOCK_TEMPLATES = {}


def _block_template(m, n, norm, nmax):
    """Where each polynomial's coefficient lands in the [|m|][k] planes, and its (-1)^k * norm
    prefactor -- the same for every wavefront of a batch, so computed once per (m, n, norm, nmax)."""
    key = (m.tobytes(), n.tobytes(), np.asarray(norm, dtype=np.float64).tobytes(), int(nmax))
    hit = _BLOCK_TEMPLATES.get(key)
    if hit is None:
        kdim = nmax // 2 + 1
        k = (n - np.abs(m)) // 2
        pref = np.array([(-1.0) ** int(kk) * float(nrm) for kk, nrm in zip(k, norm)], dtype=np.float64)
        flat = np.abs(m) * kdim + k
        if len(_BLOCK_TEMPLATES) > 256:
            _BLOCK_TEMPLATES.clear()
        hit = _BLOCK_TEMPLATES[key] = (pref, flat[m >= 0], m >= 0, flat[m < 0], m < 0, kdim)
    return hit


def zernike_block(m, n, norm, coeffs, dx, dy, radius, wl, origin="x", offset_deg=0.0, nmax=None):
    """Per-item parameter block of paos_zernike: header + cos / sin coefficient
    planes indexed [|m|][k], k = (n - |m|)/2, with (-1)^k * norm * Z folded in."""
    m = np.asarray(m, dtype=int)
    n = np.asarray(n, dtype=int)
    if nmax is None:
        nmax = int(n.max())
    pref, at_cos, is_cos, at_sin, is_sin, kdim = _block_template(m, n, norm, nmax)
    vals = pref * np.asarray(coeffs, dtype=np.float64)  # ((-1)^k * norm) * Z, term by term
    cosp = np.zeros((nmax + 1) * kdim, dtype=np.float64)
    sinp = np.zeros((nmax + 1) * kdim, dtype=np.float64)
    np.add.at(cosp, at_cos, vals[is_cos])  # += like the loop form (repeated indices accumulate)
    np.add.at(sinp, at_sin, vals[is_sin])
    if origin not in ("x", "y"):
        raise ValueError(f"Origin {origin} not recognised. Origin shall be either x or y")
    off = np.deg2rad(offset_deg)
    head = [1.0, dx, dy, radius, 1.0 if origin == "y" else 0.0, np.cos(off), np.sin(off), 1.0 / wl]
    return np.concatenate([head, cosp, sinp]), nmax, kdim


def gram_polynomials(m, n, norm):
    """[K][4] descriptors of paos_zernike_gram: |m|, k = (n - |m|)/2, is_sin, (-1)^k norm
    (the radial part is (-1)^k rho^|m| P_k, zernike.py:245-247; m < 0 takes sin, :100-104)."""
    rows = []
    for mk, nk, nrm in zip(np.asarray(m, dtype=int), np.asarray(n, dtype=int), norm):
        k = (nk - abs(mk)) // 2
        rows.append([abs(mk), k, 1.0 if mk < 0 else 0.0, (-1.0) ** k * nrm])
    return np.array(rows, dtype=np.float64)


def orthonorm_matrix(sums, count, k):
    """M of PolyOrthoNorm: covariance = masked mean of Z_i Z_j with |.| < 1e-10 zeroed
    (zernike.py:311-316), M = inv(cholesky(cov)) with |.| < 1e-10 zeroed (zernike.py:392-395).
    ``sums`` lists i <= j row by row (paos_zernike_gram).  numpy raises LinAlgError exactly where
    the reference does (polynomials not independent over the pupil)."""
    if count <= 0:
        raise ValueError("the pupil of the orthonormal polynomials is empty")
    cov = np.empty((k, k), dtype=np.float64)
    iu = np.triu_indices(k)
    cov[iu] = np.asarray(sums, dtype=np.float64) / float(count)
    cov.T[iu] = cov[iu]
    cov[np.abs(cov) < 1e-10] = 0.0
    qt = np.linalg.cholesky(cov)
    m = np.linalg.inv(qt)
    m[np.abs(m) < 1.0e-10] = 0.0
    return m
