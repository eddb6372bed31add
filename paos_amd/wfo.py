"""Device-backed wavefront object with the PAOS ``WFO`` interface.

Same constructor, read-only properties and methods as ``paos.classes.wfo.WFO``
(reference paos/classes/wfo.py:12-654).  The N x N complex field lives in HBM
(``_lib.DeviceFields``); every method below runs the scalar pilot-beam part on
the host (``planner.PilotBeam``) and the field part as HIP kernels.  ``grid_sag``
and ``psd`` (wfo.py:656-949) build their WFE map on the host (``phase_maps``: the reference's own
NumPy / SciPy steps; the scikit-image resampling of a sag at another pixel scale is a parity-unpinned
restatement on scipy.ndimage) and apply it with the ``paos_phase_map`` kernel.
"""
import numpy as np

from . import _lib
from .aperture import bbox_misses_grid, make_aperture, EllipticalAperture
from . import phase_maps as _phase_maps
from .phase_maps import MAP_SERIAL, PsdScreen, grid_sag_map
from .planner import PilotBeam, gram_polynomials, jacobi_recurrence, orthonorm_matrix, zernike_block
from .zernike import Zernike, norm_factors


class _FieldView:
    """What ``wfo._wfo`` hands out: has ``.shape``/``.dtype`` (read by run.py:139 and
    notebooks) and converts to a NumPy array on demand (one device download)."""

    def __init__(self, owner):
        self._owner = owner
        self.shape = (owner._beam.n, owner._beam.n)
        self.dtype = np.dtype(np.complex128)

    def __array__(self, dtype=None, copy=None):
        arr = self._owner._dev.download(0, _lib.WHAT_FIELD)
        return arr if dtype is None else arr.astype(dtype)

    def copy(self):
        return np.asarray(self)


class WFO:
    def __init__(self, beam_diameter, wl, grid_size, zoom, precision="fp64", device=0):
        self._beam = PilotBeam(beam_diameter, wl, grid_size, zoom)
        self._zoom = zoom
        self._dev = _lib.DeviceFields(int(grid_size), 1, precision, device)
        self._dev.fill(1.0 + 0.0j)  # np.ones(..., complex128), wfo.py:118

    # ---- scalar read-outs (wfo.py:122-160,174-193) ----------------------------------
    wl = property(lambda self: self._beam.wl)
    z = property(lambda self: self._beam.z)
    w0 = property(lambda self: self._beam.w0)
    zw0 = property(lambda self: self._beam.zw0)
    zr = property(lambda self: self._beam.zr)
    rayleigh_factor = property(lambda self: self._beam.rayleigh_factor)
    dx = property(lambda self: self._beam.dx)
    dy = property(lambda self: self._beam.dy)
    C = property(lambda self: self._beam.C)
    fratio = property(lambda self: self._beam.fratio)
    wz = property(lambda self: self._beam.wz)
    distancetofocus = property(lambda self: self._beam.distancetofocus)
    propagator = property(lambda self: self._beam.propagator)
    extent = property(lambda self: self._beam.extent)

    # ---- field read-outs (wfo.py:162-172) ----------------------------------------------
    @property
    def wfo(self):
        return self._dev.download(0, _lib.WHAT_FIELD)

    @property
    def amplitude(self):
        return self._dev.download(0, _lib.WHAT_AMPLITUDE)

    @property
    def phase(self):
        return self._dev.download(0, _lib.WHAT_PHASE)

    @property
    def intensity(self):
        """|u|^2 -- the PSF as the reference's callers define it (plot.py:125-130)."""
        return self._dev.download(0, _lib.WHAT_INTENSITY)

    @property
    def _wfo(self):
        return _FieldView(self)

    @_wfo.setter
    def _wfo(self, value):
        self._dev.upload(0, value)

    # ---- operators ------------------------------------------------------------------------
    def make_stop(self):
        self._dev.make_stop()

    def aperture(self, xc, yc, hx=None, hy=None, r=None, shape="elliptical", tilt=None,
                 obscuration=False):
        b = self._beam
        ap = make_aperture(b.n, b.dx, b.dy, xc, yc, hx=hx, hy=hy, r=r, shape=shape, tilt=tilt)
        if bbox_misses_grid(ap, b.n):
            # photutils' to_image() gives None here and the reference dies on u *= None
            raise TypeError("aperture does not overlap the grid (mask is None in the reference)")
        code = _lib.SHAPE_ELLIPSE if isinstance(ap, EllipticalAperture) else _lib.SHAPE_RECT
        self._dev.aperture(code, [ap.block(obscuration=obscuration)])
        return ap

    def insideout(self, z=None):
        return self._beam.region(z)

    def lens(self, lens_fl):
        self._dev.phase([self._beam.lens(lens_fl)], mul2pi=True)

    def Magnification(self, My, Mx=None):
        self._beam.magnification(My, Mx)

    def ChangeMedium(self, n1n2):
        self._beam.change_medium(n1n2)

    def ptp(self, dz):
        block = self._beam.ptp(dz)
        if block is not None:
            self._dev.ptp([block])

    def stw(self, dz):
        res = self._beam.stw(dz)
        if res is not None:
            self._dev.stw([res[0]], res[1])

    def wts(self, dz):
        res = self._beam.wts(dz)
        if res is not None:
            self._dev.wts([res[0]], res[1])

    def propagate(self, dz):
        for kind, block, inverse in self._beam.propagate(dz):
            if kind == "ptp":
                self._dev.ptp([block])
            elif kind == "stw":
                self._dev.stw([block], inverse)
            else:
                self._dev.wts([block], inverse)

    def zernikes(self, index, Z, ordering, normalize, radius, offset=0.0, origin="x",
                 orthonorm=False, mask=False):
        """wfo.py:574-654.  Returns the masked wfe map like the reference.  ``mask`` (True =
        outside the pupil) restricts the map, and with ``orthonorm`` the polynomials are the
        Gram-Schmidt combinations orthonormal over the unmasked pixels (PolyOrthoNorm,
        zernike.py:320-402)."""
        index = np.asarray(index)
        assert not np.any(np.diff(index) - 1), "Zernike sequence should be continuous"
        if ordering not in ("ansi", "noll", "fringe", "standard"):
            raise AssertionError("Unrecognised ordering scheme.")
        m, n = Zernike.j2mn(len(index), ordering)
        norm = norm_factors(m, n, normalize)
        b = self._beam
        coeff = np.asarray(Z, float)
        pupil = mask is not False and mask is not None
        if pupil:
            mask = np.asarray(mask, dtype=bool)
            if mask.shape != (b.n, b.n):
                raise ValueError("mask must have the shape of the field")
            self._dev.pupil_upload(0, (~mask).astype(np.float64))

        def block_for(c):
            return zernike_block(m, n, norm, c, b.dx, b.dy, radius, b.wl, origin=origin, offset_deg=offset)

        block, nmax, kdim = block_for(coeff)
        table = jacobi_recurrence(nmax)
        if orthonorm:
            sums, counts = self._dev.zernike_gram(nmax, kdim, table, [block], gram_polynomials(m, n, norm),
                                                  pupil=pupil)
            coeff = orthonorm_matrix(sums[0], counts[0], len(m)).T @ coeff
            block, _, _ = block_for(coeff)
        wfe = self._dev.zernike(nmax, kdim, table, [block], want_wfe=True, pupil=pupil)
        outside = np.isnan(wfe)
        return np.ma.MaskedArray(data=np.where(outside, 0.0, wfe), mask=outside, fill_value=0.0)

    def grid_sag(self, sag, nx, ny, delx, dely, xdec=0.0, ydec=0.0):
        """wfo.py:656-871: add a user-specified sag (metres) to the wavefront; returns the masked
        WFE map.  Maps must come at the wavefront's pixel scale (see ``phase_maps``)."""
        b = self._beam
        wfe = grid_sag_map(sag, nx, ny, delx, dely, xdec, ydec, (b.n, b.n), b.dx, b.dy)
        self._dev.phase_map(0, wfe.filled(0), b.wl)
        return wfe

    def psd(self, A=10.0, B=0.0, C=0.0, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="m"):
        """wfo.py:873-949: add a WFE screen drawn from a power spectral density plus surface
        roughness; returns the WFE map.  The draw uses NumPy's legacy global generator like the
        reference (``np.random.seed`` makes it reproducible)."""
        b = self._beam
        screen = PsdScreen((b.n, b.n), b.dx, b.dy, A, B, C, fknee, fmin, fmax, SR, units)
        if _phase_maps.psd_on_device(self._dev, b.n):
            # round 5: fft2 -> filter -> ifft2 -> roughness on the library's own passes (paos_psd_screen)
            key = next(MAP_SERIAL)
            out = self._dev.psd_screen(screen.noise, screen.rough, screen.params, key=key, want_map=True)
            self._dev.phase_map_items(None, [0], [b.wl], key=key)
            return np.ma.masked_array(out, mask=np.zeros(out.shape, dtype=bool))
        wfe = screen.host_map()
        self._dev.phase_map(0, np.ma.filled(wfe, 0.0), b.wl)
        return wfe
