"""Host half of the tabulated phase screens: ``WFO.grid_sag`` and ``WFO.psd``.

Both reference methods build an N x N wavefront-error map on the host and end with the same field
operation, ``u *= exp(2 pi i wfe / wl)`` (paos/classes/wfo.py:869-871 and :945-949).  The field
operation is a HIP kernel (``paos_phase_map``); this module restates the map construction with the
reference's own NumPy / SciPy calls in the reference's order, so the maps are bit-identical on the
same NumPy build:

* :func:`grid_sag_map` -- wfo.py:745-867: masking of non-finite / zero samples, the optional
  sub-pixel recentring (``scipy.ndimage.fourier_shift``), zero padding or cropping to the extent of
  the grid: bit-exact against vectors of the reference.  The steps that call scikit-image (``rescale`` for an
  odd size difference or a different pixel scale, ``resize`` for a final shape mismatch; wfo.py:702-716, 739-750,
  786-800, 845-859) run on :func:`_ski_resize`, a restatement of ``skimage.transform.resize`` 0.24.0
  (poetry.lock:3268; the package is neither under /root/reference nor in this image) from its published
  algorithm on ``scipy.ndimage`` -- pinned by the reference's own recorded run of these branches
  (``notebook/ComputeGridSag.ipynb`` cells 10, 12, 15, 17-19: the sums of an anti-aliased rescale to 1e-14 / bit-exact,
  the refine -> crop -> rescale shape trail, RMS / PV to the printed digits: ``tests/test_grid_sag_known_answers.py``).
* :func:`psd_map` -- paos/classes/psd.py:100-160: a white-noise field from NumPy's legacy global
  generator (``np.random.randn`` exactly like the reference, so ``np.random.seed`` makes a run
  reproducible -- and comparable with the reference), filtered by the power-law PSD, plus surface
  roughness.  ``PSD.sfe_rms`` (a sympy integral, psd.py:166-193) only feeds a debug log line in
  the reference and is not evaluated.
"""
import itertools

import numpy as np

# keys under which a context keeps ONE map on the device (paos_phase_map_items / paos_psd_screen): a serial number per map,
# never an object id (ids are handed out again)
MAP_SERIAL = itertools.count(1)

_UNIT_TO_M = {"m": 1.0, "mm": 1.0e-3, "um": 1.0e-6, "micron": 1.0e-6, "nm": 1.0e-9}


def _unit_factor(units):
    """``units.to(u.m)`` of psd.py:160 for a unit name, an astropy unit or a plain factor."""
    if units is None:
        return 1.0
    if isinstance(units, (int, float)):
        return float(units)
    if isinstance(units, str):
        if units not in _UNIT_TO_M:
            raise ValueError(f"unknown length unit {units!r}")
        return _UNIT_TO_M[units]
    name = getattr(units, "name", None)
    if name in _UNIT_TO_M:
        return _UNIT_TO_M[name]
    if hasattr(units, "to"):  # an astropy unit, when astropy is installed
        import astropy.units as u

        return float(units.to(u.m))
    raise ValueError(f"cannot interpret units {units!r}")


def _ski_resize(image, output_shape, anti_aliasing):
    """``skimage.transform.resize(image, output_shape, order=3, anti_aliasing=...)`` of scikit-image 0.24.0 with its
    defaults (``mode="reflect"``, ``cval=0``, ``clip=True``, float input), restated: when shrinking and asked to, a
    Gaussian pre-filter of sigma = (input / output - 1) / 2 per axis; cubic-spline ``ndimage.zoom`` by output / input
    with ``grid_mode=True`` in ndimage's "mirror" boundary mode (what scikit-image maps its "reflect" to); the
    result clipped to the input's value range.  PARITY UNPINNED (no scikit-image to run against)."""
    from scipy import ndimage as ndi

    image = np.asarray(image, dtype=np.float64)
    shape = tuple(int(v) for v in output_shape)
    factors = np.divide(image.shape, shape)
    filtered = image
    if anti_aliasing:
        filtered = ndi.gaussian_filter(image, np.maximum(0, (factors - 1) / 2), cval=0, mode="mirror")
    out = ndi.zoom(filtered, [1 / f for f in factors], order=3, mode="mirror", cval=0, grid_mode=True)
    return np.clip(out, np.nanmin(image), np.nanmax(image))


def _ski_rescale(image, scale_x, scale_y):
    """``skimage.transform.rescale(image, scale=(scale_y, scale_x), anti_aliasing=shrinking, order=3)`` as
    ``rescale_map`` of wfo.py:697-716 calls it: output shape = round(scale * shape), at least 1."""
    shape = np.maximum(np.round(np.array([scale_y, scale_x]) * np.asarray(image.shape)), 1)
    return _ski_resize(image, shape, anti_aliasing=(scale_x < 1.0 or scale_y < 1.0))


def _as_masked(sag):
    """Samples that are not finite or exactly zero carry no information (wfo.py:745-752)."""
    if isinstance(sag, np.ma.MaskedArray):
        return sag
    return np.ma.MaskedArray(sag, mask=~np.isfinite(sag) | (sag == 0))


def _recentre(image, xdec, ydec):
    """Sub-pixel decentre through the Fourier shift theorem (wfo.py:765-770; the shift tuple is the reference's)."""
    from scipy.ndimage import fourier_shift

    return np.fft.ifft2(fourier_shift(np.fft.fft2(image), shift=(-xdec, -ydec))).real


def _fit_extent(planes, fills, axis, excess):
    """Bring every plane to the grid's extent along ``axis``: ``excess`` samples too many are cropped (split
    left/right like wfo.py:812-821, 832-841), ``-excess`` too few are padded with the plane's fill value
    (wfo.py:804-810, 824-830: the sag with 0, its mask with 1)."""
    if excess == 0:
        return planes
    if excess < 0:
        lack = -excess
        before = lack // 2
        width = [(0, 0), (0, 0)]
        width[axis] = (before, lack - before)
        return [np.pad(p, width, mode="constant", constant_values=f) for p, f in zip(planes, fills)]
    first = excess // 2
    last = planes[0].shape[axis] - (excess - first)
    window = [slice(None), slice(None)]
    window[axis] = slice(first, last)
    return [p[tuple(window)] for p in planes]


def grid_sag_map(sag, nx, ny, delx, dely, xdec, ydec, shape, dx, dy):
    """The masked WFE map ``WFO.grid_sag`` applies and returns (wfo.py:745-867)."""
    assert sag.ndim == 2, "sag shall be a 2D array"
    assert sag.shape == (ny, nx)
    sag = _as_masked(sag)
    planes = [sag.filled(0.0), sag.mask.astype(float)]  # heights (masked samples count as 0) and the mask as 0 / 1
    if (xdec != 0) or (ydec != 0):
        planes = [_recentre(p, xdec, ydec) for p in planes]

    # how many samples the map overhangs the grid by, per axis (wfo.py:773-783)
    rows, cols = planes[0].shape
    excess_x = int(np.floor((cols * delx - shape[1] * dx) / delx))
    excess_y = int(np.floor((rows * dely - shape[0] * dy) / dely))
    # an odd overhang cannot be split evenly: sample that axis twice as finely first (wfo.py:785-800)
    up_x, up_y = (2 if excess_x % 2 == 1 else 1), (2 if excess_y % 2 == 1 else 1)
    if up_x != 1 or up_y != 1:
        planes = [_ski_rescale(p, up_x, up_y) for p in planes]
        delx, dely = delx / up_x, dely / up_y
        excess_x, excess_y = excess_x * up_x, excess_y * up_y
    planes = _fit_extent(planes, (0, 1), 1, excess_x)
    planes = _fit_extent(planes, (0, 1), 0, excess_y)

    # the map's pixel scale -> the wavefront's (wfo.py:843-849), then a last nudge to the exact shape (:851-859)
    if (delx / dx != 1) or (dely / dy != 1):
        planes = [_ski_rescale(p, delx / dx, dely / dy) for p in planes]
    if planes[0].shape != tuple(shape):
        shrink = shape[1] / planes[0].shape[1] < 1.0 or shape[0] / planes[0].shape[0] < 1.0
        planes = [_ski_resize(p, shape, anti_aliasing=shrink) for p in planes]
    heights, coverage = planes
    return np.ma.MaskedArray(heights, mask=coverage > 0.1)


def _radial_frequency(shape, dx, dy):
    """|f| on the FFT grid of the pupil, with the origin nudged off zero (wfo.py:908-913)."""
    gx, gy = np.meshgrid(np.fft.fftfreq(shape[1], dx), np.fft.fftfreq(shape[0], dy))
    rho = np.sqrt(gx**2 + gy**2)
    rho[rho == 0] = 1e-100
    return rho


def _psd_band(shape, dx, dy, fmin, fmax):
    """[fmin, fmax] with the reference's defaults and its check (wfo.py:915-925)."""
    nyquist = 0.5 * np.sqrt(dx**-2 + dy**-2)
    if fmax is None:
        fmax = nyquist
    else:
        assert fmax <= nyquist, f"fmax must be less than or equal to f_Nyq ({nyquist})"
    if fmin is None:
        fmin = 1 / (shape[0] * np.max([dx, dy]))
    return fmin, fmax


def psd_map(shape, dx, dy, A=10.0, B=0.0, C=0.0, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="m"):
    """The WFE map ``WFO.psd`` applies and returns (wfo.py:908-943 + psd.py:100-160).  The two draws come from
    NumPy's legacy global generator in the reference's order (screen first, roughness second)."""
    fmin, fmax = _psd_band(shape, dx, dy, fmin, fmax)  # (the reference's check comes before its draws)
    noise, rough = psd_draws(shape)
    return psd_map_from_draws(noise, rough, shape, dx, dy, A, B, C, fknee, fmin, fmax, SR, units)


def psd_map_from_draws(noise, rough, shape, dx, dy, A=10.0, B=0.0, C=0.0, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="m"):
    """psd.py:113-160 on the two white-noise arrays of :func:`psd_draws` (``rough`` may be None when SR is 0)."""
    rho = _radial_frequency(shape, dx, dy)
    fmin, fmax = _psd_band(shape, dx, dy, fmin, fmax)
    n0, n1 = shape  # the reference draws an array of the pupil's shape, whatever it calls the two numbers (psd.py:103)
    spectrum = np.fft.fft2(noise)
    cell = (rho[0, 2] - rho[0, 1]) * (rho[2, 0] - rho[1, 0])  # frequency-bin area
    density = A / (B + (rho / fknee) ** C) / (2 * np.pi * rho) * cell
    spectrum *= np.sqrt(density) * np.sqrt(n0 * n1)
    spectrum[np.logical_or(rho < fmin, rho > fmax)] = 0.0
    screen = np.ma.masked_array(np.fft.ifft2(spectrum).real, mask=np.zeros((n0, n1)).astype(bool))
    screen += SR * (rough if rough is not None else 0.0)
    screen *= 2
    screen *= _unit_factor(units)
    return screen


# Grids from this size up build their PSD screens on the device (below it the host's two small FFTs cost nothing, and
# the host path is the one the reference's vectors pin bit for bit: tests/golden/r2_phase_maps.npz)
PSD_ON_DEVICE_FROM = 1024


def psd_on_device(dev, n):
    """Does ``dev`` build the screen itself?  complex128 contexts of the library, n >= PSD_ON_DEVICE_FROM."""
    return hasattr(dev, "psd_screen") and getattr(dev, "precision", "fp64") == "fp64" and n >= PSD_ON_DEVICE_FROM


class PsdScreen:
    """One ``WFO.psd`` call's screen before it is built: the two draws (taken when the call is planned, so the generator is
    consumed in the reference's order) and the call's arguments.  ``run`` builds it on the device when the context can
    (``DeviceFields.psd_screen``: complex128), on the host otherwise."""

    __slots__ = ("noise", "rough", "args", "params")

    def __init__(self, shape, dx, dy, A=10.0, B=0.0, C=0.0, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="m"):
        self.args = (shape, dx, dy, A, B, C, fknee, fmin, fmax, SR, units)
        self.params = psd_device_params(*self.args)  # (the reference's fmax check fires here, before the draws)
        self.noise, rough = psd_draws(shape)
        self.rough = rough if SR != 0.0 else None

    def host_map(self):
        return psd_map_from_draws(self.noise, self.rough, *self.args)


def psd_device_params(shape, dx, dy, A=10.0, B=0.0, C=0.0, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="m"):
    """The twelve numbers ``paos_psd_screen`` takes (include/paos_hip.h) for the screen :func:`psd_map` would build: the
    frequency steps as ``np.fft.fftfreq`` forms them, the band, the frequency-bin area from the same three corner values of
    ``rho`` and ``sqrt(n0 n1)`` -- each computed with the host path's own expressions, so the device multiplies by the same
    doubles."""
    n0, n1 = shape
    fmin, fmax = _psd_band(shape, dx, dy, fmin, fmax)
    fx, fy = np.fft.fftfreq(n1, dx)[:3], np.fft.fftfreq(n0, dy)[:3]
    rho = np.sqrt(np.add.outer(fy**2, fx**2))  # rho[:3, :3] of _radial_frequency (the nudged origin is not read)
    cell = (rho[0, 2] - rho[0, 1]) * (rho[2, 0] - rho[1, 0])
    return np.array([1.0 / (n1 * dx), 1.0 / (n0 * dy), A, B, C, fknee, fmin, fmax, cell, np.sqrt(n0 * n1), SR,
                     float(_unit_factor(units))], dtype=np.float64)


def psd_draws(shape):
    """The two white-noise arrays of one ``WFO.psd`` call, from NumPy's legacy global generator in the reference's order
    (psd.py:103 the screen, psd.py:148 the roughness -- drawn whatever SR is, like the reference)."""
    n0, n1 = shape
    return np.random.randn(n0, n1), np.random.randn(n0, n1)
