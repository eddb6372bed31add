"""Host half of the tabulated phase screens: ``WFO.grid_sag`` and ``WFO.psd``.

Both reference methods build an N x N wavefront-error map on the host and end with the same field
operation, ``u *= exp(2 pi i wfe / wl)`` (paos/classes/wfo.py:869-871 and :945-949).  The field
operation is a HIP kernel (``paos_phase_map``); this module restates the map construction with the
reference's own NumPy / SciPy calls in the reference's order, so the maps are bit-identical on the
same NumPy build:

* :func:`grid_sag_map` -- wfo.py:745-867: masking of non-finite / zero samples, the optional
  sub-pixel recentring (``scipy.ndimage.fourier_shift``), zero padding or cropping to the extent of
  the grid.  The two steps that call scikit-image (``rescale`` for an odd size difference or a
  different pixel scale, ``resize`` for a final shape mismatch; wfo.py:702-716, 739-750) are NOT
  restated -- scikit-image is not in this image, so there is nothing to pin them against -- and raise
  ``NotImplementedError``: maps must come at the pixel scale of the wavefront.
* :func:`psd_map` -- paos/classes/psd.py:100-160: a white-noise field from NumPy's legacy global
  generator (``np.random.randn`` exactly like the reference, so ``np.random.seed`` makes a run
  reproducible -- and comparable with the reference), filtered by the power-law PSD, plus surface
  roughness.  ``PSD.sfe_rms`` (a sympy integral, psd.py:166-193) only feeds a debug log line in
  the reference and is not evaluated.
"""
import numpy as np

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


def grid_sag_map(sag, nx, ny, delx, dely, xdec, ydec, shape, dx, dy):
    """The masked WFE map ``WFO.grid_sag`` applies and returns (wfo.py:745-867)."""
    assert sag.ndim == 2, "sag shall be a 2D array"
    assert sag.shape == (ny, nx)
    if not isinstance(sag, np.ma.MaskedArray):
        mask = ~np.isfinite(sag) | (sag == 0)
        sag = np.ma.MaskedArray(sag, mask=mask)
    mask = sag.mask.astype(float)
    sag = sag.filled(0.0)

    if (xdec != 0) or (ydec != 0):  # wfo.py:765-770
        from scipy.ndimage import fourier_shift

        sag = fourier_shift(np.fft.fft2(sag), shift=(-xdec, -ydec))
        sag = np.fft.ifft2(sag).real
        mask = fourier_shift(np.fft.fft2(mask), shift=(-xdec, -ydec))
        mask = np.fft.ifft2(mask).real

    target_width = shape[1] * dx
    target_height = shape[0] * dy
    current_width = sag.shape[1] * delx
    current_height = sag.shape[0] * dely
    width_diff = int(np.floor((current_width - target_width) / delx))
    height_diff = int(np.floor((current_height - target_height) / dely))
    if width_diff % 2 == 1 or height_diff % 2 == 1:
        raise NotImplementedError("grid_sag: an odd size difference needs skimage.transform.rescale "
                                  "(wfo.py:786-800), which is not restated")

    def pad_map(s, m, padding):
        return (np.pad(s, padding, mode="constant", constant_values=0),
                np.pad(m, padding, mode="constant", constant_values=1))

    if width_diff < 0.0:
        pad_width = abs(width_diff)
        pad_left = pad_width // 2
        sag, mask = pad_map(sag, mask, ((0, 0), (pad_left, pad_width - pad_left)))
    elif width_diff > 0.0:
        crop_left = width_diff // 2
        crop_right = sag.shape[1] - (width_diff - crop_left)
        sag = sag[:, crop_left:crop_right]
        mask = mask[:, crop_left:crop_right]
    if height_diff < 0.0:
        pad_height = abs(height_diff)
        pad_top = pad_height // 2
        sag, mask = pad_map(sag, mask, ((pad_top, pad_height - pad_top), (0, 0)))
    elif height_diff > 0.0:
        crop_top = height_diff // 2
        crop_bottom = sag.shape[0] - (height_diff - crop_top)
        sag = sag[crop_top:crop_bottom, :]
        mask = mask[crop_top:crop_bottom, :]

    if (delx / dx != 1) or (dely / dy != 1):
        raise NotImplementedError("grid_sag: a map at another pixel scale needs skimage.transform.rescale "
                                  "(wfo.py:845-849), which is not restated: resample the sag to the "
                                  "wavefront's dx, dy first")
    if sag.shape != tuple(shape):
        raise NotImplementedError("grid_sag: a residual shape mismatch needs skimage.transform.resize "
                                  "(wfo.py:855-859), which is not restated")
    mask = mask > 0.1
    return np.ma.MaskedArray(sag, mask=mask)


def psd_map(shape, dx, dy, A=10.0, B=0.0, C=0.0, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="m"):
    """The WFE map ``WFO.psd`` applies and returns (wfo.py:908-943 + psd.py:100-160)."""
    fx = np.fft.fftfreq(shape[1], dx)
    fy = np.fft.fftfreq(shape[0], dy)
    fxx, fyy = np.meshgrid(fx, fy)
    f = np.sqrt(fxx**2 + fyy**2)
    f[f == 0] = 1e-100
    f_nyq = 0.5 * np.sqrt(dx**-2 + dy**-2)
    if fmax is None:
        fmax = f_nyq
    else:
        assert fmax <= f_nyq, f"fmax must be less than or equal to f_Nyq ({f_nyq})"
    if fmin is None:
        fmin = 1 / (shape[0] * np.max([dx, dy]))

    nx_, ny_ = shape  # psd.py:103 names them Nx, Ny = pupil.shape
    wfe = np.random.randn(nx_, ny_)
    ft_wfe = np.fft.fft2(wfe)
    dfx = f[0, 2] - f[0, 1]
    dfy = f[2, 0] - f[1, 0]
    psd2d = A / (B + (f / fknee) ** C) / (2 * np.pi * f) * (dfx * dfy)
    ft_wfe *= np.sqrt(psd2d) * np.sqrt(nx_ * ny_)
    idx = np.logical_or(f < fmin, f > fmax)
    ft_wfe[idx] = 0.0
    wfe = np.fft.ifft2(ft_wfe).real
    wfe = np.ma.masked_array(wfe, mask=np.zeros((nx_, ny_)).astype(bool))
    wfe += SR * np.random.randn(nx_, ny_)
    wfe *= 2
    wfe *= _unit_factor(units)
    return wfe
