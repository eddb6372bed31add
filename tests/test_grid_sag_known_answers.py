"""Known answers for the scikit-image branches of ``WFO.grid_sag`` (paos/classes/wfo.py:697-716, 739-750, 786-806, 845-859),
held by the reference itself: the recorded cell outputs of ``notebook/ComputeGridSag.ipynb`` (executed by the authors with
scikit-image 0.24.0, which is neither under /root/reference nor in this image).

* cell 7:  ``WFO(1.1, 0.55e-6, 1024, 4)`` -> elliptical aperture -> stop -> ``zernikes(arange(6), [0, 10, 0, -30, 20, 0] nm,
  "noll", True, 0.55)``: ``WFE RMS = 3.7394904478041395e-08`` (already pinned bit-exactly elsewhere; the input of what follows).
* cell 10: ``skimage.transform.rescale(wfe, (0.73901, 1.11977), anti_aliasing=True, order=3)``:
  ``Sum of wfe: 1.9025694787001765e-06``; with the rescaled mask (> 0.5): ``Sum of wfe after mask: 3.9501582313044445e-06``.
* cells 10-12: roll by (-101, 79), pads ((0, 0), (3, 8)) and ((70, 11), (0, 0)) -> ``sag.npy`` of shape (838, 1158)
  (cell 12), ``WFE RMS: 37.30 nm``, ``WFE PV: 168.44 nm`` (cell 15); nx, ny, delx, dely, xdec, ydec as logged in cell 22.
* cell 17 (``WFO.grid_sag`` on that file): the "sample more finely" path, logged shapes (838, 1158) -> (1676, 2316) ->
  (1514, 2294) -> (1024, 1024); cells 18-19: returned shape (1024, 1024), ``WFE RMS: 37.12 nm``, ``WFE PV: 177.68 nm``.

``paos_amd.phase_maps._ski_resize`` restates scikit-image's published algorithm on scipy.ndimage; these numbers pin it (and
the order of steps of ``grid_sag_map``) to the reference's own run.  The Zernike map comes from the oracle -- this is a test.
"""
import numpy as np
import pytest

# the notebook's recorded outputs (cells 7, 10, 12, 15, 17-19, 22)
KAT_ZERNIKE_STD = 3.7394904478041395e-08
KAT_SUM_RESCALED = 1.9025694787001765e-06
KAT_SUM_RESCALED_MASKED = 3.9501582313044445e-06
KAT_SAG_SHAPE = (838, 1158)
KAT_SAG_RMS_NM, KAT_SAG_PV_NM = "37.30", "168.44"
KAT_DELX, KAT_DELY = 0.003837283549300303, 0.0058143665173678305
KAT_XDEC, KAT_YDEC = 108.5, -103.5
KAT_SHAPE_TRAIL = [(838, 1158), (1676, 2316), (1514, 2294), (1024, 1024)]
KAT_OUT_RMS_NM, KAT_OUT_PV_NM = "37.12", "177.68"
SCALE_X, SCALE_Y = 1.11977, 0.73901
BEAM, WL, GRID, ZOOM = 1.1, 0.55e-6, 1024, 4
SEMIMAJ, SEMIMIN = BEAM / 2, BEAM / 2 / 1.1 * 0.73


def _zernike_map():
    """Cell 7 on the oracle: the masked WFE map and the wavefront's sampling."""
    from oracle.pop_numpy import RefWFO

    wfo = RefWFO(BEAM, WL, GRID, ZOOM)
    wfo.aperture(xc=0.0, yc=0.0, hx=SEMIMAJ, hy=SEMIMIN, shape="elliptical")
    wfo.make_stop()
    wfe = wfo.zernikes(np.arange(0, 6), np.array([0, 10, 0, -30.0, 20.0, 0.0]) * 1.0e-9, "noll", True, 0.55)
    return wfe, wfo.dx, wfo.dy


def _pad_map(sag, mask, padding):  # the notebook's helper (cell 10)
    return (np.pad(sag, padding, mode="constant", constant_values=0),
            np.pad(mask, padding, mode="constant", constant_values=1))


@pytest.fixture(scope="module")
def notebook_sag():
    """Cells 10-12: the masked array the notebook saves as ``sag.npy`` (metres), plus the intermediate sums."""
    from paos_amd.phase_maps import _ski_rescale

    wfe, dx, dy = _zernike_map()
    assert np.std(wfe) == KAT_ZERNIKE_STD
    mask = _ski_rescale(wfe.mask.astype(float), SCALE_X, SCALE_Y) > 0.5
    rescaled = _ski_rescale(wfe, SCALE_X, SCALE_Y)
    masked = np.ma.MaskedArray(rescaled, mask=mask)
    sums = (float(np.sum(rescaled)), float(np.sum(masked)))
    out = np.roll(masked, (-101, 79), axis=(1, 0))
    a, mk = _pad_map(out, out.mask, ((0, 0), (3, 8)))
    out = np.ma.masked_array(a, mask=mk)
    a, mk = _pad_map(out, out.mask, ((70, 11), (0, 0)))
    out = np.ma.masked_array(a, mask=mk)
    data_nm = out * 1.0e9          # what the file holds (cell 12) ...
    return {"sag": data_nm * 1.0e-9, "sums": sums, "dx": dx, "dy": dy}  # ... and what cell 17 passes on


def test_ski_rescale_reproduces_the_notebooks_sums(notebook_sag):
    """cell 10: anti-aliased cubic down- / up-scaling by (0.73901, 1.11977)."""
    total, masked = notebook_sag["sums"]
    assert abs(total - KAT_SUM_RESCALED) <= 1.0e-12 * abs(KAT_SUM_RESCALED), total  # measured: 1.2e-14
    assert masked == KAT_SUM_RESCALED_MASKED  # bit-exact: the masked part holds the samples that matter


def test_the_notebooks_sag_file_is_rebuilt(notebook_sag):
    """cells 12, 15, 22: shape, statistics and sampling of ``sag.npy`` to the printed digits."""
    sag = notebook_sag["sag"]
    assert sag.shape == KAT_SAG_SHAPE
    assert f"{1e9 * np.std(sag):.2f}" == KAT_SAG_RMS_NM and f"{1e9 * np.ptp(sag):.2f}" == KAT_SAG_PV_NM
    assert notebook_sag["dx"] / SCALE_X == KAT_DELX and notebook_sag["dy"] / SCALE_Y == KAT_DELY


def test_grid_sag_map_refine_crop_rescale_path(notebook_sag, monkeypatch):
    """cells 17-19: ``WFO.grid_sag`` on the rebuilt file -- odd overhang on both axes, so the map is sampled twice as
    finely (wfo.py:786-806), cropped on width and height (:817-843) and rescaled to the wavefront's pixels (:845-854);
    shapes exact, RMS / PV to the printed digits."""
    import paos_amd.phase_maps as pm

    trail = []
    real = pm._ski_resize

    def spy(image, output_shape, anti_aliasing):
        out = real(image, output_shape, anti_aliasing)
        trail.append((tuple(np.shape(image)), tuple(out.shape), bool(anti_aliasing)))
        return out

    monkeypatch.setattr(pm, "_ski_resize", spy)
    sag = notebook_sag["sag"]
    out = pm.grid_sag_map(sag, sag.shape[1], sag.shape[0], KAT_DELX, KAT_DELY, KAT_XDEC, KAT_YDEC, (GRID, GRID),
                          notebook_sag["dx"], notebook_sag["dy"])
    # heights and mask each: (838, 1158) -> (1676, 2316) without anti-aliasing, (1514, 2294) -> (1024, 1024) with
    assert trail == [(KAT_SHAPE_TRAIL[0], KAT_SHAPE_TRAIL[1], False)] * 2 + [(KAT_SHAPE_TRAIL[2], KAT_SHAPE_TRAIL[3], True)] * 2
    assert out.shape == KAT_SHAPE_TRAIL[3] and isinstance(out, np.ma.MaskedArray)
    assert f"{1e9 * np.std(out):.2f}" == KAT_OUT_RMS_NM and f"{1e9 * np.ptp(out):.2f}" == KAT_OUT_PV_NM


@pytest.mark.gpu
def test_wfo_grid_sag_on_the_device_with_the_notebooks_file(notebook_sag):
    """cell 17 through the drop-in ``WFO`` on the GPU: the returned map carries the notebook's numbers and the field is
    the stopped aperture times ``exp(2 pi i sag.filled(0) / wl)`` (wfo.py:869-871) -- ``paos_phase_map``."""
    from oracle.pop_numpy import RefWFO
    from paos_amd.wfo import WFO

    sag = notebook_sag["sag"]
    w = WFO(BEAM, WL, GRID, ZOOM)
    w.aperture(xc=0.0, yc=0.0, hx=SEMIMAJ, hy=SEMIMIN, shape="elliptical")
    w.make_stop()
    out = w.grid_sag(sag, sag.shape[1], sag.shape[0], KAT_DELX, KAT_DELY, KAT_XDEC, KAT_YDEC)
    assert out.shape == (GRID, GRID)
    assert f"{1e9 * np.std(out):.2f}" == KAT_OUT_RMS_NM and f"{1e9 * np.ptp(out):.2f}" == KAT_OUT_PV_NM
    ref = RefWFO(BEAM, WL, GRID, ZOOM)
    ref.aperture(xc=0.0, yc=0.0, hx=SEMIMAJ, hy=SEMIMIN, shape="elliptical")
    ref.make_stop()
    want = ref.wfo * np.exp(2.0 * np.pi * 1j * out.filled(0) / WL)
    got = np.asarray(w.wfo)
    assert np.max(np.abs(got - want)) <= 1.0e-13 * np.max(np.abs(want))
