"""Round-5 GPU tests: state carried between the steps of a walked sweep at the headline shape, and the round's library
change of the parameter arena (fenced slabs) bit for bit.  (The one-line workgroups of the fused launches are covered by
tests/test_gpu_r4.py::test_two_passes_in_one_launch_change_no_bit[4096-fp64]: fused against single launches, bit for bit.)
Everything goes through the C ABI of libpaoship.so."""
import numpy as np
import pytest

from conftest import l2_rel_err, rel_err

pytestmark = pytest.mark.gpu

ON_AXIS = {"us": 0.0, "ut": 0.0}
PSF_TOL = 1.0e-10  # north star: |PSF_gpu - PSF_ref| / |PSF_ref| < 1e-10 (max-norm and L2-relative)


def test_second_walked_step_at_the_headline_shape_vs_oracle():
    """the timed region of bench.py is a WALKED sweep on one long-lived context -- aperture record
    sets kept between batches, PSF zeros known from the previous storing pass, rows that merely stand for zeros, the
    parameter arena's slabs reused.  Two consecutive steps issued exactly as ``bench.measure`` issues them (one
    ``DeviceFields(4096, 32)``, ``outputs=()``, ``keep_psf=True``, ``sync=False``, the next block of the 512-wavelength
    sweep each step), then items 0 and 31 of the SECOND step against the oracle: PSF max-norm and L2-relative < 1e-10,
    the power of the saved last surface against the oracle's sum."""
    from oracle.run_np import run as oracle_run
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    n, nb = 4096, 32
    dev = _lib.DeviceFields(n, nb)
    try:
        chains = [syn20_chain() for _ in range(nb)]
        res = None
        sets0 = dev.record_set_stats()
        for g in (0, 1):
            if res is not None:  # bench.measure: the tickets of a step nobody reads are given back unsynchronised
                for t in {rec["power_ticket"] for r in res for rec in r.values() if "power_ticket" in rec}:
                    dev.norm2_release(t)
            wls = [syn20_wavelength((g * nb + i) % 512) for i in range(nb)]
            res = run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=(), dev=dev, sync=False, keep_psf=True)
        found, rendered = (b - a for a, b in zip(sets0, dev.record_set_stats()))
        assert rendered >= 6  # (the second step renders its own aperture records: the walk changes the sampling)
        last = max(res[0])
        powers = dev.norm2_fetch(res[0][last]["power_ticket"])  # one reduction answers for every item
        worst = 0.0
        for i in (0, nb - 1):
            psf = dev.psf_fetch(i)
            ref = oracle_run(1.0, wls[i], n, 4, ON_AXIS, syn20_chain(), light=True)[last]
            want = ref["amplitude"] ** 2
            e, e2 = rel_err(psf, want), l2_rel_err(psf, want)
            assert e < PSF_TOL and e2 < PSF_TOL, (i, e, e2)
            assert abs(powers[i] - want.sum()) <= 1e-12 * want.sum(), (i, powers[i], want.sum())
            for key in ("dx", "dy", "wl", "fratio", "wz", "distancetofocus", "propagator"):
                assert res[i][last][key] == ref[key], (i, key)
            worst = max(worst, e)
        print(f"second walked step at 4096^2 x 32 (items 0, 31) PSF error vs oracle: {worst:.2e}; record sets found {found} rendered {rendered}")
    finally:
        dev.close()


def test_many_steps_run_ahead_of_the_gpu_without_corrupting_parameters():
    """The parameter arena is four slabs fenced by events (round 5): the host enqueues several steps before the GPU has
    run the first -- every step's parameter blocks must still be intact when its launches run.  Twelve unsynchronised
    steps of a 256-wavefront batch at 1024^2 (each needs most of a slab, so every slab is refilled three times), every
    step with its own wavelengths; afterwards the LAST step's PSFs equal those of the same step run alone on a fresh
    context, bit for bit, and the powers of an early step fetched late are that step's."""
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    n, nb, steps = 1024, 256, 12
    chains = [syn20_chain() for _ in range(nb)]
    wl_of = lambda g: [syn20_wavelength((g * nb + 7 * i) % 512) for i in range(nb)]  # noqa: E731
    dev = _lib.DeviceFields(n, nb)
    try:
        kept = None
        for g in range(steps):
            res = run_batch(1.0, wl_of(g), n, 4, ON_AXIS, chains, outputs=(), dev=dev, sync=False, keep_psf=True)
            tickets = {rec["power_ticket"] for r in res for rec in r.values() if "power_ticket" in rec}
            if g == 1:
                kept = (res, tickets)  # fetched after the run
            else:
                for t in tickets:
                    dev.norm2_release(t)
        last = max(kept[0][0])
        early_powers = dev.norm2_fetch(kept[0][0][last]["power_ticket"]).copy()
        got = [dev.psf_fetch(i) for i in (0, 100, nb - 1)]
    finally:
        dev.close()
    fresh = _lib.DeviceFields(n, nb)
    try:
        run_batch(1.0, wl_of(steps - 1), n, 4, ON_AXIS, chains, outputs=(), dev=fresh, sync=True, keep_psf=True)
        for k, i in enumerate((0, 100, nb - 1)):
            assert np.array_equal(got[k], fresh.psf_fetch(i)), i
        alone = run_batch(1.0, wl_of(1), n, 4, ON_AXIS, chains, outputs=(), dev=fresh, sync=True, keep_psf=True)
        assert np.array_equal(early_powers, np.array([alone[i][last]["power"] for i in range(nb)]))
    finally:
        fresh.close()


@pytest.mark.parametrize("n", [1024, 4096])
def test_start_box_on_a_poisoned_buffer(n):
    """Round 5: the lean start writes the first field inside its aperture's bounding box only -- rows AND
    columns (paos_start_box) -- the first surface's power is summed over that box (paos_norm2_enqueue_box) and the first pass
    loads nothing else.  On a buffer poisoned with NaN everywhere: PSFs and the powers of both saved surfaces equal those of
    whole-row starts (PAOS_START_BOX=0) on a clean context bit for bit, and a run that needs the whole field first (a
    stop on the second surface: paos_zero_outside_box) agrees with the ordinary walk."""
    import paos_amd.run as prun
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    wls = [syn20_wavelength(k) for k in (7, 333)]
    chains = [syn20_chain() for _ in wls]
    assert prun.START_BOX is True
    prun.START_BOX = False
    try:
        ref_dev = _lib.DeviceFields(n, len(wls))
        try:
            whole = run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=(), dev=ref_dev, keep_psf=True)
            ref_psf = [ref_dev.psf_fetch(i) for i in range(len(wls))]
        finally:
            ref_dev.close()
    finally:
        prun.START_BOX = True
    dev = _lib.DeviceFields(n, len(wls))
    try:
        for i in range(len(wls)):
            dev.upload(i, np.full((n, n), complex(np.nan, np.nan)))
        box = run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=(), dev=dev, keep_psf=True)
        for i in range(len(wls)):
            assert np.array_equal(dev.psf_fetch(i), ref_psf[i]), i
            for k in whole[i]:
                assert box[i][k]["power"] == whole[i][k]["power"], (i, k)  # zeros add nothing: the same sum bit for bit
        # something needs the whole field before any program runs: the box is made real zeros first
        chain = syn20_chain()
        chain[2] = dict(chain[2], is_stop=True)
        for i in range(len(wls)):
            dev.upload(i, np.full((n, n), complex(np.nan, np.nan)))
        got = run_batch(1.0, wls, n, 4, ON_AXIS, [chain, chain], outputs=(), dev=dev, keep_psf=True)
        psf = [dev.psf_fetch(i) for i in range(len(wls))]
        want = run_batch(1.0, wls, n, 4, ON_AXIS, [chain, chain], outputs=("psf",), dev=dev, keep_psf=True)
        for i in range(len(wls)):
            assert np.isfinite(psf[i]).all() and rel_err(psf[i], want[i][20]["psf"]) < 1e-13, i
            assert abs(got[i][20]["power"] - want[i][20]["power"]) <= 1e-13 * want[i][20]["power"]
    finally:
        dev.close()


def test_a_shared_grid_sag_screen_is_uploaded_once_and_equals_the_per_item_path():
    """Round 5: SYN20 behind ONE white-noise grid-sag screen for all items (bench.py's `extra.dense`
    chain): the map is built once (``run._sag_map_once``) and applied by ``paos_phase_map_items`` -- one upload for the
    batch, none for the next batch.  PSFs equal those of the per-item path (every item its own copy of the array ->
    ``paos_phase_map`` per item) bit for bit, on the same context and on the next step."""
    import bench_extras
    from paos_amd import _lib
    from paos_amd.chains import syn20_wavelength
    from paos_amd.run import run_batch

    n, nb = 1024, 4
    shared = bench_extras.dense_chain(n)
    own = []
    for _ in range(nb):
        c = bench_extras.dense_chain(n)
        own.append({k: (dict(v, grid_sag=v["grid_sag"].copy()) if v["type"] == "Grid Sag" else v) for k, v in c.items()})
    dev = _lib.DeviceFields(n, nb)
    try:
        for g in (0, 1):
            wls = [syn20_wavelength((g * nb + i) % 512) for i in range(nb)]
            run_batch(1.0, wls, n, 4, ON_AXIS, [shared] * nb, outputs=(), dev=dev, keep_psf=True)
            a = [dev.psf_fetch(i) for i in range(nb)]
            run_batch(1.0, wls, n, 4, ON_AXIS, own, outputs=(), dev=dev, keep_psf=True)
            for i in range(nb):
                assert np.array_equal(a[i], dev.psf_fetch(i)), (g, i)
    finally:
        dev.close()


def test_psd_screen_built_on_the_device_vs_reference_vectors():
    """Round 5: ``paos_psd_screen`` -- fft2(noise) -> power-law filter -> ifft2 -> roughness on the
    library's own passes -- against the reference's vectors (tests/golden/r2_phase_maps.npz; the host path reproduces them
    bit for bit, test_gpu_r2.py): the maps to 1e-13 of their peak, the fields behind them to 1e-12."""
    from conftest import load_golden
    from paos_amd import phase_maps
    from paos_amd.wfo import WFO

    g = load_golden("r2_phase_maps.npz")
    cases = {"powerlaw": dict(A=7.0, B=0.0, C=1.5, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="nm"),
             "knee_sr": dict(A=12.0, B=1.0, C=2.2, fknee=3.0, fmin=0.5, fmax=6.0, SR=2.0, units="nm")}
    keep = phase_maps.PSD_ON_DEVICE_FROM
    try:
        phase_maps.PSD_ON_DEVICE_FROM = 64
        for tag, kw in cases.items():
            for anam in (False, True):
                w = WFO(1.0, 2.0e-6, 64, 4)
                if anam:
                    w.Magnification(1.25, 0.8)
                w._wfo = g["u0"]
                np.random.seed(1234)
                ret = w.psd(**kw)
                key = f"psd_{tag}{'_anam' if anam else ''}"
                want = g[key + "_wfe"]
                assert isinstance(ret, np.ma.MaskedArray) and not np.ma.getmaskarray(ret).any()
                assert np.abs(np.ma.filled(ret, 0.0) - want).max() <= 1e-13 * np.abs(want).max(), key
                assert rel_err(w.wfo, g[key + "_u"]) < 1e-12, key
    finally:
        phase_maps.PSD_ON_DEVICE_FROM = keep


@pytest.mark.parametrize("n", [1024, 4096])
def test_psd_screen_on_the_device_equals_the_host_path(n):
    """... and at the sizes that take the device path by default, against the host path (NumPy's FFTs) on the same draws:
    maps to 1e-13 of their peak, the generator left in the same state, the kept map applied to two items of a batch with
    their own wavelengths; the error paths of the two entry points."""
    from paos_amd import _lib, phase_maps

    kw = dict(A=12.0, B=1.0, C=2.2, fknee=3.0, fmin=None, fmax=None, SR=0.5, units="nm")
    dx = dy = 4.0 / n
    np.random.seed(5)
    screen = phase_maps.PsdScreen((n, n), dx, dy, **kw)
    state = np.random.get_state()[1].copy()
    np.random.seed(5)
    want = np.ma.filled(phase_maps.psd_map((n, n), dx, dy, **kw), 0.0)
    assert np.array_equal(state, np.random.get_state()[1])
    dev = _lib.DeviceFields(n, 2)
    try:
        assert phase_maps.psd_on_device(dev, n)
        dev.fill(1.0)
        got = dev.psd_screen(screen.noise, screen.rough, screen.params, key=41, want_map=True)
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
        wls = [1.0e-6, 1.7e-6]
        dev.phase_map_items(None, [0, 1], wls, key=41)
        for i, wl in enumerate(wls):
            assert rel_err(dev.download(i), np.exp(2j * np.pi * want / wl)) < 1e-12, i
        with pytest.raises(_lib.PaosHipError, match="no map kept"):
            dev.phase_map_items(None, [0], [1.0e-6], key=42)
        with pytest.raises(_lib.PaosHipError):
            dev.psd_screen(screen.noise, None, screen.params, key=0)
        bad = screen.params.copy()
        bad[2] = np.nan
        with pytest.raises(_lib.PaosHipError, match="NaN"):
            dev.psd_screen(screen.noise, None, bad, key=43)
        inf = screen.noise.copy()
        inf[3, 5] = np.inf
        with pytest.raises(_lib.PaosHipError, match="non-finite"):
            dev.psd_screen(inf, None, screen.params, key=44)
        with pytest.raises(_lib.PaosHipError, match="no map kept"):  # ... and a failed build leaves no map behind
            dev.phase_map_items(None, [0], [1.0e-6], key=44)
    finally:
        dev.close()
    if n == 1024:
        f32 = _lib.DeviceFields(n, 1, precision="fp32")
        try:
            assert not phase_maps.psd_on_device(f32, n)
            with pytest.raises(_lib.PaosHipError, match="complex128"):
                f32.psd_screen(screen.noise, None, screen.params, key=45)
        finally:
            f32.close()


def test_run_with_a_psd_surface_builds_the_screen_on_the_device():
    """``run()`` of SYN20 behind a PSD surface at 1024^2: the surface's `wfe` and every saved field equal those of the
    host-built screen (same seed) to 1e-13 / 1e-11."""
    from paos_amd import phase_maps
    from paos_amd.abcd import ABCD
    from paos_amd.chains import syn20_chain
    from paos_amd.run import run

    n = 1024
    chain = {}
    for item in syn20_chain().values():
        num = len(chain) + 1
        chain[num] = dict(item, num=num, save=True)
        if item["name"] == "Z1":
            num = len(chain) + 1
            chain[num] = dict(A=9.0, B=0.0, C=1.8, fknee=1.0, fmin=None, fmax=None, SR=0.3, units="nm", num=num, type="PSD",
                              name="SCREEN", is_stop=False, save=True, ABCDt=ABCD(thickness=0.0, curvature=0.0),
                              ABCDs=ABCD(thickness=0.0, curvature=0.0))
    key = [k for k, v in chain.items() if v["name"] == "SCREEN"][0]
    keep = phase_maps.PSD_ON_DEVICE_FROM
    try:
        np.random.seed(9)
        got = run(1.0, 1.1e-6, n, 4, ON_AXIS, chain)
        phase_maps.PSD_ON_DEVICE_FROM = 1 << 20
        np.random.seed(9)
        want = run(1.0, 1.1e-6, n, 4, ON_AXIS, chain)
    finally:
        phase_maps.PSD_ON_DEVICE_FROM = keep
    a, b = np.ma.filled(got[key]["wfe"], 0.0), np.ma.filled(want[key]["wfe"], 0.0)
    assert np.abs(b).max() > 0 and np.abs(a - b).max() <= 1e-13 * np.abs(b).max()
    for k in want:
        assert rel_err(got[k]["wfo"], want[k]["wfo"]) < 1e-11, k


@pytest.mark.parametrize("n", [1024, 4096])
def test_zernike_like_equals_zernike_bit_for_bit(n):
    """Round 5: ``paos_zernike_like`` -- items that the caller knows to hold copies of one field (the surface right behind
    the start of a sweep) and that share their wfe map are served by one load per pixel.  Eleven items in three start
    groups (two apertures, one of them with and without the stop) and two coefficient sets -- so some wfe groups are twins
    throughout, one is mixed and takes the ordinary loads: fields bit for bit those of ``paos_zernike``; a hint that
    names no item raises."""
    from paos_amd import _lib
    from paos_amd.aperture import make_aperture
    from paos_amd.planner import jacobi_recurrence, zernike_block
    from paos_amd.zernike import Zernike, norm_factors

    nb, k = 11, 15
    dx = 4.0 / n
    rng = np.random.default_rng(n)
    m, nn = Zernike.j2mn(k, "noll")
    norm = norm_factors(m, nn, True)
    coef = [rng.standard_normal(k) * 40e-9, rng.standard_normal(k) * 25e-9]
    aps = [make_aperture(n, dx, dx, 0.0, 0.0, hx=0.5, hy=0.5, shape="elliptical"),
           make_aperture(n, dx, dx, 0.01, -0.02, hx=0.45, hy=0.4, shape="elliptical")]
    start_group = [0, 0, 0, 0, 1, 1, 1, 2, 2, 0, 1]          # (aperture, stop) -> three distinct fields
    which_ap = {0: 0, 1: 1, 2: 1}
    stop_of = {0: 1.0, 1: 1.0, 2: 0.0}
    coef_of = [0, 0, 0, 1, 0, 0, 1, 1, 1, 0, 0]               # wfe groups: items with coefficient set 0 / 1
    wls = [0.8e-6 + 0.1e-6 * i for i in range(nb)]
    blocks = []
    nmax = kdim = None
    for i in range(nb):
        b, nmax, kdim = zernike_block(m, nn, norm, coef[coef_of[i]], dx, dx, 0.5, wls[i])
        blocks.append(b)
    table = jacobi_recurrence(nmax)
    same_as = [start_group.index(g) for g in start_group]
    dev = _lib.DeviceFields(n, nb)
    try:
        out = {}
        for mode in ("plain", "like"):
            dev.start(1.0, _lib.SHAPE_ELLIPSE, [aps[which_ap[g]].block() for g in start_group], [stop_of[g] for g in start_group])
            if mode == "like":
                dev.zernike(nmax, kdim, table, blocks, same_as=same_as)
            else:
                dev.zernike(nmax, kdim, table, blocks)
            out[mode] = [dev.download(i) for i in range(nb)]
        for i in range(nb):
            assert np.array_equal(out["plain"][i], out["like"][i]), i
            assert not np.array_equal(out["plain"][i], out["plain"][(i + 1) % nb])
        with pytest.raises(_lib.PaosHipError, match="item indices"):
            dev.zernike(nmax, kdim, table, blocks, same_as=[nb] * nb)
    finally:
        dev.close()


def test_start_power_sums_are_found_again_only_for_the_same_start():
    """Round 5: the power sums of a start (exact pixel overlaps of the aperture, per distinct aperture of the batch) are kept
    with everything they depend on; the next start with the same shape, constant, records and stop flags copies them.  A
    second start equals the first bit for bit, a start with another aperture / other flags / another constant equals that of
    a fresh context, and so does the first one again afterwards."""
    from paos_amd import _lib
    from paos_amd.aperture import make_aperture

    n, nb = 1024, 3
    dx = 4.0 / n
    a1 = [make_aperture(n, dx, dx, 0.0, 0.0, hx=0.5, hy=0.5, shape="elliptical").block() for _ in range(nb)]
    a2 = [make_aperture(n, dx, dx, 0.0, 0.01 * i, hx=0.45, hy=0.4, shape="elliptical").block() for i in range(nb)]
    cases = [(1.0, a1, [1.0] * nb), (1.0, a1, [1.0] * nb), (1.0, a2, [1.0] * nb), (1.0, a2, [1.0, 0.0, 1.0]), (0.5 + 0.25j, a2, [1.0, 0.0, 1.0]),
             (1.0, a1, [1.0] * nb)]

    def fields(dev, case):
        value, aps, stops = case
        dev.start(value, _lib.SHAPE_ELLIPSE, aps, stops)
        return [dev.download(i) for i in range(nb)]

    dev = _lib.DeviceFields(n, nb)
    try:
        for k, case in enumerate(cases):
            got = fields(dev, case)
            fresh = _lib.DeviceFields(n, nb)
            try:
                want = fields(fresh, case)
            finally:
                fresh.close()
            for i in range(nb):
                assert np.array_equal(got[i], want[i]), (k, i)
            p = np.array([np.sum(np.abs(g) ** 2) for g in got])
            assert np.allclose(p[np.array(case[2]) != 0.0], 1.0, rtol=1e-12)
    finally:
        dev.close()


@pytest.mark.parametrize("n", [1024, 4096])
def test_rectangle_records_rendered_64_lines_at_a_time_equal_the_one_line_renderer(n):
    """Round 5: the line records of a rectangular aperture (a field stop, a slit) are rendered 64 lines per wave from ONE
    column profile (pointwise.h: mask_rect_block_render); PAOS_MASK_RECT_BLOCKS=0 keeps the one-line-per-wave renderer,
    which scans the profile for every line.  Random rectangles -- wide and narrow, centred and near the edge of the grid,
    partly off it -- riding on a row pass and on a column pass over a random field: the same field bit for bit, and what
    comes back is the masked field."""
    import os

    from paos_amd import _lib

    rng = np.random.default_rng(11 * n)
    u = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    dev = _lib.DeviceFields(n, 1, "fp64")
    try:
        for trial in range(8):
            w, h = rng.uniform(3.0, 0.8 * n, 2)
            if trial == 3:
                h = 2.4  # a slit a few pixels high
            xc, yc = (n / 2 + rng.uniform(-3, 3, 2)) if trial % 2 == 0 else rng.uniform(0.05 * n, 0.95 * n, 2)
            blocks = np.array([[[1.0, xc, yc, w, h]], [[0.0, 0.0, 32.0, float(_lib.SHAPE_RECT), 0.0]], [[1.0, 0.0, 0.0, 0.0, 0.0]]])
            got = {}
            for mode, env in (("blocks", {}), ("lines", {"PAOS_MASK_RECT_BLOCKS": "0"})):
                os.environ.update(env)
                try:
                    out = []
                    rendered = dev.record_set_stats()[1]
                    for axis in (0, 1):
                        dev.upload(0, u)
                        dev.run_passes([{"axis": axis, "fft1": 2, "fft2": -1, "pre": [(_lib.PW_MASK, 0, 0)], "mid": [], "post": []}], blocks)
                        out.append(dev.download(0))
                    assert dev.record_set_stats()[1] == rendered + 2, mode
                    got[mode] = out
                finally:
                    for k in env:
                        os.environ.pop(k, None)
            for x, y in zip(got["blocks"], got["lines"]):
                assert np.array_equal(x, y), (trial, w, h, xc, yc)
            back = np.fft.ifft(got["blocks"][0], axis=1)
            yy, xx = np.mgrid[0:n, 0:n]
            outside = (np.abs(xx - xc) > w / 2 + 1.5) | (np.abs(yy - yc) > h / 2 + 1.5)
            inside = (np.abs(xx - xc) < w / 2 - 1.5) & (np.abs(yy - yc) < h / 2 - 1.5)
            assert (np.abs(back[outside]) < 1e-9).all() and np.allclose(back[inside], u[inside], rtol=0, atol=1e-9)
    finally:
        dev.close()
