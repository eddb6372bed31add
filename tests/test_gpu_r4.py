"""GPU tests added in round 4: the separable pass programs (passes.py: SeparableCompiler) and the two-axis planner
behind them (csrc/paos_hip.hip: plan_pruning).

* SYN20 at 4096^2 (the benchmark's size) from the separable programs against the operator-by-operator ones -- which
  tests/test_gpu_r3.py and tests/test_gpu.py pin to the oracle at that size -- fp64 and fp32, and against the oracle
  directly at 2048^2.
* What the plan has a SYN20 step move: every launch skips the tiles of dead or unwanted lines, the step's planned bytes
  are an eighth of the dense ones, and with the pruning off every launch is a full pass again with the same PSF.
* A field buffer poisoned with NaN wherever the programs are not supposed to look (every stretch, not only the start).
* The half checkerboards (PAOS_PWF_X_ONLY / PAOS_PWF_Y_ONLY) through the C ABI on the frugal AND the generic kernels.
* Two or three consecutive passes of a row / column chain in one launch (LONG builds) against one launch per pass: bit-identical,
  complex128 and complex64 (whose fused launches read from tables the factors single passes evaluate).
* Aperture line records through two boundary windows against the chunk scan: bit-identical fields for random ellipses.
"""
import numpy as np
import pytest

from conftest import l2_rel_err, rel_err

pytestmark = pytest.mark.gpu

ON_AXIS = {"us": 0.0, "ut": 0.0}


def _by_operator(fn):
    import paos_amd.passes as ppasses

    assert ppasses.SEPARABLE is True
    ppasses.SEPARABLE = False
    try:
        return fn()
    finally:
        ppasses.SEPARABLE = True


@pytest.mark.parametrize("n,precision,tol", [(4096, "fp64", 1e-12), (4096, "fp32", 2e-5), (1024, "fp64", 1e-12)])
def test_separable_programs_equal_operator_by_operator_at_the_benchmark_size(n, precision, tol):
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    wls = [syn20_wavelength(k) for k in (0, 31, 300)]
    chains = [syn20_chain() for _ in wls]
    st_sep, st_ref = {}, {}
    sep = run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=("psf",), precision=precision, stats=st_sep)
    ref = _by_operator(lambda: run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=("psf",), precision=precision, stats=st_ref))
    assert st_sep["fused_passes"] == st_ref["fused_passes"] == 24
    worst = 0.0
    for a, b in zip(sep, ref):
        for k in a:
            e, e2 = rel_err(a[k]["psf"], b[k]["psf"]), l2_rel_err(a[k]["psf"], b[k]["psf"])
            worst = max(worst, e, e2)
            assert e < tol and e2 < tol, (k, e, e2)
            assert abs(a[k]["power"] / b[k]["power"] - 1.0) < (1e-12 if precision == "fp64" else 1e-5)
    print(f"SYN20 {n}^2 {precision}: separable vs operator-by-operator PSF {worst:.2e}")


def test_separable_syn20_vs_oracle():
    from oracle.run_np import run as oracle_run
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    n, wl = 2048, syn20_wavelength(77)
    got = run_batch(1.0, [wl], n, 4, ON_AXIS, [syn20_chain()], outputs=("psf", "wfo"))
    ref = oracle_run(1.0, wl, n, 4, ON_AXIS, syn20_chain(), light=True)
    for k in ref:
        assert rel_err(got[0][k]["wfo"], ref[k]["wfo"]) < 1e-11 and l2_rel_err(got[0][k]["wfo"], ref[k]["wfo"]) < 1e-11
        assert rel_err(got[0][k]["psf"], ref[k]["amplitude"] ** 2) < 1e-10
        assert l2_rel_err(got[0][k]["psf"], ref[k]["amplitude"] ** 2) < 1e-10


def test_what_the_plan_has_a_syn20_step_move():
    """Every pass launch of a lean SYN20 step skips tiles (bit 0 of its tag); the planned bytes of the step are far
    below 24 dense passes (zoom 4: a quarter of the lines, a quarter to all of the positions); with the pruning off
    every launch is dense -- planned bytes = batch x grid x 32 B -- and the PSFs are the same bit for bit."""
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    n, nb = 2048, 2
    wls = [syn20_wavelength(k) for k in (3, 200)]
    chains = [syn20_chain() for _ in wls]
    dev = _lib.DeviceFields(n, nb, "fp64")
    try:
        def step():
            dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=256)
            run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=(), dev=dev, keep_psf=True)
            planned = dev.profile_planned_bytes()
            ms, tags = dev.profile_end_launches()
            return planned, tags, [dev.psf_fetch(i) for i in range(nb)]

        planned, tags, psf = step()
        dense = 2.0 * 16 * n * n * nb
        # (a launch may run two or three passes of a row / column chain -- bits 4 / 5 of its tag: 24 passes in fewer launches)
        assert tags.size == planned.size and tags.size + int(np.sum((tags & 16) != 0)) + 2 * int(np.sum((tags & 32) != 0)) == 24
        assert np.all(tags & 1), tags
        assert np.all(planned > 0.0) and np.all(planned < 0.5 * dense)
        assert planned.sum() < 0.2 * 24 * dense, planned.sum() / (24 * dense)
        dev.set_pruning(False)
        try:
            planned_d, tags_d, psf_d = step()
        finally:
            dev.set_pruning(True)
        assert tags_d.size == 24 and np.all(tags_d[:-1] == 0) and np.all(planned_d[:-1] == dense)
        for a, b in zip(psf, psf_d):
            assert np.array_equal(a, b)
    finally:
        dev.close()


def test_nothing_outside_the_plan_is_ever_read():
    """The field buffer is poisoned with NaN between two steps of a sweep on one context, wherever the next step's
    programs are not supposed to look: the start writes the live rows only, every later pass loads what the pass in
    front of it stored.  The PSFs are those of a fresh context, bit for bit."""
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    n = 1024
    wls = [syn20_wavelength(k) for k in (5, 400)]
    chains = [syn20_chain() for _ in wls]
    fresh = run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=("psf",))
    dev = _lib.DeviceFields(n, len(wls), "fp64")
    try:
        run_batch(1.0, [syn20_wavelength(9), syn20_wavelength(10)], n, 4, ON_AXIS, chains, outputs=(), dev=dev, keep_psf=True)
        for i in range(len(wls)):
            dev.upload(i, np.full((n, n), complex(np.nan, np.nan)))
        run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=(), dev=dev, keep_psf=True)
        for i in range(len(wls)):
            assert np.array_equal(dev.psf_fetch(i), fresh[i][20]["psf"]), i
    finally:
        dev.close()


@pytest.mark.parametrize("n", [256, 1024])
def test_half_checkerboards_through_the_c_abi(n):
    """PAOS_PW_SIGN with PAOS_PWF_X_ONLY / PAOS_PWF_Y_ONLY in front of a transform, between two and in a stand-alone
    pass (axis -1), against NumPy: 256^2 runs on the generic kernels, 1024^2 on the frugal ones."""
    from paos_amd import _lib

    rng = np.random.default_rng(n)
    u = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    i = np.arange(n)
    sx, sy = np.where(i[None, :] & 1, -1.0, 1.0), np.where(i[:, None] & 1, -1.0, 1.0)
    on = np.array([[[1.0, 0.0, 0.0, 0.0, 0.0]], [[1.0, 0.0, 0.0, 0.0, 0.0]]])  # block 0: the signs' enable, block 1: fft control (forward)
    X, Y = (_lib.PW_SIGN, _lib.PWF_X_ONLY, 0), (_lib.PW_SIGN, _lib.PWF_Y_ONLY, 0)
    dev = _lib.DeviceFields(n, 1, "fp64")
    try:
        for axis in (0, 1):
            ax = 1 if axis == 0 else 0
            dev.upload(0, u)
            dev.run_passes([{"axis": axis, "fft1": 1, "fft2": 1, "pre": [X], "mid": [Y, X, X], "post": []}], on)
            want = np.fft.fft(np.fft.fft(u * sx, axis=ax) * sy, axis=ax)
            assert rel_err(dev.download(0), want) < 1e-13, axis
        dev.upload(0, u)
        dev.run_passes([{"axis": -1, "pre": [X]}, {"axis": -1, "pre": [Y, (_lib.PW_SIGN, 0, 0)]}], on)
        assert np.array_equal(dev.download(0), u * sx * sy * (sx * sy))
    finally:
        dev.close()


@pytest.mark.parametrize("n,precision", [(1024, "fp64"), (4096, "fp64"), (2048, "fp32"), (4096, "fp32")])
def test_two_passes_in_one_launch_change_no_bit(n, precision):
    """Where two (or three) consecutive passes of a row / column chain allow it the library runs them in one launch (frugal_pass.h:
    LONG builds; PAOS_FUSE_PAIRS=0 switches it off): the tile stays in registers between them -- the same arithmetic in
    the same order, so the PSFs are equal bit for bit, in fewer launches."""
    import os

    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    wls = [syn20_wavelength(k) for k in (1, 255)]
    chains = [syn20_chain() for _ in wls]
    dev = _lib.DeviceFields(n, len(wls), precision)
    try:
        def step():
            dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=256)
            res = run_batch(1.0, wls, n, 4, ON_AXIS, chains, outputs=(), dev=dev, keep_psf=True, precision=precision)
            _, tags = dev.profile_end_launches()
            return tags, [dev.psf_fetch(i) for i in range(len(wls))], [r[20]["power"] for r in res]

        tags, psf, power = step()
        assert os.environ.get("PAOS_FUSE_PAIRS") is None
        os.environ["PAOS_FUSE_PAIRS"] = "0"
        try:
            tags0, psf0, power0 = step()
        finally:
            del os.environ["PAOS_FUSE_PAIRS"]
        assert tags0.size == 24 and not np.any(tags0 & 48)
        assert tags.size < 24 and tags.size + int(np.sum((tags & 16) != 0)) + 2 * int(np.sum((tags & 32) != 0)) == 24
        # (complex64 too: a fused launch reads from tables the very factors a single pass evaluates -- slot_factor32)
        for a, b in zip(psf, psf0):
            assert np.array_equal(a, b)
        assert power == power0
    finally:
        dev.close()


@pytest.mark.parametrize("n", [1024, 4096])
def test_two_window_aperture_records_equal_the_scan(n):
    """The aperture line records of an ellipse are rendered with one exact-overlap evaluation per line where the two
    boundary runs of the line fit two 32-pixel windows (pointwise.h: mask_lines_kernel) -- round 5: per two lines through
    16-pixel windows, per four through 8-pixel ones --, and by the chunk scan of rounds 2-3 elsewhere; PAOS_MASK_SCAN=1
    forces the scan, PAOS_MASK_PAIRS=0 / 1 the wider windows.  Random ellipses -- round and flat, centred and near the edge of
    the grid, apertures and obscurations -- riding on a row pass and on a column pass over a random field: the two
    renderings give the same field bit for bit."""
    import os

    from paos_amd import _lib

    rng = np.random.default_rng(7 * n)
    u = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    dev = _lib.DeviceFields(n, 1, "fp64")
    try:
        for trial in range(10):
            a, b = rng.uniform(4.0, 0.45 * n, 2)
            if trial % 3 == 0:
                b = a  # round
            xc, yc = (n / 2 + rng.uniform(-3, 3, 2)) if trial % 2 == 0 else rng.uniform(0.1 * n, 0.9 * n, 2)
            obsc = 1.0 if trial in (4, 7) else 0.0
            if 2.0 * max(a, b) * np.sqrt(3.0 / min(a, b)) + 8.0 > 192:  # (the bound of lower_frugal: records must fit)
                a = b = max(a, b)
            blocks = np.array([[[1.0, xc, yc, a, b]], [[0.0, obsc, 1.0, float(_lib.SHAPE_ELLIPSE), 0.0]], [[1.0, 0.0, 0.0, 0.0, 0.0]]])
            got = {}
            # default: four lines per wave through 8-pixel windows where they fit (round 5), else two through 16-pixel ones,
            # else one through 32-pixel ones, else the scan; the switches force the older paths.  The context keys its kept
            # record sets by the mode too, so every mode renders (checked: `rendered` grows)
            for mode, env in (("windows", {}), ("pairs", {"PAOS_MASK_PAIRS": "1"}), ("lines", {"PAOS_MASK_PAIRS": "0"}),
                              ("scan", {"PAOS_MASK_SCAN": "1"})):
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
            for mode in ("pairs", "lines", "scan"):
                for x, y in zip(got["windows"], got[mode]):
                    assert np.array_equal(x, y), (mode, trial, a, b, xc, yc, obsc)
            # ... and the mask is what the stand-alone aperture kernel applies: |F^-1| of the row pass is the masked field
            if obsc == 0.0:
                back = np.fft.ifft(got["windows"][0], axis=1)
                dark = np.abs(back) < 1e-9
                yy, xx = np.mgrid[0:n, 0:n]
                outside = ((xx - xc) / (a + 1.5)) ** 2 + ((yy - yc) / (b + 1.5)) ** 2 > 1.0
                inside = ((xx - xc) / max(a - 1.5, 0.1)) ** 2 + ((yy - yc) / max(b - 1.5, 0.1)) ** 2 < 1.0
                assert dark[outside].all() and np.allclose(back[inside], u[inside], rtol=0, atol=1e-9)
    finally:
        dev.close()
