"""GPU parity: HIP path (through the C ABI) vs the reference's own golden vectors
and vs the CPU oracle on the same seeded inputs.  Run with ``-m gpu`` on an MI355X.

Tolerances: the north star asks |PSF_gpu - PSF_ref| / |PSF_ref| < 1e-10 in fp64
(BASELINE.json); fields are checked at 1e-11 of max|u| (observed ~1e-14), aperture
classification {0, partial, 1} bit-exactly, fp32 mode at 2e-5 (c64 FFT chain).
"""
import copy
import os

import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

FIELD_TOL = 1e-11
PSF_TOL = 1e-10
ORDERINGS = ("ansi", "noll", "fringe", "standard")
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


@pytest.fixture(scope="module")
def WFO():
    from paos_amd.wfo import WFO as cls

    return cls


def scalars(w):
    return np.array([w.wl, w.z, w.w0, w.zw0, w.zr, w.dx, w.dy, w.C, w.fratio, w.wz,
                     w.distancetofocus])


def fresh(WFO, g, wl=3.0e-6, anam=False, n=64):
    w = WFO(1.0, wl, n, 4)
    if anam:
        w.Magnification(1.25, 0.8)
    w._wfo = g["u0"]
    return w


def check(g, tag, w, before=None):
    if before is not None:
        assert np.array_equal(before, g[tag + "_before"]), tag
    assert np.array_equal(scalars(w), g[tag + "_after"]), tag
    assert rel_err(w.wfo, g[tag + "_u"]) < FIELD_TOL, (tag, rel_err(w.wfo, g[tag + "_u"]))


def test_library_loaded_is_in_tree():
    from paos_amd import _lib

    lib = _lib.load()
    assert os.path.dirname(_lib.LIB_PATH).endswith("paos_amd")
    assert b"gfx950" in lib.paos_build_info()


def test_upload_download_roundtrip(WFO):
    g = load_golden("primitives.npz")
    w = fresh(WFO, g)
    assert np.array_equal(w.wfo, g["u0"])
    assert np.array_equal(np.asarray(w._wfo), g["u0"]) and w._wfo.shape == (64, 64)
    assert rel_err(w.amplitude, np.abs(g["u0"])) < 1e-15
    assert rel_err(w.phase, np.angle(g["u0"])) < 1e-15
    w2 = WFO(1.0, 1e-6, 128, 4)
    assert np.array_equal(w2.wfo, np.ones((128, 128), dtype=complex))


@pytest.mark.parametrize("anam", [False, True])
def test_primitives_vs_reference_vectors(WFO, anam):
    """Every field primitive against outputs of the reference's own wfo.py."""
    g = load_golden("primitives.npz")
    sfx = "_anam" if anam else ""
    w = fresh(WFO, g, anam=anam)
    b = scalars(w)
    w.make_stop()
    check(g, "make_stop" + sfx, w, b)
    for fl in (10.0, -3.0, 0.4):
        w = fresh(WFO, g, anam=anam)
        b = scalars(w)
        w.lens(fl)
        check(g, f"lens_{fl}{sfx}", w, b)
    for dz in (0.5, -0.25, 1.0e-8):
        w = fresh(WFO, g, anam=anam)
        b = scalars(w)
        w.ptp(dz)
        check(g, f"ptp_{dz}{sfx}", w, b)
    for fl in (10.0, -7.0):
        w = fresh(WFO, g, anam=anam)
        w.lens(fl)
        w._wfo = g["u0"]
        b = scalars(w)
        dz = w.zw0 - w.z
        assert dz == g[f"stw_{fl}{sfx}_dz"]
        w.stw(dz)
        check(g, f"stw_{fl}{sfx}", w, b)
    for dz in (2.0, -1.5):
        w = fresh(WFO, g, anam=anam)
        b = scalars(w)
        w.wts(dz)
        check(g, f"wts_{dz}{sfx}", w, b)


def test_propagate_regimes(WFO):
    g = load_golden("primitives.npz")
    for tag, (fl, dist) in {"II": (None, 1.0), "OI": (10.0, 10.0), "IO": (None, 4.0e6),
                            "OO": (10.0, 20.0)}.items():
        w = WFO(1.0, 3.0e-6, 64, 4)
        if fl is not None:
            w.lens(fl)
        w._wfo = g["u0"]
        b = scalars(w)
        w.propagate(dist)
        assert w.propagator == tag
        check(g, "propagate_" + tag, w, b)
    w = fresh(WFO, g)
    w.lens(5.0)
    w.Magnification(1.3, 0.7)
    assert np.array_equal(scalars(w), g["magnification_after"])
    w = fresh(WFO, g)
    w.lens(5.0)
    w.ChangeMedium(0.66)
    assert np.array_equal(scalars(w), g["changemedium_after"])


def test_guards_mirror_reference(WFO):
    w = WFO(1.0, 3.0e-6, 64, 4)
    u = w.wfo
    w.ptp(1e-10)
    assert np.array_equal(w.wfo, u) and w.z == 0.0
    w.lens(10.0)
    with pytest.raises(ValueError):
        w.ptp(1.0)
    with pytest.raises(ValueError):
        WFO(1.0, 3.0e-6, 64, 4).stw(1.0)
    with pytest.raises(AssertionError):
        WFO(1.0, 3.0e-6, 100, 4)
    with pytest.raises(ValueError):
        w.aperture(0, 0, hx=1, hy=1, shape="hexagonal")


@pytest.mark.parametrize("ordering", ORDERINGS)
def test_zernike_vs_reference_vectors(WFO, ordering):
    g = load_golden("zernike_maps.npz")
    for norm in (True, False):
        for origin in ("x", "y"):
            w = WFO(1.0, 1.0e-6, 64, 2)
            wfe = w.zernikes(np.arange(36), g["coef"], ordering, norm, 0.5, origin=origin)
            key = f"{ordering}_{int(norm)}_{origin}"
            assert np.array_equal(np.ma.getmaskarray(wfe), g[key + "_mask"]), key
            assert rel_err(wfe.filled(0.0), g[key + "_wfe"]) < 1e-13, key
            assert rel_err(w.wfo, g[key + "_u"]) < FIELD_TOL, key


def test_zernike_anamorphic_and_kat(WFO):
    g = load_golden("zernike_maps.npz")
    w = WFO(1.0, 2.0e-6, 64, 2)
    w.Magnification(1.5, 0.75)
    wfe = w.zernikes(np.arange(15), g["coef"][:15], "noll", True, 0.62, origin="x")
    assert rel_err(wfe.filled(0.0), g["anam_wfe"]) < 1e-13
    assert rel_err(w.wfo, g["anam_u"]) < FIELD_TOL
    # the reference's notebook pin (notebook/ComputeGridSag.ipynb:131)
    w2 = WFO(1.1, 0.55e-6, 1024, 4)
    wfe = w2.zernikes(np.arange(6), np.array([0, 10, 0, -30, 20, 0]) * 1e-9, "noll", True, 0.55)
    assert abs(np.std(wfe) - 3.7394904478041395e-08) < 1e-20


def test_aperture_masks_vs_oracle(WFO):
    """Mask VALUES are parity-unpinned (photutils absent); the GPU must at least
    agree with the CPU restatement: classification bit-exact, values to 1e-14."""
    from oracle import aperture_np
    from paos_amd.aperture import EllipticalAperture, RectangularAperture

    cases = [
        (EllipticalAperture((128.0, 128.0), 32.0, 32.0), aperture_np.ellipse_mask, (128.0, 128.0, 32.0, 32.0, 0.0)),
        (EllipticalAperture((120.3, 131.1), 40.7, 25.2, 0.6), aperture_np.ellipse_mask, (120.3, 131.1, 40.7, 25.2, 0.6)),
        (EllipticalAperture((30.2, 250.9), 40.7, 25.2), aperture_np.ellipse_mask, (30.2, 250.9, 40.7, 25.2, 0.0)),
        (EllipticalAperture((100.2, 99.9), 0.3, 0.2, 0.3), aperture_np.ellipse_mask, (100.2, 99.9, 0.3, 0.2, 0.3)),
        (RectangularAperture((128.0, 128.0), 80.5, 33.25), aperture_np.rectangle_mask, (128.0, 128.0, 80.5, 33.25, 0.0)),
        (RectangularAperture((127.6, 130.2), 50.5, 21.25, 0.3), aperture_np.rectangle_mask, (127.6, 130.2, 50.5, 21.25, 0.3)),
    ]
    for ap, fn, args in cases:
        got = ap.to_mask(method="exact" if isinstance(ap, EllipticalAperture) else "subpixel").to_image((256, 256))
        ref = fn((256, 256), *args)
        assert np.array_equal(got == 0.0, ref == 0.0), ap
        assert np.array_equal(got == 1.0, ref == 1.0), ap
        assert np.max(np.abs(got - ref)) < 1e-14, (ap, np.max(np.abs(got - ref)))
    g = load_golden("primitives.npz")
    w = fresh(WFO, g)
    w.aperture(0.1, -0.2, hx=0.9, hy=0.6, shape="elliptical")
    assert rel_err(w.wfo, g["aperture_ell_u"]) < 1e-14
    w = fresh(WFO, g)
    w.aperture(0.0, 0.0, hx=0.7, hy=0.3, shape="rectangular", obscuration=True)
    assert rel_err(w.wfo, g["aperture_rect_obsc_u"]) < 1e-14


CHAINS = {"Hubble_simple": 128, "Excite_TEL": 64, "Ariel_AIRS-CH0": 64, "Ariel_FGS-FGS1": 64}


def _spec(name):
    from paos_amd.chains import syn20_chain
    from paos_amd.parse_config import parse_config

    if name == "SYN20":
        return dict(pup=1.0, wl=1.0e-6, zoom=4, field={"us": 0.0, "ut": 0.0}, chain=syn20_chain()), 128
    pup, par, wls, fields, chains = parse_config(os.path.join(DATA, "lens", name + ".ini"))
    return dict(pup=pup, wl=1.0e-6 * wls[0], zoom=par["zoom"], field=fields[0], chain=chains[0]), CHAINS[name]


@pytest.mark.parametrize("name", list(CHAINS) + ["SYN20"])
def test_run_vs_reference_vectors(name):
    """run() end to end against the reference's run() outputs (tests/golden)."""
    from paos_amd.run import run

    spec, n = _spec(name)
    gs = load_golden(f"scalars_{name}.npz")
    chain = copy.deepcopy(spec["chain"])
    for item in chain.values():
        item["save"] = True
    ret = run(spec["pup"], spec["wl"], 64, spec["zoom"], spec["field"], chain)
    assert np.array_equal(sorted(ret), gs["nums"])
    for row, k in zip(gs["table"], gs["nums"]):
        r = ret[k]
        got = [r["wl"], r["dx"], r["dy"], r["wz"], r["distancetofocus"], r["fratio"]]
        assert np.array_equal(got, row), (name, k, got, row)
    assert [ret[k]["propagator"] for k in gs["nums"]] == list(gs["propagator"])
    assert np.array_equal(np.array([ret[k]["ABCDt"]() for k in gs["nums"]]), gs["ABCDt"])
    assert np.array_equal(np.array([ret[k]["extent"] for k in gs["nums"]]), gs["extent"])

    gr = load_golden(f"run_{name}.npz")
    ret = run(spec["pup"], spec["wl"], n, spec["zoom"], spec["field"], spec["chain"])
    assert np.array_equal(sorted(ret), gr["nums"])
    for k in gr["nums"]:
        ref = gr[f"S{k:02d}_wfo"]
        assert rel_err(ret[k]["wfo"], ref) < FIELD_TOL, (name, k, rel_err(ret[k]["wfo"], ref))
        assert rel_err(ret[k]["amplitude"] ** 2, np.abs(ref) ** 2) < PSF_TOL
        assert rel_err(ret[k]["amplitude"], np.abs(ref)) < FIELD_TOL
        if f"S{k:02d}_wfe" in gr:
            assert rel_err(ret[k]["wfe"].filled(0.0), gr[f"S{k:02d}_wfe"]) < 1e-13


@pytest.mark.parametrize("n", [256, 512])
def test_run_vs_oracle_mid_sizes(n):
    """SYN20 and the anamorphic AIRS-CH0 chain against the oracle at sizes it finishes in seconds."""
    from oracle.run_np import run as oracle_run
    from paos_amd.run import run

    for name in ("SYN20", "Ariel_AIRS-CH0"):
        spec, _ = _spec(name)
        got = run(spec["pup"], spec["wl"], n, spec["zoom"], spec["field"], spec["chain"])
        ref = oracle_run(spec["pup"], spec["wl"], n, spec["zoom"], spec["field"], spec["chain"], light=True)
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert rel_err(got[k]["wfo"], ref[k]["wfo"]) < FIELD_TOL, (name, n, k)
            assert rel_err(got[k]["amplitude"] ** 2, ref[k]["amplitude"] ** 2) < PSF_TOL
            for key in ("dx", "dy", "wl", "fratio", "wz", "distancetofocus", "propagator"):
                assert got[k][key] == ref[k][key], (name, k, key)
            # phase only where the amplitude is significant (atan2 of noise elsewhere)
            sig = ref[k]["amplitude"] > 1e-6 * ref[k]["amplitude"].max()
            dphi = np.angle(np.exp(1j * (got[k]["phase"] - ref[k]["phase"])))
            assert np.max(np.abs(dphi[sig])) < 1e-8


def test_wfe_injection_vs_reference_vectors():
    from paos_amd.chains import inject_wfe, syn20_chain
    from paos_amd.run import run

    for col in (0, 1):
        g = load_golden(f"run_SYN20_wfe{col}.npz")
        ret = run(1.0, 1.0e-6, 64, 4, {"us": 0.0, "ut": 0.0}, inject_wfe(syn20_chain(), g["draw_nm"]))
        assert rel_err(ret[20]["wfo"], g["wfo"]) < FIELD_TOL
        assert np.array_equal([ret[20]["dx"], ret[20]["dy"], ret[20]["fratio"]], g["scal"])


def test_run_batch_matches_single_runs():
    """Wavelength sweep + WFE draws in one batch == item-by-item run() (bitwise: same kernels)."""
    from paos_amd.chains import inject_wfe, read_wfe_table, syn20_chain, syn20_wavelength
    from paos_amd.run import run, run_batch

    _, _, _, table = read_wfe_table(os.path.join(DATA, "wfe", "wfe_realization_SN20210914.csv"))
    field = {"us": 0.0, "ut": 0.0}
    wls = [syn20_wavelength(k * 100) for k in range(3)]
    chains = [inject_wfe(syn20_chain(), table[:, k]) for k in range(3)]
    batch = run_batch(1.0, wls, 128, 4, field, chains, outputs=("psf", "wfo"))
    for i in range(3):
        single = run(1.0, wls[i], 128, 4, field, chains[i])
        assert sorted(batch[i]) == sorted(single)
        for k in single:
            assert np.array_equal(batch[i][k]["wfo"], single[k]["wfo"])
            assert rel_err(batch[i][k]["psf"], single[k]["amplitude"] ** 2) < 1e-15
            assert batch[i][k]["dx"] == single[k]["dx"] and batch[i][k]["fratio"] == single[k]["fratio"]
            assert abs(batch[i][k]["power"] - np.sum(single[k]["amplitude"] ** 2)) < 1e-12


def test_batch_with_divergent_propagator_regimes():
    """One batch, two wavelengths whose planners pick different primitives for the same
    surface (OO = stw + wts vs OI = stw + ptp): per-item enable flags keep them correct."""
    from oracle.run_np import run as oracle_run
    from paos_amd.run import run_batch
    from test_host_logic import _two_regime_chain

    field = {"us": 0.0, "ut": 0.0}
    wls = [1.0e-6, 1.0e-5]
    chains = [_two_regime_chain(), _two_regime_chain()]
    # 1024: the frugal kernels (per-item data, disabled phases as coef = 0); 256: the generic ones
    for n in (1024, 256):
        _check_divergent(run_batch, oracle_run, _two_regime_chain, wls, field, n)


def _check_divergent(run_batch, oracle_run, _two_regime_chain, wls, field, n):
    chains = [_two_regime_chain(), _two_regime_chain()]
    got = run_batch(1.0, wls, n, 4, field, chains, outputs=("wfo",))
    props = []
    for i in range(2):
        ref = oracle_run(1.0, wls[i], n, 4, field, chains[i], light=True)
        props.append(ref[3]["propagator"])
        for k in ref:
            assert rel_err(got[i][k]["wfo"], ref[k]["wfo"]) < FIELD_TOL, (i, k)
            assert got[i][k]["propagator"] == ref[k]["propagator"] and got[i][k]["dx"] == ref[k]["dx"]
    assert props == ["OO", "OI"]


def test_aperture_fusion_variants_agree(monkeypatch):
    """Apertures as stand-alone kernels, as per-line records riding on frugal passes (the
    default at N >= 1024), and as rendered weight maps in the generic kernel (forced at 256)
    give the same fields; the Hubble chain adds obscurations and off-centre pads, the field
    stop of SYN20 a rectangle."""
    import paos_amd.run as prun
    from paos_amd.chains import syn20_chain
    from paos_amd.parse_config import parse_config

    field = {"us": 0.0, "ut": 0.0}
    pup, par, wls, fields, chains = parse_config(os.path.join(DATA, "lens", "Hubble_simple.ini"))
    cases = [((1.0, 1.0e-6, 1024, 4, field), syn20_chain), ((pup, 1e-6 * wls[0], 1024, par["zoom"], fields[0]), lambda: chains[0]),
             ((1.0, 1.0e-6, 256, 4, field), syn20_chain)]
    for args, chain in cases:
        monkeypatch.setattr(prun, "FUSE_APERTURES", False)
        base = prun.run(*args, chain())
        for mode in ("auto", True):
            monkeypatch.setattr(prun, "FUSE_APERTURES", mode)
            fused = prun.run(*args, chain())
            for k in base:
                assert rel_err(fused[k]["wfo"], base[k]["wfo"]) < 1e-13, (args[2], mode, k)


def test_psf_metrics_on_device(WFO):
    """Power, centroid, peak and encircled energy of |u|^2 against NumPy on the downloaded PSF."""
    n = 512
    w = WFO(1.0, 1.0e-6, n, 4)
    w.aperture(0.0, 0.0, hx=0.5, hy=0.4, shape="elliptical")
    w.make_stop()
    w.lens(10.0)
    w.propagate(10.0)
    psf = w.intensity
    radii = [2.0, 5.5, 20.0, 300.0]
    m = w._dev.psf_metrics(radii)[0]
    yy, xx = np.mgrid[0:n, 0:n]
    assert abs(m["power"] - psf.sum()) < 1e-13 and abs(m["peak"] - psf.max()) < 1e-18
    assert abs(m["centroid"][0] - (psf * xx).sum() / psf.sum()) < 1e-9
    assert abs(m["centroid"][1] - (psf * yy).sum() / psf.sum()) < 1e-9
    d2 = (xx - n / 2) ** 2 + (yy - n / 2) ** 2
    for r, ee in zip(radii, m["encircled"]):
        assert abs(ee - psf[d2 <= r * r].sum()) < 1e-13
    assert m["encircled"][-1] > 0.999 * m["power"]


def test_fp32_mode_tolerance():
    """c64 storage / FFT arithmetic with fp64 phase arguments: expected ~3e-6."""
    from paos_amd.chains import syn20_chain
    from paos_amd.run import run

    field = {"us": 0.0, "ut": 0.0}
    for n in (256, 2048):  # generic kernels / the c64 frugal pass kernels
        r64 = run(1.0, 1.0e-6, n, 4, field, syn20_chain())
        r32 = run(1.0, 1.0e-6, n, 4, field, syn20_chain(), precision="fp32")
        e = rel_err(r32[20]["amplitude"] ** 2, r64[20]["amplitude"] ** 2)
        assert 1e-9 < e < 2e-5, (n, e)


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_full_size_properties(WFO, n):
    """Size-independent checks at BASELINE grid sizes: unit power after make_stop is
    conserved by ptp/stw/wts (ortho FFTs + unimodular phases); ptp(dz) o ptp(-dz) and
    stw o its inverse return the input field."""
    w = WFO(1.0, 1.0e-6, n, 4)
    w.aperture(0.0, 0.0, hx=0.5, hy=0.5, shape="elliptical")
    w.make_stop()
    dev = w._dev
    assert abs(dev.norm2()[0] - 1.0) < 1e-13
    u0 = w.wfo
    w.ptp(3.0)
    assert abs(dev.norm2()[0] - 1.0) < 1e-12
    w.ptp(-3.0)
    assert rel_err(w.wfo, u0) < 1e-12
    w.lens(10.0)
    w.propagate(10.0)  # OI: stw + ptp
    assert w.propagator == "OI" and abs(dev.norm2()[0] - 1.0) < 1e-12
    # Airy pattern: the peak of a uniform circular pupil's PSF sits at the grid centre
    psf = w.intensity
    assert np.unravel_index(np.argmax(psf), psf.shape) == (n // 2, n // 2)
    assert np.allclose(psf, psf[::-1, ::-1][np.ix_(np.r_[n - 1, 0:n - 1], np.r_[n - 1, 0:n - 1])], atol=1e-18)


def test_full_size_4096_vs_oracle_single_ptp(WFO):
    """One 4096^2 ptp (the headline step) against NumPy on the same seeded input."""
    n = 4096
    rng = np.random.default_rng(5)
    u0 = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    from oracle.pop_numpy import RefWFO

    w = WFO(1.0, 1.0e-6, n, 4)
    w._wfo = u0
    w.ptp(2.5)
    r = RefWFO(1.0, 1.0e-6, n, 4)
    r._wfo = u0.copy()
    r.ptp(2.5)
    assert rel_err(w.wfo, r._wfo) < FIELD_TOL


@pytest.mark.gpu
def test_wfo_zernikes_orthonorm_vs_reference_vectors(WFO):
    """WFO.zernikes(orthonorm=True, mask=...) -- wfo.py:574-654 with PolyOrthoNorm
    (zernike.py:320-402): Gram sums on the GPU, Cholesky / inverse on the host."""
    g = load_golden("orthonorm.npz")
    w = WFO(1.0, 1.2e-6, 64, 2)
    w._wfo = g["wfo_in"]
    wfe = w.zernikes(np.arange(10), g["wfo_coeff"], "noll", True, 0.5, origin="x", orthonorm=True,
                     mask=g["wfo_zmask"])
    assert np.array_equal(np.ma.getmaskarray(wfe), g["wfo_wfe_mask"])
    assert rel_err(wfe.filled(0.0), g["wfo_wfe"]) < 1e-11
    assert rel_err(w.wfo, g["wfo_out"]) < FIELD_TOL
    # a plain expansion restricted by a mask (no orthonormalisation)
    from oracle.pop_numpy import RefWFO

    ref = RefWFO(1.0, 1.2e-6, 64, 2)
    ref._wfo = g["wfo_in"].copy()
    want = ref.zernikes(np.arange(10), g["wfo_coeff"], "noll", True, 0.5, mask=g["wfo_zmask"].copy())
    w2 = WFO(1.0, 1.2e-6, 64, 2)
    w2._wfo = g["wfo_in"]
    got = w2.zernikes(np.arange(10), g["wfo_coeff"], "noll", True, 0.5, mask=g["wfo_zmask"])
    assert np.array_equal(np.ma.getmaskarray(got), np.ma.getmaskarray(want))
    assert rel_err(w2.wfo, ref._wfo) < FIELD_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 128, 512])
def test_run_zorthonorm(n):
    """run() with Zorthonorm=True on an elliptical pupil (run.py:133-141): against the
    reference's vectors at 64 / 128, against the oracle at 512."""
    from paos_amd.chains import syn20_orthonorm_chain
    from paos_amd.run import run

    chain = syn20_orthonorm_chain()
    chain[2]["save"] = True
    ret = run(1.0, 1.0e-6, n, 4, {"us": 0.0, "ut": 0.0}, chain)
    if n <= 128:
        g = load_golden("orthonorm.npz")
        want_wfe, want_mask = g[f"run{n}_S02_wfe"], g[f"run{n}_S02_wfe_mask"]
        want2, want20 = g[f"run{n}_S02_wfo"], g[f"run{n}_S20_wfo"]
    else:
        from oracle.run_np import run as oracle_run

        ref = oracle_run(1.0, 1.0e-6, n, 4, {"us": 0.0, "ut": 0.0}, chain)
        want_wfe, want_mask = ref[2]["wfe"].filled(0.0), np.ma.getmaskarray(ref[2]["wfe"])
        want2, want20 = ref[2]["wfo"], ref[20]["wfo"]
    assert np.array_equal(np.ma.getmaskarray(ret[2]["wfe"]), want_mask)
    assert rel_err(ret[2]["wfe"].filled(0.0), want_wfe) < 1e-11
    assert rel_err(ret[2]["wfo"], want2) < FIELD_TOL
    assert rel_err(ret[20]["wfo"], want20) < FIELD_TOL
    psf, ref_psf = np.abs(ret[20]["wfo"]) ** 2, np.abs(want20) ** 2
    assert rel_err(psf, ref_psf) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 128])
def test_run_zorthonorm_rectangular_pupil(n):
    """Zorthonorm over a RECTANGULAR aperture (run.py:133-141 with RectangularAperture.to_mask("exact"),
    which photutils serves with the 32 x 32 sub-pixel rule): against the reference's run()."""
    from paos_amd.chains import syn20_orthonorm_chain
    from paos_amd.run import run, run_batch

    chain = syn20_orthonorm_chain()
    chain[2]["aperture"] = {"shape": "rectangular", "type": "aperture", "xrad": 0.9, "yrad": 0.6, "xc": 0.0, "yc": 0.0}
    chain[2]["save"] = True
    g = load_golden("r2_orthonorm_rect.npz")
    ret = run(1.0, 1.0e-6, n, 4, {"us": 0.0, "ut": 0.0}, chain)
    assert np.array_equal(np.ma.getmaskarray(ret[2]["wfe"]), g[f"run{n}_S02_wfe_mask"])
    assert rel_err(ret[2]["wfe"].filled(0.0), g[f"run{n}_S02_wfe"]) < 1e-11
    assert rel_err(ret[2]["wfo"], g[f"run{n}_S02_wfo"]) < FIELD_TOL
    assert rel_err(ret[20]["wfo"], g[f"run{n}_S20_wfo"]) < FIELD_TOL
    # a batch mixing a rectangular and an elliptical pupil
    ell = syn20_orthonorm_chain()
    ell[2]["save"] = True
    both = run_batch(1.0, [1.0e-6, 1.0e-6], n, 4, {"us": 0.0, "ut": 0.0}, [chain, ell], outputs=("wfo",))
    assert rel_err(both[0][20]["wfo"], g[f"run{n}_S20_wfo"]) < FIELD_TOL
    assert rel_err(both[1][20]["wfo"], load_golden("orthonorm.npz")[f"run{n}_S20_wfo"]) < FIELD_TOL


@pytest.mark.gpu
def test_zorthonorm_batch_and_errors():
    from paos_amd.chains import syn20_orthonorm_chain, syn20_chain
    from paos_amd.run import run, run_batch

    wls = [1.0e-6, 1.3e-6]
    chain = syn20_orthonorm_chain()
    field = {"us": 0.0, "ut": 0.0}
    batch = run_batch(1.0, wls, 128, 4, field, [chain, chain], outputs=("wfo",))
    for wl, got in zip(wls, batch):
        one = run(1.0, wl, 128, 4, field, chain)
        assert rel_err(got[20]["wfo"], one[20]["wfo"]) < 1e-13
    with pytest.raises(NotImplementedError):
        run_batch(1.0, wls, 128, 4, field, [chain, syn20_chain()])
    bad = syn20_orthonorm_chain()
    del bad[2]["aperture"]
    with pytest.raises(AssertionError):
        run(1.0, 1.0e-6, 64, 4, field, bad)


@pytest.mark.gpu
def test_zernike_gram_full_size_properties():
    """4096^2: the Gram sums obey what the polynomials guarantee -- piston^2 sums to the pixel
    count, the count is the number of pupil pixels, products of opposite parity cancel on a centred
    elliptical pupil, and the orthonormalised set has unit covariance."""
    import time

    from paos_amd import _lib
    from paos_amd.aperture import make_aperture
    from paos_amd.planner import (gram_polynomials, jacobi_recurrence, orthonorm_matrix,
                                  zernike_block)
    from paos_amd.zernike import Zernike, norm_factors

    n, k = 4096, 36
    dx = 1.0 / 1024
    dev = _lib.DeviceFields(n, 1)
    try:
        m, nn = Zernike.j2mn(k, "noll")
        norm = norm_factors(m, nn, True)
        block, nmax, kdim = zernike_block(m, nn, norm, np.zeros(k), dx, dx, 0.5, 1.0e-6)
        ap = make_aperture(n, dx, dx, 0.0, 0.0, hx=0.5, hy=0.3, shape="elliptical")
        dev.pupil_aperture(_lib.SHAPE_ELLIPSE, [ap.block()])
        poly = gram_polynomials(m, nn, norm)
        dev.zernike_gram(nmax, kdim, jacobi_recurrence(nmax), [block], poly)  # warm-up
        t0 = time.perf_counter()
        sums, counts = dev.zernike_gram(nmax, kdim, jacobi_recurrence(nmax), [block], poly)
        dt = time.perf_counter() - t0
        print(f"zernike_gram 4096^2 K=36: {dt * 1e3:.2f} ms")
        mask = dev.aperture_mask(_lib.SHAPE_ELLIPSE, ap.block())
        ax = (np.arange(n) - n // 2) * dx
        rho = np.sqrt(ax[None, :] ** 2 + ax[:, None] ** 2) / 0.5
        assert counts[0] == np.count_nonzero((mask != 0) & (rho <= 1.0))
        iu = np.triu_indices(k)
        cov = np.zeros((k, k))
        cov[iu] = sums[0] / counts[0]
        assert sums[0][0] == counts[0]  # piston: Z_0 = 1 everywhere
        # x -> -x flips cos(m phi) terms of odd m and sin(m phi) terms of even m
        odd_x = np.where(m >= 0, np.abs(m) % 2 == 1, np.abs(m) % 2 == 0)
        cross = odd_x[:, None] != odd_x[None, :]
        assert np.max(np.abs(cov[np.triu(cross)])) < 1e-12
        mm = orthonorm_matrix(sums[0], counts[0], k)
        full = cov + np.triu(cov, 1).T
        assert np.allclose(mm @ full @ mm.T, np.eye(k), atol=1e-9)
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Ariel_AIRS-CH1", "Ariel_FGS-FGS2", "Ariel_FGS-NIRSpec", "Ariel_FGS-VISPhot",
                                  "lens_file_TA_Ground", "lens_file_TA_OGSE_Ground", "lens_file_template",
                                  "periscope"])
def test_remaining_lens_files_vs_reference_vectors(name):
    """Every other runnable shipped lens file, first and last wavelength, against the reference's
    own output at 64^2 and against the oracle at 256^2 (PSF to the north-star tolerance)."""
    from oracle.run_np import run as oracle_run
    from paos_amd.parse_config import parse_config
    from paos_amd.run import run

    g = load_golden("run_more_chains.npz")
    pup, par, wls, fields, chains = parse_config(os.path.join(DATA, "lens", name + ".ini"))
    for tag, iw in (("first", 0), ("last", len(wls) - 1)):
        key = f"{name}_{tag}"
        ret = run(pup, 1.0e-6 * wls[iw], 64, par["zoom"], fields[0], chains[iw])
        nums = sorted(ret.keys())
        assert np.array_equal(nums, g[key + "_nums"])
        table = np.array([[ret[k][f] for f in ("wl", "dx", "dy", "wz", "distancetofocus", "fratio")] for k in nums])
        assert np.array_equal(table, g[key + "_table"]), key
        assert [ret[k]["propagator"] for k in nums] == list(g[key + "_propagator"])
        assert rel_err(ret[nums[-1]]["wfo"], g[key + "_wfo"]) < FIELD_TOL, key
    ret = run(pup, 1.0e-6 * wls[0], 256, par["zoom"], fields[0], chains[0])
    ref = oracle_run(pup, 1.0e-6 * wls[0], 256, par["zoom"], fields[0], chains[0])
    last = sorted(ret.keys())[-1]
    assert rel_err(np.abs(ret[last]["wfo"]) ** 2, np.abs(ref[last]["wfo"]) ** 2) < 1e-10


@pytest.mark.gpu
def test_self_regression_like_the_reference_test():
    """tests/regressionTest.py:35-90 runs the Hubble_simple pipeline twice and demands bit-equal
    datasets.  Same here: two runs, every array of every saved surface bitwise identical."""
    from paos_amd.parse_config import parse_config
    from paos_amd.run import run

    pup, par, wls, fields, chains = parse_config(os.path.join(DATA, "lens", "Hubble_simple.ini"))
    first = run(pup, 1.0e-6 * wls[0], par["grid_size"], par["zoom"], fields[0], chains[0])
    second = run(pup, 1.0e-6 * wls[0], par["grid_size"], par["zoom"], fields[0], chains[0])
    assert sorted(first) == sorted(second)
    for k in first:
        for name, val in first[k].items():
            other = second[k][name]
            if isinstance(val, np.ndarray):
                assert np.array_equal(np.ma.getdata(val), np.ma.getdata(other)), (k, name)
            elif isinstance(val, (float, int, str)):
                assert val == other, (k, name)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_start_equals_fill_aperture_stop(precision):
    """paos_start (ones -> aperture -> make_stop in one write) leaves bit for bit the field of
    paos_fill + paos_aperture + paos_make_stop, per item flags included."""
    from paos_amd import _lib
    from paos_amd.aperture import make_aperture

    n, nb = 256, 4
    dx = 1.0 / 64
    cases = [
        (_lib.SHAPE_ELLIPSE, [make_aperture(n, dx, dx, 0.01 * i, -0.02 * i, hx=0.5 + 0.01 * i, hy=0.4, shape="elliptical")
                              for i in range(nb)], [False, True, False, True]),
        (_lib.SHAPE_RECT, [make_aperture(n, dx, dx, 0.0, 0.03 * i, hx=0.3, hy=0.2 + 0.01 * i, shape="rectangular")
                           for i in range(nb)], [False, False, True, False]),
    ]
    a = _lib.DeviceFields(n, nb, precision)
    b = _lib.DeviceFields(n, nb, precision)
    try:
        for code, handles, obsc in cases:
            for stop in ([1.0, 0.0, 1.0, 1.0], None):
                blocks = [h.block(obscuration=o) for h, o in zip(handles, obsc)]
                blocks[3][0] = 0.0  # item 3: no aperture at all
                value = 0.75 - 0.5j
                a.fill(value)
                a.aperture(code, blocks)
                if stop is not None:
                    a.make_stop(stop)
                b.start(value, code, blocks, stop)
                for i in range(nb):
                    assert np.array_equal(a.download(i), b.download(i)), (code, stop, i)
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_large_downloads_use_page_locked_arrays_and_recycle_them():
    """Arrays above 4 MiB come back in page-locked memory that returns to a pool when the array
    dies; the values are those of the ordinary path."""
    import gc

    from paos_amd import _lib

    n = 1024
    dev = _lib.DeviceFields(n, 1)
    try:
        rng = np.random.default_rng(5)
        u = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        dev.upload(0, u)
        a = dev.download(0)
        assert np.array_equal(a, u) and a.flags.writeable
        amp = dev.download(0, _lib.WHAT_AMPLITUDE)
        assert rel_err(amp, np.abs(u)) < 1e-15
        live = _lib._pin_live
        assert live >= a.nbytes + amp.nbytes
        view = a[10:20]
        del a
        gc.collect()
        assert _lib._pin_live == live  # a view keeps the buffer
        assert np.array_equal(view, u[10:20])
        del view, amp
        gc.collect()
        assert _lib._pin_live == live - u.nbytes - u.nbytes // 2
        b = dev.download(0)  # recycled buffer
        assert np.array_equal(b, u)
    finally:
        dev.close()


@pytest.mark.gpu
def test_edge_cases_empty_and_rejected_inputs():
    """Empty inputs and the argument checks of WFO.__init__ (wfo.py:100-103)."""
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain
    from paos_amd.run import run, run_batch

    field = {"us": 0.0, "ut": 0.0}
    assert run(1.0, 1.0e-6, 64, 4, field, {}) == {}
    assert run_batch(1.0, [], 64, 4, field, []) == []
    unsaved = syn20_chain()
    for item in unsaved.values():
        item["save"] = False
    assert run(1.0, 1.0e-6, 64, 4, field, unsaved) == {}
    for bad in (dict(gridsize=100), dict(zoom=0), dict(pupil_diameter=-1.0), dict(wavelength=0.0)):
        kw = dict(pupil_diameter=1.0, wavelength=1.0e-6, gridsize=64, zoom=4)
        kw.update(bad)
        with pytest.raises(AssertionError):
            run(kw["pupil_diameter"], kw["wavelength"], kw["gridsize"], kw["zoom"], field, syn20_chain())
    with pytest.raises(_lib.PaosHipError):  # beyond what the kernels are instantiated for
        run(1.0, 1.0e-6, 8192, 4, field, syn20_chain())
    with pytest.raises(ValueError):
        run_batch(1.0, [1.0e-6], 64, 4, field, [syn20_chain(), syn20_chain()])
    with pytest.raises(AssertionError):
        run(1.0, 1.0e-6, 64, 4, field, [])


@pytest.mark.gpu
def test_run_sharded_single_process_matches_run_batch():
    """run_sharded without a process group: batches of 2 (with a shorter tail) == one run_batch."""
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.dist import run_sharded
    from paos_amd.run import run_batch

    field = {"us": 0.0, "ut": 0.0}
    wls = [syn20_wavelength(60 * k) for k in range(5)]
    chains = [syn20_chain() for _ in wls]
    whole = run_batch(1.0, wls, 128, 4, field, chains, outputs=("psf",))
    parts = run_sharded(1.0, wls, 128, 4, field, chains, batch=2, outputs=("psf",))
    assert [i for i, _ in parts] == [0, 1, 2, 3, 4]
    for (i, got), want in zip(parts, whole):
        assert np.array_equal(got[20]["psf"], want[20]["psf"]) and got[20]["power"] == want[20]["power"]


@pytest.mark.gpu
def test_keep_psf_on_device_matches_download():
    """run_batch(keep_psf=True): the PSFs of the last surface stay in HBM and read back equal to
    the ordinary intensity download."""
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.run import run_batch

    field = {"us": 0.0, "ut": 0.0}
    wls = [syn20_wavelength(100 * k) for k in range(3)]
    chains = [syn20_chain() for _ in wls]
    dev = _lib.DeviceFields(256, 3)
    try:
        res = run_batch(1.0, wls, 256, 4, field, chains, outputs=("psf",), dev=dev, keep_psf=True)
        for i in range(3):
            assert np.array_equal(dev.psf_fetch(i), res[i][20]["psf"])
            assert abs(res[i][20]["psf"].sum() - res[i][20]["power"]) < 1e-12
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Hubble_simple", "Ariel_AIRS-CH0", "SYN20"])
def test_off_axis_field_point_vs_reference_vectors(name):
    """Off-axis field point (decentred apertures, run.py:96-121) against the reference's output."""
    from paos_amd.chains import syn20_chain
    from paos_amd.parse_config import parse_config
    from paos_amd.run import run

    g = load_golden("run_offaxis.npz")
    field = {"us": float(g["us"]), "ut": float(g["ut"])}
    if name == "SYN20":
        pup, wl, zoom, chain = 1.0, 1.0e-6, 4, syn20_chain()
    else:
        pup, par, wls, _, chains = parse_config(os.path.join(DATA, "lens", name + ".ini"))
        wl, zoom, chain = 1.0e-6 * wls[0], par["zoom"], chains[0]
    ret = run(pup, wl, 64, zoom, field, chain)
    nums = sorted(ret.keys())
    assert np.array_equal(nums, g[name + "_nums"])
    table = np.array([[ret[k][f] for f in ("wl", "dx", "dy", "wz", "distancetofocus", "fratio")] for k in nums])
    assert np.array_equal(table, g[name + "_table"])
    assert rel_err(ret[nums[0]]["wfo"], g[name + "_first_wfo"]) < FIELD_TOL
    assert rel_err(ret[nums[-1]]["wfo"], g[name + "_wfo"]) < FIELD_TOL
