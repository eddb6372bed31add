"""GPU parity of the PRODUCTION kernels on the BASELINE.json configurations, against the CPU oracle.

The frugal pass kernels serve complex128 at N >= 1024 and complex64 at N >= 2048 (csrc/frugal_pass.h);
every test here runs at those sizes with the default aperture fusion ("auto": apertures ride on passes as
line records, all KPRE / KMID shapes of the chains occur) and compares the saved surfaces with
``oracle.run_np.run`` on the same inputs.  Oracle cost on the GPU box's host: ~2.5 s per wavefront at
1024^2, ~11 s at 2048^2, ~50 s at 4096^2 (results are cached per module so each oracle run happens once).

    configs[1]  Ariel_AIRS-CH0, 1 wavelength, 1024^2            test_end_to_end_vs_oracle[AIRS-1024]
    configs[2]  Ariel_AIRS-CH0 wavelength batch, 2048^2         test_airs_wavelength_batch_vs_oracle (+ AIRS-2048)
    configs[3]  Ariel_FGS-FGS1 + WFE table, Monte-Carlo batch   test_fgs1_monte_carlo_* (reference fixture + oracle)
    configs[4]  Excite_TEL, fp32 vs fp64                        test_excite_fp32_fp64_study
    headline    SYN20 4096^2 complex128                         test_end_to_end_vs_oracle[SYN20-4096]

Tolerances: field 1e-11 of max|u| and PSF 1e-10 of max (north star) in fp64; fp32 mode 2e-5 on the PSF
(c64 FFT chain, SURVEY 8d predicted ~3e-6).
"""
import os

import numpy as np
import pytest

from conftest import l2_rel_err, load_golden, rel_err

pytestmark = pytest.mark.gpu

FIELD_TOL = 1e-11
PSF_TOL = 1e-10
FP32_PSF_TOL = 2e-5
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
LENS = os.path.join(DATA, "lens")
WFE = os.path.join(DATA, "wfe", "wfe_realization_SN20210914.csv")
ON_AXIS = {"us": 0.0, "ut": 0.0}

_ORACLE = {}


def oracle(tag, *args):
    """oracle.run_np.run(*args, light=True), once per ``tag``."""
    if tag not in _ORACLE:
        from oracle.run_np import run as oracle_run

        _ORACLE[tag] = oracle_run(*args, light=True)
    return _ORACLE[tag]


def compare(got, ref, where, field_tol=FIELD_TOL, psf_tol=PSF_TOL, scalars=True):
    assert sorted(got) == sorted(ref), where
    worst = 0.0
    for k in ref:
        if "wfo" in got[k]:
            e = rel_err(got[k]["wfo"], ref[k]["wfo"])
            worst = max(worst, e)
            assert e < field_tol, (where, k, "field", e)
            e2 = l2_rel_err(got[k]["wfo"], ref[k]["wfo"])  # SURVEY 8d's second gate: L2-relative, same bound
            assert e2 < field_tol, (where, k, "field, L2-relative", e2)
        if "psf" in got[k]:
            psf = got[k]["psf"]
        elif "amplitude" in got[k]:
            psf = got[k]["amplitude"] ** 2
        else:
            psf = np.abs(got[k]["wfo"]) ** 2
        e = rel_err(psf, ref[k]["amplitude"] ** 2)
        assert e < psf_tol, (where, k, "psf", e)
        e2 = l2_rel_err(psf, ref[k]["amplitude"] ** 2)
        assert e2 < psf_tol, (where, k, "psf, L2-relative", e2)
        if scalars:
            for key in ("dx", "dy", "wl", "fratio", "wz", "distancetofocus", "propagator"):
                assert got[k][key] == ref[k][key], (where, k, key, got[k][key], ref[k][key])
    return worst


def _airs(wavelengths_um=None):
    from paos_amd.chains import parse_config_variant

    return parse_config_variant(os.path.join(LENS, "Ariel_AIRS-CH0.ini"), wavelengths_um)


def _spec(name):
    from paos_amd.chains import syn20_chain
    from paos_amd.parse_config import parse_config

    if name == "SYN20":
        return 1.0, 1.0e-6, 4, ON_AXIS, syn20_chain()
    pup, par, wls, fields, chains = parse_config(os.path.join(LENS, name + ".ini"))
    return pup, 1.0e-6 * wls[0], par["zoom"], fields[0], chains[0]


@pytest.mark.parametrize("name,n", [("SYN20", 1024), ("Ariel_AIRS-CH0", 1024), ("SYN20", 2048),
                                    ("Ariel_AIRS-CH0", 2048), ("SYN20", 4096)],
                         ids=["SYN20-1024", "AIRS-1024", "SYN20-2048", "AIRS-2048", "SYN20-4096"])
def test_end_to_end_vs_oracle(name, n):
    """run() through the frugal kernels with apertures riding on passes, every saved surface against
    the oracle; SYN20-4096 is the headline workload of bench.py, one wavefront of it."""
    import paos_amd.run as prun

    assert prun.FUSE_APERTURES == "auto"
    pup, wl, zoom, field, chain = _spec(name)
    stats = {}
    got = prun.run_batch(pup, [wl], n, zoom, field, [chain], outputs=("wfo", "amplitude"), stats=stats)[0]
    ref = oracle((name, n, 0), pup, wl, n, zoom, field, chain)
    compare(got, ref, (name, n))
    assert stats["fused_passes"] > 0
    # run() itself (the drop-in entry point) returns the same arrays as the batch of one
    if n <= 2048:
        single = prun.run(pup, wl, n, zoom, field, chain)
        for k in ref:
            assert np.array_equal(single[k]["wfo"], got[k]["wfo"]), (name, n, k)


def test_airs_wavelength_batch_vs_oracle():
    """BASELINE configs[2] (scaled to what the oracle finishes in seconds): a batch of Ariel_AIRS-CH0
    wavelengths spanning the channel -- every wavelength has its own chain (glass indices, the
    anamorphic prism magnification) -- through ONE sequence of launches; each item against the oracle."""
    from paos_amd.run import run_batch

    sweep = np.linspace(1.95, 3.9, 5)
    pup, par, wls, fields, chains = _airs(sweep)
    w = [1.0e-6 * x for x in wls]
    n = 1024
    got = run_batch(pup, w, n, par["zoom"], fields[0], chains, outputs=("wfo", "psf"))
    for i in range(len(w)):
        ref = oracle(("AIRS-sweep", n, i), pup, w[i], n, par["zoom"], fields[0], chains[i])
        compare(got[i], ref, ("AIRS batch", i))
        last = max(ref)
        assert abs(got[i][last]["power"] - float(np.sum(ref[last]["amplitude"] ** 2))) < 1e-11
    # two of them again at 2048^2 (the size configs[2] is quoted on), batched
    got = run_batch(pup, w[::4], 2048, par["zoom"], fields[0], chains[::4], outputs=("psf",))
    for j, i in enumerate(range(0, len(w), 4)):
        ref = oracle(("AIRS-sweep", 2048, i), pup, w[i], 2048, par["zoom"], fields[0], chains[i])
        compare(got[j], ref, ("AIRS batch 2048", i))


def _fgs1():
    from paos_amd.chains import parse_config_variant, read_wfe_table

    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Ariel_FGS-FGS1.ini"), unignore=("Z1",))
    _, _, _, table = read_wfe_table(WFE)
    return pup, par, wls, fields, chains, table


def test_fgs1_monte_carlo_vs_reference_vectors():
    """BASELINE configs[3] against the reference itself: Ariel_FGS-FGS1 with ``Z1`` un-ignored and WFE
    columns 0..3 injected (pipeline.py:116-129), 64^2, fixture made by tests/golden_tools/make_golden_r2.py."""
    from paos_amd.chains import inject_wfe
    from paos_amd.run import run, run_batch

    g = load_golden("r2_fgs1_mc.npz")
    pup, par, wls, fields, chains, table = _fgs1()
    assert pup == g["pup"] and wls[0] == g["wl_um"] and par["zoom"] == g["zoom"]
    mc = [inject_wfe(chains[0], table[:, c]) for c in range(4)]
    for c in range(4):
        assert np.array_equal(table[:, c], g[f"c{c}_draw_nm"])
    batch = run_batch(pup, [1.0e-6 * wls[0]] * 4, 64, par["zoom"], fields[0], mc, outputs=("wfo",))
    for c in range(4):
        nums = g[f"c{c}_nums"]
        assert sorted(batch[c]) == list(nums)
        tab = np.array([[batch[c][k][a] for a in ("wl", "dx", "dy", "wz", "distancetofocus", "fratio")] for k in nums])
        assert np.array_equal(tab, g[f"c{c}_table"]), c
        assert [batch[c][k]["propagator"] for k in nums] == list(g[f"c{c}_propagator"])
        assert rel_err(batch[c][nums[-1]]["wfo"], g[f"c{c}_wfo"]) < FIELD_TOL, c
    single = run(pup, 1.0e-6 * wls[0], 64, par["zoom"], fields[0], mc[2])
    assert rel_err(single[max(single)]["wfo"], g["c2_wfo"]) < FIELD_TOL
    zk = [k for k in single if "wfe" in single[k]]
    if zk and "c2_wfe" in g:
        assert rel_err(single[zk[0]]["wfe"].filled(0.0), g["c2_wfe"]) < 1e-13


@pytest.mark.parametrize("n", [512, 1024])
def test_fgs1_monte_carlo_batch_vs_oracle(n):
    """The same Monte-Carlo batch on the generic (512^2) and the frugal (1024^2) kernels, each draw
    against the oracle, plus the on-device PSF metrics the study uses."""
    from paos_amd.chains import inject_wfe
    from paos_amd.run import run_batch

    pup, par, wls, fields, chains, table = _fgs1()
    wl = 1.0e-6 * wls[0]
    mc = [inject_wfe(chains[0], table[:, c]) for c in range(4)]
    radii = np.geomspace(2.0, 128.0, 8)
    got = run_batch(pup, [wl] * 4, n, par["zoom"], fields[0], mc, outputs=("wfo",), metrics_radii_px=radii)
    for c in range(4):
        ref = oracle(("FGS1-mc", n, c), pup, wl, n, par["zoom"], fields[0], mc[c])
        compare(got[c], ref, ("FGS1 draw", n, c))
        last = max(ref)
        psf = ref[last]["amplitude"] ** 2
        m = got[c][last]["metrics"]
        assert abs(m["power"] - psf.sum()) < 1e-11 * psf.sum()
        yy, xx = np.mgrid[0:n, 0:n]
        d2 = (xx - n / 2) ** 2 + (yy - n / 2) ** 2
        for r, ee in zip(radii, m["encircled"]):
            assert abs(ee - psf[d2 <= r * r].sum()) < 1e-10 * psf.sum(), (c, r)
    # the draws differ: the batch is not four copies of one wavefront
    last = max(got[0])
    assert rel_err(got[0][last]["wfo"], got[1][last]["wfo"]) > 1e-6


@pytest.mark.parametrize("n", [1024, 2048])
def test_excite_fp32_fp64_study(n):
    """BASELINE configs[4]: Excite_TEL wavelength sweep, fp64 and fp32 against the oracle.  Stated
    bounds: fp64 PSF < 1e-10, fp32 PSF < 2e-5 of the peak (fields are stored and transformed in
    complex64; phase arguments stay fp64)."""
    from paos_amd.chains import parse_config_variant
    from paos_amd.run import run_batch

    sweep = np.linspace(1.0, 4.0, 512)[::170]  # 4 of the 512 wavelengths of the sweep
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Excite_TEL.ini"), sweep)
    w = [1.0e-6 * x for x in wls]
    r64 = run_batch(pup, w, n, par["zoom"], fields[0], chains, outputs=("wfo", "psf"))
    r32 = run_batch(pup, w, n, par["zoom"], fields[0], chains, outputs=("psf",), precision="fp32")
    worst32 = 0.0
    for i in range(len(w)):
        ref = oracle(("Excite", n, i), pup, w[i], n, par["zoom"], fields[0], chains[i])
        compare(r64[i], ref, ("Excite fp64", n, i))
        compare(r32[i], ref, ("Excite fp32", n, i), psf_tol=FP32_PSF_TOL, scalars=True)
        last = max(ref)
        e = rel_err(r32[i][last]["psf"], ref[last]["amplitude"] ** 2)
        worst32 = max(worst32, e)
        assert rel_err(r32[i][last]["psf"], r64[i][last]["psf"]) < FP32_PSF_TOL
    assert worst32 > 1e-9, "fp32 mode is expected to differ measurably from fp64"
