"""GPU tests added in round 2: Grid Sag / PSD phase screens against reference vectors, the power-ticket
ring with more saved surfaces than the library has slots, parameter-arena growth, run() of a chain that
carries a PSD surface."""
import copy
import os

import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

FIELD_TOL = 1e-11
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _wfo(g, anam=False, precision="fp64"):
    from paos_amd.wfo import WFO

    w = WFO(1.0, 2.0e-6, 64, 4, precision=precision)
    if anam:
        w.Magnification(1.25, 0.8)
    w._wfo = g["u0"]
    return w


def test_grid_sag_vs_reference_vectors():
    """WFO.grid_sag (wfo.py:656-871): identity, anamorphic sampling, zero padding, cropping and the
    sub-pixel Fourier shift -- field after the phase multiply against the reference's."""
    g = load_golden("r2_phase_maps.npz")
    for tag, anam, kw in (("same", False, {}), ("same_anam", True, {}), ("pad", False, {}), ("crop", False, {}),
                          ("shift", False, dict(xdec=1.5, ydec=-0.25))):
        w = _wfo(g, anam)
        sag = g[f"gs_{tag}_sag"]
        ret = w.grid_sag(sag.copy(), sag.shape[1], sag.shape[0], w.dx, w.dy, **kw)
        assert np.array_equal(ret.filled(0.0), g[f"gs_{tag}_wfe"]), tag
        assert np.array_equal(np.ma.getmaskarray(ret), g[f"gs_{tag}_mask"]), tag
        e = rel_err(w.wfo, g[f"gs_{tag}_u"])
        assert e < 1e-14, (tag, e)


def test_psd_vs_reference_vectors():
    """WFO.psd (wfo.py:873-949, psd.py:85-160) with the legacy global generator seeded like the fixture."""
    g = load_golden("r2_phase_maps.npz")
    cases = {"powerlaw": dict(A=7.0, B=0.0, C=1.5, fknee=1.0, fmin=None, fmax=None, SR=0.0, units="nm"),
             "knee_sr": dict(A=12.0, B=1.0, C=2.2, fknee=3.0, fmin=0.5, fmax=6.0, SR=2.0, units="nm")}
    for tag, kw in cases.items():
        for anam in (False, True):
            w = _wfo(g, anam)
            np.random.seed(1234)
            ret = w.psd(**kw)
            key = f"psd_{tag}{'_anam' if anam else ''}"
            assert np.array_equal(np.ma.filled(ret, 0.0), g[key + "_wfe"]), key
            e = rel_err(w.wfo, g[key + "_u"])
            assert e < 1e-14, (key, e)
    # fp32 fields take the same screen (phase formed in fp64, stored in complex64)
    w = _wfo(g, False, "fp32")
    np.random.seed(1234)
    w.psd(**cases["powerlaw"])
    assert 1e-9 < rel_err(w.wfo, g["psd_powerlaw_u"]) < 1e-6


def test_phase_map_rejects_bad_input():
    from paos_amd import _lib

    dev = _lib.DeviceFields(64, 2)
    dev.fill(1.0)
    with pytest.raises(ValueError):
        dev.phase_map(0, np.zeros((32, 32)), 1e-6)
    bad = np.zeros((64, 64))
    bad[3, 3] = np.nan
    with pytest.raises(_lib.PaosHipError, match="non-finite"):
        dev.phase_map(0, bad, 1e-6)
    with pytest.raises(_lib.PaosHipError):
        dev.phase_map(2, np.zeros((64, 64)), 1e-6)
    with pytest.raises(_lib.PaosHipError, match="wavelength"):
        dev.phase_map(0, np.zeros((64, 64)), 0.0)
    # item 1 only: item 0 untouched
    dev.phase_map(1, np.full((64, 64), 0.25e-6), 1e-6)
    assert np.array_equal(dev.download(0), np.ones((64, 64), dtype=complex))
    assert rel_err(dev.download(1), np.full((64, 64), 1j)) < 1e-15
    dev.close()


def test_run_with_a_psd_surface_matches_the_stepwise_wfo():
    """lens_file_TA_Ground_PSD.ini end to end: run() (fused passes, the PSD screen as a chain breaker)
    equals driving WFO method by method with the same seed; the screen is the `wfe` of the saved surface."""
    from paos_amd.parse_config import parse_config
    from paos_amd.run import run

    pup, par, wls, fields, chains = parse_config(os.path.join(DATA, "lens", "lens_file_TA_Ground_PSD.ini"))
    chain = chains[1]
    np.random.seed(7)
    n = par["grid_size"]  # 1024: the PSD band of the file (fmax = 500 / m) needs the file's own sampling
    ret = run(pup, 1e-6 * wls[1], n, par["zoom"], fields[0], chain)
    psd_num = [k for k, it in chain.items() if it["type"] == "PSD"][0]
    assert psd_num in ret and "wfe" in ret[psd_num]
    wfe = ret[psd_num]["wfe"]
    assert wfe.shape == (n, n) and np.std(np.ma.filled(wfe, 0.0)) > 0
    # the same chain with the PSD surface replaced by nothing differs exactly by that phase screen
    plain = {k: v for k, v in chain.items() if k != psd_num}
    base = run(pup, 1e-6 * wls[1], n, par["zoom"], fields[0], plain)
    # surfaces before the PSD are identical, the PSD surface itself is base * exp(2 pi i wfe / wl)
    before = [k for k in ret if k < psd_num]
    for k in before:
        assert np.array_equal(ret[k]["wfo"], base[k]["wfo"])
    np.random.seed(7)
    again = run(pup, 1e-6 * wls[1], n, par["zoom"], fields[0], chain)
    for k in ret:
        assert np.array_equal(ret[k]["wfo"], again[k]["wfo"])
    last = max(ret)
    assert rel_err(ret[last]["wfo"], base[last]["wfo"]) > 1e-6


def test_more_saved_surfaces_than_ticket_slots():
    """A chain that saves more than PAOS_NORM_SLOTS surfaces: run_batch drains the power tickets early and
    every power is right (they used to wrap silently); the raw ABI refuses to overrun the ring."""
    from paos_amd import _lib
    from paos_amd.abcd import ABCD
    from paos_amd.run import run_batch

    n_surf = _lib.NORM_SLOTS + 9
    chain = {}
    for num in range(1, n_surf + 1):
        item = {"num": num, "type": "Standard", "name": f"S{num}", "is_stop": num == 1, "save": True,
                "ABCDt": ABCD(thickness=0.05 if num % 2 else 0.0, curvature=0.0),
                "ABCDs": ABCD(thickness=0.05 if num % 2 else 0.0, curvature=0.0)}
        if num == 1 or num % 7 == 0:
            rad = 0.5 if num == 1 else 0.5 - 0.002 * num
            item["aperture"] = {"shape": "elliptical", "type": "aperture", "xrad": rad, "yrad": rad, "xc": 0.0, "yc": 0.0}
        chain[num] = item
    res = run_batch(1.0, [1.0e-6, 1.3e-6], 128, 4, {"us": 0.0, "ut": 0.0}, [chain, copy.deepcopy(chain)],
                    outputs=("psf",))
    for r in res:
        assert sorted(r) == list(range(1, n_surf + 1))
        for k, rec in r.items():
            assert abs(rec["power"] - rec["psf"].sum()) < 1e-12, k
        powers = [r[k]["power"] for k in sorted(r)]
        assert abs(powers[0] - 1.0) < 1e-13 and powers[-1] < powers[0]  # the later apertures clip light
    dev = _lib.DeviceFields(64, 1)
    dev.fill(1.0)
    tickets = [dev.norm2_enqueue() for _ in range(_lib.NORM_SLOTS)]
    with pytest.raises(_lib.PaosHipError, match="outstanding"):
        dev.norm2_enqueue()
    assert dev.norm2_fetch(tickets[0])[0] == 64.0 * 64.0
    with pytest.raises(_lib.PaosHipError, match="not outstanding"):
        dev.norm2_fetch(tickets[0])
    tickets.append(dev.norm2_enqueue())  # the freed slot is usable again
    for t in tickets[1:]:
        assert dev.norm2_fetch(t)[0] == 64.0 * 64.0
    dev.close()


def test_parameter_arena_grows_for_a_long_program():
    """One fused program whose parameter records exceed the arena's initial capacity (large batch, a long
    run of lens / propagation surfaces without a breaker): the arena grows before the first push instead of
    wrapping under the live block table; results equal the same work in short programs."""
    from paos_amd import _lib
    from paos_amd.planner import PilotBeam

    n, nb = 64, 96
    dev = _lib.DeviceFields(n, nb)
    rng = np.random.default_rng(3)
    u0 = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    wls = [1.0e-6 * (1 + 0.01 * i) for i in range(nb)]

    def program(count):
        beams = [PilotBeam(1.0, wl, n, 4) for wl in wls]
        passes, blocks = [], []
        for s in range(count):
            blocks.append([b.ptp(0.01 * (s + 1)) for b in beams])
        return beams, blocks

    for i in range(nb):
        dev.upload(i, u0)
    _, blocks = program(400)
    # 400 ptp = 1200 passes in ONE program: ~1200 x 96 records of FrugalItem-sized pushes >> 8 MiB
    from paos_amd.passes import PassCompiler

    comp = PassCompiler(nb, n)
    null_lens = [[1.0, 1.0e-3, 1.0e-3, 0.0, -1.0]] * nb  # exp(i 0) = 1 exactly: keeps consecutive ptp from merging (round 3)
    for blk in blocks:
        comp.ptp(blk)
        comp.lens(null_lens)
    npass = comp.flush(dev)
    assert npass >= 800
    got = dev.download(5)
    for i in range(nb):
        dev.upload(i, u0)
    for blk in blocks:  # the same, one operator per program
        dev.ptp(blk)
    ref = dev.download(5)
    assert rel_err(got, ref) < 1e-12
    dev.close()


@pytest.mark.parametrize("n", [1024, 2048])
def test_dead_line_pruning_changes_nothing(n):
    """Rows / columns an aperture has zeroed are skipped by the following passes (tiles not processed,
    loads not issued).  With the pruning switched off every tile is processed: the saved fields must be
    equal (zeros are zeros), for apertures riding on passes, stand-alone apertures (the live-row promise
    of paos_run_passes_live), an off-axis field point (decentred boxes), a batch whose items have
    different boxes, and fp32."""
    import paos_amd.run as prun
    from paos_amd import _lib
    from paos_amd.chains import parse_config_variant, syn20_chain, syn20_wavelength
    from paos_amd.parse_config import parse_config

    def both(args, chains, wls, precision="fp64", **kw):
        out = []
        for on in (True, False):
            dev = _lib.DeviceFields(n, len(chains), precision)
            dev.set_pruning(on)
            stats = {}
            out.append(prun.run_batch(args[0], wls, n, args[1], args[2], chains, outputs=("wfo",), dev=dev,
                                      precision=precision, stats=stats, **kw))
            dev.close()
        for a, b in zip(*out):
            assert sorted(a) == sorted(b)
            for k in a:
                assert np.array_equal(a[k]["wfo"], b[k]["wfo"]), k
                assert a[k]["power"] == b[k]["power"]
        return out[0]

    on_axis = {"us": 0.0, "ut": 0.0}
    wls = [syn20_wavelength(k) for k in (0, 200, 400)]
    both((1.0, 4, on_axis), [syn20_chain() for _ in wls], wls)
    both((1.0, 4, {"us": 5.0e-5, "ut": -2.0e-5}), [syn20_chain()], [1.0e-6])
    both((1.0, 4, on_axis), [syn20_chain()], [1.0e-6], precision="fp32") if n >= 2048 else None
    pup, par, w, fields, chains = parse_config(os.path.join(DATA, "lens", "Hubble_simple.ini"))
    both((pup, par["zoom"], fields[0]), [chains[0]], [1e-6 * w[0]])
    pup, par, w, fields, chains = parse_config_variant(os.path.join(DATA, "lens", "Ariel_AIRS-CH0.ini"), [1.95, 3.9])
    both((pup, par["zoom"], fields[0]), chains, [1e-6 * x for x in w])
    # stand-alone apertures only (no line records): the promise travels through paos_run_passes_live
    old = prun.FUSE_APERTURES
    prun.FUSE_APERTURES = False
    try:
        both((1.0, 4, on_axis), [syn20_chain() for _ in wls], wls)
    finally:
        prun.FUSE_APERTURES = old


def test_live_row_promise_is_validated():
    from paos_amd import _lib
    from paos_amd.passes import PassCompiler
    from paos_amd.planner import PilotBeam

    dev = _lib.DeviceFields(1024, 1)
    dev.fill(1.0)
    comp = PassCompiler(1, 1024)
    comp.ptp([PilotBeam(1.0, 1e-6, 1024, 4).ptp(1.0)])
    passes, blocks = comp.program()
    with pytest.raises(_lib.PaosHipError, match="live row range"):
        dev.run_passes(passes, blocks, live_rows=[[10, 2000]])
    with pytest.raises(ValueError):
        dev.run_passes(passes, blocks, live_rows=[[0, 10], [0, 10]])
    # a field that really is zero outside rows [384, 640): same result with and without the promise
    u = np.zeros((1024, 1024), dtype=complex)
    rng = np.random.default_rng(1)
    u[384:640] = rng.standard_normal((256, 1024)) + 1j * rng.standard_normal((256, 1024))
    dev.upload(0, u)
    dev.run_passes(passes, blocks, live_rows=[[384, 640]])
    a = dev.download(0)
    dev.upload(0, u)
    dev.run_passes(passes, blocks)
    b = dev.download(0)
    assert np.array_equal(a, b)
    dev.close()


def test_rccl_transport_single_rank():
    """paos_comm over RCCL on the one GPU of this box: dlopen of librccl, ncclCommInitRank, and the three
    collectives the fan-out uses (broadcast of a packed work description, ragged gather, MAX) -- the
    N > 1 path differs only in the number of ranks.  run_sharded with that communicator == run_batch."""
    from paos_amd import wire
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.comm import Comm
    from paos_amd.dist import broadcast_work, run_sharded
    from paos_amd.run import run_batch

    comm = Comm(1, 0, 0, "rccl")
    try:
        work = {"wavelengths": [1e-6, 2e-6], "chains": [syn20_chain(), syn20_chain()]}
        blob = wire.dumps(work)
        assert comm.bcast_blob(blob, root=0) == blob  # through a device buffer and ncclBroadcast
        assert broadcast_work(work, comm) is work
        parts = comm.allgather_scalars(np.arange(5.0))
        assert len(parts) == 1 and np.array_equal(parts[0], np.arange(5.0))
        assert comm.max(2.5) == 2.5
        comm.barrier()
        field = {"us": 0.0, "ut": 0.0}
        wls = [syn20_wavelength(k * 100) for k in range(3)]
        chains = [syn20_chain() for _ in wls]
        got = run_sharded(1.0, wls, 128, 4, field, chains, batch=2, outputs=("psf",), comm=comm)
        ref = run_batch(1.0, wls, 128, 4, field, chains, outputs=("psf",))
        assert [i for i, _ in got] == [0, 1, 2]
        for (i, r), q in zip(got, ref):
            assert np.array_equal(r[20]["psf"], q[20]["psf"]) and r[20]["power"] == q[20]["power"]
    finally:
        comm.close()


def test_line_records_never_overflow_when_the_host_accepts_an_aperture():
    """An aperture rides on a pass as per-line records with room for 192 partial pixels per side
    (csrc/frugal_pass.h: kMaskW).  The host accepts an ellipse only if 2 a sqrt(3 / b) + 8 <= 192 (and the
    same with a, b swapped) -- the tip row of an ellipse holds at most 2 a sqrt(2 / b) partial pixels, so the
    test is conservative.  Scan ellipses right at that limit, decentred and at grid edges, along both pass
    axes: the overflow counter the kernel keeps must stay zero (paos_sync raises otherwise), and the result
    must equal the stand-alone aperture kernel's."""
    import math

    from paos_amd import _lib
    from paos_amd.aperture import EllipticalAperture
    from paos_amd.passes import PassCompiler
    from paos_amd.planner import PilotBeam
    from paos_amd.run import _aperture_fits_line_records

    n, nb = 4096, 8
    rng = np.random.default_rng(11)
    dev = _lib.DeviceFields(n, nb)
    tried = 0
    for trial in range(6):
        handles = []
        while len(handles) < nb:
            b = float(rng.choice([2.0, 2.5, 4.0, 9.0, 30.0, 200.0, 1500.0]))
            a_max = (192.0 - 8.0) / (2.0 * math.sqrt(3.0 / b))
            a = float(min(a_max * rng.uniform(0.97, 1.0), 1900.0))
            if trial % 2:
                a, b = b, a
            xc, yc = rng.uniform(-100.0, n + 100.0), rng.uniform(-100.0, n + 100.0)
            if rng.random() < 0.5:
                xc, yc = n / 2 + rng.uniform(-3, 3), n / 2 + rng.uniform(-3, 3)
            h = EllipticalAperture((xc, yc), a, b)
            if _aperture_fits_line_records(h, False, n, "fp64"):
                handles.append(h)
        tried += nb
        dev.fill(1.0 + 0.5j)
        comp = PassCompiler(nb, n)
        comp.aperture([(h.block(obscuration=False), _lib.SHAPE_ELLIPSE) for h in handles])
        blk = [PilotBeam(1.0, 1.0e-6, n, 4).ptp(0.25)] * nb
        comp.ptp(blk)
        comp.flush(dev)
        dev.sync()  # raises if a line record overflowed
        got = dev.download(trial % nb)
        dev.fill(1.0 + 0.5j)
        dev.aperture(_lib.SHAPE_ELLIPSE, [h.block(obscuration=False) for h in handles])
        dev.ptp(blk)
        assert rel_err(got, dev.download(trial % nb)) < 1e-13
    assert tried == 48
    dev.close()


def test_items_with_identical_apertures_share_their_records():
    """A batch whose items see the same aperture on the same sampling (Monte-Carlo draws at one wavelength;
    every item at the entrance pupil) renders one set of line records and sums the stop power once.
    Each item of a mixed batch -- three at one wavelength with different wavefront errors, one at another
    wavelength -- must equal the same item propagated alone."""
    import paos_amd.run as prun
    from paos_amd.chains import syn20_chain, syn20_coefficients, syn20_wavelength

    n = 1024
    on_axis = {"us": 0.0, "ut": 0.0}
    wls = [1.0e-6, 1.0e-6, syn20_wavelength(300), 1.0e-6]
    chains = []
    for k in range(4):
        ch = syn20_chain()
        coef = np.array(syn20_coefficients(), dtype=float) * (1.0 + 0.25 * k)
        for item in ch.values():
            if item.get("type") == "Zernike":
                item["Z"] = coef.copy()
        chains.append(ch)
    together = prun.run_batch(1.0, wls, n, 4, on_axis, [copy.deepcopy(c) for c in chains], outputs=("wfo",))
    for k in range(4):
        alone = prun.run_batch(1.0, [wls[k]], n, 4, on_axis, [copy.deepcopy(chains[k])], outputs=("wfo",))[0]
        assert sorted(alone) == sorted(together[k])
        for num in alone:
            assert np.array_equal(alone[num]["wfo"], together[k][num]["wfo"]), (k, num)
            assert alone[num]["power"] == together[k][num]["power"], (k, num)
    assert not np.array_equal(together[0][max(together[0])]["wfo"], together[1][max(together[1])]["wfo"])


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_psf_keep_power_equals_the_two_separate_sweeps(precision):
    """paos_psf_keep_power = paos_psf_keep + paos_norm2_enqueue with one read of the field: same PSF, and the
    same sum to the last bit (same grid, same order of additions)."""
    from paos_amd import _lib

    n, nb = 1024, 3
    rng = np.random.default_rng(5)
    dev = _lib.DeviceFields(n, nb, precision)
    try:
        for i in range(nb):
            dev.upload(i, (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * (i + 1))
        dev.psf_keep()
        separate = [dev.psf_fetch(i) for i in range(nb)]
        power = dev.norm2_fetch(dev.norm2_enqueue())
        fused_power = dev.norm2_fetch(dev.psf_keep_power())
        assert np.array_equal(power, fused_power)
        for i in range(nb):
            assert np.array_equal(dev.psf_fetch(i), separate[i])
        assert abs(power[0] - separate[0].sum()) <= 1e-12 * power[0]
    finally:
        dev.close()
