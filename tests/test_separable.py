"""The separable pass programs of round 4 (paos_amd/passes.py: SeparableCompiler) on the CPU: the reordering of an
aperture-to-aperture stretch into row factors then column factors is checked against the operator-by-operator
compiler (PassCompiler, which the other files of this directory pin to the reference's golden vectors) on the NumPy
model of the pass semantics -- SYN20, the shipped prescriptions and random chains -- plus the packing rules."""
import copy

import numpy as np
import pytest

from conftest import rel_err
from fakes import ModelDevice
from paos_amd import _lib
from paos_amd import passes as ppasses
from test_host_logic import FIELD, _model_run, _random_chain, _spec


class _Recorder(ModelDevice):
    """The model device, keeping every program it is handed."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.programs = []

    def run_passes(self, passes, blocks, **kw):
        self.programs.append((copy.deepcopy(passes), np.array(blocks), dict(kw)))
        return super().run_passes(passes, blocks, **kw)


def _both(spec, n, **kw):
    assert ppasses.SEPARABLE
    sep, _, st_sep = _model_run(spec, n, **kw)
    ppasses.SEPARABLE = False
    try:
        ref, _, st_ref = _model_run(spec, n, **kw)
    finally:
        ppasses.SEPARABLE = True
    return sep, ref, st_sep, st_ref


@pytest.mark.parametrize("name", ["SYN20", "Hubble_simple", "Excite_TEL", "Ariel_AIRS-CH0", "Ariel_FGS-FGS1"])
@pytest.mark.parametrize("light", [False, True])
def test_separable_programs_equal_operator_by_operator(name, light):
    """Every saved surface of the shipped prescriptions (every surface saved, and `light_output`: only the image plane,
    the long programs) from the separable programs and from the operator-by-operator ones: equal to rounding."""
    import paos_amd.run as prun

    spec = _spec(name)
    chain = copy.deepcopy(spec["chain"])
    for item in chain.values():
        item["save"] = (item["name"] == "IMAGE_PLANE") if light else True
    prun.FUSE_APERTURES = True
    try:
        sep, ref, st_sep, st_ref = _both(dict(spec, chain=chain), 128)
    finally:
        prun.FUSE_APERTURES = "auto"
    assert sorted(sep[0]) == sorted(ref[0])
    for k in sep[0]:
        assert rel_err(sep[0][k]["wfo"], ref[0][k]["wfo"]) < 1e-12, (name, k)


@pytest.mark.parametrize("seed", range(16))
def test_separable_programs_on_random_chains(seed):
    """Random prescriptions (tests/test_host_logic.py: _random_chain), two wavelengths per batch so that items disagree
    about which hops they take, apertures riding or stand-alone: every saved surface equal to ~1e-12."""
    import paos_amd.run as prun

    rng = np.random.default_rng(4000 + seed)
    chain = _random_chain(rng, int(rng.integers(6, 16)))
    wls = [1.0e-6, float(rng.choice([1.0e-6, 1.7e-6, 2.3e-6]))]
    spec = dict(pup=1.0, wl=wls[0], zoom=4, field=FIELD, chain=chain)
    prun.FUSE_APERTURES = bool(rng.integers(0, 2)) or "auto"
    try:
        try:
            sep, ref, st_sep, st_ref = _both(spec, 64, chains=[chain, chain], wls=wls)
        except (ValueError, AssertionError, TypeError) as exc:
            pytest.skip(f"the planner refuses this draw like the reference would ({type(exc).__name__}: {exc})")
    finally:
        prun.FUSE_APERTURES = "auto"
    for i in sep:
        assert sorted(sep[i]) == sorted(ref[i])
        for k in sep[i]:
            a, b = sep[i][k]["wfo"], ref[i][k]["wfo"]
            assert np.isfinite(a).all() and np.isfinite(b).all()
            assert rel_err(a, b) < 1e-11, (seed, i, k, rel_err(a, b), st_sep, st_ref)


def _syn20_programs(n=128):
    from paos_amd.chains import syn20_chain
    from paos_amd.run import _Item, _walk
    import paos_amd.run as prun

    dev = _Recorder(n, 1)
    prun.FUSE_APERTURES = True
    try:
        _walk(dev, [_Item(1.0, 1.0e-6, n, 4, FIELD)], [syn20_chain()], lambda *a: None, fresh=1.0 + 0.0j)
    finally:
        prun.FUSE_APERTURES = "auto"
    return dev.programs


def test_syn20_runs_rows_then_columns_between_apertures():
    """The structure bench.py's headline rests on: SYN20 with its apertures riding is ONE program of 24 passes; between
    two apertures the row passes (axis 0) come first, then the column passes (axis 1); an aperture rides in front of the
    first transform of the row pass that opens the next stretch -- or behind the last transform of a stretch with an odd
    number of transforms; every phase a row pass carries is the row factor of its operator (sy = 0), every phase of a
    column pass the column factor (sx = 0); the checkerboards are split into their halves."""
    programs = _syn20_programs()
    passes, blocks, _ = max(programs, key=lambda p: len(p[0]))
    assert len(passes) == 24 and all(p["axis"] in (0, 1) for p in passes)
    # stretches: cut in front of every pass that has an aperture in `pre`, behind every pass that has one in `mid`
    stretches, cur = [], []
    for ps in passes:
        if any(op[0] == _lib.PW_MASK for op in ps["pre"]) and cur:
            stretches.append(cur)
            cur = []
        cur.append(ps)
        if any(op[0] == _lib.PW_MASK for op in ps["mid"]):
            stretches.append(cur)
            cur = []
    if cur:
        stretches.append(cur)
    assert len(stretches) == 6  # stop -> relay 1, four relay-to-relay stretches, the last relay -> field stop
    for st in stretches:
        axes = [p["axis"] for p in st]
        assert axes == sorted(axes), axes  # rows first, then columns
        assert 0 in axes and 1 in axes
    masks_pre = [p for p in passes if any(op[0] == _lib.PW_MASK for op in p["pre"])]
    masks_mid = [p for p in passes if any(op[0] == _lib.PW_MASK for op in p["mid"])]
    assert len(masks_pre) == 5 and all(p["axis"] == 0 for p in masks_pre)
    assert len(masks_mid) == 1 and masks_mid[0] is passes[-1] and passes[-1].get("fft2", -1) < 0
    for ps in passes:
        for slot in ("pre", "mid"):
            for kind, flags, blk in ps[slot]:
                if kind in (_lib.PW_QPHASE_CENTRED, _lib.PW_QPHASE_NATURAL):
                    on = blocks[blk][:, 0] != 0.0
                    dead = blocks[blk][on, 2 if ps["axis"] == 0 else 1]
                    assert np.all(dead == 0.0), "a phase factor of the other axis rides on this pass"
                if kind == _lib.PW_SIGN:
                    assert flags in (_lib.PWF_X_ONLY, _lib.PWF_Y_ONLY)
    assert not any(p["axis"] == -1 for prog in programs for p in prog[0]), "a stand-alone sweep was needed"


def test_a_program_that_ends_on_an_even_stretch_needs_no_extra_sweep():
    """stw + wts (two transforms per axis) and then the end of the program: the half checkerboard that trails the column
    factors has no later pass to ride on -- the column chain is packed [1][1] instead of [2] so that it sits behind the
    last transform; no stand-alone pointwise pass (a whole-grid sweep) is emitted."""
    comp = ppasses.SeparableCompiler(1, 64)
    blk = [[1.0, 1.0e-3, 1.0e-3, 2.0, 1.0]]
    comp.stw(blk, False)
    comp.wts([[1.0, 2.0e-3, 2.0e-3, -1.5, 1.0]], True)
    passes, blocks = comp.program()
    assert [p["axis"] for p in passes] == [0, 1, 1]
    assert passes[0].get("fft2", -1) >= 0 and passes[1].get("fft2", -1) < 0 and passes[2].get("fft2", -1) < 0
    assert passes[2]["mid"] and passes[2]["mid"][-1][0] == _lib.PW_SIGN
    # the same with an aperture behind it: the aperture takes the free slot too
    comp.stw(blk, False)
    comp.wts([[1.0, 2.0e-3, 2.0e-3, -1.5, 1.0]], True)
    comp.aperture([([1.0, 32.0, 32.0, 8.0, 8.0, 0.0, 0.0, 1.0], _lib.SHAPE_ELLIPSE)])
    passes, blocks = comp.program()
    assert [p["axis"] for p in passes] == [0, 1, 1]
    assert passes[2]["mid"][-1][0] == _lib.PW_MASK


def test_open_takes_mask_counts_the_transforms_since_the_last_aperture():
    comp = ppasses.SeparableCompiler(1, 64)
    assert not comp.open_takes_mask()
    comp.stw([[1.0, 1.0e-3, 1.0e-3, 2.0, 1.0]], False)
    assert comp.open_takes_mask()          # one transform per axis: the last column pass has its second slot free
    comp.ptp([[1.0, 1.0, 1.0, 0.3, -1.0]])
    assert comp.open_takes_mask()          # three
    comp.wts([[1.0, 2.0e-3, 2.0e-3, -1.5, 1.0]], True)
    assert not comp.open_takes_mask()      # four: both passes of the column chain are full
    comp.aperture([([1.0, 32.0, 32.0, 8.0, 8.0, 0.0, 0.0, 1.0], _lib.SHAPE_ELLIPSE)])
    assert not comp.open_takes_mask()      # one aperture per pass
    comp.program()


def test_identities_apply_to_the_operator_stream():
    """ptp(+d) ptp(-d) cancel, consecutive ptp share their transforms, a wts and the stw that undoes it go -- as in
    PassCompiler, before the stream is split into row and column factors."""
    comp = ppasses.SeparableCompiler(2, 64)
    h = np.array([[1.0, 3.0, 3.0, 0.25, -1.0], [1.0, 3.0, 3.0, 0.5, -1.0]])
    comp.ptp(h)
    comp.ptp(h * [1.0, 1.0, 1.0, -1.0, 1.0])
    assert not comp.pending()
    comp.ptp(h)
    comp.ptp(h * [1.0, 1.0, 1.0, 2.0, 1.0])
    passes, blocks = comp.program()
    assert len(passes) == 2 and [p["axis"] for p in passes] == [0, 1]
    assert sum(op[0] == _lib.PW_QPHASE_NATURAL for op in passes[0]["mid"]) == 2
    w = np.array([[1.0, 1.0e-3, 1.0e-3, 2.0, 1.0], [1.0, 1.0e-3, 1.0e-3, 3.0, 1.0]])
    comp.wts(w, [False, False])
    comp.stw(w * [1.0, 1.0, 1.0, -1.0, 1.0], [True, True])
    assert not comp.pending()
