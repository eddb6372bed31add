"""Pass compiler: turn a run of lens / stw / ptp / wts operators into fused HBM passes.

Every operator of the reference between two "breakers" (aperture, stop, Zernike,
saving a surface) is a chain of diagonal multiplications and 1-D transforms along
the rows or the columns of the field:

    lens : diag(exp(-2 pi i c (x^2+y^2)))                       wfo.py:359-366
    stw  : S . Qc . F_B F_A . S          (F = fft or ifft)        wfo.py:491-509
    wts  : S . F_B F_A . S . P                                    wfo.py:528-545
    ptp  : F_A^-1 F_B^-1 . H . F_B F_A                            wfo.py:462-472

where {A, B} = {rows, columns} in EITHER order (a 2-D transform is separable), S is the
checkerboard that replaces the fftshift / ifftshift pair and Qc / P / H are quadratic
phases.  One GPU pass can do ``diag, F_A, diag, F_A', diag`` on a resident tile, so the
compiler picks A for each operator = the axis the previous operator ended on and glues
the boundary transforms of neighbours into one pass.  Cost per 2-D transform drops from
two passes (plus one per lens) to about one.

The compiler is batch-aware: every operator carries one parameter block per batch item
whose first entry is an enable flag, so items whose planners disagree (a ptp skipped
below wl/1000, II vs OI regimes) still share the pass sequence.
"""
import numpy as np

from . import _lib

import os

_OFF = [0.0, 0.0, 0.0, 0.0, 0.0]
MAX_MERGED_PTP = 3  # transfer functions one middle pass can carry (kFrugalMaxMid of csrc/frugal_pass.h)
# The two identities the compiler uses on consecutive ptp (see PassCompiler.ptp): fft2(ifft2(X)) = X and
# H(-d) H(d) = 1.  PAOS_PTP_ALGEBRA=0 (or setting this to False) runs every ptp of the reference as its own three
# passes again -- same results to rounding, 44 instead of 24 passes for SYN20; bench.py reports that rate beside the
# headline one.
PTP_ALGEBRA = os.environ.get("PAOS_PTP_ALGEBRA", "1") != "0"
# The axis of a pass that has no open pass to glue onto (the first pass of a program).  Every later pass alternates
# from there, so this decides which of the chain's passes run along rows and which along columns.
# Largest residual phase (rad, at the corner of the grid) a wts -> stw pair may leave behind and still be dropped as
# the identity (PassCompiler._undoes): a tenth of the 1e-10 field gate the parity tests assert.
PARITY_GATE = 1.0e-10
UNDO_MAX_RESIDUAL = 1.0e-11  # = PARITY_GATE / 10
# A pair whose transforms cancel but whose two phases leave MORE than that (the separately computed coefficients of a
# real prescription differ by a few 1e-15 of a 1e4 rad corner phase: 3e-11 ... 9e-11 rad for Ariel_FGS-FGS1 at 4096^2)
# still loses its two transforms -- they are exact inverses whatever the phases do -- and the residual phase is KEPT,
# as one more diagonal operator, so nothing above UNDO_MAX_RESIDUAL is ever dropped.  Beyond this many radians the two
# hops are not each other's inverse in any sense (another distance, a magnification in between) and both run.
UNDO_COMPENSATE_MAX = 1.0e-6
FIRST_AXIS = 1 if os.environ.get("PAOS_FIRST_AXIS", "0") == "1" else 0


def _rows(rows, batch):
    """[batch][5] array of parameter blocks from either such an array (enable = 0: step off) or a list
    of 5-double rows with None for "off"."""
    if isinstance(rows, np.ndarray):
        if rows.shape != (batch, 5):
            raise ValueError(f"expected blocks of shape {(batch, 5)}, got {rows.shape}")
        return rows
    if len(rows) != batch:
        raise ValueError("one block (or None) per batch item is required")
    return np.array([r if r is not None else _OFF for r in rows], dtype=np.float64).reshape(batch, 5)


def _aperture_rows(records, batch):
    """The two [batch][5] block sets of an aperture operator from per-item records (None, or the 8-double block of
    paos_aperture plus the shape code) -- or, for a batch planned with array arithmetic (run.py: _BatchApertures), from
    the ([batch][8] blocks, [batch] shape codes) pair itself.  None when no item carries an aperture."""
    if isinstance(records, tuple) and len(records) == 2 and isinstance(records[0], np.ndarray):
        blocks, codes = records
        if blocks.shape != (batch, 8):
            raise ValueError("aperture blocks must be [batch][8]")
        on = blocks[:, 0] != 0.0
        if not on.any():
            return None
        first = np.ascontiguousarray(blocks[:, :5])
        second = np.zeros((batch, 5), dtype=np.float64)
        second[:, :3] = blocks[:, 5:8]
        second[:, 3] = codes
        first[~on] = 0.0
        second[~on] = 0.0
        return first, second
    if all(r is None for r in records):
        return None
    first = [list(r[0][:5]) if r is not None else None for r in records]
    second = [[r[0][5], r[0][6], r[0][7], float(r[1]), 0.0] if r is not None else None for r in records]
    return _rows(first, batch), _rows(second, batch)


class PassCompiler:
    def __init__(self, batch, n):
        self.batch, self.n = int(batch), int(n)
        self._reset()

    def _reset(self):
        self.blocks = []    # each: [batch][5] array
        self.passes = []    # emitted, in order
        self.open = None    # last pass, if its second transform slot is still free
        self.tail = []      # pointwise operators not yet attached to a pass
        self.last_ptp = None  # the ptp just emitted, while another one may still merge with it (see ptp)
        self.last_single = None  # the stw / wts just emitted, while the next operator may still undo it (see _single)

    # ---- helpers -------------------------------------------------------------------
    def _block(self, rows):
        self.blocks.append(_rows(rows, self.batch))
        return len(self.blocks) - 1

    def _derived(self, arr, v1=0.0, v2=0.0, v3=0.0, v4=0.0):
        """Block with the enable flags of ``arr`` and a payload (scalars, or one value per item)."""
        out = np.zeros((self.batch, 5), dtype=np.float64)
        on = arr[:, 0] != 0.0
        for col, v in ((1, v1), (2, v2), (3, v3), (4, v4)):
            out[:, col] = v
        out[:, 0] = 1.0
        out[~on] = 0.0
        return self._block(out)

    def _standalone(self, ops):
        for i in range(0, len(ops), _lib.MAX_PW):
            self.passes.append({"axis": -1, "pre": list(ops[i:i + _lib.MAX_PW])})

    def _close_open(self):
        """Attach pending pointwise work to the open pass (or emit it alone)."""
        if self.open is not None:
            if len(self.open["mid"]) + len(self.tail) <= _lib.MAX_PW:
                self.open["mid"] += self.tail
                self.tail = []
            self.open = None
        if self.tail:
            self._standalone(self.tail)
            self.tail = []

    def _first_pass(self, pre_ops, ctl):
        """First 1-D pass of an operator: glue onto the open pass when there is one."""
        o = self.open
        if o is not None and len(o["mid"]) + len(self.tail) + len(pre_ops) <= _lib.MAX_PW:
            o["mid"] += self.tail + pre_ops
            o["fft2"] = ctl
            self.tail = []
            self.open = None
            return o["axis"]
        if o is not None:
            self._close_open()
        pre = list(pre_ops)
        if len(self.tail) + len(pre) <= _lib.MAX_PW:
            pre = self.tail + pre  # pending lens phases ride on this pass's load
            self.tail = []
        else:
            self._close_open()
        self.passes.append({"axis": FIRST_AXIS, "fft1": ctl, "pre": pre, "mid": [], "post": []})
        return FIRST_AXIS

    def _open_pass(self, axis, ctl, mid_ops):
        p = {"axis": axis, "fft1": ctl, "fft2": -1, "pre": [], "mid": list(mid_ops), "post": []}
        self.passes.append(p)
        self.open = p

    # ---- operators (rows = one 5-double block or None per batch item) --------------------
    def aperture(self, records):
        """records: per item None or the 8-double aperture block of paos_aperture
        [enable, xc, yc, a|w, b|h, theta, obscuration, subpixels] plus the shape code.  The
        mask rides on the next pass (its weight map is rendered right before it)."""
        rows = _aperture_rows(records, self.batch)
        if rows is None:
            return
        if any(op[0] == _lib.PW_MASK for op in self.tail) or (
                self.open is not None and any(op[0] == _lib.PW_MASK for op in self.open["mid"])):
            self._close_open()  # one aperture per pass
        first = self._block(rows[0])
        self._block(rows[1])
        self.tail.append((_lib.PW_MASK, 0, first))

    def lens(self, rows):
        arr = _rows(rows, self.batch)
        if not arr[:, 0].any():
            return
        self.tail.append((_lib.PW_QPHASE_CENTRED, _lib.PWF_MUL2PI, self._block(arr)))

    def _single(self, rows, inverse, kind):
        arr = _rows(rows, self.batch)
        if not arr[:, 0].any():
            return
        inv = np.where(np.asarray(inverse, dtype=bool), 1.0, 0.0) * np.ones(self.batch)
        # A wts straight into an stw that undoes it (wfo.py:566-570: two outside-to-outside hops in a row, the second
        # starting where the first ended, with no power, aperture or saved surface between them -- the flat windows
        # and fold mirrors of a real prescription): the first ends with q_w = exp(i pi r^2 / (lambda dz)), FFT2 towards
        # z; the second starts with FFT2^-1 back to the waist and q_s = exp(-i pi lambda dz f^2), which on the same
        # pixels is 1 / q_w.  fft2(ifft2(X)) = X, the four checkerboards and the two 1/N with the N^2 of the
        # transform pair are exactly 1, and the two phases multiply to 1 up to the rounding noise of their separately
        # rounded arguments (~1e-12 rad at 1e4 rad, which the reference carries and this does not): both operators
        # go, and the stw in front meets the wts behind -- the two hops become the one hop they are.
        ls = self.last_single if PTP_ALGEBRA else None
        if (ls is not None and kind == "stw" and ls["kind"] == "wts" and self.open is ls["open"] and not self.tail and
                self.open["mid"] == ls["open_mid"]):
            if self._undoes(ls["arr"], ls["inv"], arr, inv):
                self._restore(ls["undo"])
                return
            if self._undoes(ls["arr"], ls["inv"], arr, inv, max_residual=UNDO_COMPENSATE_MAX):
                leftovers = self._residual_phase(ls["arr"], arr)
                self._restore(ls["undo"])
                for blk in leftovers:
                    self.tail.append((_lib.PW_QPHASE_CENTRED, 0, self._block(blk)))
                self.last_single = None
                return
        undo = self._snapshot()
        par = self._block(arr)
        ctl = self._derived(arr, v1=inv)
        scl = self._derived(arr, v3=1.0 / self.n)
        sign = (_lib.PW_SIGN, 0, par)
        phase = (_lib.PW_QPHASE_CENTRED, 0, par)
        pre = [phase, sign] if kind == "wts" else [sign]
        post = [sign, phase] if kind == "stw" else [sign]
        axis = self._first_pass(pre, ctl)
        self._open_pass(1 - axis, ctl, post + [(_lib.PW_SCALE, 0, scl)])
        self.last_single = {"kind": kind, "arr": arr.copy(), "inv": inv.copy(), "open": self.open,
                            "open_mid": list(self.open["mid"]), "undo": undo}

    def _undoes(self, first, first_inv, second, second_inv, max_residual=None):
        """Does the stw ``second`` undo the wts ``first`` for every item?  Same items, opposite transform directions,
        and phases exp(i c (sx^2 x^2 + sy^2 y^2)) (x, y in pixels from the centre) whose coefficients cancel: the
        residual phase at the corner of the grid stays below ``max_residual`` rad (default UNDO_MAX_RESIDUAL = the
        1e-10 parity gate on the field / 10: a residual phase d changes the field by |exp(i d) - 1| = d, so what is
        dropped stays an order of magnitude inside the contract; rounding noise of the two separately rounded
        coefficients is ~1e-12 at the 1e4 rad these phases reach; anything physical -- a different distance, a
        magnification between the two -- is many radians).  A pair above the threshold simply keeps both operators."""
        if max_residual is None:
            max_residual = UNDO_MAX_RESIDUAL
        on = second[:, 0] != 0.0
        if not np.array_equal(first[:, 0] != 0.0, on) or not np.array_equal(first_inv[on], 1.0 - second_inv[on]):
            return False
        half = (self.n / 2.0) ** 2
        for a, b in ((first[on], second[on]),):
            rx = a[:, 3] * a[:, 4] * a[:, 1] ** 2 + b[:, 3] * b[:, 4] * b[:, 1] ** 2
            ry = a[:, 3] * a[:, 4] * a[:, 2] ** 2 + b[:, 3] * b[:, 4] * b[:, 2] ** 2
            if not np.all(np.isfinite(rx)) or not np.all(np.isfinite(ry)):
                return False
            if np.any((np.abs(rx) + np.abs(ry)) * half >= max_residual):
                return False
        return True

    def _residual_phase(self, first, second):
        """What the phases of a wts and the stw behind it leave when their transforms have cancelled:
        exp(i (rx gx^2 + ry gy^2)) (g: pixels from the centre), rx = c1 sx1^2 + c2 sx2^2 per item, as one centred
        quadratic-phase block [enable, sqrt|rx|, sqrt|ry|, 1, sign] -- or two (an x-only and a y-only one) when the
        two leftovers differ in sign, which noise-level leftovers do."""
        on = second[:, 0] != 0.0
        rx = np.where(on, first[:, 3] * first[:, 4] * first[:, 1] ** 2 + second[:, 3] * second[:, 4] * second[:, 1] ** 2, 0.0)
        ry = np.where(on, first[:, 3] * first[:, 4] * first[:, 2] ** 2 + second[:, 3] * second[:, 4] * second[:, 2] ** 2, 0.0)
        out = []
        same = (rx >= 0.0) == (ry >= 0.0)
        if np.all(same | (rx == 0.0) | (ry == 0.0)):
            sgn = np.where((rx < 0.0) | (ry < 0.0), -1.0, 1.0)
            out.append(np.stack([on.astype(float), np.sqrt(np.abs(rx)), np.sqrt(np.abs(ry)), np.ones_like(rx), sgn], axis=1))
        else:
            zero = np.zeros_like(rx)
            out.append(np.stack([on.astype(float), np.sqrt(np.abs(rx)), zero, np.ones_like(rx), np.where(rx < 0.0, -1.0, 1.0)], axis=1))
            out.append(np.stack([on.astype(float), zero, np.sqrt(np.abs(ry)), np.ones_like(rx), np.where(ry < 0.0, -1.0, 1.0)], axis=1))
        return [b for b in out if np.any(b[:, 0] != 0.0) and np.any((b[:, 1] != 0.0) | (b[:, 2] != 0.0))]

    def stw(self, rows, inverse):
        self._single(rows, inverse, "stw")

    def wts(self, rows, inverse):
        self._single(rows, inverse, "wts")

    def ptp(self, rows):
        arr = _rows(rows, self.batch)
        if not arr[:, 0].any():
            return
        # Two ptp in a row (OI then IO, or II then II: wfo.py:560-570) with nothing between them:
        #   F^-1 H2 F . F^-1 H1 F  =  F^-1 (H2 H1) F        since fft2(ifft2(X)) = X,
        # so the second one only adds its transfer function to the first one's middle pass -- two passes per
        # junction less, and the same numbers up to the rounding noise of the transform pair that is not run
        # (~1e-16 relative; each phase keeps its own, separately rounded argument).  Items that take only one of
        # the two have the other phase switched off (factor 1); the transforms run for the union.
        lp = self.last_ptp if PTP_ALGEBRA else None
        if (lp is not None and self.open is lp["tail"] and not self.tail and lp["tail"]["mid"] == lp["tail_mid"] and
                sum(op[0] == _lib.PW_QPHASE_NATURAL for op in lp["middle"]["mid"]) < MAX_MERGED_PTP):
            first = self.blocks[lp["par"]]
            on = arr[:, 0] != 0.0
            # ... and when the second one undoes the first (a surface a hair past a waist: OI steps stw, ptp(+d) and the
            # next IO step starts with ptp(-d); SYN20 does this at every focus with d = 1.6 nm): H(-d) H(d) = 1 for
            # every item, so both go -- and the stw in front and the wts behind meet in ONE pass.  (Items that take
            # neither ptp are covered: the pair is the identity for them too.)
            if (lp["single"] and np.array_equal(first[:, 0] != 0.0, on) and
                    np.array_equal(first[on, 1:3], arr[on, 1:3]) and np.array_equal(first[on, 3], -arr[on, 3]) and
                    np.array_equal(first[on, 4], arr[on, 4])):
                self._restore(lp["undo"])
                return
            par = self._block(arr)
            for blk, col, val in ((lp["fwd"], 1, 0.0), (lp["inv"], 1, 1.0), (lp["scl"], 3, 1.0 / self.n)):
                b = self.blocks[blk]
                new = on & (b[:, 0] == 0.0)
                b[new, :] = 0.0
                b[new, 0] = 1.0
                b[new, col] = val
            mid = lp["middle"]["mid"]
            mid.insert(len(mid) - 1, (_lib.PW_QPHASE_NATURAL, 0, par))  # in front of the 1/N
            lp["single"] = False
            return
        undo = self._snapshot()
        par = self._block(arr)
        fwd = self._derived(arr, v1=0.0)
        inv = self._derived(arr, v1=1.0)
        scl = self._derived(arr, v3=1.0 / self.n)
        axis = self._first_pass([], fwd)
        middle = {"axis": 1 - axis, "fft1": fwd, "fft2": inv, "pre": [], "post": [],
                  "mid": [(_lib.PW_QPHASE_NATURAL, 0, par), (_lib.PW_SCALE, 0, scl)]}
        self.passes.append(middle)
        self.open = None
        self._open_pass(axis, inv, [(_lib.PW_SCALE, 0, scl)])
        self.last_ptp = {"middle": middle, "tail": self.open, "tail_mid": list(self.open["mid"]), "fwd": fwd, "inv": inv,
                         "scl": scl, "par": par, "single": True, "undo": undo}

    def _snapshot(self):
        """What ``_restore`` needs to take the compiler back to this point (before an operator was queued)."""
        o = self.open
        return {"n_blocks": len(self.blocks), "n_passes": len(self.passes), "open": o, "tail": list(self.tail),
                "open_state": None if o is None else (list(o["mid"]), o.get("fft2", -1)), "last_ptp": self.last_ptp,
                "last_single": self.last_single}

    def _restore(self, snap):
        del self.blocks[snap["n_blocks"]:]
        del self.passes[snap["n_passes"]:]
        self.open, self.tail, self.last_ptp = snap["open"], list(snap["tail"]), snap["last_ptp"]
        self.last_single = snap["last_single"]
        if self.open is not None:
            self.open["mid"], self.open["fft2"] = list(snap["open_state"][0]), snap["open_state"][1]

    def open_takes_mask(self):
        """True when a pass is still open whose operator slot after its transform can take an
        aperture (none there yet, room left): an aperture queued now rides on that pass even if the
        program is flushed right afterwards."""
        o = self.open
        if o is None or self.tail:
            return False
        if any(op[0] == _lib.PW_MASK for op in o["mid"]):
            return False
        return len(o["mid"]) + 1 <= _lib.MAX_PW

    # ---- execution -----------------------------------------------------------------------
    def pending(self):
        return bool(self.passes or self.tail or self.open)

    def program(self):
        """Finish the current stretch; returns (passes, blocks[n][batch][5])."""
        self._close_open()
        passes = self.passes
        blocks = np.stack(self.blocks) if self.blocks else np.zeros((0, self.batch, 5), dtype=np.float64)
        self._reset()
        return passes, blocks

    def flush(self, dev, live_rows=None, rows_stale=False, final_intensity=False, live_cols=None):
        """``live_rows`` ([batch][2], optional): rows of each item outside [lo, hi) are exactly zero in
        memory on entry (a stand-alone aperture has just been applied) -- the library then skips them;
        ``rows_stale``: they hold old data standing for zeros instead.  ``final_intensity``: the last pass
        stores |u|^2 (PSF buffer) instead of the field (True / 1), or stores the field and sums its power on the way (2);
        returns (passes run, power ticket -- None when there was no pass to do it)."""
        if not self.pending():
            return (0, None) if final_intensity else 0
        passes, blocks = self.program()
        ticket = None
        if passes:
            if rows_stale and live_cols is not None:  # (the start field was written inside its aperture's box only)
                ticket = dev.run_passes(passes, blocks, live_rows=live_rows, rows_stale=True, final_intensity=final_intensity,
                                        live_cols=live_cols)
            elif rows_stale or final_intensity:
                ticket = dev.run_passes(passes, blocks, live_rows=live_rows, rows_stale=rows_stale,
                                        final_intensity=final_intensity)
            elif live_rows is None:
                dev.run_passes(passes, blocks)
            else:
                dev.run_passes(passes, blocks, live_rows=live_rows)
        elif final_intensity and final_intensity != 2:
            raise RuntimeError("a program without passes cannot store the PSF")
        return (len(passes), ticket) if final_intensity else len(passes)


# The separable pass programs (round 4).  PAOS_SEPARABLE=0 (or SEPARABLE = False): the compiler above.
SEPARABLE = os.environ.get("PAOS_SEPARABLE", "1") != "0"


class SeparableCompiler(PassCompiler):
    """Every operator between two apertures is a product of a factor that acts along the rows and a factor that acts
    along the columns:

        lens, Qc, P, H : exp(i c (x^2 + y^2)) = exp(i c x^2) exp(i c y^2)          wfo.py:359-366, 462-545
        checkerboard S : (-1)^(row + column)  = (-1)^column (-1)^row
        fft2 / ifft2   : F_rows F_columns

    and an operator along the rows commutes with every operator along the columns, so a whole stretch

        aperture_k . [lens ptp lens stw wts ...] . aperture_k+1   =   aperture_k . X . Y . aperture_k+1

    with X = all the row factors in their order, Y = all the column factors in theirs.  X runs first, on the rows
    aperture_k left alive only (rows of zeros stay rows of zeros under X); Y then runs on the columns aperture_k+1 will
    keep only (a column nobody reads need not be computed), reading the live rows and storing the rows that aperture keeps.
    For SYN20 at zoom 4 (a 1024-pixel pupil on a 4096 grid) that is 4 x 1024 + 4 x 1024 line transforms per relay
    instead of the 2 x 1024 + 6 x 4096 of the operator-by-operator order (PassCompiler), with every pass touching a
    sixteenth of the grid on the way in and on the way out.  Which lines are alive / wanted is the library's planner's
    business (csrc/paos_hip.hip: plan_pruning); this class only orders the factors.

    The identities of PassCompiler (consecutive ptp share their transforms, ptp(+d) ptp(-d) = 1, a wts and the stw that
    undoes it) are applied to the operator stream before it is split.  Arguments: the reference rounds
    c (x^2 + y^2) once, the two factors round c x^2 and c y^2 separately -- a few 1e-12 rad where the light is (measured
    end to end: tests/test_host_logic.py, tests/test_gpu_r4.py)."""

    def _reset(self):
        super()._reset()
        self.ops = []  # the operator stream since the last program(): diag / ptp / single / mask entries

    # ---- operators -----------------------------------------------------------------------
    def aperture(self, records):
        rows = _aperture_rows(records, self.batch)
        if rows is None:
            return
        self.ops.append({"k": "mask", "first": rows[0], "second": rows[1]})

    def lens(self, rows):
        arr = _rows(rows, self.batch)
        if arr[:, 0].any():
            self.ops.append({"k": "diag", "kind": _lib.PW_QPHASE_CENTRED, "flags": _lib.PWF_MUL2PI, "arr": arr.copy()})

    def _single(self, rows, inverse, kind):
        arr = _rows(rows, self.batch)
        if not arr[:, 0].any():
            return
        inv = np.where(np.asarray(inverse, dtype=bool), 1.0, 0.0) * np.ones(self.batch)
        last = self.ops[-1] if self.ops else None
        if PTP_ALGEBRA and kind == "stw" and last is not None and last["k"] == "single" and last["kind"] == "wts":
            # (see PassCompiler._single: a wts straight into the stw that undoes it)
            if self._undoes(last["arr"], last["inv"], arr, inv):
                self.ops.pop()
                return
            if self._undoes(last["arr"], last["inv"], arr, inv, max_residual=UNDO_COMPENSATE_MAX):
                leftovers = self._residual_phase(last["arr"], arr)
                self.ops.pop()
                for blk in leftovers:
                    self.ops.append({"k": "diag", "kind": _lib.PW_QPHASE_CENTRED, "flags": 0, "arr": blk})
                return
        self.ops.append({"k": "single", "kind": kind, "arr": arr.copy(), "inv": inv})

    def ptp(self, rows):
        arr = _rows(rows, self.batch)
        if not arr[:, 0].any():
            return
        on = arr[:, 0] != 0.0
        last = self.ops[-1] if self.ops else None
        if PTP_ALGEBRA and last is not None and last["k"] == "ptp" and len(last["H"]) < MAX_MERGED_PTP:
            first = last["H"][0]
            # (see PassCompiler.ptp: H(-d) H(d) = 1; otherwise the second transfer function joins the first)
            if (len(last["H"]) == 1 and np.array_equal(first[:, 0] != 0.0, on) and
                    np.array_equal(first[on, 1:3], arr[on, 1:3]) and np.array_equal(first[on, 3], -arr[on, 3]) and
                    np.array_equal(first[on, 4], arr[on, 4])):
                self.ops.pop()
                return
            last["H"].append(arr.copy())
            last["on"] = last["on"] | on
            return
        self.ops.append({"k": "ptp", "H": [arr.copy()], "on": on.copy()})

    def open_takes_mask(self):
        """An aperture queued now ends up behind the last transform of the last pass -- when that pass has one transform
        only (an odd number of transforms since the last aperture) and carries no aperture yet."""
        count = 0
        for op in reversed(self.ops):
            if op["k"] == "mask":
                return False if count == 0 else count % 2 == 1
            count += {"ptp": 2, "single": 1}.get(op["k"], 0)
        return count % 2 == 1

    def pending(self):
        return bool(self.ops)

    # ---- lowering ------------------------------------------------------------------------
    def _axis_block(self, arr, axis):
        """The factor of a quadratic phase along one axis: the other axis' sampling set to zero."""
        out = arr.copy()
        out[:, 2 if axis == 0 else 1] = 0.0
        out[out[:, 0] == 0.0] = 0.0
        return self._block(out)

    def _chains(self, stretch):
        """The row factors (axis 0) and the column factors (axis 1) of a run of separable operators: lists of
        ("d", (kind, flags, block)) and ("t", control block)."""
        only = {0: _lib.PWF_X_ONLY, 1: _lib.PWF_Y_ONLY}
        chains = {0: [], 1: []}
        for op in stretch:
            if op["k"] == "diag":
                for ax in (0, 1):
                    chains[ax].append(("d", (op["kind"], op["flags"], self._axis_block(op["arr"], ax))))
            elif op["k"] == "single":
                arr, kind = op["arr"], op["kind"]
                par = self._block(arr)  # (the sign reads the enable flags only)
                ctl = self._derived(arr, v1=op["inv"])
                scl = self._derived(arr, v3=1.0 / self.n)
                for ax in (0, 1):
                    sign = ("d", (_lib.PW_SIGN, only[ax], par))
                    phase = ("d", (_lib.PW_QPHASE_CENTRED, 0, self._axis_block(arr, ax)))
                    ch = chains[ax]
                    ch += [phase, sign] if kind == "wts" else [sign]
                    if ax == 0:
                        ch.append(("d", (_lib.PW_SCALE, 0, scl)))  # 1 / N for the 2-D pair: once (a scalar: in front)
                    ch.append(("t", ctl))
                    ch += [sign, phase] if kind == "stw" else [sign]
            else:  # ptp: F . H1 [H2 H3] . F^-1, the transforms on for every item that takes any of them
                union = np.zeros((self.batch, 5))
                union[:, 0] = np.where(op["on"], 1.0, 0.0)
                fwd = self._derived(union, v1=0.0)
                inv = self._derived(union, v1=1.0)
                scl = self._derived(union, v3=1.0 / self.n)
                for ax in (0, 1):
                    ch = chains[ax]
                    ch.append(("t", fwd))
                    for h in op["H"]:
                        ch.append(("d", (_lib.PW_QPHASE_NATURAL, 0, self._axis_block(h, ax))))
                    ch.append(("d", (_lib.PW_SCALE, 0, scl)))
                    ch.append(("t", inv))
        return chains

    @staticmethod
    def _phases(ops):
        return sum(op[0] in (_lib.PW_QPHASE_CENTRED, _lib.PW_QPHASE_NATURAL) for op in ops)

    def _room(self, slot_ops, extra, limit):
        ops = slot_ops + extra
        return (len(ops) <= _lib.MAX_PW and self._phases(ops) <= limit and
                sum(op[0] == _lib.PW_MASK for op in ops) <= 1)

    def _emit_chain(self, axis, chain, lone_first=False):
        """Pack one axis' factors into passes of up to two transforms.  ``lone_first``: the first pass takes one
        transform only, so that a chain with an even number of them ends on a pass whose slot behind the transform is
        free for what trails the chain (the program ends there: no later pass could carry it)."""
        cur = None
        lone = lone_first
        for what, val in chain:
            if what == "d":
                if cur is not None and cur["fft2"] == -1 and not self.tail and self._room(cur["mid"], [val], _MAX_MID) and \
                        not (val[0] == _lib.PW_MASK and self._has_mask(cur)):
                    cur["mid"].append(val)
                else:
                    cur = None
                    self.tail.append(val)
            else:
                if cur is not None and cur["fft2"] == -1 and not self.tail and not lone:
                    cur["fft2"] = val
                    continue
                if cur is not None:
                    lone = False
                if not self._room([], self.tail, _MAX_PRE):
                    self._standalone(self.tail)
                    self.tail = []
                cur = {"axis": axis, "fft1": val, "fft2": -1, "pre": self.tail, "mid": [], "post": []}
                self.tail = []
                self.passes.append(cur)
        self.open = cur if cur is not None and cur["fft2"] == -1 and not self.tail else None

    @staticmethod
    def _has_mask(ps):
        return any(op[0] == _lib.PW_MASK for op in ps["pre"] + ps["mid"])

    def _lower(self):
        # segments: a run of separable operators and the aperture behind it (None at the end of the program)
        segments, stretch = [], []
        for op in self.ops:
            if op["k"] == "mask":
                segments.append((stretch, op))
                stretch = []
            else:
                stretch.append(op)
        if stretch:
            segments.append((stretch, None))
        for s, (stretch, mask_op) in enumerate(segments):
            if stretch:
                chains = self._chains(stretch)
                self._emit_chain(0, chains[0])
                # The program ends with this stretch [and its aperture]: what trails the column factors (a checkerboard
                # half, a phase, the aperture) needs a slot behind the last transform.
                count = sum(w == "t" for w, _ in chains[1])
                trails = bool(chains[1]) and chains[1][-1][0] == "d"
                self._emit_chain(1, chains[1], lone_first=(s == len(segments) - 1 and count > 0 and count % 2 == 0 and
                                                            (trails or mask_op is not None)))
            if mask_op is None:
                continue
            first = self._block(mask_op["first"])
            self._block(mask_op["second"])
            mask = (_lib.PW_MASK, 0, first)
            o = self.open
            if o is not None and not self._has_mask(o) and self._room(o["mid"], [mask], _MAX_MID):
                o["mid"].append(mask)  # behind the last transform of the pass that ends the stretch
            else:
                self.open = None
                if any(t[0] == _lib.PW_MASK for t in self.tail):  # two apertures with nothing between them
                    self._standalone(self.tail)
                    self.tail = []
                self.tail.append(mask)
        self.ops = []

    def _close_open(self):
        self.open = None
        if self.tail:
            self._standalone(self.tail)
            self.tail = []

    def program(self):
        self._lower()
        return super().program()


_MAX_PRE, _MAX_MID = 2, 3  # phases a slot of the frugal kernels can carry (kFrugalMaxPre / kFrugalMaxMid of csrc/frugal_pass.h)
