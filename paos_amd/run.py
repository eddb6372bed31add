"""Propagation loop: ``run`` (drop-in) and ``run_batch`` (many wavefronts per launch).

``run(pupil_diameter, wavelength, gridsize, zoom, field, opt_chain)`` keeps the
signature, the per-surface order of operations and the returned dictionary of
``paos.core.run.run`` (reference paos/core/run.py:30-228).  All scalar decisions
are taken on the host exactly as there (coordinate break :81-91, aperture
geometry :96-122, Zernike radius :131, gating of Magnification / ChangeMedium /
lens / propagate :181-207, ray and ABCD bookkeeping :209-219); the field never
leaves HBM except for the arrays of surfaces marked ``save``.

``run_batch`` runs B independent wavefronts (wavelengths of one lens file, or
Monte-Carlo WFE realisations -- what the reference fans out over joblib workers,
paos/core/pipeline.py:140-150) through ONE sequence of kernel launches: every
launch carries a per-item parameter block with an enable flag, so items whose
planners disagree (a ptp skipped for one wavelength, II vs OI) still share
launches.
"""
import atexit
import gc
import math
import threading
from copy import deepcopy

import numpy as np

from . import _lib
from . import abcd as _abcd
from .abcd import ABCD
from .aperture import EllipticalAperture, RectangularAperture, bbox_misses_grid, make_aperture
from .coordinate_break import coordinate_break
from . import passes as _passes
from .passes import PassCompiler, SeparableCompiler
from . import phase_maps as _phase_maps
from .phase_maps import MAP_SERIAL as _MAP_SERIAL, PsdScreen, grid_sag_map
from .planner import (BeamBatch, gram_polynomials, jacobi_recurrence, orthonorm_matrix,
                      zernike_block)
from .zernike import zernike_tables

# Apertures ride on FFT passes as PW_MASK operators when the library can hold them as per-line
# records (csrc/frugal_pass.h: MaskLine; complex128, N >= 1024, untilted ellipse whose partial
# runs fit, or untilted rectangular aperture): no stand-alone pass and the fusion chain does
# not break.  PAOS_FUSE_APERTURES=0 turns it off, =1 forces it for every aperture (the generic
# kernel then uses a rendered weight map, which measured slower than a stand-alone pass).
import os as _os

FUSE_APERTURES = {"0": False, "1": True}.get(_os.environ.get("PAOS_FUSE_APERTURES", "auto"), "auto")
# The power of a saved surface whose field is exactly what its pass program stored is summed by that program's last pass
# (csrc/frugal_pass.h: STORE = 2) instead of by a reduction that reads the field back; PAOS_POWER_ON_STORE=0 switches it off.
POWER_ON_STORE = _os.environ.get("PAOS_POWER_ON_STORE", "1") != "0"
# ... and a stop right behind a pass program scales by the power that program's last pass has summed (make_stop's own
# reduction would read the field back); PAOS_STOP_FROM_PROGRAM=0 switches it off.
STOP_FROM_PROGRAM = _os.environ.get("PAOS_STOP_FROM_PROGRAM", "1") != "0"
# ... and the pass program that reaches the last surface that does anything (a saved slit with the image plane right behind
# it) stores the PSF for the inert surfaces behind it as well; PAOS_INERT_TAIL=0 switches it off.
INERT_TAIL = _os.environ.get("PAOS_INERT_TAIL", "1") != "0"
# ... and such a stop leaves even its scaling to the next pass program's first pass (paos_stop_defer_last_power);
# PAOS_STOP_DEFERRED=0 runs the scaling sweep at the stop.
STOP_DEFERRED = _os.environ.get("PAOS_STOP_DEFERRED", "1") != "0"
# ... and a lean walk's first field is written inside the aperture's bounding BOX only (rows and columns: paos_start_box,
# round 5; rounds 3-4 wrote whole rows); PAOS_START_BOX=0 writes whole rows again.
START_BOX = _os.environ.get("PAOS_START_BOX", "1") != "0"
# Items that start from the same constant under the same aperture hold copies of one field until something that depends on the
# wavelength touches them: a Zernike surface right behind the start reads one field per group of copies (paos_zernike_like,
# round 5).  PAOS_TWIN_FIELDS=0: every item's field is read.
TWIN_FIELDS = _os.environ.get("PAOS_TWIN_FIELDS", "1") != "0"
_MASK_RUN = 192  # kMaskW of csrc/frugal_pass.h


def _aperture_fits_line_records(handle, obscuration, n, precision):
    if n < (1024 if precision == "fp64" else 2048) or handle.theta != 0.0:
        return False
    if isinstance(handle, EllipticalAperture):
        a, b = handle.a, handle.b
        if not (a >= 2.0 and b >= 2.0):
            return False
        return (2.0 * a * math.sqrt(3.0 / b) + 8.0 <= _MASK_RUN and
                2.0 * b * math.sqrt(3.0 / a) + 8.0 <= _MASK_RUN)
    return not obscuration and handle.w > 0.0 and handle.h > 0.0

_OFF_PHASE = [0.0] * _lib.PHASE_STRIDE
_OFF_APERTURE = [0.0] * _lib.APERTURE_STRIDE


class _Reduction:
    """One reduction of sum |u|^2 per item that is outstanding on the device (a ticket slot of paos_norm2_enqueue* /
    paos_run_program).  The library frees the slot on the first fetch and hands it out again, so the values are fetched
    at most once and kept here; every record that the reduction answers for shares this object."""

    __slots__ = ("dev", "slot", "raw", "released")

    def __init__(self, dev, slot):
        self.dev, self.slot, self.raw, self.released = dev, int(slot), None, False

    def values(self):
        if self.raw is None:
            if self.released:
                raise RuntimeError("this power was given back unread (PowerTicket.release)")
            self.raw = self.dev.norm2_fetch(self.slot)  # synchronises the stream
        return self.raw

    def release(self):
        if self.raw is None and not self.released:
            self.dev.norm2_release(self.slot)
            self.released = True


class PowerTicket:
    """What ``run_batch(..., sync=False)`` leaves under ``'power_ticket'``: a handle to the powers (one per batch item)
    of a saved surface that are still being reduced on the GPU.  ``fetch()`` waits for them and returns the array --
    any number of times, from any of the records that share the reduction (several saved surfaces can be answered by
    one reduction: a saved slit and the image plane right behind it); ``release()`` gives the slot back unread.
    ``dev.norm2_fetch(ticket)`` / ``dev.norm2_release(ticket)`` accept the handle as well.  A power that is derived
    from the reduced one (the power BEHIND a stop that scales by it) is derived at fetch time: nothing synchronises
    before the caller asks."""

    __slots__ = ("_red", "_post")

    def __init__(self, reduction, post=None):
        self._red, self._post = reduction, post

    def fetch(self):
        v = self._red.values()
        return self._post(v) if self._post is not None else v

    def release(self):
        self._red.release()

    def __int__(self):  # the slot, for diagnostics
        return self._red.slot

    def __repr__(self):
        state = "fetched" if self._red.raw is not None else ("released" if self._red.released else "outstanding")
        return f"PowerTicket(slot {self._red.slot}, {state}{', derived' if self._post is not None else ''})"


class _BatchApertures:
    """The apertures of one surface for every wavefront of a batch as arrays (round 5: ``_plan_batch``): pixel-unit centres
    and sizes, shape, obscuration flag -- what the library's parameter blocks, the line-record test and the live-row
    bookkeeping need, without a Python object per wavefront.  ``handle(i)`` makes the photutils-like object of item i
    when somebody wants to see it (a saved surface's record, an orthonormal Zernike pupil)."""

    def __init__(self, ixc, iyc, ea, eb, rect, obsc):
        self.ixc, self.iyc, self.ea, self.eb, self.rect, self.obsc = ixc, iyc, ea, eb, rect, obsc
        self.batch = len(ixc)
        self._handles = {}

    def handle(self, i):
        h = self._handles.get(i)
        if h is None:
            cls = RectangularAperture if self.rect[i] else EllipticalAperture
            h = self._handles[i] = cls((float(self.ixc[i]), float(self.iyc[i])), float(self.ea[i]), float(self.eb[i]), 0.0)
        return h

    def blocks(self):
        """[batch][8] parameter blocks of paos_aperture / paos_start (aperture.py: block) and the shape code per item."""
        b = np.zeros((self.batch, _lib.APERTURE_STRIDE), dtype=np.float64)
        b[:, 0] = 1.0
        b[:, 1], b[:, 2], b[:, 3], b[:, 4] = self.ixc, self.iyc, self.ea, self.eb
        b[:, 6] = self.obsc
        b[:, 7] = np.where(self.rect, 32.0, 0.0)
        return b, np.where(self.rect, float(_lib.SHAPE_RECT), float(_lib.SHAPE_ELLIPSE))

    def fits_line_records(self, n, precision):
        """``_aperture_fits_line_records`` for every item."""
        if n < (1024 if precision == "fp64" else 2048):
            return False
        a, b = self.ea, self.eb
        with np.errstate(all="ignore"):
            ell = (a >= 2.0) & (b >= 2.0) & (2.0 * a * np.sqrt(3.0 / b) + 8.0 <= _MASK_RUN) & (2.0 * b * np.sqrt(3.0 / a) + 8.0 <= _MASK_RUN)
        rec = ~self.obsc & (a > 0.0) & (b > 0.0)
        return bool(np.all(np.where(self.rect, rec, ell)))

    def narrow_rows(self, live, n):
        """``_live_rows_after`` for every item: rows outside a clear aperture's bounding box (widened by a pixel) are zero."""
        ext = np.where(self.rect, self.eb / 2.0, self.eb)
        ok = ~self.obsc & np.isfinite(self.iyc) & np.isfinite(ext) & (ext > 0.0)
        with np.errstate(all="ignore"):
            lo = np.maximum(0, np.floor(self.iyc - ext + 0.5) - 1)
            hi = np.minimum(n, np.ceil(self.iyc + ext + 0.5) + 1)
        for i in np.nonzero(ok & (lo < hi))[0].tolist():
            r = live[i]
            r[0], r[1] = max(r[0], int(lo[i])), min(r[1], int(hi[i]))
            if r[0] >= r[1]:
                r[0], r[1] = 0, 0


def _live_cols_of(plans, n):
    """Per item the columns [lo, hi) outside which a clear aperture leaves exact zeros (photutils' bounding box widened by
    a pixel, like ``_live_rows_after`` for the rows), or None when some item has no such aperture."""
    out = []
    for i, p in enumerate(plans):
        ap = p["aperture"]
        if ap is None or ap[1] or ap[0].theta != 0.0:
            return None
        h = ap[0]
        ext = h.a if isinstance(h, EllipticalAperture) else h.w / 2.0
        xc = h._xy[0]
        if not (math.isfinite(xc) and math.isfinite(ext) and ext > 0.0):
            return None
        lo = max(0, int(math.floor(xc - ext + 0.5)) - 1)
        hi = min(n, int(math.ceil(xc + ext + 0.5)) + 1)
        if lo >= hi:
            return None
        out.append([lo, hi])
    return out


class _LazyAperture:
    """``plan["aperture"]`` of an item of a ``_BatchApertures`` surface: behaves like the (handle, obscuration) pair
    ``_plan_host`` makes; the handle is built when it is looked at."""

    __slots__ = ("src", "i")

    def __init__(self, src, i):
        self.src, self.i = src, i

    def __getitem__(self, k):
        if k == 0:
            return self.src.handle(self.i)
        if k == 1:
            return bool(self.src.obsc[self.i])
        raise IndexError(k)

    def __iter__(self):
        yield self.src.handle(self.i)
        yield bool(self.src.obsc[self.i])

    def __len__(self):
        return 2


class _Plans(list):
    """The per-item plans of one surface; ``ap`` is the surface's ``_BatchApertures`` when ``_plan_batch`` planned the
    apertures of ALL items with array arithmetic (None otherwise: the consumers then read the per-item entries).
    ``summary()``: (any stop, any Zernike, any phase map, any aperture) over the items -- looked at a dozen times per
    surface by the walk, scanned once."""

    ap = None
    _summary = None

    def summary(self):
        if self._summary is None:
            self._summary = (any(p["stop"] for p in self), any(p["zernike"] is not None for p in self),
                             any(p["phase_map"] is not None for p in self), any(p["aperture"] is not None for p in self))
        return self._summary


class _Item:
    """Host state of one wavefront while it walks the chain: the two paraxial rays of the field
    point and the ray-transfer factors met so far.  (The pilot beams of all wavefronts live together
    in a ``planner.BeamBatch``.)"""

    def __init__(self, pupil_diameter, wavelength, gridsize, zoom, field):
        # the argument checks of WFO.__init__ (wfo.py:100-103)
        assert np.log2(gridsize).is_integer(), "Grid size should be 2**n"
        assert zoom > 0, "zoom factor should be positive"
        assert pupil_diameter > 0, "beam diameter should be positive"
        assert wavelength > 0, "a wavelength should be positive"
        self.pupil_diameter, self.wavelength, self.gridsize, self.zoom = pupil_diameter, wavelength, gridsize, zoom
        self.vt = np.array([0.0, field["ut"]])
        self.vs = np.array([0.0, field["us"]])
        # an on-axis field point: both rays are [0, 0] and stay so under every ABCD matrix (A @ 0 = 0)
        # until a coordinate break moves them -- no need to multiply
        self.still = field["ut"] == 0.0 and field["us"] == 0.0
        self.factors_t, self.factors_s = [], []


class ABCDProduct(ABCD):
    """The accumulated ray-transfer matrix ``f_k * ... * f_1 * ABCD()`` of a saved surface
    (run.py:215-219), multiplied out when it is first read -- most callers never look at it."""

    def __init__(self, factors):  # noqa: D107 -- deliberately does not build a matrix
        object.__setattr__(self, "_factors", factors)

    def __getattr__(self, name):  # only reached while the product has not been formed
        if name in ("_mat", "_cin", "_cout", "_cache"):
            acc = ABCD()
            for f in object.__getattribute__(self, "_factors"):
                acc = f * acc
            self._mat, self._cin, self._cout, self._cache = acc._mat, acc._cin, acc._cout, {}
            return object.__getattribute__(self, name)
        raise AttributeError(name)


# Round 5: a measured surface map is one array that every wavelength of a sweep and every draw of a Monte-Carlo batch passes
# through WFO.grid_sag with the same sampling: the map the reference would build per wavefront (wfo.py:745-867: masking,
# Fourier shift, padding / cropping, resampling -- seconds at 4096^2) is built once per distinct (array, geometry) and shared;
# ``_launch_phase_maps`` then uploads it once for all the items that carry it.  An entry keeps its input array alive (so that
# the id stays unique) and is matched by the array's identity plus a digest of ALL of its bytes (xxh3: ~5 ms for a 4096^2 map
# on the GPU box's host, taken once per walk and array -- ``_SAG_SEEN``), so an array edited in place between two runs,
# wherever, is rebuilt.
_SAG_MAPS = {}


def forget_maps():
    """Drop what the module keeps between calls: the Grid Sag maps (``_sag_map_once``) with their zero-filled copies, and the
    gate arrays remembered per column of ABCD objects (``_gate_arrays``: they keep those objects alive)."""
    _SAG_MAPS.clear()
    _SAG_SEEN.clear()
    _FILLED_MAPS.clear()
    _GATE_COLUMNS.clear()


_SAG_SEEN = {}  # id(array) -> (array, fingerprint) within ONE walk (cleared when a walk starts): the items of a batch share it


def _digest(buf):
    """A 64-bit digest of every byte of a contiguous uint8 view (xxh3: ~5 ms for the 128 MiB of a 4096^2 map on the GPU
    box's host).  Without the xxhash module: the wrap-around uint64 sums of the rows and of the columns of the bytes laid
    out as a matrix, hashed -- an edit, a flip or a roll changes one of them."""
    try:
        import xxhash

        return xxhash.xxh3_64_intdigest(buf)
    except ImportError:
        import zlib

        words = buf[: buf.size // 8 * 8].view(np.uint64)
        side = max(1, int(np.sqrt(words.size)))
        mat = words[: words.size // side * side].reshape(-1, side)
        tail = bytes(words[mat.size:]) + bytes(buf[buf.size // 8 * 8:])
        return zlib.crc32(mat.sum(axis=0, dtype=np.uint64).tobytes() + mat.sum(axis=1, dtype=np.uint64).tobytes() + tail)


def _sag_fingerprint(sag):
    """Shape, type, masked-or-not and a digest of every byte of the array (and of its mask), once per walk and array."""
    seen = _SAG_SEEN.get(id(sag))
    if seen is not None and seen[0] is sag:
        return seen[1]
    a = np.ascontiguousarray(np.ma.getdata(sag))
    digests = [_digest(a.reshape(-1).view(np.uint8))]
    if isinstance(sag, np.ma.MaskedArray) and sag.mask is not np.ma.nomask:
        digests.append(_digest(np.ascontiguousarray(np.ma.getmaskarray(sag)).reshape(-1).view(np.uint8)))
    fp = (a.shape, a.dtype.str, isinstance(sag, np.ma.MaskedArray), tuple(digests))
    _SAG_SEEN[id(sag)] = (sag, fp)
    return fp


def _sag_map_once(sag, nx, ny, delx, dely, xdec, ydec, n, dx, dy):
    key = (id(sag), nx, ny, delx, dely, xdec, ydec, n, dx, dy)
    fp = _sag_fingerprint(sag)
    hit = _SAG_MAPS.get(key)
    if hit is not None and hit[0] is sag and hit[1] == fp:
        return hit[2]
    m = grid_sag_map(sag, nx, ny, delx, dely, xdec, ydec, (n, n), dx, dy)
    if len(_SAG_MAPS) >= 2:  # (a 4096^2 map with its mask is 144 MiB: keep the last two)
        _SAG_MAPS.pop(next(iter(_SAG_MAPS)))
    _SAG_MAPS[key] = (sag, fp, m)
    return m


def _plan_host(st, item, n, dx, dy, wl, wz_of, aperture_plan=False):
    """Host half of one loop iteration of run.py:77-178 for one wavefront -- everything but the
    pilot-beam scalars (those are advanced for the whole batch in one C call, planner.BeamBatch).
    ``dx, dy, wl`` are the wavefront's current sampling and wavelength, ``wz_of()`` its beam radius."""
    plan = {"aperture": None, "stop": bool(item["is_stop"]), "zernike": None, "phase_map": None}
    kind = item["type"]

    if kind == "Coordinate Break":
        st.vt, st.vs = coordinate_break(st.vt, st.vs, item["xdec"], item["ydec"], item["xrot"],
                                        item["yrot"], 0.0)
        st.still = not (np.any(st.vt) or np.any(st.vs))

    ap = item.get("aperture")
    if aperture_plan is not False:  # (_plan_batch has done the aperture)
        plan["aperture"] = aperture_plan
    elif ap is not None:
        xdec = ap["xc"] if math.isfinite(ap["xc"]) else st.vs[0]
        ydec = ap["yc"] if math.isfinite(ap["yc"]) else st.vt[0]
        xrad = ap["xrad"]
        yrad = ap["yrad"]
        xrad *= math.sqrt(1 / (st.vs[1] ** 2 + 1))  # sqrt is correctly rounded either way
        yrad *= math.sqrt(1 / (st.vt[1] ** 2 + 1))
        xaper = xdec - st.vs[0]
        yaper = ydec - st.vt[0]
        if math.isfinite(xrad) and math.isfinite(yrad):
            handle = make_aperture(n, dx, dy, xaper, yaper, hx=xrad, hy=yrad, shape=ap["shape"])
            if bbox_misses_grid(handle, n):
                raise TypeError("aperture does not overlap the grid (mask is None in the reference)")
            plan["aperture"] = (handle, ap["type"] != "aperture")

    if kind == "Zernike":
        radius = item["Zradius"] if math.isfinite(item["Zradius"]) else wz_of()
        pupil = None
        if item["Zorthonorm"]:  # run.py:133-141: orthonormal over THIS surface's aperture object
            assert "aperture" in item, "Zorthonorm requires aperture"
            if plan["aperture"] is None:
                raise KeyError("aperture")  # the reference finds no _retval_["aperture"]
            # to_mask("exact") of a RectangularAperture: photutils has no exact rectangle overlap and
            # maps that request to the 32 x 32 sub-pixel rule (its _translate_mask_mode), i.e. the very
            # mask WFO.aperture applies -- so both shapes are served by the aperture kernel
            pupil = plan["aperture"][0]
        index = np.asarray(item["Zindex"])
        assert not np.any(np.diff(index) - 1), "Zernike sequence should be continuous"
        ordering = item["Zordering"]
        if ordering not in ("ansi", "noll", "fringe", "standard"):
            raise AssertionError("Unrecognised ordering scheme.")
        m, nn, norm = zernike_tables(len(index), ordering, bool(item["Znormalize"]))
        plan["zernike"] = dict(m=m, n=nn, norm=norm, Z=np.asarray(item["Z"], dtype=np.float64), dx=dx, dy=dy,
                               radius=radius, wl=wl, origin=item["Zorigin"], pupil=pupil)
    elif kind == "Grid Sag":  # run.py:154-164
        plan["phase_map"] = (_sag_map_once(item["grid_sag"], item["nx"], item["ny"], item["delx"], item["dely"],
                                           item["xdec"], item["ydec"], n, dx, dy), wl)
    elif kind == "PSD":  # run.py:166-177
        # (the draws are taken here, in the reference's order; the screen itself is built when the surface is launched: on the
        # device where the context can -- _launch_phase_maps)
        plan["phase_map"] = (PsdScreen((n, n), dx, dy, item["A"], item["B"], item["C"], item["fknee"], item["fmin"],
                                       item["fmax"], item["SR"], item["units"]), wl)
    return plan


def _plan_batch(states, items, n, dxs, dys, wls, wz_of, all_still=None):
    """``_plan_host`` for every wavefront of a batch.  The common case -- every item on axis (both paraxial rays zero:
    no coordinate break has moved them), the same kind of surface for all, no Grid Sag / PSD map to build -- is planned
    with array arithmetic (round 5: at 1024^2 x 256 wavefronts the per-item Python of this function was 40 % of a
    step that the GPU finishes in half the time the host needed to describe it).  The numbers are the ones
    ``_plan_host`` forms: with the rays at zero its ``xrad *= sqrt(1 / (0**2 + 1))`` and ``xdec - 0.0`` are exact
    identities, the pixel quantities are the same IEEE divisions and sums.  Anything else takes the per-item path."""
    kind = items[0]["type"]
    if all_still is None:
        all_still = all(st.still for st in states)
    fast = kind not in ("Coordinate Break", "Grid Sag", "PSD") and all_still and all(it["type"] == kind for it in items)
    if not fast:
        return _Plans(_plan_host(st, it, n, dxs[i], dys[i], wls[i], lambda i=i: wz_of(i))
                      for i, (st, it) in enumerate(zip(states, items)))
    plans = _Plans({"aperture": None, "stop": bool(it["is_stop"]), "zernike": None, "phase_map": None} for it in items)
    aps = [it.get("aperture") for it in items]
    have = [i for i, a in enumerate(aps) if a is not None]
    if have:
        sel = aps if len(have) == len(aps) else [aps[i] for i in have]
        geo = np.array([(a["xc"], a["yc"], a["xrad"], a["yrad"]) for a in sel], dtype=np.float64)
        xc, yc, xrad, yrad = geo[:, 0], geo[:, 1], geo[:, 2], geo[:, 3]
        if len(have) == len(aps):
            dx, dy = np.asarray(dxs, dtype=np.float64), np.asarray(dys, dtype=np.float64)
        else:
            dx, dy = np.array([dxs[i] for i in have], dtype=np.float64), np.array([dys[i] for i in have], dtype=np.float64)
        shapes = [a["shape"] for a in sel]
        rect = np.array([sh == "rectangular" for sh in shapes])
        known = all(sh in ("elliptical", "rectangular") for sh in shapes)
        xaper = np.where(np.isfinite(xc), xc, 0.0)  # (a missing centre falls back on the chief ray: zero here)
        yaper = np.where(np.isfinite(yc), yc, 0.0)
        with np.errstate(all="ignore"):
            ixc, iyc = xaper / dx + n / 2, yaper / dy + n / 2
            ea, eb = xrad / dx, yrad / dy   # semi-axes of an ellipse, FULL widths of a rectangle (wfo.py:224,261-264)
            xe, ye = np.where(rect, np.abs(ea / 2.0), np.abs(ea)), np.where(rect, np.abs(eb / 2.0), np.abs(eb))
            x0, x1 = np.floor(ixc - xe + 0.5), np.ceil(ixc + xe + 0.5)
            y0, y1 = np.floor(iyc - ye + 0.5), np.ceil(iyc + ye + 0.5)
            miss = ~(np.isfinite(ixc) & np.isfinite(iyc) & np.isfinite(xe) & np.isfinite(ye))
            miss |= (np.maximum(x0, 0) >= np.minimum(x1, n)) | (np.maximum(y0, 0) >= np.minimum(y1, n))
        finite = np.isfinite(xrad) & np.isfinite(yrad)
        obsc = np.array([a["type"] != "aperture" for a in sel])
        if known and len(have) == len(aps) and bool(finite.all()):
            # every item carries an aperture of a shape the kernels render: no per-item objects at all
            if bool(miss.any()):
                raise TypeError("aperture does not overlap the grid (mask is None in the reference)")
            plans.ap = _BatchApertures(ixc, iyc, ea, eb, rect, obsc)
            for i, p in enumerate(plans):
                p["aperture"] = _LazyAperture(plans.ap, i)
        else:
            ixc, iyc, ea, eb = ixc.tolist(), iyc.tolist(), ea.tolist(), eb.tolist()
            for k, i in enumerate(have):
                if not finite[k]:
                    continue  # run.py:112: skipped unless both radii are finite
                shape = shapes[k]
                if shape == "elliptical":
                    handle = EllipticalAperture((ixc[k], iyc[k]), ea[k], eb[k], 0.0)
                elif shape == "rectangular":
                    handle = RectangularAperture((ixc[k], iyc[k]), ea[k], eb[k], 0.0)
                else:  # ("circular" needs r, which run() never passes: make_aperture raises as the reference does)
                    handle = make_aperture(n, dxs[i], dys[i], float(xaper[k]), float(yaper[k]), hx=float(xrad[k]),
                                           hy=float(yrad[k]), shape=shape)
                if miss[k]:
                    raise TypeError("aperture does not overlap the grid (mask is None in the reference)")
                plans[i]["aperture"] = (handle, bool(obsc[k]))
    plans._summary = (any(it["is_stop"] for it in items), kind == "Zernike", False,
                      plans.ap is not None or any(p["aperture"] is not None for p in plans) if have else False)
    if kind == "Zernike":
        # (run.py:124-152 per item; what depends only on the index array / ordering / normalisation -- the continuity
        # check, the (m, n, norm) tables -- is looked up once per distinct index array of the batch)
        tables = {}
        for i, it in enumerate(items):
            radius = it["Zradius"] if math.isfinite(it["Zradius"]) else wz_of(i)
            pupil = None
            if it["Zorthonorm"]:
                assert "aperture" in it, "Zorthonorm requires aperture"
                if plans[i]["aperture"] is None:
                    raise KeyError("aperture")  # the reference finds no _retval_["aperture"]
                pupil = plans[i]["aperture"][0]
            index = np.asarray(it["Zindex"])
            key = (index.tobytes(), index.dtype.str, it["Zordering"], bool(it["Znormalize"]))
            hit = tables.get(key)
            if hit is None:
                assert not np.any(np.diff(index) - 1), "Zernike sequence should be continuous"
                if it["Zordering"] not in ("ansi", "noll", "fringe", "standard"):
                    raise AssertionError("Unrecognised ordering scheme.")
                hit = tables[key] = zernike_tables(len(index), it["Zordering"], bool(it["Znormalize"]))
            m, nn, norm = hit
            plans[i]["zernike"] = dict(m=m, n=nn, norm=norm, Z=np.asarray(it["Z"], dtype=np.float64), dx=dxs[i], dy=dys[i],
                                       radius=radius, wl=wls[i], origin=it["Zorigin"], pupil=pupil)
    return plans


def _surface_gates(items):
    """Mt, Ms, fl, T, n1n2 of one surface for every item -- the quantities run.py:181-190 reads off the
    surface's ABCD matrices (fl = cout / power, inf for a powerless surface; T = cout * thickness)."""
    Mt, fl, T, n1n2 = zip(*[it["ABCDt"].gates() for it in items])
    Ms = [it["ABCDs"].M for it in items]
    return Mt, Ms, fl, T, n1n2


# The same for the columns of ABCD objects the walk has gathered anyway, as arrays, remembered per column: the batches of a
# sweep or of a Monte-Carlo study meet the SAME matrix objects at every step (parse_config's, shared by the shallow copies
# inject_wfe makes), and asking 2 x 256 objects per surface again was a tenth of the host's time per step at 256 wavefronts.
# An entry is valid while the column holds the very same objects (list equality on objects without __eq__ is identity, at C
# speed) and no matrix has been edited in place since (abcd.EPOCH).
_GATE_COLUMNS = {}


def _gate_arrays(col_t, col_s):
    key = (id(col_t[0]), id(col_s[0]), len(col_t))
    hit = _GATE_COLUMNS.get(key)
    if hit is not None and hit[0] == _abcd.EPOCH[0] and hit[1] == col_t and hit[2] == col_s:
        return hit[3]
    Mt, fl, T, n1n2 = zip(*[a.gates() for a in col_t])
    arrays = tuple(np.array(v, dtype=np.float64) for v in (Mt, [a.M for a in col_s], fl, T, n1n2))
    if len(_GATE_COLUMNS) >= 1024:
        _GATE_COLUMNS.clear()
    _GATE_COLUMNS[key] = (_abcd.EPOCH[0], list(col_t), list(col_s), arrays)
    return arrays


def _inert(item):
    """A surface that does nothing to the field whatever the beam: no aperture, no stop, no phase, and identity ABCD
    matrices (zero thickness, no power, unit magnification, no change of medium) -- the image plane right behind a slit,
    a coordinate break.  (What run.py:181-207 would gate on: Mt = Ms = 1, fl = inf, T = 0, n1n2 = 1.)"""
    if item.get("aperture") is not None or item.get("is_stop") or item["type"] in ("Zernike", "Grid Sag", "PSD"):
        return False
    Mt, fl, T, n1n2 = item["ABCDt"].gates()
    return Mt == 1.0 and item["ABCDs"].M == 1.0 and math.isinf(fl) and T == 0.0 and n1n2 == 1.0


def _live_rows_after(plans, live, n):
    """Rows that may be non-zero after the stand-alone apertures of ``plans`` were applied: a clear
    aperture (not an obscuration) leaves exact zeros outside its bounding box (photutils' box,
    widened by a pixel here).  ``live`` is updated in place, one [lo, hi) per item."""
    if getattr(plans, "ap", None) is not None:
        plans.ap.narrow_rows(live, n)
        return
    for i, p in enumerate(plans):
        ap = p["aperture"]
        if ap is None or ap[1] or ap[0].theta != 0.0:
            continue
        h = ap[0]
        ext = h.b if isinstance(h, EllipticalAperture) else h.h / 2.0
        yc = h._xy[1]
        if not (math.isfinite(yc) and math.isfinite(ext) and ext > 0.0):
            continue
        lo = max(0, int(math.floor(yc - ext + 0.5)) - 1)
        hi = min(n, int(math.ceil(yc + ext + 0.5)) + 1)
        if lo >= hi:
            continue
        live[i][0] = max(live[i][0], lo)
        live[i][1] = min(live[i][1], hi)
        if live[i][0] >= live[i][1]:  # disjoint boxes: everything is zero; keep a token range
            live[i][0], live[i][1] = 0, 0


def _launch_apertures(dev, plans):
    if getattr(plans, "ap", None) is not None:
        blocks, codes = plans.ap.blocks()
        for code in (_lib.SHAPE_ELLIPSE, _lib.SHAPE_RECT):
            mine = codes == float(code)
            if mine.any():
                b = blocks.copy()
                b[~mine] = 0.0
                dev.aperture(code, b)
        return
    for code, cls_is_ellipse in ((_lib.SHAPE_ELLIPSE, True), (_lib.SHAPE_RECT, False)):
        blocks, any_on = [], False
        for p in plans:
            ap = p["aperture"]
            if ap is not None and isinstance(ap[0], EllipticalAperture) == cls_is_ellipse:
                blocks.append(ap[0].block(obscuration=ap[1]))
                any_on = True
            else:
                blocks.append(_OFF_APERTURE)
        if any_on:
            dev.aperture(code, blocks)


def _launch_zernike(dev, plans, want_wfe=False, same_as=None):
    if isinstance(plans, _Plans) and not plans.summary()[1]:
        return None
    zs = [p["zernike"] for p in plans]
    if not any(z is not None for z in zs):
        return None
    nmax = max(int(z["n"].max()) for z in zs if z is not None)
    kdim = nmax // 2 + 1
    stride = _lib.ZERNIKE_HEAD + 2 * (nmax + 1) * kdim
    table = jacobi_recurrence(nmax)
    ortho = [z is not None and z["pupil"] is not None for z in zs]

    def build(coeffs):
        blocks = np.zeros((len(plans), stride), dtype=np.float64)
        # (items of a sweep share polynomial tables and coefficients and differ in the header only -- sampling, radius,
        # wavelength: the coefficient planes are formed once per distinct (tables, coefficients) and copied)
        planes = {}
        for i, z in enumerate(zs):
            if z is None:
                continue
            c = np.asarray(coeffs[i], dtype=np.float64)
            key = (id(z["m"]), id(z["n"]), id(z["norm"]), c.tobytes(), z["origin"])
            hit = planes.get(key)
            if hit is None:
                hit = planes[key] = zernike_block(z["m"], z["n"], z["norm"], c, 1.0, 1.0, 1.0, 1.0, origin=z["origin"], nmax=nmax)[0]
            row = blocks[i]
            row[:] = hit
            row[1], row[2], row[3], row[7] = z["dx"], z["dy"], z["radius"], 1.0 / z["wl"]
        return blocks

    coeffs = [z["Z"] if z is not None else None for z in zs]
    if not any(ortho):
        if same_as is not None:  # (the items named hold copies of one field: the surface right behind the start)
            return dev.zernike(nmax, kdim, table, build(coeffs), want_wfe=want_wfe, same_as=same_as)
        return dev.zernike(nmax, kdim, table, build(coeffs), want_wfe=want_wfe)

    # PolyOrthoNorm (zernike.py:320-402): U = M Z, so sum_k c_k U_k = sum_n (M^T c)_n Z_n -- the
    # ordinary expansion with transformed coefficients, restricted to the pupil.
    live = [z for z in zs if z is not None]
    if not all(ortho[i] for i, z in enumerate(zs) if z is not None):
        raise NotImplementedError("a batch must use Zorthonorm on all of its items or on none")
    if any(not (np.array_equal(z["m"], live[0]["m"]) and np.array_equal(z["n"], live[0]["n"])
                and np.array_equal(z["norm"], live[0]["norm"])) for z in live):
        raise NotImplementedError("batched Zorthonorm surfaces must share index range, ordering and normalisation")
    for code, is_ellipse in ((_lib.SHAPE_ELLIPSE, True), (_lib.SHAPE_RECT, False)):
        ap = np.zeros((len(plans), _lib.APERTURE_STRIDE), dtype=np.float64)
        for i, z in enumerate(zs):
            if z is not None and isinstance(z["pupil"], EllipticalAperture) == is_ellipse:
                ap[i] = z["pupil"].block(obscuration=False)
        if ap[:, 0].any():  # items of the other shape keep enable = 0 and are left alone
            dev.pupil_aperture(code, ap)
    k = len(live[0]["m"])
    sums, counts = dev.zernike_gram(nmax, kdim, table, build(coeffs), gram_polynomials(live[0]["m"], live[0]["n"],
                                                                                      live[0]["norm"]), pupil=True)
    for i, z in enumerate(zs):
        if z is not None:
            coeffs[i] = orthonorm_matrix(sums[i], counts[i], k).T @ np.asarray(z["Z"], dtype=np.float64)
    return dev.zernike(nmax, kdim, table, build(coeffs), want_wfe=want_wfe, pupil=True)


def _launch_phase_maps(dev, plans, wfe):
    """Grid Sag / PSD surfaces (run.py:154-177): each item's host-built WFE map multiplies its field
    (paos_phase_map).  Returns what the surface's ``wfe`` entry holds for a single wavefront: the map
    of the LAST of Zernike / Grid Sag / PSD (each assignment at run.py:143,156,168 overwrites)."""
    if isinstance(plans, _Plans) and not plans.summary()[2]:
        return wfe
    groups = {}  # items that carry the very same map object (round 5: _sag_map_once): one upload for all of them
    for i, p in enumerate(plans):
        if p["phase_map"] is not None:
            m, wl = p["phase_map"]
            if isinstance(m, PsdScreen):  # a random screen of this item's own
                if _phase_maps.psd_on_device(dev, m.args[0][0]):
                    # fft2 -> power-law filter -> ifft2 -> roughness on the library's passes (paos_psd_screen); the map stays on
                    # the device and comes back only where the reference returns it (one wavefront: the surface's `wfe`)
                    serial = next(_MAP_SERIAL)
                    out = dev.psd_screen(m.noise, m.rough, m.params, key=serial, want_map=len(plans) == 1)
                    dev.phase_map_items(None, [i], [wl], key=serial)
                    if len(plans) == 1:
                        wfe = np.ma.masked_array(out, mask=np.zeros(out.shape, dtype=bool))
                    p["phase_map"] = None  # (its two draws: 2 x 8 n^2 bytes per item)
                    continue
                m = m.host_map()  # complex64 contexts: the screen is built in doubles on the host
                p["phase_map"] = None
            groups.setdefault(id(m), (m, [], []))
            groups[id(m)][1].append(i)
            groups[id(m)][2].append(wl)
            if len(plans) == 1:
                # (a map _sag_map_once keeps is handed out as a copy: the caller owns what run() returns, like the reference's)
                wfe = m.copy() if any(hit[2] is m for hit in _SAG_MAPS.values()) else m
    for m, idx, wls in groups.values():
        if len(idx) > 1 and hasattr(dev, "phase_map_items"):
            filled = _FILLED_MAPS.get(id(m))
            if filled is None or filled[0] is not m:
                if len(_FILLED_MAPS) >= 2:
                    _FILLED_MAPS.pop(next(iter(_FILLED_MAPS)))
                filled = _FILLED_MAPS[id(m)] = (m, np.ascontiguousarray(np.ma.filled(m, 0.0), dtype=np.float64), next(_MAP_SERIAL))
            # (the key under which the library keeps the copy on the device: a serial number given to this map object when it was
            # first sent -- never the object's id, which the allocator hands out again once an evicted map has been freed)
            dev.phase_map_items(filled[1], idx, wls, key=filled[2])
        else:
            for i, wl in zip(idx, wls):
                dev.phase_map(i, np.ma.filled(m, 0.0), wl)
    return wfe


_FILLED_MAPS = {}  # id(map) -> (map, its zero-filled contiguous copy: what crosses PCIe, serial number = the library's content key)


def _queue_apertures(comp, plans):
    """Apertures as pass operators (their weight maps are rendered right before the pass
    they ride on, csrc/paos_hip.hip: launch_one_pass)."""
    if getattr(plans, "ap", None) is not None:
        comp.aperture(plans.ap.blocks())
        return
    recs = []
    for p in plans:
        ap = p["aperture"]
        if ap is None:
            recs.append(None)
        else:
            code = _lib.SHAPE_ELLIPSE if isinstance(ap[0], EllipticalAperture) else _lib.SHAPE_RECT
            recs.append((ap[0].block(obscuration=ap[1]), code))
    comp.aperture(recs)


def _queue_steps(comp, lens, stw, ptp, wts, inv_stw, inv_wts):
    """Lens and stw / ptp / wts of one surface go to the pass compiler, in the fixed slot
    order every regime respects (OI: stw, ptp; IO: ptp, wts; OO: stw, wts; II: ptp --
    wfo.py:560-570).  Blocks are [batch][5] arrays with enable = 0 where an item skips the step.
    Nothing is launched here: consecutive surfaces fuse (passes.py)."""
    comp.lens(lens)
    comp.stw(stw, inv_stw != 0.0)
    comp.ptp(ptp)
    comp.wts(wts, inv_wts != 0.0)


def _start_field(dev, plans, value, write_rows=None, write_cols=None):
    """The wavefront is still the constant ``value`` (wfo.py:118).  When every item opens with a
    stand-alone aperture of one shape, ones -> aperture -> [make_stop] is a single write of the
    field (paos_start); otherwise fill and let the surface run as usual.  Returns False, or -- when the
    first surface's aperture and stop are done -- per item the index of the first item with the same field.  ``write_rows`` ([lo, hi) per item): only these rows are
    written, the others are left standing for zeros (paos_start_rows)."""
    batch_ap = getattr(plans, "ap", None)
    if batch_ap is not None:
        if bool(batch_ap.rect.any()) and not bool(batch_ap.rect.all()):
            dev.fill(value)
            return None
        code = _lib.SHAPE_RECT if bool(batch_ap.rect.all()) else _lib.SHAPE_ELLIPSE
        blocks = batch_ap.blocks()[0].tolist()
    else:
        aps = [p["aperture"] for p in plans]
        if any(a is None for a in aps) or len({isinstance(a[0], EllipticalAperture) for a in aps}) != 1:
            dev.fill(value)
            return None
        code = _lib.SHAPE_ELLIPSE if isinstance(aps[0][0], EllipticalAperture) else _lib.SHAPE_RECT
        blocks = [a[0].block(obscuration=a[1]) for a in aps]
    stops = [1.0 if p["stop"] else 0.0 for p in plans]
    if write_rows is None:
        dev.start(value, code, blocks, stops)
    elif write_cols is None:
        dev.start(value, code, blocks, stops, write_rows=write_rows)
    else:
        dev.start(value, code, blocks, stops, write_rows=write_rows, write_cols=write_cols)
    # which items now hold the same field: same aperture record, stop flag and row window
    seen, same_as = {}, []
    for i, (b, st) in enumerate(zip(blocks, stops)):
        key = (tuple(float(x) for x in b), st, tuple(write_rows[i]) if write_rows is not None else None,
               tuple(write_cols[i]) if write_cols is not None else None)
        same_as.append(seen.setdefault(key, i))
    return same_as


class _WalkState:
    """What ``on_saved`` may want to know about the field at a saved surface of a lean walk:
    ``rows`` -- per item [lo, hi) outside which the field is zero (or stands for zero), or None;
    ``psf_ticket`` -- set when the pass program that ended at this surface has already written |u|^2 to the
    PSF buffer and enqueued its sum (the field itself is then undefined);
    ``same_as`` -- per item the index of an item whose field is known to be identical (the first surface of a sweep:
    the start field does not depend on the wavelength), or None."""

    def __init__(self):
        self.rows, self.psf_ticket, self.same_as = None, None, None
        self.cols = None  # with ``rows``: the columns outside which those rows stand for zeros as well (the start box)
        # set once a pass program has stored the PSF for good (nothing but inert surfaces follows): the saved surfaces
        # still to come report this ticket's power, and the PSF is already where keep_psf wants it
        self.final_ticket = None


def _walk(dev, states, chains, on_saved, stats=None, fresh=None, lean=None, psf_at=None, power_state=None):
    """Drive all items through their chains in lock-step, one surface at a time.  ``fresh``: the
    constant the field is meant to hold but has not been filled with yet (see _start_field).
    ``lean`` (a _WalkState, optional): the caller reads no arrays at saved surfaces, only powers (through
    ``lean.rows`` / ``lean.psf_ticket``) -- the walk may then leave dead rows unwritten at the start and, when
    ``psf_at`` names the last surface, have the last pass store the PSF instead of the field."""
    _SAG_SEEN.clear()  # (a Grid Sag array is hashed once per walk, not once per item)
    keys = [list(c.keys()) for c in chains]
    if any(k != keys[0] for k in keys[1:]):
        raise ValueError("batched chains must list the same surfaces (same keys, same order)")
    st0 = states[0]
    if any((st.pupil_diameter, st.gridsize, st.zoom) != (st0.pupil_diameter, st0.gridsize, st0.zoom) for st in states):
        raise ValueError("the wavefronts of a batch share beam diameter, grid and zoom")
    beams = BeamBatch(st0.pupil_diameter, [st.wavelength for st in states], st0.gridsize, st0.zoom)
    state, n = beams.state, beams.n
    comp = (SeparableCompiler if _passes.SEPARABLE else PassCompiler)(len(states), dev.n)
    npass = 0
    twins = None  # per item the index of an item whose field is a copy of its own (None: nothing known)
    # rows of each item known to be exactly zero in memory (outside [lo, hi)): set by stand-alone
    # apertures, kept by stops / Zernike / phase screens (they multiply), handed to the next pass
    # program (which skips them) and forgotten once that program has run
    live = [[0, dev.n] for _ in states]
    stale = [False]  # rows outside ``live`` hold old data that stands for zeros (lean start)
    stale_cols = [None]  # ... and, while ``stale``, so do the columns outside these [lo, hi) per item (paos_start_box)
    dead = [False]   # the field has been given up for its PSF (lean end): nothing may run on it any more

    prog_power = [None]  # ticket of the power of the field a program has just stored (flush(final_power=True))
    factor_cols_t, factor_cols_s = [], []  # [surface][item] ABCD factors met so far (run.py:215-219)
    all_still = [all(st.still for st in states)]  # every item on axis: nothing moves the rays (refreshed behind a coordinate break)

    def known_rows():
        return [list(r) for r in live] if any(r[0] > 0 or r[1] < dev.n for r in live) else None

    def settle():
        """Before anything reads whole fields: rows that stand for zeros become zeros."""
        if stale[0]:
            if stale_cols[0] is not None:
                dev.zero_outside_rows(live, stale_cols[0])
            else:
                dev.zero_outside_rows(live)
            stale[0], stale_cols[0] = False, None

    def keep_reads_field():
        """The saved LAST surface of a lean walk that keeps its PSFs, reached without a pass program having stored
        them (a single-surface chain; every hop shorter than lambda / 1000): ``on_saved`` then takes |u|^2 of the WHOLE
        field (paos_psf_keep / paos_psf_keep_power), so rows that merely stand for zeros have to be zeros first
        (the PSF and its power came out NaN on the model device)."""
        if psf_at is not None and key == psf_at and lean.psf_ticket is None:
            settle()

    def flush(final_intensity=False, final_power=False):
        if dead[0]:
            comp.program()  # drop what was queued behind the PSF store
            return 0
        rows = known_rows()
        if not comp.pending():
            return 0
        cols = stale_cols[0] if (stale[0] and rows is not None) else None
        if final_intensity:
            done, ticket = comp.flush(dev, live_rows=rows, rows_stale=stale[0] and rows is not None, final_intensity=True,
                                      live_cols=cols)
            lean.psf_ticket = lean.final_ticket = ticket
            dead[0] = True
        elif final_power:
            # the surface reached by this program is saved (or a stop) and nothing but the program touches its field
            # first: the last pass sums |u|^2 while it stores (paos_run_program: final_intensity = 2) -- no sweep that
            # reads the field back
            if power_state is not None:
                power_state["before"]()  # (the ticket is taken inside: room in the ring first)
            done, ticket = comp.flush(dev, live_rows=rows, rows_stale=stale[0] and rows is not None, final_intensity=2,
                                      live_cols=cols)
            prog_power[0] = ticket if done else None
        else:
            done = comp.flush(dev, live_rows=rows, rows_stale=stale[0] and rows is not None, live_cols=cols)
        if done:
            stale[0], stale_cols[0] = False, None
            for r in live:
                r[0], r[1] = 0, dev.n
        return done

    # from which surface on nothing touches the field any more (index into keys[0]; len = never)
    order = keys[0]
    inert_from = len(order)
    while inert_from > 0 and all(_inert(c[order[inert_from - 1]]) for c in chains):
        inert_from -= 1
    for pos, key in enumerate(order):
        items = [c[key] for c in chains]
        if power_state is not None:
            if power_state["ticket"] is not None and not power_state.get("used"):
                dev.norm2_release(power_state["ticket"])  # (taken for a stop on a surface nobody saved)
            power_state["ticket"], power_state["post"], power_state["used"] = None, None, False  # (a ticket belongs to its surface)
        dxs, dys, wls = state[:, beams.DX].tolist(), state[:, beams.DY].tolist(), state[:, beams.WL].tolist()
        readout = []

        def wz_dtf():
            if not readout:
                readout.extend(beams.readout())
            return readout

        plans = _plan_batch(states, items, n, dxs, dys, wls, lambda i: wz_dtf()[0][i], all_still[0])
        if not all_still[0] or items[0]["type"] == "Coordinate Break" or any(it["type"] == "Coordinate Break" for it in items):
            all_still[0] = all(st.still for st in states)
        saved = any(it["save"] for it in items)
        if saved:  # push_results scalars (run.py:12-27) are those BEFORE magnification / lens / propagate
            wz, dtf = wz_dtf()
            fr, props, ext = state[:, beams.FRATIO].tolist(), beams.propagators(), beams.extents()
            for i, (it, p) in enumerate(zip(items, plans)):
                if it["save"]:
                    p["scalars"] = {"wz": float(wz[i]), "distancetofocus": float(dtf[i]), "fratio": fr[i],
                                    "dx": dxs[i], "dy": dys[i], "wl": wls[i], "extent": ext[i],
                                    "propagator": props[i]}
        # the pilot beams of the whole batch through this surface (one C call), the rays and the
        # accumulated ABCD factors of each item
        # (the ray-transfer factors met so far, one column per surface; a saved surface gets its product -- lazily)
        col_t, col_s = [it["ABCDt"] for it in items], [it["ABCDs"] for it in items]
        factor_cols_t.append(col_t)
        factor_cols_s.append(col_s)
        lens, stw, ptp, wts, inv_stw, inv_wts = beams.surface(*_gate_arrays(col_t, col_s))
        if not all_still[0]:
            for i, st in enumerate(states):
                if not st.still:
                    st.vt = col_t[i]() @ st.vt
                    st.vs = col_s[i]() @ st.vs
        if saved:
            for i, (it, p) in enumerate(zip(items, plans)):
                if it["save"]:
                    p["ABCDt"] = ABCDProduct(tuple(col[i] for col in factor_cols_t))
                    p["ABCDs"] = ABCDProduct(tuple(col[i] for col in factor_cols_s))

        if fresh is not None and not saved and all(_inert(it) for it in items):
            # (coordinate breaks in front of the first mirror: the field is still the constant it will be filled with;
            # the first surface that does something -- usually aperture + stop -- then writes it in one go, _start_field)
            continue
        if fresh is not None:
            value, fresh = fresh, None
            rows0 = cols0 = None
            if lean is not None and hasattr(dev, "zero_outside_rows"):
                trial = [[0, dev.n] for _ in states]
                _live_rows_after(plans, trial, dev.n)
                # worth it only when every item's aperture leaves whole rows dark
                if all(r[0] > 0 or r[1] < dev.n for r in trial) and all(r[0] < r[1] for r in trial):
                    rows0 = trial
                    if START_BOX:  # ... and whole columns: the field is written inside the aperture's box only
                        cols0 = _live_cols_of(plans, dev.n)
            same_as = _start_field(dev, plans, value, write_rows=rows0, write_cols=cols0)
            if same_as is not None:
                _live_rows_after(plans, live, dev.n)
                stale[0] = rows0 is not None
                stale_cols[0] = cols0 if stale[0] else None
                want_wfe = len(plans) == 1 and bool(items[0]["save"])
                # still copies of each other unless this surface puts a wavefront error on them
                untouched = not any(p["zernike"] is not None or p["phase_map"] is not None for p in plans)
                copies = same_as if len(set(same_as)) < len(same_as) else None
                wfe = _launch_zernike(dev, plans, want_wfe=want_wfe, same_as=copies if TWIN_FIELDS else None)
                wfe = _launch_phase_maps(dev, plans, wfe)
                if saved:
                    if lean is not None:
                        keep_reads_field()
                        lean.rows = known_rows()
                        lean.cols = stale_cols[0] if (stale[0] and lean.rows is not None) else None
                        lean.same_as = copies if untouched else None
                    on_saved(key, items, plans, wfe)
                    if lean is not None:
                        lean.same_as, lean.cols = None, None
                _queue_steps(comp, lens, stw, ptp, wts, inv_stw, inv_wts)
                # (round 5) ... and they stay copies until something wavelength-dependent is applied or queued: a Zernike surface
                # right behind the start (SYN20's S02) reads one field per group of them (paos_zernike_like)
                twins = copies if (TWIN_FIELDS and untouched and not comp.pending()) else None
                continue
        fuse_ap = FUSE_APERTURES
        any_stop, any_zern, any_map, any_ap = plans.summary()
        own_breaker = saved or any_stop or any_zern or any_map
        if fuse_ap == "auto":
            # an aperture followed on its own surface by a stop / Zernike / save could only ride
            # on a transform-free pass: the stand-alone aperture kernel is cheaper there
            # -- unless the previous propagation left a pass open: then the aperture becomes the
            # last operator of that pass (applied while the tile is stored) at no extra traffic
            if plans.ap is not None:
                fuse_ap = (not own_breaker or comp.open_takes_mask()) and plans.ap.fits_line_records(dev.n, dev.precision)
            else:
                aps = [p["aperture"] for p in plans if p["aperture"] is not None]
                fuse_ap = (bool(aps) and (not own_breaker or comp.open_takes_mask()) and
                           all(_aperture_fits_line_records(h, o, dev.n, dev.precision) for h, o in aps))
        if fuse_ap:
            _queue_apertures(comp, plans)
        breaker = own_breaker or (any_ap and not fuse_ap)
        if breaker:
            # the field must be current before a non-fusable operator -- unless all this surface wants of it is the
            # PSF of every item (a lean walk's last, saved surface with nothing else on it): then the last pass
            # stores |u|^2 instead
            only_saved = (saved and all(it["save"] for it in items) and
                          not (any_stop or any_zern or any_map or (any_ap and not fuse_ap)))
            # ... the last surface -- or a surface behind which only inert ones follow (the image plane right behind a
            # saved slit): what the program stores here is what the chain ends with
            ends_here = key == psf_at or (INERT_TAIL and pos + 1 >= inert_from and psf_at == order[-1])
            as_psf = (lean is not None and psf_at is not None and ends_here and only_saved and comp.pending()
                      and not dead[0])
            # a stop right behind the program needs the power of what the program stores; a saved surface whose
            # field IS what the program stores reports it
            stop_rides = (STOP_FROM_PROGRAM and any_stop and comp.pending() and not dead[0] and
                          not (any_ap and not fuse_ap))
            npass += flush(final_intensity=as_psf,
                           final_power=(not as_psf and comp.pending() and not dead[0] and
                                        (stop_rides or (power_state is not None and only_saved))))
        if not fuse_ap and any_ap:
            _launch_apertures(dev, plans)
            if not comp.pending():
                _live_rows_after(plans, live, dev.n)
        if any_stop:
            settle()
            flags = [1.0 if p["stop"] else 0.0 for p in plans]
            if prog_power[0] is not None:  # the program's last pass has summed the power on its way out
                dev.make_stop(flags, power_known=True, defer=STOP_DEFERRED)
                if power_state is not None:
                    # should the surface be saved: the power BEHIND the stop is P (1 / sqrt P)^2 where the stop applies
                    power_state["ticket"] = prog_power[0]
                    power_state["post"] = lambda v, f=np.array(flags): np.where(f != 0.0, v * (1.0 / np.sqrt(v)) ** 2, v)
                else:
                    dev.norm2_release(prog_power[0])
                prog_power[0] = None
            else:
                dev.make_stop(flags)
        elif prog_power[0] is not None:
            if power_state is not None:
                power_state["ticket"], power_state["post"] = prog_power[0], None
            else:
                dev.norm2_release(prog_power[0])
            prog_power[0] = None
        want_wfe = len(plans) == 1 and bool(items[0]["save"])
        if twins is not None and (comp.pending() or fuse_ap or any_ap or any_stop):
            twins = None  # (something has been applied to the fields, or waits to be)
        wfe = _launch_zernike(dev, plans, want_wfe=want_wfe, same_as=twins)
        if any_zern or any_map:
            twins = None
        wfe = _launch_phase_maps(dev, plans, wfe)
        if saved:
            if lean is not None:
                keep_reads_field()
                lean.rows = known_rows() if not comp.pending() else None
                lean.cols = stale_cols[0] if (stale[0] and lean.rows is not None) else None
            else:
                settle()
            on_saved(key, items, plans, wfe)
            if lean is not None:
                lean.rows, lean.psf_ticket, lean.cols = None, None, None
        _queue_steps(comp, lens, stw, ptp, wts, inv_stw, inv_wts)
        if comp.pending():
            twins = None
    if power_state is not None and power_state["ticket"] is not None and not power_state.get("used"):
        dev.norm2_release(power_state["ticket"])  # (taken for a stop on the last surface, which nobody saved)
        power_state["ticket"] = None
    if fresh is not None:  # an empty chain still yields the initial wavefront
        dev.fill(fresh)
    npass += flush()
    settle()
    if stats is not None:
        stats["fused_passes"] = npass


# One idle single-wavefront context per (grid, precision, device) is kept between run() calls:
# creating and destroying one (HBM, pinned staging, twiddles) costs ~7 ms, several times the
# propagation of a small grid.  A context is either in the pool or in use by exactly one call, so
# concurrent run() calls from several threads each get their own.
_IDLE_CONTEXTS = {}
_IDLE_LOCK = threading.Lock()


def _borrow_context(n, precision, device):
    with _IDLE_LOCK:
        dev = _IDLE_CONTEXTS.pop((n, precision, device), None)
    return dev if dev is not None else _lib.DeviceFields(n, 1, precision, device)


def _return_context(dev, n, precision, device):
    with _IDLE_LOCK:
        if (n, precision, device) not in _IDLE_CONTEXTS:
            _IDLE_CONTEXTS[(n, precision, device)] = dev
            return
    dev.close()


def release_contexts():
    """Free the idle contexts kept by run() (also done at interpreter exit)."""
    with _IDLE_LOCK:
        devs = list(_IDLE_CONTEXTS.values())
        _IDLE_CONTEXTS.clear()
    for dev in devs:
        dev.close()


atexit.register(release_contexts)


def run(pupil_diameter, wavelength, gridsize, zoom, field, opt_chain, precision="fp64", device=0):
    """Drop-in for ``paos.core.run.run``: same arguments, same returned dict
    ``{num: {aperture, [wfe], amplitude, wz, distancetofocus, fratio, phase, dx, dy, wfo,
    wl, extent, propagator, ABCDt, ABCDs}}`` for surfaces with ``save`` set."""
    assert isinstance(opt_chain, dict), "opt_chain must be a dict"
    retval = {}
    state = _Item(pupil_diameter, wavelength, gridsize, zoom, field)
    dev = _borrow_context(int(gridsize), precision, device)

    def on_saved(key, items, plans, wfe):
        item, plan = items[0], plans[0]
        rec = {"aperture": plan["aperture"][0] if plan["aperture"] else None}
        if isinstance(wfe, np.ma.MaskedArray):  # Grid Sag / PSD: the host-built map itself
            rec["wfe"] = wfe
        elif wfe is not None:
            outside = np.isnan(wfe)
            rec["wfe"] = np.ma.MaskedArray(data=np.where(outside, 0.0, wfe), mask=outside,
                                           fill_value=0.0)
        s = plan["scalars"]
        rec.update({
            "amplitude": dev.download(0, _lib.WHAT_AMPLITUDE),
            "wz": s["wz"], "distancetofocus": s["distancetofocus"], "fratio": s["fratio"],
            "phase": dev.download(0, _lib.WHAT_PHASE),
            "dx": s["dx"], "dy": s["dy"],
            "wfo": dev.download(0, _lib.WHAT_FIELD),
            "wl": s["wl"], "extent": s["extent"], "propagator": s["propagator"],
        })
        rec["_plan"] = plan  # ABCDs are attached once the surface is fully processed
        retval[item["num"]] = rec

    try:
        _walk(dev, [state], [opt_chain], on_saved, fresh=1.0 + 0.0j)
        dev.sync()
    except BaseException:
        dev.close()  # whatever state it is in, it does not go back to the pool
        raise
    _return_context(dev, int(gridsize), precision, device)
    for rec in retval.values():
        plan = rec.pop("_plan")
        rec["ABCDt"] = deepcopy(plan["ABCDt"])
        rec["ABCDs"] = deepcopy(plan["ABCDs"])
    return retval


def run_batch(pupil_diameter, wavelengths, gridsize, zoom, field, opt_chains, precision="fp64",
              device=0, outputs=("psf",), dev=None, sync=True, stats=None, metrics_radii_px=None,
              keep_psf=False, power=True):
    """Propagate ``B = len(opt_chains)`` wavefronts together on one GPU.

    ``wavelengths[i]`` / ``opt_chains[i]`` describe wavefront ``i`` (chains must
    contain the same surfaces).  Returns a list of ``B`` dicts
    ``{num: {scalars..., 'power': sum|u|^2, ['psf'], ['wfo'], ['amplitude'], ['phase']}}``
    for saved surfaces (with ``sync=False`` the power is left as ``'power_ticket'``, a :class:`PowerTicket`:
    ``ticket.fetch()[i]`` -- or ``dev.norm2_fetch(ticket)[i]`` -- is item i's power; records whose powers come from one
    reduction share one handle state, so every record may be fetched, in any order, any number of times).  ``outputs`` picks which N x N arrays are copied back to the
    host per saved surface and item ('psf' = |u|^2, plot.py:125-130); ``()`` keeps
    every array on the GPU and returns scalars and the power only -- the mode the
    throughput benchmark uses.  ``metrics_radii_px`` (up to 16 radii, pixels) adds
    ``'metrics'`` = power, centroid, peak and encircled power per radius of |u|^2 computed on
    the GPU (about the grid centre) -- what a Monte-Carlo encircled-energy study needs, without
    moving a PSF.  ``keep_psf`` writes |u|^2 of every item at the LAST surface of the chain (when it
    is saved) into the context's device-resident PSF buffer (``dev.psf_fetch(i)`` reads one back):
    the final intensity write of a run whose PSFs stay in HBM.  ``power=False`` skips the sum |u|^2
    reduction per saved surface (the reference does not return it; chains that save a dozen
    surfaces spend 7 % of their time there).  ``dev`` may pass a pre-allocated
    ``DeviceFields(gridsize, B)`` to reuse across calls.
    """
    nb = len(opt_chains)
    if len(wavelengths) != nb:
        raise ValueError("one wavelength per chain is required")
    if nb == 0:
        return []
    unknown = set(outputs) - {"psf", "wfo", "amplitude", "phase"}
    if unknown:
        raise ValueError(f"unknown outputs {sorted(unknown)}")
    states = [_Item(pupil_diameter, wl, gridsize, zoom, field) for wl in wavelengths]
    own = dev is None
    if own:
        dev = _lib.DeviceFields(int(gridsize), nb, precision, device)
    elif dev.batch != nb or dev.n != int(gridsize):
        raise ValueError("supplied DeviceFields does not match the batch")
    results = [dict() for _ in range(nb)]
    what = {"psf": _lib.WHAT_INTENSITY, "wfo": _lib.WHAT_FIELD, "amplitude": _lib.WHAT_AMPLITUDE,
            "phase": _lib.WHAT_PHASE}

    last_key = list(opt_chains[0].keys())[-1] if len(opt_chains[0]) else None
    # nobody reads an array at a saved surface: the walk may skip writing dead rows at the start and store the PSF
    # straight from the last pass (csrc/frugal_pass.h: STORE)
    lean = _WalkState() if (not outputs and metrics_radii_px is None) else None
    tickets = []  # (_Reduction, [(item index, record)], post): powers are fetched after the walk, so the
    # host keeps planning while the GPU works (no mid-chain synchronisation) -- except when a chain
    # saves more surfaces than the library has ticket slots: then the oldest are fetched early
    drained = [0]
    # The reduction a pass program left behind when it stored the PSF for good answers for every saved surface from
    # there on (lean.final_ticket): those entries share ONE _Reduction, fetched once -- also when a mid-walk drain has
    # already read it.  Every other ticket is a reduction of its own (a slot number alone does not say which: the
    # library hands a slot out again as soon as it was fetched or released).
    final_reduction = {}

    def reduction_of(ticket, final=False):
        if not final:
            return _Reduction(dev, ticket)
        if ticket not in final_reduction:
            final_reduction[ticket] = _Reduction(dev, ticket)
        return final_reduction[ticket]

    def outstanding():
        return sum(1 for e in tickets[drained[0]:] if e[0].raw is None and not e[0].released)

    def drain():
        for red, pending, post in tickets[drained[0]:]:
            values = red.values()
            if post is not None:
                values = post(values)
            for i, rec in pending:
                rec["power"] = float(values[i])
        drained[0] = len(tickets)

    def room_in_the_ring():
        if outstanding() >= _lib.NORM_SLOTS - 2:
            drain()

    # the power of a saved surface summed by the pass that stores its field (csrc/frugal_pass.h: pow_partial): _walk
    # leaves the ticket here when the surface's field is exactly what its pass program stored
    power_state = ({"ticket": None, "post": None, "used": False, "before": room_in_the_ring}
                   if (power and POWER_ON_STORE) else None)

    def on_saved(key, items, plans, wfe):
        pending = []
        for i, (item, plan) in enumerate(zip(items, plans)):
            if not item["save"]:
                continue
            rec = dict(plan["scalars"])
            rec["aperture"] = plan["aperture"][0] if plan["aperture"] else None
            for name in outputs:
                rec[name] = dev.download(i, what[name])
            rec["_plan"] = plan
            results[i][item["num"]] = rec
            pending.append((i, rec))
        if metrics_radii_px is not None:
            met = dev.psf_metrics(metrics_radii_px)
            for i, rec in pending:
                rec["metrics"] = met[i]
        keep = keep_psf and key == last_key
        # the last pass has stored |u|^2 and enqueued its sum -- for this surface, or for good (inert surfaces behind it)
        fused = (lean.psf_ticket if lean.psf_ticket is not None else lean.final_ticket) if lean is not None else None
        rows = lean.rows if lean is not None else None
        if power:
            if outstanding() >= _lib.NORM_SLOTS - 1 and fused is None:
                drain()  # the ticket ring of the library is about to fill: fetch what is pending
            # the saved last surface of a run that keeps its PSFs: |u|^2 written and summed in one sweep
            if fused is not None:
                tickets.append((reduction_of(fused, final=True), pending, None))
            elif power_state is not None and power_state["ticket"] is not None and not keep:
                tickets.append((reduction_of(power_state["ticket"]), pending, power_state.get("post")))
                power_state["used"] = True
            elif keep:
                if power_state is not None and power_state["ticket"] is not None:  # (the kept PSF brings its own sum)
                    dev.norm2_release(power_state["ticket"])
                    power_state["used"] = True
                tickets.append((reduction_of(dev.psf_keep_power()), pending, None))
            else:
                like = lean.same_as if lean is not None else None
                cols = lean.cols if lean is not None else None
                if rows is not None and cols is not None:
                    tickets.append((reduction_of(dev.norm2_enqueue(rows, same_as=like, live_cols=cols)), pending, None))
                elif rows is not None and like is not None:
                    tickets.append((reduction_of(dev.norm2_enqueue(rows, same_as=like)), pending, None))
                else:
                    tickets.append((reduction_of(dev.norm2_enqueue(rows) if rows is not None else dev.norm2_enqueue()),
                                    pending, None))
        elif fused is not None:
            reduction_of(fused, final=True).release()  # (once, however many surfaces it stands for)
        elif keep:
            dev.psf_keep()

    # The walk allocates a few thousand small containers per surface of a large batch (plans, blocks, handles); a
    # generation-2 pass of the cycle collector in the middle of it costs as much as planning several surfaces (round 5:
    # 10 ms of a 29 ms walk at 256 wavefronts).  Nothing the walk builds is cyclic garbage worth collecting before it
    # ends, so the collector is paused for its duration (and left as it was found).
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        _walk(dev, states, list(opt_chains), on_saved, stats=stats, fresh=1.0 + 0.0j, lean=lean,
              psf_at=last_key if (keep_psf and lean is not None) else None, power_state=power_state)
        if sync or own:
            drain()
        else:  # caller synchronises later: hand out handles to the reductions still outstanding (nothing waits here)
            for red, pending, post in tickets[drained[0]:]:
                handle = PowerTicket(red, post)
                for i, rec in pending:
                    rec["power_ticket"] = handle
    finally:
        if gc_was_on:
            gc.enable()
        if own:
            dev.close()
    for res in results:
        for rec in res.values():
            plan = rec.pop("_plan")
            rec["ABCDt"] = plan["ABCDt"]
            rec["ABCDs"] = plan["ABCDs"]
    return results
