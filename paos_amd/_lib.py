"""ctypes binding of libpaoship.so (include/paos_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` / ``make``.
There is deliberately no fallback: if the library or a GPU is missing, creating a
device context raises.
"""
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PAOS_LIB: another build of the same library (an experiment variant from tools/build_variant.sh) -- for A/B runs
# of bench.py; the shipped library is never overwritten.  The path is printed once so that no run uses one unawares.
LIB_PATH = os.environ.get("PAOS_LIB") or os.path.join(_HERE, "libpaoship.so")

PAOS_F64, PAOS_F32 = 0, 1
SHAPE_ELLIPSE, SHAPE_RECT = 0, 1
WHAT_FIELD, WHAT_AMPLITUDE, WHAT_PHASE, WHAT_INTENSITY = 0, 1, 2, 3
KERNEL_PASS_ROWS, KERNEL_PASS_COLS, KERNEL_PASS_ANY = 0, 1, 2
PW_SIGN, PW_QPHASE_CENTRED, PW_QPHASE_NATURAL, PW_SCALE, PW_MASK = 1, 2, 3, 4, 5
PWF_MUL2PI = 1
PWF_X_ONLY, PWF_Y_ONLY = 2, 4  # PW_SIGN: (-1)^column / (-1)^row instead of the checkerboard (-1)^(row + column)
MAX_PW = 6
NORM_SLOTS = 64  # PAOS_NORM_SLOTS: tickets of paos_norm2_enqueue that may be outstanding


class PwOp(ctypes.Structure):
    """paos_pw_op of include/paos_hip.h"""
    _fields_ = [("kind", ctypes.c_int), ("flags", ctypes.c_int), ("block", ctypes.c_int)]


class ProgramOpts(ctypes.Structure):
    """paos_program_opts of include/paos_hip.h"""
    _fields_ = [("live_rows", ctypes.POINTER(ctypes.c_double)), ("rows_stale", ctypes.c_int),
                ("final_intensity", ctypes.c_int), ("power_ticket", ctypes.POINTER(ctypes.c_int)),
                ("live_cols", ctypes.POINTER(ctypes.c_double))]


class Pass(ctypes.Structure):
    """paos_pass of include/paos_hip.h"""
    _fields_ = [("axis", ctypes.c_int), ("fft1", ctypes.c_int), ("fft2", ctypes.c_int),
                ("n_pre", ctypes.c_int), ("n_mid", ctypes.c_int), ("n_post", ctypes.c_int),
                ("pre", PwOp * MAX_PW), ("mid", PwOp * MAX_PW), ("post", PwOp * MAX_PW)]

PHASE_STRIDE = 5
APERTURE_STRIDE = 8
ZERNIKE_HEAD = 8

# every symbol include/paos_hip.h declares: (name, restype, argtypes)
_c_ctx = ctypes.c_void_p
_dbl_p = ctypes.POINTER(ctypes.c_double)
SYMBOLS = {
    "paos_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_c_ctx)]),
    "paos_ctx_destroy": (ctypes.c_int, [_c_ctx]),
    "paos_last_error": (ctypes.c_char_p, [_c_ctx]),
    "paos_sync": (ctypes.c_int, [_c_ctx]),
    "paos_build_info": (ctypes.c_char_p, []),
    "paos_stream": (ctypes.c_void_p, [_c_ctx]),
    "paos_profile_begin": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int]),
    "paos_profile_end": (ctypes.c_int, [_c_ctx, ctypes.POINTER(ctypes.c_int), _dbl_p]),
    "paos_profile_end_split": (ctypes.c_int, [_c_ctx, ctypes.POINTER(ctypes.c_int), _dbl_p, ctypes.POINTER(ctypes.c_int), _dbl_p]),
    "paos_source_hash": (ctypes.c_char_p, []),
    "paos_profile_planned_bytes": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p, ctypes.POINTER(ctypes.c_int)]),
    "paos_profile_line_transforms": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p, ctypes.POINTER(ctypes.c_int)]),
    "paos_profile_end_launches": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p, ctypes.POINTER(ctypes.c_int),
                                                 ctypes.POINTER(ctypes.c_int)]),
    "paos_fill": (ctypes.c_int, [_c_ctx, ctypes.c_double, ctypes.c_double]),
    "paos_import": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_void_p]),
    "paos_export": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "paos_aperture": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p]),
    "paos_aperture_render": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p, _dbl_p]),
    "paos_make_stop": (ctypes.c_int, [_c_ctx, _dbl_p]),
    "paos_stop_scale_last_power": (ctypes.c_int, [_c_ctx, _dbl_p]),
    "paos_stop_defer_last_power": (ctypes.c_int, [_c_ctx, _dbl_p]),
    "paos_norm2": (ctypes.c_int, [_c_ctx, _dbl_p]),
    "paos_norm2_enqueue": (ctypes.c_int, [_c_ctx, ctypes.POINTER(ctypes.c_int)]),
    "paos_norm2_fetch": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p]),
    "paos_norm2_release": (ctypes.c_int, [_c_ctx, ctypes.c_int]),
    "paos_psf_metrics": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p, ctypes.c_double, ctypes.c_double, _dbl_p]),
    "paos_phase": (ctypes.c_int, [_c_ctx, _dbl_p, ctypes.c_int]),
    "paos_phase_map": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p, ctypes.c_double]),
    "paos_ptp": (ctypes.c_int, [_c_ctx, _dbl_p]),
    "paos_stw": (ctypes.c_int, [_c_ctx, _dbl_p, ctypes.c_int]),
    "paos_wts": (ctypes.c_int, [_c_ctx, _dbl_p, ctypes.c_int]),
    "paos_run_passes": (ctypes.c_int, [_c_ctx, ctypes.POINTER(Pass), ctypes.c_int, _dbl_p, ctypes.c_int]),
    "paos_copy_yardstick": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p, _dbl_p]),
    "paos_record_set_stats": (ctypes.c_int, [_c_ctx, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]),
    "paos_ctx_set_pruning": (ctypes.c_int, [_c_ctx, ctypes.c_int]),
    "paos_run_passes_live": (ctypes.c_int, [_c_ctx, ctypes.POINTER(Pass), ctypes.c_int, _dbl_p, ctypes.c_int, _dbl_p]),
    "paos_zernike": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _dbl_p, _dbl_p, ctypes.c_int, _dbl_p]),
    "paos_psf_keep": (ctypes.c_int, [_c_ctx]),
    "paos_psf_keep_power": (ctypes.c_int, [_c_ctx, ctypes.POINTER(ctypes.c_int)]),
    "paos_psf_fetch": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p]),
    "paos_host_alloc": (ctypes.c_int, [ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_void_p)]),
    "paos_host_free": (ctypes.c_int, [ctypes.c_void_p]),
    "paos_export_pinned": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "paos_start": (ctypes.c_int, [_c_ctx, ctypes.c_double, ctypes.c_double, ctypes.c_int, _dbl_p, _dbl_p]),
    "paos_pupil_aperture": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p]),
    "paos_pupil_upload": (ctypes.c_int, [_c_ctx, ctypes.c_int, _dbl_p]),
    "paos_zernike_like": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _dbl_p, _dbl_p, ctypes.c_int, _dbl_p, _dbl_p]),
    "paos_zernike_gram": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _dbl_p, _dbl_p, ctypes.c_int,
                                         ctypes.c_int, _dbl_p, ctypes.c_int, _dbl_p]),
    "paos_zernike_pupil": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _dbl_p, _dbl_p, ctypes.c_int, _dbl_p]),
    "paos_start_rows": (ctypes.c_int, [_c_ctx, ctypes.c_double, ctypes.c_double, ctypes.c_int, _dbl_p, _dbl_p, _dbl_p]),
    "paos_zero_outside_rows": (ctypes.c_int, [_c_ctx, _dbl_p]),
    "paos_phase_map_items": (ctypes.c_int, [_c_ctx, _dbl_p, ctypes.c_ulonglong, ctypes.c_int, _dbl_p, _dbl_p]),
    "paos_psd_screen": (ctypes.c_int, [_c_ctx, _dbl_p, _dbl_p, _dbl_p, ctypes.c_ulonglong, _dbl_p]),
    "paos_start_box": (ctypes.c_int, [_c_ctx, ctypes.c_double, ctypes.c_double, ctypes.c_int, _dbl_p, _dbl_p, _dbl_p, _dbl_p]),
    "paos_zero_outside_box": (ctypes.c_int, [_c_ctx, _dbl_p, _dbl_p]),
    "paos_norm2_enqueue_box": (ctypes.c_int, [_c_ctx, _dbl_p, _dbl_p, _dbl_p, ctypes.POINTER(ctypes.c_int)]),
    "paos_norm2_enqueue_rows": (ctypes.c_int, [_c_ctx, _dbl_p, ctypes.POINTER(ctypes.c_int)]),
    "paos_norm2_enqueue_rows_like": (ctypes.c_int, [_c_ctx, _dbl_p, _dbl_p, ctypes.POINTER(ctypes.c_int)]),
    "paos_run_program": (ctypes.c_int, [_c_ctx, ctypes.POINTER(Pass), ctypes.c_int, _dbl_p, ctypes.c_int,
                                        ctypes.POINTER(ProgramOpts)]),
}

_lib = None


class PaosHipError(RuntimeError):
    pass


def load():
    """Load libpaoship.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PaosHipError(
            f"{LIB_PATH} not found: build it with `make` or "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    if os.environ.get("PAOS_LIB"):
        import sys

        print(f"paos_amd: using the library variant {LIB_PATH} (PAOS_LIB)", file=sys.stderr)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _dptr(arr):
    return arr.ctypes.data_as(_dbl_p)


def as_blocks(blocks, batch, stride):
    arr = np.ascontiguousarray(blocks, dtype=np.float64)
    if arr.shape != (batch, stride):
        raise ValueError(f"expected parameter blocks of shape {(batch, stride)}, got {arr.shape}")
    return arr


# ---- page-locked result arrays ---------------------------------------------------------------
# Measured on the MI355X host (4096^2 PSF, 128 MiB): pageable destination 6 ms (21 GB/s, with a
# 36 ms outlier now and then when the pages are fresh), page-locked and recycled 2.5 ms, page-locked
# and freshly allocated 27 ms (hipHostMalloc pins at ~5 GB/s).  Fresh pinning only pays below a few
# tens of MiB, where the pageable path's fixed costs and outliers dominate (Ariel AIRS 1024^2 with
# 12 saved surfaces: 28 -> 13 ms per run()).  So arrays between PINNED_MIN_BYTES and
# PINNED_MAX_BYTES are backed by hipHostMalloc memory -- one DMA, no CPU copy -- and larger ones
# only when a recycled buffer of their size is waiting in the pool.  The memory goes back to the
# pool when the array (and every view of it) has been garbage-collected; at most PINNED_LIVE_BYTES
# are handed out at a time, beyond that -- or if the allocation fails -- the caller falls back to
# pageable memory.
PINNED_MIN_BYTES = 4 << 20
PINNED_MAX_BYTES = 32 << 20
PINNED_LIVE_BYTES = int(os.environ.get("PAOS_PINNED_BYTES", 8 << 30))
PINNED_POOL_BYTES = 2 << 30
_pin_lock = threading.Lock()
_pin_free = {}  # nbytes -> [address]
_pin_live = 0
_pin_pooled = 0


def _pin_release(address, nbytes):
    global _pin_live, _pin_pooled
    with _pin_lock:
        _pin_live -= nbytes
        if _pin_pooled + nbytes <= PINNED_POOL_BYTES:
            _pin_free.setdefault(nbytes, []).append(address)
            _pin_pooled += nbytes
            return
    try:
        load().paos_host_free(ctypes.c_void_p(address))
    except Exception:  # interpreter shutdown
        pass


def pinned_empty(shape, dtype):
    """Uninitialised array in page-locked host memory, or None when the request is small, the
    budget is used up or the allocation fails (the caller then uses ordinary memory)."""
    global _pin_live, _pin_pooled
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes <= PINNED_MIN_BYTES:
        return None
    address = None
    with _pin_lock:
        if _pin_live + nbytes > PINNED_LIVE_BYTES:
            return None
        stack = _pin_free.get(nbytes)
        if stack:
            address = stack.pop()
            _pin_pooled -= nbytes
        elif nbytes > PINNED_MAX_BYTES:
            return None  # pinning this much afresh costs more than the pageable copy
        _pin_live += nbytes
    if address is None:
        ptr = ctypes.c_void_p()
        if load().paos_host_alloc(nbytes, ctypes.byref(ptr)) != 0 or not ptr.value:
            with _pin_lock:
                _pin_live -= nbytes
            return None
        address = ptr.value
    buf = (ctypes.c_char * nbytes).from_address(address)
    weakref.finalize(buf, _pin_release, address, nbytes)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def release_pinned_pool():
    """Free the cached (idle) page-locked buffers."""
    global _pin_pooled
    with _pin_lock:
        items = [(a, n) for n, lst in _pin_free.items() for a in lst]
        _pin_free.clear()
        _pin_pooled = 0
    for address, _ in items:
        load().paos_host_free(ctypes.c_void_p(address))


_HIP_TOUCHED = [False]


def hip_initialised():
    """Has this process asked the library for a device context yet (the first HIP call of the process)?"""
    return _HIP_TOUCHED[0]


class DeviceFields:
    """``batch`` n x n complex fields in HBM plus the stream that owns them."""

    def __init__(self, n, batch=1, precision="fp64", device=0):
        self._lib = load()
        _HIP_TOUCHED[0] = True
        self._ctx = _c_ctx()
        self.n, self.batch = int(n), int(batch)
        if precision not in ("fp64", "fp32"):
            raise ValueError("precision must be 'fp64' or 'fp32'")
        self.precision = precision
        rc = self._lib.paos_ctx_create(int(device), self.n, self.batch,
                                       PAOS_F64 if precision == "fp64" else PAOS_F32,
                                       ctypes.byref(self._ctx))
        if rc != 0:
            msg = self._lib.paos_last_error(None).decode()
            self._ctx = _c_ctx()
            raise PaosHipError(f"paos_ctx_create failed ({rc}): {msg}")

    # -- plumbing ---------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise PaosHipError(f"{what} failed ({rc}): {self._lib.paos_last_error(self._ctx).decode()}")

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.paos_ctx_destroy(self._ctx)
            self._ctx = _c_ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self._lib.paos_sync(self._ctx), "paos_sync")

    @property
    def stream(self):
        return self._lib.paos_stream(self._ctx)

    def build_info(self):
        return self._lib.paos_build_info().decode()

    def profile_begin(self, kernel_kind, max_launches=4096):
        self._check(self._lib.paos_profile_begin(self._ctx, int(kernel_kind), int(max_launches)),
                    "paos_profile_begin")

    def profile_end(self):
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        self._check(self._lib.paos_profile_end(self._ctx, ctypes.byref(n), ctypes.byref(ms)),
                    "paos_profile_end")
        return n.value, ms.value

    def profile_end_split(self):
        """(launches, total ms, pruned launches, their ms): pruned = skipped dead tiles / loads."""
        n, ms, pn, pms = ctypes.c_int(0), ctypes.c_double(0.0), ctypes.c_int(0), ctypes.c_double(0.0)
        self._check(self._lib.paos_profile_end_split(self._ctx, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(pn),
                                                     ctypes.byref(pms)), "paos_profile_end_split")
        return n.value, ms.value, pn.value, pms.value

    def profile_planned_bytes(self, capacity=1 << 16):
        """Bytes the pruning plan had every launch timed so far load + store (call before profile_end_launches)."""
        out = np.empty(capacity, dtype=np.float64)
        n = ctypes.c_int(0)
        self._check(self._lib.paos_profile_planned_bytes(self._ctx, int(capacity), _dptr(out), ctypes.byref(n)),
                    "paos_profile_planned_bytes")
        return out[:n.value].copy()

    def profile_line_transforms(self, capacity=1 << 16):
        """1-D line transforms every launch timed so far ran (call before profile_end_launches)."""
        out = np.empty(capacity, dtype=np.float64)
        n = ctypes.c_int(0)
        self._check(self._lib.paos_profile_line_transforms(self._ctx, int(capacity), _dptr(out), ctypes.byref(n)),
                    "paos_profile_line_transforms")
        return out[:n.value].copy()

    def profile_end_launches(self, capacity=1 << 16):
        """(ms[i], tag[i]) of every timed launch, in launch order; tag bits: 1 skipped tiles, 2 skipped loads,
        4 skipped stores, 8 stored the PSF instead of the field."""
        ms = np.empty(capacity, dtype=np.float64)
        tags = np.empty(capacity, dtype=np.int32)
        n = ctypes.c_int(0)
        self._check(self._lib.paos_profile_end_launches(self._ctx, int(capacity), _dptr(ms),
                                                        tags.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(n)),
                    "paos_profile_end_launches")
        return ms[:n.value].copy(), tags[:n.value].copy()

    # -- field I/O ----------------------------------------------------------------
    def fill(self, value=1.0 + 0.0j):
        value = complex(value)
        self._check(self._lib.paos_fill(self._ctx, value.real, value.imag), "paos_fill")

    def upload(self, item, field):
        arr = np.ascontiguousarray(field, dtype=np.complex128)
        if arr.shape != (self.n, self.n):
            raise ValueError(f"field must have shape {(self.n, self.n)}")
        self._check(self._lib.paos_import(self._ctx, int(item), arr.ctypes.data_as(ctypes.c_void_p)),
                    "paos_import")

    def download(self, item=0, what=WHAT_FIELD):
        dtype = np.complex128 if what == WHAT_FIELD else np.float64
        out = pinned_empty((self.n, self.n), dtype)
        if out is not None:  # large array: one DMA into page-locked memory owned by the array
            self._check(self._lib.paos_export_pinned(self._ctx, int(item), int(what),
                                                     out.ctypes.data_as(ctypes.c_void_p)), "paos_export_pinned")
            return out
        out = np.empty((self.n, self.n), dtype=dtype)
        self._check(self._lib.paos_export(self._ctx, int(item), int(what),
                                          out.ctypes.data_as(ctypes.c_void_p)), "paos_export")
        return out

    def psf_keep(self):
        """|u|^2 of every item into the context's device-resident PSF buffer (no host copy)."""
        self._check(self._lib.paos_psf_keep(self._ctx), "paos_psf_keep")

    def psf_keep_power(self):
        """``psf_keep`` and ``norm2_enqueue`` in one sweep over the field; returns the power ticket."""
        t = ctypes.c_int(-1)
        self._check(self._lib.paos_psf_keep_power(self._ctx, ctypes.byref(t)), "paos_psf_keep_power")
        return t.value

    def psf_fetch(self, item=0):
        out = np.empty((self.n, self.n), dtype=np.float64)
        self._check(self._lib.paos_psf_fetch(self._ctx, int(item), _dptr(out)), "paos_psf_fetch")
        return out

    # -- operators ------------------------------------------------------------------
    def aperture(self, shape, blocks):
        b = as_blocks(blocks, self.batch, APERTURE_STRIDE)
        self._check(self._lib.paos_aperture(self._ctx, int(shape), _dptr(b)), "paos_aperture")

    def aperture_mask(self, shape, block):
        b = as_blocks([block], 1, APERTURE_STRIDE)
        out = np.empty((self.n, self.n), dtype=np.float64)
        self._check(self._lib.paos_aperture_render(self._ctx, int(shape), _dptr(b), _dptr(out)),
                    "paos_aperture_render")
        return out

    def make_stop(self, enable=None, power_known=False, defer=False):
        """``power_known``: the pass program that stored the field has just reduced its power (run_passes with
        final_intensity = 2, nothing since): only the scaling sweep runs -- or, with ``defer``, not even that: the
        next pass program's first pass takes the factor along (the library applies it earlier if anything else touches
        the field first)."""
        fn, name = ((self._lib.paos_stop_defer_last_power, "paos_stop_defer_last_power") if (power_known and defer)
                    else (self._lib.paos_stop_scale_last_power, "paos_stop_scale_last_power") if power_known
                    else (self._lib.paos_make_stop, "paos_make_stop"))
        if enable is None:
            self._check(fn(self._ctx, None), name)
        else:
            e = np.ascontiguousarray(enable, dtype=np.float64).reshape(self.batch)
            self._check(fn(self._ctx, _dptr(e)), name)

    def norm2(self):
        out = np.empty(self.batch, dtype=np.float64)
        self._check(self._lib.paos_norm2(self._ctx, _dptr(out)), "paos_norm2")
        return out

    def _rows(self, rows):
        lr = np.ascontiguousarray(rows, dtype=np.float64)
        if lr.shape != (self.batch, 2):
            raise ValueError("row ranges must be [batch][2]")
        return lr

    def norm2_enqueue(self, live_rows=None, same_as=None, live_cols=None):
        """``live_rows`` ([batch][2], optional): rows outside [lo, hi) are zero (or stand for zeros) and are not read.
        ``same_as`` ([batch] indices, with ``live_rows``): item i's field is a copy of item same_as[i]'s -- summed once.
        ``live_cols`` ([batch][2], with ``live_rows``): ... and neither are the columns outside [lo, hi) of those rows."""
        t = ctypes.c_int(-1)
        if live_rows is not None and live_cols is not None:
            like = None
            if same_as is not None:
                like = np.ascontiguousarray(same_as, dtype=np.float64)
                if like.shape != (self.batch,):
                    raise ValueError("same_as must be [batch]")
            self._check(self._lib.paos_norm2_enqueue_box(self._ctx, _dptr(self._rows(live_rows)), _dptr(self._rows(live_cols)),
                                                         _dptr(like) if like is not None else None, ctypes.byref(t)),
                        "paos_norm2_enqueue_box")
            return t.value
        if live_rows is not None and same_as is not None:
            like = np.ascontiguousarray(same_as, dtype=np.float64)
            if like.shape != (self.batch,):
                raise ValueError("same_as must be [batch]")
            self._check(self._lib.paos_norm2_enqueue_rows_like(self._ctx, _dptr(self._rows(live_rows)), _dptr(like),
                                                               ctypes.byref(t)), "paos_norm2_enqueue_rows_like")
            return t.value
        if live_rows is not None:
            self._check(self._lib.paos_norm2_enqueue_rows(self._ctx, _dptr(self._rows(live_rows)), ctypes.byref(t)),
                        "paos_norm2_enqueue_rows")
            return t.value
        self._check(self._lib.paos_norm2_enqueue(self._ctx, ctypes.byref(t)), "paos_norm2_enqueue")
        return t.value

    def zero_outside_rows(self, live_rows, live_cols=None):
        """Rows (and, with ``live_cols``, columns of the rows in between) that merely stand for zeros
        (``start(..., write_rows=..., write_cols=...)``) become zeros."""
        if live_cols is not None:
            self._check(self._lib.paos_zero_outside_box(self._ctx, _dptr(self._rows(live_rows)), _dptr(self._rows(live_cols))),
                        "paos_zero_outside_box")
            return
        self._check(self._lib.paos_zero_outside_rows(self._ctx, _dptr(self._rows(live_rows))), "paos_zero_outside_rows")

    def norm2_fetch(self, ticket):
        if hasattr(ticket, "fetch"):  # a run.PowerTicket: fetched once, however many records share it
            return ticket.fetch()
        out = np.empty(self.batch, dtype=np.float64)
        self._check(self._lib.paos_norm2_fetch(self._ctx, int(ticket), _dptr(out)), "paos_norm2_fetch")
        return out

    def norm2_release(self, ticket):
        if hasattr(ticket, "release"):
            return ticket.release()
        self._check(self._lib.paos_norm2_release(self._ctx, int(ticket)), "paos_norm2_release")

    def psf_metrics(self, radii_px=(), centre=None):
        """Per item: dict(power, centroid (col,row), peak, encircled power per radius) of |u|^2,
        computed on the GPU.  ``centre`` defaults to the grid centre (n/2, n/2)."""
        r = np.ascontiguousarray(radii_px, dtype=np.float64).reshape(-1)
        cxp, cyp = (self.n / 2, self.n / 2) if centre is None else centre
        out = np.empty((self.batch, 4 + r.size), dtype=np.float64)
        self._check(self._lib.paos_psf_metrics(self._ctx, int(r.size), _dptr(r) if r.size else None,
                                               float(cxp), float(cyp), _dptr(out)), "paos_psf_metrics")
        res = []
        for row in out:
            p = row[0]
            res.append({"power": p, "centroid": (row[1] / p, row[2] / p) if p > 0 else (np.nan, np.nan),
                        "peak": row[3], "encircled": row[4:].copy()})
        return res

    def phase(self, blocks, mul2pi):
        b = as_blocks(blocks, self.batch, PHASE_STRIDE)
        self._check(self._lib.paos_phase(self._ctx, _dptr(b), int(bool(mul2pi))), "paos_phase")

    def phase_map(self, item, wfe, wl):
        """u[item] *= exp(2 pi i wfe / wl) for a host map (metres) -- wfo.py:869-871, 945-949."""
        w = np.ascontiguousarray(wfe, dtype=np.float64)
        if w.shape != (self.n, self.n):
            raise ValueError(f"phase map must have shape {(self.n, self.n)}")
        self._check(self._lib.paos_phase_map(self._ctx, int(item), _dptr(w), float(wl)), "paos_phase_map")

    def phase_map_items(self, wfe, items, wls, key=0):
        """u[i] *= exp(2 pi i wfe / wl_i) for the listed items, which share the host map ``wfe`` (metres): one upload.
        ``key`` != 0 names the map's content: the same key again re-uses the copy already on the device."""
        w = None
        if wfe is not None:  # (None: the map the device keeps under ``key`` -- psd_screen)
            w = np.ascontiguousarray(wfe, dtype=np.float64)
            if w.shape != (self.n, self.n):
                raise ValueError(f"phase map must have shape {(self.n, self.n)}")
        it = np.ascontiguousarray(items, dtype=np.float64).reshape(-1)
        wl = np.ascontiguousarray(wls, dtype=np.float64).reshape(-1)
        if it.size != wl.size or it.size < 1:
            raise ValueError("one wavelength per listed item is required")
        self._check(self._lib.paos_phase_map_items(self._ctx, _dptr(w) if w is not None else None, int(key) & 0xFFFFFFFFFFFFFFFF,
                                                   int(it.size), _dptr(it), _dptr(wl)), "paos_phase_map_items")

    def psd_screen(self, noise, rough, params, key, want_map=False):
        """The random screen of ``WFO.psd`` from the host's two white-noise draws (``paos_psd_screen``: fft2, power-law filter,
        ifft2, roughness on the library's own passes); it stays on the device under ``key`` for ``phase_map_items(None, ...,
        key=key)``.  ``params``: the twelve numbers of ``phase_maps.psd_device_params``.  Returns the map when asked to."""
        nz = np.ascontiguousarray(noise, dtype=np.float64)
        if nz.shape != (self.n, self.n):
            raise ValueError(f"the noise must have shape {(self.n, self.n)}")
        rg = None
        if rough is not None:
            rg = np.ascontiguousarray(rough, dtype=np.float64)
            if rg.shape != (self.n, self.n):
                raise ValueError(f"the roughness draw must have shape {(self.n, self.n)}")
        pr = np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        if pr.size != 12:
            raise ValueError("twelve PSD parameters are required")
        out = np.empty((self.n, self.n), dtype=np.float64) if want_map else None
        self._check(self._lib.paos_psd_screen(self._ctx, _dptr(nz), _dptr(rg) if rg is not None else None, _dptr(pr),
                                              int(key) & 0xFFFFFFFFFFFFFFFF, _dptr(out) if out is not None else None), "paos_psd_screen")
        return out

    def ptp(self, blocks):
        b = as_blocks(blocks, self.batch, PHASE_STRIDE)
        self._check(self._lib.paos_ptp(self._ctx, _dptr(b)), "paos_ptp")

    def stw(self, blocks, inverse):
        b = as_blocks(blocks, self.batch, PHASE_STRIDE)
        self._check(self._lib.paos_stw(self._ctx, _dptr(b), int(bool(inverse))), "paos_stw")

    def wts(self, blocks, inverse):
        b = as_blocks(blocks, self.batch, PHASE_STRIDE)
        self._check(self._lib.paos_wts(self._ctx, _dptr(b), int(bool(inverse))), "paos_wts")

    def copy_yardstick(self, reps=10):
        """(ms per launch, bytes per launch) of an in-place copy of the whole batch (measurement aid)."""
        ms, nbytes = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._check(self._lib.paos_copy_yardstick(self._ctx, int(reps), ctypes.byref(ms), ctypes.byref(nbytes)),
                    "paos_copy_yardstick")
        return ms.value, nbytes.value

    def record_set_stats(self):
        """(found, rendered): aperture line records found in the context's kept sets / rendered anew, since creation."""
        found, rendered = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        self._check(self._lib.paos_record_set_stats(self._ctx, ctypes.byref(found), ctypes.byref(rendered)),
                    "paos_record_set_stats")
        return int(found.value), int(rendered.value)

    def set_pruning(self, on):
        """Dead-line pruning of the pass programs on (default) / off -- results are identical."""
        self._check(self._lib.paos_ctx_set_pruning(self._ctx, 1 if on else 0), "paos_ctx_set_pruning")

    def run_passes(self, passes, blocks, live_rows=None, rows_stale=False, final_intensity=False, live_cols=None):
        """passes: list of dicts {axis, fft1, fft2, pre, mid, post} with operator tuples
        (kind, flags, block); blocks: array [n_blocks][batch][PHASE_STRIDE].  ``live_rows``
        ([batch][2], optional): rows outside [lo, hi) of item i are exactly zero in memory -- or, with
        ``rows_stale``, hold old data that stands for zeros.  ``final_intensity``: True / 1 -- the last pass writes |u|^2 to
        the PSF buffer instead of the field (which is undefined afterwards); 2 -- the field is stored as usual and its
        sum |u|^2 is reduced by the last pass on the way (the power of a saved surface without reading the field
        back); either way the power ticket is returned.  ``live_cols`` (with ``rows_stale``): the columns outside stand
        for zeros as well (``start(..., write_cols=...)``)."""
        b = np.ascontiguousarray(blocks, dtype=np.float64)
        if b.ndim != 3 or b.shape[1:] != (self.batch, PHASE_STRIDE):
            raise ValueError("blocks must be [n_blocks][batch][5]")
        arr = (Pass * len(passes))()
        for dst, src in zip(arr, passes):
            dst.axis, dst.fft1, dst.fft2 = src["axis"], src.get("fft1", -1), src.get("fft2", -1)
            for name in ("pre", "mid", "post"):
                ops = src.get(name, ())
                if len(ops) > MAX_PW:
                    raise ValueError("too many operators in one pass slot")
                setattr(dst, "n_" + name, len(ops))
                lst = getattr(dst, name)
                for i, (kind, flags, block) in enumerate(ops):
                    lst[i].kind, lst[i].flags, lst[i].block = kind, flags, block
        if rows_stale or final_intensity:
            lr = self._rows(live_rows) if live_rows is not None else None
            lc = self._rows(live_cols) if (live_cols is not None and rows_stale and lr is not None) else None
            ticket = ctypes.c_int(-1)
            opts = ProgramOpts(_dptr(lr) if lr is not None else None, 1 if (rows_stale and lr is not None) else 0,
                               int(final_intensity), ctypes.pointer(ticket), _dptr(lc) if lc is not None else None)
            self._check(self._lib.paos_run_program(self._ctx, arr, len(passes), _dptr(b), b.shape[0], ctypes.byref(opts)),
                        "paos_run_program")
            return ticket.value if final_intensity else None
        if live_rows is not None:
            lr = self._rows(live_rows)
            self._check(self._lib.paos_run_passes_live(self._ctx, arr, len(passes), _dptr(b), b.shape[0], _dptr(lr)),
                        "paos_run_passes_live")
            return None
        self._check(self._lib.paos_run_passes(self._ctx, arr, len(passes), _dptr(b), b.shape[0]),
                    "paos_run_passes")
        return None

    def zernike(self, nmax, kdim, table, blocks, want_wfe=False, pupil=False, same_as=None):
        """``pupil=True``: only pixels inside the pupil set by pupil_aperture / pupil_upload.  ``same_as`` ([batch] item
        indices, without ``pupil``): items known to hold copies of one field (paos_zernike_like)."""
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(blocks, dtype=np.float64)
        if b.ndim != 2 or b.shape[0] != self.batch:
            raise ValueError("zernike blocks must be [batch][stride]")
        out = np.empty((self.n, self.n), dtype=np.float64) if want_wfe else None
        if same_as is not None and not pupil:
            like = np.ascontiguousarray(same_as, dtype=np.float64).reshape(-1)
            if like.size != self.batch:
                raise ValueError("same_as must name an item per item")
            self._check(self._lib.paos_zernike_like(self._ctx, int(nmax), int(kdim), _dptr(t), _dptr(b), int(b.shape[1]),
                                                    _dptr(like), _dptr(out) if want_wfe else None), "paos_zernike_like")
            return out
        fn = self._lib.paos_zernike_pupil if pupil else self._lib.paos_zernike
        self._check(fn(self._ctx, int(nmax), int(kdim), _dptr(t), _dptr(b), int(b.shape[1]),
                       _dptr(out) if want_wfe else None),
                    "paos_zernike_pupil" if pupil else "paos_zernike")
        return out

    def start(self, value, shape, blocks, stop=None, write_rows=None, write_cols=None):
        """fill(value) + aperture(shape, blocks) + make_stop(stop) in one write of the field.  ``write_rows``
        ([batch][2]): only these rows are written, the others stand for zeros (see paos_start_rows); ``write_cols``
        (with ``write_rows``): ... and only these columns of them (paos_start_box)."""
        b = np.ascontiguousarray(blocks, dtype=np.float64)
        if b.shape != (self.batch, APERTURE_STRIDE):
            raise ValueError("aperture blocks must be [batch][8]")
        st = None if stop is None else np.ascontiguousarray(stop, dtype=np.float64)
        if st is not None and st.shape != (self.batch,):
            raise ValueError("stop flags must be [batch]")
        v = complex(value)
        if write_rows is not None and write_cols is not None:
            self._check(self._lib.paos_start_box(self._ctx, v.real, v.imag, int(shape), _dptr(b),
                                                 _dptr(st) if st is not None else None, _dptr(self._rows(write_rows)),
                                                 _dptr(self._rows(write_cols))), "paos_start_box")
            return
        if write_rows is not None:
            self._check(self._lib.paos_start_rows(self._ctx, v.real, v.imag, int(shape), _dptr(b),
                                                  _dptr(st) if st is not None else None, _dptr(self._rows(write_rows))),
                        "paos_start_rows")
            return
        self._check(self._lib.paos_start(self._ctx, v.real, v.imag, int(shape), _dptr(b),
                                         _dptr(st) if st is not None else None), "paos_start")

    def pupil_aperture(self, shape, blocks):
        """Pupil = pixels where the exact mask of the aperture object is non-zero (run.py:136-141)."""
        b = np.ascontiguousarray(blocks, dtype=np.float64)
        if b.shape != (self.batch, APERTURE_STRIDE):
            raise ValueError("aperture blocks must be [batch][8]")
        self._check(self._lib.paos_pupil_aperture(self._ctx, int(shape), _dptr(b)), "paos_pupil_aperture")

    def pupil_upload(self, item, weights):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.shape != (self.n, self.n):
            raise ValueError("pupil weights must be [n][n]")
        self._check(self._lib.paos_pupil_upload(self._ctx, int(item), _dptr(w)), "paos_pupil_upload")

    def zernike_gram(self, nmax, kdim, table, blocks, poly, pupil=True):
        """(sums[batch][K(K+1)/2], count[batch]): sums of Z_i Z_j (i <= j, row by row) over the
        unmasked pixels and their number -- the raw material of Zernike.cov (zernike.py:293-318)."""
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        b = np.ascontiguousarray(blocks, dtype=np.float64)
        q = np.ascontiguousarray(poly, dtype=np.float64)
        if b.ndim != 2 or b.shape[0] != self.batch:
            raise ValueError("zernike blocks must be [batch][stride]")
        if q.ndim != 2 or q.shape[1] != 4:
            raise ValueError("poly must be [K][4]")
        k = q.shape[0]
        out = np.empty((self.batch, k * (k + 1) // 2 + 1), dtype=np.float64)
        self._check(self._lib.paos_zernike_gram(self._ctx, int(nmax), int(kdim), _dptr(t), _dptr(b),
                                                int(b.shape[1]), int(k), _dptr(q), 1 if pupil else 0, _dptr(out)),
                    "paos_zernike_gram")
        return out[:, :-1], out[:, -1]
