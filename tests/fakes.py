"""CPU stand-ins used by the ``not gpu`` tests.

``ModelDevice`` mimics ``paos_amd._lib.DeviceFields`` with NumPy arrays and executes
pass programs by the documented semantics of include/paos_hip.h (operators, per-item
enable flags, transform control blocks).  It is a *model of the device contract* for
testing the host logic (planner, pass compiler, batching) without a GPU -- test
infrastructure only, never imported by the product.  Aperture masks and Zernike maps
come from the oracle.
"""
import numpy as np

from oracle import aperture_np
from paos_amd import _lib


class ModelDevice:
    def __init__(self, n, batch=1, precision="fp64", device=0):
        self.n, self.batch, self.precision = int(n), int(batch), precision
        self.u = np.zeros((self.batch, self.n, self.n), dtype=np.complex128)
        self.log = []  # (name, detail) per launch
        self.pass_count = 0

    def close(self):
        pass

    def sync(self):
        pass

    def build_info(self):
        return "numpy model"

    def fill(self, value=1.0 + 0.0j):
        self.u[:] = value

    def upload(self, item, field):
        self.u[item] = field

    def download(self, item=0, what=_lib.WHAT_FIELD):
        u = self.u[item]
        if what == _lib.WHAT_FIELD:
            return u.copy()
        if what == _lib.WHAT_AMPLITUDE:
            return np.abs(u)
        if what == _lib.WHAT_PHASE:
            return np.angle(u)
        return u.real**2 + u.imag**2

    def norm2(self):
        return np.array([np.sum(np.abs(u) ** 2) for u in self.u])

    def psf_metrics(self, radii_px=(), centre=None):
        n = self.n
        cxp, cyp = (n / 2, n / 2) if centre is None else centre
        yy, xx = np.mgrid[0:n, 0:n]
        d2 = (xx - cxp) ** 2 + (yy - cyp) ** 2
        out = []
        for u in self.u:
            I = u.real**2 + u.imag**2
            p = I.sum()
            out.append({"power": p, "centroid": ((I * xx).sum() / p, (I * yy).sum() / p), "peak": I.max(),
                        "encircled": np.array([I[d2 <= r * r].sum() for r in radii_px])})
        return out

    def norm2_enqueue(self, live_rows=None, same_as=None, live_cols=None):
        if live_cols is not None:  # the columns outside are not read either (they may hold NaN in this model)
            assert live_rows is not None
            return self._new_ticket(np.array([np.sum(np.abs(u[int(lo):int(hi), (int(cl) // 4) * 4:-(-int(ch) // 4) * 4]) ** 2)
                                              for u, (lo, hi), (cl, ch) in zip(self.u, live_rows, live_cols)]))
        if same_as is not None:  # the caller claims copies: hold it to that
            for i, j in enumerate(same_as):
                lo, hi = (int(x) for x in live_rows[i])
                assert int(same_as[int(j)]) == int(j) and np.array_equal(self.u[i][lo:hi], self.u[int(j)][lo:hi]), (i, j)
        if live_rows is None:
            return self._new_ticket(self.norm2())
        # rows outside [lo, hi) are not read (they may hold stale data: NaN in this model)
        return self._new_ticket(np.array([np.sum(np.abs(u[int(lo):int(hi)]) ** 2) for u, (lo, hi) in zip(self.u, live_rows)]))

    # The library's ticket ring (paos_hip.hip: next_norm_slot): NORM_SLOTS slots, the next FREE one is handed out, a slot
    # is free again as soon as it was fetched or released -- so a slot NUMBER does not identify a reduction, and
    # fetching one twice is an error (or, worse, somebody else's value).  The model is as strict.
    def _new_ticket(self, values):
        ring = self.__dict__.setdefault("_ring", {})
        start = self.__dict__.get("_ring_next", 0)
        for k in range(_lib.NORM_SLOTS):
            slot = (start + k) % _lib.NORM_SLOTS
            if slot not in ring:
                ring[slot] = np.array(values, dtype=np.float64)
                self._ring_next = (slot + 1) % _lib.NORM_SLOTS
                return slot
        raise RuntimeError(f"{_lib.NORM_SLOTS} tickets outstanding")

    def norm2_release(self, ticket):
        if hasattr(ticket, "release"):
            return ticket.release()
        if int(ticket) not in self.__dict__.get("_ring", {}):
            raise RuntimeError("ticket is not outstanding")
        del self._ring[int(ticket)]

    def zero_outside_rows(self, live_rows, live_cols=None):
        self.log.append(("zero_outside_rows", None))
        for i, (u, (lo, hi)) in enumerate(zip(self.u, live_rows)):
            u[:int(lo)] = 0.0
            u[int(hi):] = 0.0
            if live_cols is not None:
                cl, ch = (int(live_cols[i][0]) // 4) * 4, -(-int(live_cols[i][1]) // 4) * 4
                u[:, :cl] = 0.0
                u[:, ch:] = 0.0

    def norm2_fetch(self, ticket):
        if hasattr(ticket, "fetch"):
            return ticket.fetch()
        if int(ticket) not in self.__dict__.get("_ring", {}):
            raise RuntimeError("ticket is not outstanding")
        return self._ring.pop(int(ticket))

    def make_stop(self, enable=None, power_known=False, defer=False):
        # (defer: the real device leaves the scaling to the next pass; the model has no such thing as a cost and scales now)
        self.log.append(("make_stop", "power_known" if power_known else None))
        if power_known:  # the precondition the library cannot check: a program with final_intensity = 2 came right before
            assert self.log[-2][0] == "power_on_store", self.log[-3:]
        for i in range(self.batch):
            if enable is None or enable[i]:
                self.u[i] /= np.sqrt(np.sum(np.abs(self.u[i]) ** 2))

    def psf_keep(self):
        self.log.append(("psf_keep", None))
        self.psf = self.u.real**2 + self.u.imag**2

    def psf_keep_power(self):
        self.psf_keep()
        return self.norm2_enqueue()

    def psf_fetch(self, item=0):
        return self.psf[item].copy()

    def start(self, value, shape, blocks, stop=None, write_rows=None, write_cols=None):
        """paos_start: fill -> aperture -> make_stop on the flagged items.  ``write_rows``: the rows outside
        are NOT written and merely stand for zeros -- the model poisons them so that any read shows."""
        self.log.append(("start", shape))
        self.u[:] = value
        n0 = len(self.log)
        self.aperture(shape, blocks)
        if stop is not None and any(stop):
            self.make_stop(stop)
        del self.log[n0:]
        if write_rows is not None:
            for u, (lo, hi) in zip(self.u, write_rows):
                lo, hi = (int(lo) // 4) * 4, min(self.n, -(-int(hi) // 4) * 4)
                assert not u[:lo].any() and not u[hi:].any(), "write_rows must contain every non-zero row"
                u[:lo] = np.nan
                u[hi:] = np.nan
            if write_cols is not None:
                for u, (lo, hi) in zip(self.u, write_cols):
                    lo, hi = (int(lo) // 4) * 4, min(self.n, -(-int(hi) // 4) * 4)
                    live = np.nan_to_num(u, nan=0.0)
                    assert not live[:, :lo].any() and not live[:, hi:].any(), "write_cols must contain every non-zero column"
                    u[:, :lo] = np.nan
                    u[:, hi:] = np.nan

    def aperture(self, shape, blocks):
        self.log.append(("aperture", shape))
        for i, b in enumerate(blocks):
            if not b[0]:
                continue
            _, xc, yc, a, bb, theta, obsc, _sub = b
            if shape == _lib.SHAPE_ELLIPSE:
                mask = aperture_np.ellipse_mask((self.n, self.n), xc, yc, a, bb, theta)
            else:
                mask = aperture_np.rectangle_mask((self.n, self.n), xc, yc, a, bb, theta)
            self.u[i] *= (1 - mask) if obsc else mask

    def _zernike_terms(self, nmax, kdim, table, b):
        """Yield (am, k, rho_pow * P_k, cos(am phi), sin(am phi)) maps and rho, like the kernels."""
        n = self.n
        _, dx, dy, radius, origin_y, co, so, _ = b[:8]
        x = (np.arange(n) - n // 2) * dx
        y = (np.arange(n) - n // 2) * dy
        xx, yy = np.meshgrid(x, y)
        rr = np.sqrt(xx**2 + yy**2)
        rho = rr / radius
        with np.errstate(invalid="ignore", divide="ignore"):
            c1 = np.where(rr > 0, (yy if origin_y else xx) / rr, 1.0)
            s1 = np.where(rr > 0, (xx if origin_y else yy) / rr, 0.0)
        cr, sr = c1 * co - s1 * so, s1 * co + c1 * so
        xj = 1.0 - 2.0 * rho * rho
        terms = []
        rho_pow, cm, sm = np.ones_like(rho), np.ones_like(rho), np.zeros_like(rho)
        for am in range(nmax + 1):
            pkm1, pk = np.zeros_like(rho), np.ones_like(rho)
            for k in range((nmax - am) // 2 + 1):
                if k > 0:
                    a_, b_, c_ = table[am, k]
                    pkm1, pk = pk, (a_ * xj + b_) * pk - c_ * pkm1
                terms.append((am, k, rho_pow * pk, cm, sm))
            rho_pow = rho_pow * rho
            cm, sm = cm * cr - sm * sr, sm * cr + cm * sr
        return rho, terms

    def zernike(self, nmax, kdim, table, blocks, want_wfe=False, pupil=False, same_as=None):
        """Mirror of csrc/pointwise.h zernike_kernel (same recurrences, NumPy).  ``same_as``: the library takes the caller's
        word that those items hold copies of one field (paos_zernike_like); the model checks it on every pixel the kernel
        reads (the unit disk)."""
        self.log.append(("zernike", nmax) if same_as is None else ("zernike_like", nmax))
        table = np.asarray(table).reshape(nmax + 1, kdim, 3)
        wfe0 = None
        if same_as is not None:
            before = self.u.copy()
            for i, b in enumerate(np.asarray(blocks)):
                j = int(same_as[i])
                if b[0] and j != i:
                    inside = self._zernike_terms(nmax, kdim, table, b)[0] <= 1.0
                    if not np.array_equal(before[i][inside], before[j][inside]):
                        raise AssertionError(f"zernike(same_as=...): item {i} does not hold a copy of item {j}'s field")
        for i, b in enumerate(np.asarray(blocks)):
            if not b[0]:
                continue
            inv_wl = b[7]
            cc = b[8:8 + (nmax + 1) * kdim].reshape(nmax + 1, kdim)
            ss = b[8 + (nmax + 1) * kdim:8 + 2 * (nmax + 1) * kdim].reshape(nmax + 1, kdim)
            rho, terms = self._zernike_terms(nmax, kdim, table, b)
            wfe = np.zeros_like(rho)
            for am, k, base, cm, sm in terms:
                wfe += base * (cc[am, k] * cm + ss[am, k] * sm)
            masked = rho > 1.0
            if pupil:
                masked = masked | (self.pupil[i] == 0.0)
            wfe = np.where(masked, 0.0, wfe)
            self.u[i] = self.u[i] * np.exp(1j * ((6.283185307179586 * wfe) * inv_wl))
            if i == 0:
                wfe0 = np.where(masked, np.nan, wfe)
        return wfe0 if want_wfe else None

    def pupil_aperture(self, shape, blocks):
        """Weights of the aperture OBJECT (obscuration flag ignored), csrc: paos_pupil_aperture."""
        self.log.append(("pupil_aperture", shape))
        if not hasattr(self, "pupil"):
            self.pupil = np.ones((self.batch, self.n, self.n))
        for i, b in enumerate(np.asarray(blocks)):
            if not b[0]:
                continue
            _, xc, yc, a, bb, theta, _obsc, _sub = b
            if shape != _lib.SHAPE_ELLIPSE:
                raise NotImplementedError
            m = aperture_np.ellipse_mask((self.n, self.n), xc, yc, a, bb, theta)
            self.pupil[i] = np.zeros((self.n, self.n)) if m is None else m

    def pupil_upload(self, item, weights):
        self.log.append(("pupil_upload", item))
        if not hasattr(self, "pupil"):
            self.pupil = np.ones((self.batch, self.n, self.n))
        self.pupil[item] = np.asarray(weights, dtype=np.float64)

    def zernike_gram(self, nmax, kdim, table, blocks, poly, pupil=True):
        """Mirror of zernike_gram_kernel: sums of Z_i Z_j over the unmasked pixels + their count."""
        self.log.append(("zernike_gram", len(poly)))
        table = np.asarray(table).reshape(nmax + 1, kdim, 3)
        poly = np.asarray(poly)
        k = len(poly)
        sums = np.zeros((self.batch, k * (k + 1) // 2))
        counts = np.zeros(self.batch)
        iu = np.triu_indices(k)
        for i, b in enumerate(np.asarray(blocks)):
            if not b[0]:
                continue
            rho, terms = self._zernike_terms(nmax, kdim, table, b)
            lookup = {(am, kk): (base, cm, sm) for am, kk, base, cm, sm in terms}
            valid = rho <= 1.0
            if pupil:
                valid = valid & (self.pupil[i] != 0.0)
            z = []
            for am, kk, is_sin, fac in poly:
                base, cm, sm = lookup[(int(am), int(kk))]
                z.append(np.where(valid, fac * base * (sm if is_sin else cm), 0.0))
            z = np.array(z).reshape(k, -1)
            sums[i] = (z @ z.T)[iu]
            counts[i] = valid.sum()
        return sums, counts

    # ---- pass programs -----------------------------------------------------------------
    def _apply(self, u, op, p):
        kind, flags, _ = op
        n = self.n
        if kind == _lib.PW_SIGN:
            i = np.arange(n)
            if flags & _lib.PWF_X_ONLY:
                return u * np.where(i[None, :] & 1, -1.0, 1.0)
            if flags & _lib.PWF_Y_ONLY:
                return u * np.where(i[:, None] & 1, -1.0, 1.0)
            return u * np.where((i[:, None] + i[None, :]) & 1, -1.0, 1.0)
        if kind == _lib.PW_SCALE:
            return u * p[3]
        if kind == _lib.PW_MASK:
            raise AssertionError("handled by the caller (needs two blocks)")
        i = np.arange(n)
        g = (i - n // 2) if kind == _lib.PW_QPHASE_CENTRED else np.where(i < n // 2, i, i - n)
        x, y = g * p[1], g * p[2]
        xx, yy = np.meshgrid(x, y)
        q = p[3] * (xx**2 + yy**2)
        if flags & _lib.PWF_MUL2PI:
            q = 6.283185307179586 * q
        return u * (np.cos(q) + 1j * p[4] * np.sin(q))

    def run_passes(self, passes, blocks, live_rows=None, rows_stale=False, final_intensity=False, live_cols=None):
        # live_rows: a traffic hint, results are the same -- unless rows_stale: then the rows outside hold garbage
        # that stands for zeros and the program must behave as if they were zeros
        blocks = np.asarray(blocks, dtype=np.float64)
        assert blocks.ndim == 3 and blocks.shape[1:] == (self.batch, 5)
        n_ = self.n
        if live_rows is not None:
            for k, (u, (lo, hi)) in enumerate(zip(self.u, live_rows)):
                lo, hi = (int(lo) // 4) * 4, min(self.n, -(-int(hi) // 4) * 4)
                if rows_stale:
                    u[:lo] = 0.0
                    u[hi:] = 0.0
                    if live_cols is not None:
                        cl, ch = (int(live_cols[k][0]) // 4) * 4, min(self.n, -(-int(live_cols[k][1]) // 4) * 4)
                        u[:, :cl] = 0.0
                        u[:, ch:] = 0.0
                else:
                    assert not u[:lo].any() and not u[hi:].any(), "live_rows promised zeros"
        else:
            assert not rows_stale
        for ps in passes:
            self.pass_count += 1
            self.log.append(("pass", ps["axis"]))
            for i in range(self.batch):
                u = self.u[i]
                for slot, ctl in (("pre", ps.get("fft1", -1)), ("mid", ps.get("fft2", -1)), ("post", -1)):
                    for op in ps.get(slot, ()):
                        p = blocks[op[2], i]
                        if p[0] == 0.0:
                            continue
                        if op[0] == _lib.PW_MASK:
                            theta, obsc, _sub, shape, _ = blocks[op[2] + 1, i]
                            fn = aperture_np.ellipse_mask if shape == _lib.SHAPE_ELLIPSE else aperture_np.rectangle_mask
                            mask = fn((n_, n_), p[1], p[2], p[3], p[4], theta)
                            u = u * ((1 - mask) if obsc else mask)
                        else:
                            u = self._apply(u, op, p)
                    if ctl is not None and ctl >= 0 and blocks[ctl, i, 0] != 0.0:
                        assert ps["axis"] in (0, 1)
                        ax = 1 if ps["axis"] == 0 else 0  # "along rows" = NumPy axis 1
                        u = np.fft.ifft(u, axis=ax) * self.n if blocks[ctl, i, 1] else np.fft.fft(u, axis=ax)
                self.u[i] = u
        if final_intensity == 2:  # the field is kept; its power comes back as a ticket (the last pass sums it on the way)
            self.log.append(("power_on_store", None))
            return self._new_ticket(self.norm2())
        if final_intensity:  # the PSF and its sum instead of the field, which is given up
            self.log.append(("psf_store", None))
            self.psf = self.u.real**2 + self.u.imag**2
            ticket = self._new_ticket(self.psf.sum(axis=(1, 2)))
            self.u[:] = np.nan
            return ticket
        return None

    # one-operator programs, as csrc/paos_hip.hip builds them
    def _single(self, blocks, inverse, kind):
        from paos_amd.passes import PassCompiler

        comp = PassCompiler(self.batch, self.n)
        rows = [list(b) if b[0] else None for b in blocks]
        if kind == "ptp":
            comp.ptp(rows)
        elif kind == "stw":
            comp.stw(rows, [inverse] * self.batch)
        elif kind == "wts":
            comp.wts(rows, [inverse] * self.batch)
        else:
            comp.lens(rows)
        comp.flush(self)

    def phase_map(self, item, wfe, wl):
        self.log.append(("phase_map", item))
        self.u[int(item)] = self.u[int(item)] * np.exp(2.0 * np.pi * 1j * np.asarray(wfe, dtype=np.float64) / wl)

    def psd_screen(self, noise, rough, params, key, want_map=False):
        """paos_psd_screen in NumPy, from the twelve numbers alone (the arithmetic of csrc/pointwise.h: psd_filter_kernel)."""
        fx, fy, A, B, C, fknee, fmin, fmax, cell, gain, SR, unit = [float(v) for v in params]
        n = self.n
        k = np.where(np.arange(n) < n // 2, np.arange(n), np.arange(n) - n).astype(np.float64)
        gx, gy = np.meshgrid(k * fx, k * fy)
        rho = np.sqrt(gx * gx + gy * gy)
        rho[rho == 0] = 1e-100
        spec = np.fft.fft2(np.asarray(noise, dtype=np.float64))
        with np.errstate(all="ignore"):
            g = np.sqrt(A / (B + (rho / fknee) ** C) / (6.283185307179586 * rho) * cell) * gain
        spec = np.where((rho < fmin) | (rho > fmax), 0.0, spec * g)
        v = np.fft.ifft2(spec).real
        if rough is not None:
            v = v + SR * np.asarray(rough, dtype=np.float64)
        self._kept_map = (int(key), v * 2 * unit)
        self.log.append(("psd_screen", int(key)))
        return self._kept_map[1].copy() if want_map else None

    def phase_map_items(self, wfe, items, wls, key=0):
        self.log.append(("phase_map_items", len(items)))
        if wfe is None:
            kept = getattr(self, "_kept_map", None)
            if kept is None or kept[0] != int(key) or not key:
                raise RuntimeError("paos_phase_map_items failed: no host map, and no map kept on the device under this key")
            wfe = kept[1]
        for i, wl in zip(items, wls):
            self.u[int(i)] = self.u[int(i)] * np.exp(2.0 * np.pi * 1j * np.asarray(wfe, dtype=np.float64) / wl)

    def ptp(self, blocks):
        self._single(blocks, False, "ptp")

    def stw(self, blocks, inverse):
        self._single(blocks, inverse, "stw")

    def wts(self, blocks, inverse):
        self._single(blocks, inverse, "wts")

    def phase(self, blocks, mul2pi):
        assert mul2pi
        self._single(blocks, False, "lens")
