"""Zernike index tables for the host planner.

The GPU evaluates the polynomials (csrc/zernike.hip); the host only has to turn
an ordering name and a term count into integer (m, n) tables and the
normalisation constants, exactly as the reference does in
paos/classes/zernike.py:131-176 (j -> m, n), :178-207 (m, n -> j) and :77-83
(norm).  Integer arithmetic (``math.isqrt``) replaces the reference's
floating-point ceil/sqrt expressions; equality with the reference tables is
pinned for the first 400 indices of every ordering (tests/golden/zernike_index.npz).
"""
import math

import functools

import numpy as np

ORDERINGS = ("ansi", "noll", "fringe", "standard")


def _mn_ansi(j):
    n = (math.isqrt(8 * j + 1) - 1) // 2
    return 2 * j - n * (n + 2), n


def _mn_standard(j):
    m, n = _mn_ansi(j)
    return -m, n


def _mn_noll(j):
    idx = j + 1
    n = (math.isqrt(8 * idx - 7) - 3) // 2 + 1
    first = n * (n + 1) // 2 + 1
    if n % 2 == 0:
        m = (idx - first + 1) // 2 * 2
    else:
        m = (idx - first) // 2 * 2 + 1
    return (-m if idx % 2 else m), n


def _mn_fringe(j):
    idx = j + 1
    half = math.isqrt(idx - 1)  # ceil(sqrt(idx)) - 1
    start = half * half + 1
    n = half + (idx - start) // 2
    m = 2 * half - n
    return (-m if (idx - start) % 2 else m), n


_TABLE = {"ansi": _mn_ansi, "standard": _mn_standard, "noll": _mn_noll, "fringe": _mn_fringe}


class Zernike:
    """Index bookkeeping with the reference's static-method names."""

    @staticmethod
    def j2mn(N, ordering):
        m, n = _j2mn_cached(int(N), ordering)
        return m.copy(), n.copy()

    @staticmethod
    def mn2j(m, n, ordering):
        m = np.asarray(m)
        n = np.asarray(n)
        if ordering == "ansi":
            return (n * (n + 2) + m) // 2
        if ordering == "standard":
            return (n * (n + 2) - m) // 2
        if ordering == "fringe":
            am = np.abs(m)
            base = (1 + (n + am) // 2) ** 2 - 2 * am
            # reference: int(a - b - (1 + sign m)/2) + 1, i.e. +1 only for m < 0
            return base + (m < 0)
        if ordering == "noll":
            am = np.abs(m)
            low = (n % 4 == 0) | (n % 4 == 1)
            bump = np.where(low, m <= 0, m >= 0).astype(np.int64)
            return (n * (n + 1) // 2 + am + bump).astype(np.int64)
        raise NameError("Ordering not supported.")


@functools.lru_cache(maxsize=64)
def _j2mn_cached(count, ordering):
    """Index tables are pure functions of (count, ordering); the planner asks for them once per
    wavefront and Zernike surface.  The cached arrays are read-only."""
    if ordering not in _TABLE:
        raise NameError("Ordering not supported.")
    pairs = [_TABLE[ordering](j) for j in range(count)]
    m = np.array([p[0] for p in pairs], dtype=int)
    n = np.array([p[1] for p in pairs], dtype=int)
    m.setflags(write=False)
    n.setflags(write=False)
    return m, n


@functools.lru_cache(maxsize=64)
def zernike_tables(count, ordering, normalize):
    """(m, n, norm) of the first ``count`` polynomials, read-only and shared between calls."""
    m, n = _j2mn_cached(int(count), ordering)
    norm = norm_factors(m, n, normalize)
    norm.setflags(write=False)
    return m, n, norm


def norm_factors(m, n, normalize):
    """sqrt(n+1) for m == 0, sqrt(2(n+1)) otherwise when ``normalize`` is truthy
    (the pipeline passes the *string* "True", reference pipeline.py:125); ones
    otherwise -- zernike.py:77-83."""
    if normalize:
        return np.array(
            [np.sqrt(nk + 1) if mk == 0 else np.sqrt(2.0 * (nk + 1)) for mk, nk in zip(m, n)],
            dtype=np.float64,
        )
    return np.ones(len(m), dtype=np.float64)
