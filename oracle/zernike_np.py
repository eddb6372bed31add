"""Zernike index tables and polynomial maps on the CPU (NumPy + SciPy).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
paos/classes/zernike.py: index conversion j -> (m, n) (zernike.py:131-176),
(m, n) -> j (zernike.py:178-207), radial part through the Jacobi polynomial
(zernike.py:209-247), normalisation (zernike.py:77-83), the rho > 1 mask
(zernike.py:85-89) and the azimuthal part cos(m phi) / sin(|m| phi)
(zernike.py:100-104), and the Gram-Schmidt orthonormal variant PolyOrthoNorm
(zernike.py:293-402).  Pinned by tests/golden/zernike_*.npz, orthonorm_*.npz
(reference import) and by the reference notebook KAT.
"""
import numpy as np
from scipy.special import eval_jacobi

ORDERINGS = ("ansi", "noll", "fringe", "standard")


def index_to_mn(count, ordering):
    """First ``count`` (m, n) pairs of an ordering -- zernike.py:131-176."""
    j = np.arange(count, dtype=int)
    if ordering in ("ansi", "standard"):
        n = np.ceil((-3.0 + np.sqrt(9.0 + 8.0 * j)) / 2.0).astype(int)
        m = 2 * j - n * (n + 2)
        if ordering == "standard":
            m = -2 * j + n * (n + 2)
        return m, n
    if ordering == "noll":
        idx = j + 1
        n = ((0.5 * (np.sqrt(8 * idx - 7) - 3)) + 1).astype(int)
        cn = n * (n + 1) / 2 + 1
        even = n % 2 == 0
        m = np.empty(count, dtype=int)
        m[even] = (idx[even] - cn[even] + 1) // 2 * 2
        m[~even] = (idx[~even] - cn[~even]) // 2 * 2 + 1
        m = (-1) ** (idx % 2) * m
        return m, n
    if ordering == "fringe":
        idx = j + 1
        m_n = 2 * (np.ceil(np.sqrt(idx)) - 1)
        g_s = (m_n / 2) ** 2 + 1
        n = m_n / 2 + np.floor((idx - g_s) / 2)
        m = (m_n - n) * (1 - np.mod(idx - g_s, 2) * 2)
        return m.astype(int), n.astype(int)
    raise NameError("Ordering not supported.")


def mn_to_index(m, n, ordering):
    """(m, n) -> j -- zernike.py:178-207."""
    m = np.asarray(m)
    n = np.asarray(n)
    if ordering == "ansi":
        return (n * (n + 2) + m) // 2
    if ordering == "standard":
        return (n * (n + 2) - m) // 2
    if ordering == "fringe":
        a = (1 + (n + np.abs(m)) / 2) ** 2
        return (a - 2 * np.abs(m) - (1 + np.sign(m)) / 2).astype(int) + 1
    if ordering == "noll":
        p = np.zeros(n.size, dtype=np.int64)
        for k, (mk, nk) in enumerate(zip(np.atleast_1d(m), np.atleast_1d(n))):
            lo = nk % 4 in (0, 1)
            if (mk > 0 and lo) or (mk < 0 and not lo):
                p[k] = 0
            elif (mk >= 0 and not lo) or (mk <= 0 and lo):
                p[k] = 1
            else:
                raise ValueError("Invalid (m,n) in Noll indexing.")
        return (n * (n + 1) / 2 + np.abs(m) + p).astype(np.int64)
    raise NameError("Ordering not supported.")


def radial(m, n, rho):
    """R_n^|m|(rho) through P_k^(|m|,0)(1 - 2 rho^2) -- zernike.py:245-247."""
    m = abs(int(m))
    k = (n - m) // 2
    return (-1) ** k * rho**m * eval_jacobi(k, m, 0.0, (1.0 - 2.0 * rho**2))


def norms(m, n, normalize):
    """sqrt(n+1) (m == 0) / sqrt(2(n+1)) when normalised, else 1 -- zernike.py:77-83.
    ``normalize`` is used for truthiness only (the pipeline passes the string
    "True", pipeline.py:125)."""
    if normalize:
        return np.array(
            [np.sqrt(nk + 1) if mk == 0 else np.sqrt(2.0 * (nk + 1)) for mk, nk in zip(m, n)]
        )
    return np.ones(len(m), dtype=np.float64)


def zernike_stack(count, rho, phi, ordering="ansi", normalize=False):
    """(count, ...) masked array of polynomials -- zernike.py:63-109."""
    assert ordering in ORDERINGS, "Unrecognised ordering scheme."
    assert count > 0, "N shall be a positive integer"
    m, n = index_to_mn(count, ordering)
    nrm = norms(m, n, normalize)
    outside = rho > 1.0
    if isinstance(rho, np.ma.MaskedArray):
        rho.mask |= outside
    else:
        rho = np.ma.MaskedArray(data=rho, mask=outside, fill_value=0.0)
    rad = {}
    for nn in range(max(n) + 1):
        for mm in range(-nn, 1, 2):
            rad[(nn, -mm)] = radial(mm, nn, rho)
    az = {0: np.ones_like(phi)}
    for mm in range(1, m.max() + 1):
        az[mm] = np.cos(mm * phi)
        az[-mm] = np.sin(mm * phi)
    return np.ma.MaskedArray(
        [nrm[k] * rad[(n[k], abs(m[k]))] * az[m[k]] for k in range(count)],
        fill_value=0.0,
    )


def covariance(stack):
    """M[i, j] = masked mean of Z[i] * Z[j], entries below 1e-10 zeroed -- zernike.py:293-318."""
    k = stack.shape[0]
    cov = np.empty((k, k))
    for i in range(k):
        for j in range(i, k):
            cov[i, j] = cov[j, i] = np.ma.mean(stack[i] * stack[j])
    cov[np.abs(cov) < 1e-10] = 0.0
    return cov


def poly_orthonorm_stack(count, rho, phi, ordering="ansi", normalize=False, mask=False):
    """Polynomials orthonormal over the unmasked pixels: U = M Z with M the inverse of the
    Cholesky factor of the covariance -- PolyOrthoNorm.__init__, zernike.py:388-402.
    Returns (U, M)."""
    stack = zernike_stack(count, rho, phi, ordering=ordering, normalize=normalize)
    cov = covariance(stack)
    qt = np.linalg.cholesky(cov)
    m = np.linalg.inv(qt)
    m[np.where(np.abs(m) < 1.0e-10)] = 0.0
    full_mask = stack.mask | mask
    z1 = np.tensordot(m, stack.filled(fill_value=0), axes=1)
    return np.ma.MaskedArray(data=z1, mask=full_mask, fill_value=0.0), m
