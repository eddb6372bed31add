"""Sharding of independent wavefronts across GPUs (one process per GPU).

The reference parallelises over wavelengths with ``joblib.Parallel`` worker
processes on one host (paos/core/pipeline.py:140-150).  Here the unit of work is
the same -- one wavefront = one (wavelength, opt_chain) pair or one Monte-Carlo
WFE draw -- and the partition is static: rank ``r`` of ``W`` takes a contiguous
block.  The only exchange is ONE broadcast of the packed work description from
rank 0 (``paos_comm_bcast_blob``: RCCL over xGMI, or TCP in the CPU tests -- include/paos_comm.h,
no PyTorch); after that the ranks never talk until the optional gather of per-wavefront scalars.
No data-path collective.
"""
import os

import numpy as np

from . import wire


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of ``total`` items for ``rank``; blocks differ by at
    most one item and cover range(total) exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_work(work, comm=None, root=0):
    """The ONE broadcast: ``work`` (any structure ``wire`` can pack: wavelengths, chains, coefficient
    tables) from ``root`` to every rank of ``comm``; the other ranks pass None.  Single process: no-op."""
    if comm is None or comm.size == 1:
        return work
    blob = comm.bcast_blob(wire.dumps(work) if comm.rank == root else None, root=root)
    return work if comm.rank == root else wire.loads(blob)


def syn20_work(total, mode="wavelengths", wfe_table=None):
    """Work description for a SYN20 batch: ``total`` wavefronts, either a
    wavelength sweep lambda_k = 1 um (1 + k/512) with the seeded Zernike draw, or
    Monte-Carlo WFE columns at 1 um.  Small (KB): this is what rank 0 broadcasts."""
    from .chains import syn20_coefficients, syn20_wavelength

    if mode == "wavelengths":
        coef = syn20_coefficients()
        return {"wavelengths": [syn20_wavelength(k) for k in range(total)],
                "coefficients": [coef] * total}
    if mode == "wfe":
        table = np.asarray(wfe_table, dtype=np.float64)
        return {"wavelengths": [1.0e-6] * total,
                "coefficients": [np.append(np.zeros(3), table[:, k] * 1.0e-9) for k in range(total)]}
    raise ValueError(mode)


# ---- per-wavefront scalars as flat float64 records (what the gather moves) -------------------------
_PROP = ("", "II", "IO", "OI", "OO")
_SCALARS = ("wl", "dx", "dy", "wz", "distancetofocus", "fratio")


def _pack_result(index, res):
    """[index, n_surfaces, then per saved surface: num, 6 scalars, extent(4), propagator code, power,
    ABCDt(4) cin cout, ABCDs(4) cin cout, n_metrics, metrics...]"""
    rec = [float(index), float(len(res))]
    for num in sorted(res):
        r = res[num]
        rec.append(float(num))
        rec += [float(r[k]) for k in _SCALARS]
        rec += [float(x) for x in r["extent"]]
        rec.append(float(_PROP.index(r["propagator"])))
        rec.append(float(r.get("power", np.nan)))
        for key in ("ABCDt", "ABCDs"):
            m = np.asarray(r[key](), dtype=np.float64)
            rec += [m[0, 0], m[0, 1], m[1, 0], m[1, 1], float(r[key].cin), float(r[key].cout)]
        met = r.get("metrics")
        if met is None:
            rec.append(0.0)
        else:
            enc = np.asarray(met["encircled"], dtype=np.float64)
            rec.append(float(4 + enc.size))
            rec += [float(met["power"]), float(met["centroid"][0]), float(met["centroid"][1]), float(met["peak"])]
            rec += [float(x) for x in enc]
    return rec


def _unpack_results(flat):
    from .abcd import ABCD

    out, pos, flat = [], 0, np.asarray(flat, dtype=np.float64)
    while pos < flat.size:
        index, nsurf = int(flat[pos]), int(flat[pos + 1])
        pos += 2
        res = {}
        for _ in range(nsurf):
            num = int(flat[pos])
            r = dict(zip(_SCALARS, (float(x) for x in flat[pos + 1:pos + 7])))
            r["extent"] = tuple(float(x) for x in flat[pos + 7:pos + 11])
            r["propagator"] = _PROP[int(flat[pos + 11])]
            r["power"] = float(flat[pos + 12])
            pos += 13
            for key in ("ABCDt", "ABCDs"):
                m = ABCD()
                m.ABCD = np.array([[flat[pos], flat[pos + 1]], [flat[pos + 2], flat[pos + 3]]])
                m.cin, m.cout = flat[pos + 4], flat[pos + 5]
                r[key] = m
                pos += 6
            nmet = int(flat[pos])
            pos += 1
            if nmet:
                r["metrics"] = {"power": float(flat[pos]), "centroid": (float(flat[pos + 1]), float(flat[pos + 2])),
                                "peak": float(flat[pos + 3]), "encircled": flat[pos + 4:pos + nmet].copy()}
                pos += nmet
            res[num] = r
        out.append((index, res))
    return out


def run_sharded(pupil_diameter, wavelengths, gridsize, zoom, field, opt_chains, batch=8,
                precision="fp64", device=None, outputs=(), metrics_radii_px=None, gather=True,
                make_device=None, comm=None):
    """The reference's fan-out over wavelengths / Monte-Carlo draws (pipeline.py:139-150,
    joblib workers on one host) on N GPUs: call from every rank of ``comm`` (a ``paos_amd.comm.Comm``;
    None = single process).  Rank 0 supplies ``wavelengths`` and ``opt_chains`` (other ranks may pass
    None); they travel in ONE broadcast.  Every rank then propagates its contiguous shard in batches of
    ``batch`` wavefronts with ``run_batch`` -- no further communication -- and returns
    ``[(global index, result dict), ...]`` for its shard, arrays asked for in ``outputs`` included (PSFs
    stay with the rank that computed them).  With ``gather`` every rank ALSO receives the per-wavefront
    scalars of all ranks (sampling, f-ratio, power, ABCD matrices, on-device PSF metrics; < 1 KB per
    wavefront) and returns the full, index-ordered list, its own entries still carrying their arrays.

    A failure on one rank (an unsupported surface, a HIP error) is reported on every rank: the ranks
    exchange a status word before the gather, so nobody waits for a result that will not come.

    ``device``: GPU ordinal of this rank (default LOCAL_RANK, else 0).  ``make_device(n, nb)``
    lets the CPU tests substitute a model of the device."""
    from . import _lib
    from .run import run_batch

    rank = comm.rank if comm is not None else 0
    world = comm.size if comm is not None else 1
    work = {"wavelengths": list(wavelengths), "chains": list(opt_chains)} if rank == 0 else None
    work = broadcast_work(work, comm)
    total = len(work["chains"])
    if len(work["wavelengths"]) != total:
        raise ValueError("one wavelength per chain is required")
    lo, hi = shard_bounds(total, rank, world)
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    mine, failure = [], None
    dev, dev_nb = None, 0
    try:
        for start in range(lo, hi, int(batch)):
            stop = min(start + int(batch), hi)
            nb = stop - start
            if dev is None or nb != dev_nb:  # one context per batch size (the tail may be shorter)
                if dev is not None:
                    dev.close()
                dev = (make_device or (lambda n, b: _lib.DeviceFields(n, b, precision, device)))(int(gridsize), nb)
                dev_nb = nb
            res = run_batch(pupil_diameter, work["wavelengths"][start:stop], gridsize, zoom, field,
                            work["chains"][start:stop], precision=precision, outputs=outputs, dev=dev,
                            metrics_radii_px=metrics_radii_px)
            mine.extend(zip(range(start, stop), res))
    except Exception as exc:  # noqa: BLE001 -- reported to every rank below, then re-raised
        if world == 1:
            raise
        failure = exc
    finally:
        if dev is not None:
            dev.close()
    if world == 1:
        return mine
    # status word of every rank, then the messages of the ranks that failed
    status = comm.allgather_scalars([0.0 if failure is None else 1.0])
    bad = [r for r in range(world) if status[r][0] != 0.0]
    if bad:
        notes = []
        for r in bad:
            text = f"{type(failure).__name__}: {failure}" if r == rank else None
            notes.append((r, comm.bcast_blob(text.encode() if text is not None else None, root=r).decode()))
        if failure is not None:
            raise failure
        raise RuntimeError("run_sharded failed on " + "; ".join(f"rank {r}: {m}" for r, m in notes))
    if not gather:
        return mine
    flat = [x for index, res in mine for x in _pack_result(index, res)]
    parts = comm.allgather_scalars(flat)
    merged = {index: res for part in parts for index, res in _unpack_results(part)}
    for index, res in mine:  # this rank's own entries keep their arrays / aperture handles
        merged[index] = res
    return sorted(merged.items(), key=lambda p: p[0])
