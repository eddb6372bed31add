"""Sharding of independent wavefronts across GPUs (one process per GPU).

The reference parallelises over wavelengths with ``joblib.Parallel`` worker
processes on one host (paos/core/pipeline.py:140-150).  Here the unit of work is
the same -- one wavefront = one (wavelength, opt_chain) pair or one Monte-Carlo
WFE draw -- and the partition is static: rank ``r`` of ``W`` takes a contiguous
block.  The only exchange is ONE broadcast of the packed work description from
rank 0 (RCCL over xGMI when the process group is NCCL, gloo in CPU tests); after
that the ranks never talk until the timing reduction.  No data-path collective.
"""
import pickle

import numpy as np


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of ``total`` items for ``rank``; blocks differ by at
    most one item and cover range(total) exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_blob(obj, src=0, device=None):
    """Broadcast a picklable object from ``src`` to every rank with two
    ``torch.distributed.broadcast`` calls (length, then bytes).  ``device`` is the
    tensor device: a CUDA device under the NCCL(=RCCL) backend, CPU under gloo."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return obj
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rank = dist.get_rank()
    payload = pickle.dumps(obj) if rank == src else b""
    size = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    dist.broadcast(size, src)
    buf = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(buf, src)
    return pickle.loads(buf.cpu().numpy().tobytes())


def max_over_ranks(value, device=None):
    """MAX all-reduce of a scalar (the bench's time bracket)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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


def run_sharded(pupil_diameter, wavelengths, gridsize, zoom, field, opt_chains, batch=8,
                precision="fp64", device=None, outputs=(), metrics_radii_px=None, gather=True,
                make_device=None):
    """The reference's fan-out over wavelengths / Monte-Carlo draws (pipeline.py:139-150,
    joblib workers on one host) on N GPUs: call from every rank of an initialised process group
    (or from a single process).  Rank 0 supplies ``wavelengths`` and ``opt_chains`` (other ranks
    may pass None); they travel in ONE broadcast.  Every rank then propagates its contiguous
    shard in batches of ``batch`` wavefronts with ``run_batch`` -- no further communication --
    and returns ``[(global index, result dict), ...]`` for its shard; with ``gather`` the
    per-wavefront results (scalars, power, metrics, and whatever ``outputs`` asks for) are
    collected so that every rank returns the full, index-ordered list.

    ``device``: GPU ordinal of this rank (default LOCAL_RANK, else 0).  ``make_device(n, nb)``
    lets the CPU tests substitute a model of the device."""
    import os

    from . import _lib
    from .run import run_batch

    try:
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized()
    except ImportError:  # single process without torch
        dist, live = None, False
    rank = dist.get_rank() if live else 0
    world = dist.get_world_size() if live else 1
    work = {"wavelengths": list(wavelengths), "chains": list(opt_chains)} if rank == 0 else None
    work = broadcast_blob(work, src=0)
    total = len(work["chains"])
    if len(work["wavelengths"]) != total:
        raise ValueError("one wavelength per chain is required")
    lo, hi = shard_bounds(total, rank, world)
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    mine = []
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
    finally:
        if dev is not None:
            dev.close()
    if not (gather and live and world > 1):
        return mine
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    return sorted((pair for part in parts for pair in part), key=lambda p: p[0])
