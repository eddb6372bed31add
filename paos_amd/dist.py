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
