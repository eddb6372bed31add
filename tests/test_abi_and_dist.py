"""The C-ABI library loads and exports what include/paos_hip.h declares (no compute calls
without a GPU); the multi-GPU sharding helpers under a world_size-2 gloo group."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "paos_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(paos_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from paos_amd import _lib

    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/paos_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes binding table and header disagree"
    assert b"gfx950" in lib.paos_build_info()


def test_abi_struct_layout_matches_header():
    from paos_amd import _lib

    assert ctypes.sizeof(_lib.PwOp) == 12
    assert ctypes.sizeof(_lib.Pass) == 6 * 4 + 3 * _lib.MAX_PW * 12
    text = open(os.path.join(ROOT, "include", "paos_hip.h")).read()
    assert f"PAOS_MAX_PW = {_lib.MAX_PW}" in text
    assert f"PAOS_PHASE_STRIDE = {_lib.PHASE_STRIDE}" in text
    assert f"PAOS_APERTURE_STRIDE = {_lib.APERTURE_STRIDE}" in text


def test_context_creation_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device the product path raises."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from paos_amd import _lib

    with pytest.raises(_lib.PaosHipError):
        _lib.DeviceFields(64, 1)
    from paos_amd.wfo import WFO

    with pytest.raises(_lib.PaosHipError):
        WFO(1.0, 1e-6, 64, 4)


def test_shard_bounds_cover_exactly():
    from paos_amd.dist import shard_bounds

    for total in (0, 1, 7, 64, 256, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _gloo_worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fakes import ModelDevice
        from paos_amd.chains import syn20_chain
        from paos_amd.dist import broadcast_blob, max_over_ranks, shard_bounds, syn20_work
        from paos_amd.run import _Item, _walk

        total = 5
        work = syn20_work(total, "wavelengths") if rank == 0 else None
        work = broadcast_blob(work, src=0)
        lo, hi = shard_bounds(total, rank, world)
        wls = work["wavelengths"][lo:hi]
        chains = [syn20_chain(coefficients=c) for c in work["coefficients"][lo:hi]]
        dev = ModelDevice(64, len(chains))
        dev.fill(1.0)
        states = [_Item(1.0, wl, 64, 4, {"us": 0.0, "ut": 0.0}) for wl in wls]
        power = {}

        def on_saved(key, items, plans, wfe):
            p = dev.norm2()
            for i, it in enumerate(items):
                power[(lo + i, it["num"])] = float(p[i])

        _walk(dev, states, chains, on_saved)
        slowest = max_over_ranks(float(rank + 1))
        out.put((rank, lo, hi, [float(w) for w in work["wavelengths"]], power, slowest))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    """world_size 2 on CPU: one broadcast of the work description, disjoint shards that
    together cover the batch, MAX-reduction of the time bracket; each rank's wavefronts agree
    with the single-process oracle."""
    import socket

    import torch.multiprocessing as mp

    from oracle.run_np import run as oracle_run
    from paos_amd.chains import syn20_chain, syn20_wavelength

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    (r0, lo0, hi0, wl0, pw0, s0), (r1, lo1, hi1, wl1, pw1, s1) = results
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)
    assert wl0 == wl1 == [syn20_wavelength(k) for k in range(5)]
    assert s0 == s1 == 2.0
    power = {**pw0, **pw1}
    assert sorted(k[0] for k in power if k[1] == 20) == [0, 1, 2, 3, 4]
    for idx in (0, 4):
        ref = oracle_run(1.0, syn20_wavelength(idx), 64, 4, {"us": 0.0, "ut": 0.0}, syn20_chain(), light=True)
        assert abs(power[(idx, 20)] - np.sum(ref[20]["amplitude"] ** 2)) < 1e-12


def _sharded_worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fakes import ModelDevice
        from paos_amd.chains import syn20_chain, syn20_wavelength
        from paos_amd.dist import run_sharded

        total = 5
        wls = [syn20_wavelength(k * 50) for k in range(total)] if rank == 0 else None
        chains = [syn20_chain() for _ in range(total)] if rank == 0 else None
        res = run_sharded(1.0, wls, 64, 4, {"us": 0.0, "ut": 0.0}, chains, batch=2, outputs=("psf",),
                          make_device=lambda n, nb: ModelDevice(n, nb))
        out.put((rank, [(i, float(r[20]["power"]), float(r[20]["psf"].sum()), r[20]["dx"]) for i, r in res]))
    finally:
        dist.destroy_process_group()


def test_run_sharded_two_ranks_gloo():
    """run_sharded: rank 0 alone holds the work, both ranks return the full ordered result list,
    and it equals the single-process run."""
    import socket

    import torch.multiprocessing as mp

    from oracle.run_np import run as oracle_run
    from paos_amd.chains import syn20_chain, syn20_wavelength

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] == results[1]
    assert [i for i, *_ in results[0]] == [0, 1, 2, 3, 4]
    for i, power, psf_sum, dx in (results[0][0], results[0][4]):
        ref = oracle_run(1.0, syn20_wavelength(i * 50), 64, 4, {"us": 0.0, "ut": 0.0}, syn20_chain(), light=True)
        assert abs(power - np.sum(ref[20]["amplitude"] ** 2)) < 1e-12
        assert abs(psf_sum - np.sum(ref[20]["amplitude"] ** 2)) < 1e-12
        assert dx == ref[20]["dx"]
