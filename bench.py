#!/usr/bin/env python3
"""Headline benchmark: wavefronts/s through the 20-surface SYN20 chain.

    python bench.py --gpus 1 --steps 5 --warmup 1            # 4096^2 complex128 (default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" propagates one batch of ``--batch`` wavefronts (wavelength sweep
lambda_k = 1 um (1 + k/512)) per GPU through all 20 surfaces on the HIP path; fields
are created and stay in HBM.  With N GPUs every rank gets its own contiguous block of
the sweep (weak scaling) after ONE broadcast of the work description from rank 0.
Rank 0 prints one JSON line (contract in the task statement) with two extra objects:
``roofline`` (dominant kernel: the fused FFT pass, every launch HIP-event timed inside the
timed region) and ``cpu_baseline`` (the NumPy oracle on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def chain_fft_counts(wavelength, gridsize):
    """(n_ptp, n_stw, n_wts) the planner executes for SYN20 at this wavelength."""
    from paos_amd.chains import syn20_chain
    from paos_amd.run import _Item, _plan_surface

    st = _Item(1.0, wavelength, gridsize, 4, {"us": 0.0, "ut": 0.0})
    counts = {"ptp": 0, "stw": 0, "wts": 0}
    for item in syn20_chain().values():
        for step in _plan_surface(st, item)["steps"]:
            counts[step[0]] += 1
    return counts["ptp"], counts["stw"], counts["wts"]


def cpu_baseline(gridsize):
    """NumPy oracle ("port") on the host, 1 core: the full SYN20 chain for two wavelengths of the
    sweep at 2048^2 (saved surfaces only, like the reference's run()), scaled by the pixel ratio
    to the benchmark grid (flatters the CPU: its cost per pixel grows with the grid).  ~10-30 s."""
    from oracle.run_np import run as oracle_run
    from paos_amd.chains import syn20_chain, syn20_wavelength

    n_s = min(gridsize, 2048)
    sample = [syn20_wavelength(0), syn20_wavelength(256)]
    t0 = time.perf_counter()
    for wl in sample:
        oracle_run(1.0, wl, n_s, 4, {"us": 0.0, "ut": 0.0}, syn20_chain(), light=True)
    dt = (time.perf_counter() - t0) / len(sample)
    scale = (gridsize / n_s) ** 2
    return {
        "value": 1.0 / (dt * scale),
        "unit": "wavefronts/s",
        "cores": 1,
        "kind": "port",
        "sample": f"full SYN20 chain, {len(sample)} wavelengths of the sweep, {n_s}x{n_s} complex128, "
                  f"oracle/run_np.py (NumPy pocketfft, single thread): {dt:.1f} s per wavefront; scaled by "
                  f"{scale:g}x pixels to {gridsize}x{gridsize}; host has {os.cpu_count()} logical CPUs",
    }


def _oracle_seconds(task):
    """Worker of the parallel CPU baseline: one wavefront of the sweep, returns its wall time."""
    k, n_s = task
    from oracle.run_np import run as oracle_run
    from paos_amd.chains import syn20_chain, syn20_wavelength

    t0 = time.perf_counter()
    oracle_run(1.0, syn20_wavelength(k), n_s, 4, {"us": 0.0, "ut": 0.0}, syn20_chain(), light=True)
    return time.perf_counter() - t0


def cpu_baseline_parallel(gridsize, workers=8):
    """The reference's own way to use a CPU (pipeline.py:140: joblib over wavelengths): ``workers``
    processes, one wavelength of the sweep each, 2048^2, scaled like cpu_baseline.  Must run BEFORE
    this process touches the GPU (the workers are forked)."""
    import multiprocessing as mp

    n_s = min(gridsize, 2048)
    workers = max(1, min(workers, os.cpu_count() or 1))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        each = pool.map(_oracle_seconds, [(64 * k, n_s) for k in range(workers)])
    wall = time.perf_counter() - t0
    scale = (gridsize / n_s) ** 2
    return {
        "value": workers / (wall * scale),
        "unit": "wavefronts/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{workers} worker processes (the reference fans out over wavelengths with joblib, "
                  f"pipeline.py:140), one SYN20 wavefront each at {n_s}x{n_s}: {wall:.1f} s wall "
                  f"({min(each):.1f}-{max(each):.1f} s per worker); scaled by {scale:g}x pixels",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=8, help="wavefronts per GPU per step")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

    # forked CPU workers: before anything initialises the GPU in this process
    cpu_parallel = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_parallel = cpu_baseline_parallel(args.grid)

    dist = None
    if world > 1 or os.environ.get("PAOS_BENCH_FORCE_DIST") == "1":  # the flag exercises the
        # process-group path with a single rank (what a 1-GPU box can test)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist

        if os.environ.get("PAOS_BENCH_REHEARSAL") == "1":
            # several ranks sharing ONE GPU over gloo: rehearses the N > 1 control flow (broadcast,
            # shards, barrier, MAX reduction) on a single-GPU box; the numbers mean nothing
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from paos_amd import _lib
    from paos_amd.chains import syn20_chain
    from paos_amd.dist import broadcast_blob, max_over_ranks, shard_bounds, syn20_work
    from paos_amd.run import run_batch

    n, nb = args.grid, args.batch
    total = nb * world
    # rank 0 describes the whole job; one broadcast (RCCL over xGMI when N > 1)
    work = syn20_work(total, "wavelengths") if rank == 0 else None
    work = broadcast_blob(work, src=0)
    lo, hi = shard_bounds(total, rank, world)
    wavelengths = work["wavelengths"][lo:hi]
    chains = [syn20_chain(coefficients=c) for c in work["coefficients"][lo:hi]]
    field = {"us": 0.0, "ut": 0.0}

    dev = _lib.DeviceFields(n, nb, args.precision, device=local_rank)

    stats = {}

    def step():
        return run_batch(1.0, wavelengths, n, 4, field, chains, precision=args.precision,
                         outputs=(), dev=dev, sync=False, stats=stats, keep_psf=True)

    def barrier():
        dev.sync()
        if dist is not None:
            dist.barrier()
            if dist.get_backend() == "nccl":
                import torch

                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=64 * 1024)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    launches, kern_ms = dev.profile_end()
    elapsed = max_over_ranks(elapsed)

    # The north star's headline step: one ptp (fft2 -> H -> ifft2, wfo.py:462-472) on the
    # whole batch, timed alone with the pass timer (3 launches: rows | cols x2 fused | rows).
    ptp_ms = None
    if rank == 0:
        from paos_amd.planner import PilotBeam

        blk = [PilotBeam(1.0, wl, n, 4).ptp(2.5) for wl in wavelengths]
        for _ in range(2):
            dev.ptp(blk)
        dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=64)
        for _ in range(5):
            dev.ptp(blk)
        nl, tot = dev.profile_end()
        ptp_ms = tot / 5.0

    if rank == 0:
        esz = 16 if args.precision == "fp64" else 8
        value = total * args.steps / elapsed
        n_ptp, n_stw, n_wts = chain_fft_counts(wavelengths[0], n)
        ffts = 2 * n_ptp + n_stw + n_wts
        # 2 passes x (read + write) per 2-D FFT + one 8 B/px intensity write
        chain_bytes = (ffts * 4 * esz + 8) * n * n
        pass_bytes = 2 * esz * n * n * nb  # one pass over the batch: every element read + written once
        avg_ms = kern_ms / max(launches, 1)
        achieved = pass_bytes / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and n == 4096 and args.precision == "fp64":
            with open(tpath) as fh:
                traffic = json.load(fh).get("fused_pass_bytes_per_launch")
        out = {
            "metric": "wavefronts/sec (4096^2 c128, 20-surface chain) + achieved HBM GB/s",
            "value": value,
            "unit": "wavefronts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "c128 (f64)" if args.precision == "fp64" else "c64 (f32, f64 phases)",
            "data": "synthetic",
            "config": {"workload": f"SYN20 20-surface chain, {n}x{n} {args.precision}, wavelength sweep "
                                   f"1um*(1+k/512), {nb} wavefronts/GPU/step, {ffts} 2-D FFTs per wavefront "
                                   f"({n_ptp} ptp, {n_stw} stw, {n_wts} wts); the final |u|^2 of every wavefront is written "
                                   f"to HBM (8 B/px) and stays there, powers of the saved surfaces are reduced on the GPU",
                       "grid": n, "batch_per_gpu": nb, "parallelism": f"wavefront-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": "frugal_pass_kernel (every FFT pass launch, rows and columns)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "launches": launches, "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": pass_bytes,
                         "fused_passes_per_wavefront": stats.get("fused_passes"),
                         "copy_yardstick": {"in_place_copy_GBps": 5522.0, "same_tile_shape_GBps": 5080.0,
                                            "what": "tools/membench.hip: in-place read-modify-write of the same 4096^2 c128 x 8 "
                                                    "batch without the FFT (contiguous tiles / the FFT passes' tile shapes)",
                                            "source": "profiles/r01_membench_rmw_patterns.txt"} if (n == 4096 and esz == 16) else None},
            "chain_roofline": {"algorithmic_bytes_per_wavefront": chain_bytes,
                               "achieved_GBps_per_gpu": chain_bytes * (value / world) / 1e9,
                               "frac_of_hbm_peak": chain_bytes * (value / world) / 1e9 / HBM_PEAK_GBS},
            "ptp_step": {"what": "one ptp over the batch: 2 2-D FFTs + H, 3 fused passes",
                         "ms_per_wavefront": ptp_ms / nb,
                         "algorithmic_GBps": 8 * esz * n * n * nb / (ptp_ms * 1e-3) / 1e9,
                         "frac_of_hbm_peak": 8 * esz * n * n * nb / (ptp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "SURVEY 8d prices a ptp at 128 B/px (4 passes); the fused path moves 96 B/px"},
            "power_check": float(dev.norm2_fetch(res[0][20]["power_ticket"])[0]),
            "build": dev.build_info(),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
            out["cpu_baseline_parallel"] = cpu_parallel
        print(json.dumps(out), flush=True)
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
