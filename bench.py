#!/usr/bin/env python3
"""Headline benchmark: wavefronts/s through the 20-surface SYN20 chain.

    python bench.py --gpus 1 --steps 5 --warmup 1            # 4096^2 complex128 (default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" propagates one batch of ``--batch`` wavefronts (wavelength sweep lambda_k = 1 um (1 + k/512)) per GPU
through all 20 surfaces on the HIP path; fields are created and stay in HBM.  With N GPUs every rank gets its own
contiguous block of the sweep (weak scaling) after ONE broadcast of the packed work description from rank 0
(paos_comm_bcast_blob: RCCL over xGMI, driven from libpaoship.so -- torch is only the launcher, it is never
imported here).  With N > 1 the run FAILS (exit 3) when the ranks could not agree on RCCL, unless --allow-tcp is
given; `config.ranks_seen` / `devices_seen` come from the communicator.  Rank 0 prints one JSON line (contract in the task statement) with these extra objects:

  roofline           the dominant kernel (the fused FFT pass): every pass launch of the timed region is
                     bracketed by HIP events on the context's stream (paos_profile_end_launches: time and class of
                     each launch); `achieved` = 32 B/px x N^2 x batch (one read and one write of every element) / mean
                     duration of the launches that skip nothing; `classes` lists every class of launch (full, skipping
                     tiles of dead lines / loads of dead positions / stores nobody reads, storing the PSF) with its mean
                     time and -- from the counters -- the bytes it really moved; `all_launches` sums both over one
                     step; `copy_yardstick` is measured in this run (paos_copy_yardstick); `traffic` = HBM bytes per
                     full launch from FETCH_SIZE / WRITE_SIZE, collected by two child runs of this script under
                     rocprofv3 --pmc (measure_traffic; N = 1 only, --no-traffic skips it).
  chain_vs_survey_model / ptp_step   the whole chain and one ptp priced with SURVEY 8d's UNFUSED byte model (2 passes
                     per 2-D FFT) next to the bytes the launches of one step really moved under the counters
                     (`bytes_moved_per_wavefront`, every kernel of the step; `other_kernels` lists the non-pass ones).
  without_ptp_algebra  the rate (3 steps) with the pass compiler's two ptp identities switched off (PAOS_PTP_ALGEBRA=0:
                     44 passes per wavefront instead of 24), so that the headline value can be read without them.
  extra              the same chain at 2048^2 and 1024^2 (the north star's sweep), value + roofline each.
  cpu_baseline       the NumPy oracle ("port") on the host: one wavefront of the workload at the benchmark grid
                     on one core, and `cpu_baseline_parallel`: min(batch, cores, memory, 8) worker processes over
                     wavelengths at the benchmark grid -- the reference's own joblib scheme (pipeline.py:140).
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
ON_AXIS = {"us": 0.0, "ut": 0.0}


def chain_fft_counts(wavelength, gridsize):
    """(n_ptp, n_stw, n_wts) the planner executes for SYN20 at this wavelength."""
    from paos_amd.chains import syn20_chain
    from paos_amd.planner import BeamBatch
    from paos_amd.run import _surface_gates

    beams = BeamBatch(1.0, [wavelength], gridsize, 4)
    n_ptp = n_stw = n_wts = 0
    for item in syn20_chain().values():
        _, stw, ptp, wts, _, _ = beams.surface(*_surface_gates([item]))
        n_stw += int(stw[0, 0] != 0.0)
        n_ptp += int(ptp[0, 0] != 0.0)
        n_wts += int(wts[0, 0] != 0.0)
    return n_ptp, n_stw, n_wts


def _oracle_seconds(task):
    """One SYN20 wavefront of the sweep through the NumPy oracle; returns its wall time."""
    k, n = task
    from oracle.run_np import run as oracle_run
    from paos_amd.chains import syn20_chain, syn20_wavelength

    t0 = time.perf_counter()
    oracle_run(1.0, syn20_wavelength(k), n, 4, ON_AXIS, syn20_chain(), light=True)
    return time.perf_counter() - t0


def cpu_baseline(gridsize):
    """The oracle on ONE core: one wavefront of the workload at the benchmark grid (saved surfaces only, like the
    reference's run()).  ~50 s at 4096^2 on the GPU box's host."""
    dt = _oracle_seconds((0, gridsize))
    return {
        "value": 1.0 / dt,
        "unit": "wavefronts/s",
        "cores": 1,
        "kind": "port",
        "sample": f"full SYN20 chain, 1 wavelength of the sweep at the benchmark grid {gridsize}x{gridsize} complex128, "
                  f"oracle/run_np.py (NumPy pocketfft, single thread): {dt:.1f} s per wavefront; host has "
                  f"{os.cpu_count()} logical CPUs",
    }


def cpu_baseline_parallel(gridsize, batch):
    """The reference's own way to use a CPU (pipeline.py:140: joblib over wavelengths): n_jobs = min(batch, cores)
    worker processes, one wavelength of the sweep each, at the benchmark grid; n_jobs is also capped by memory (the
    oracle peaks at ~17 GB per 4096^2 wavefront: the reference's Zernike evaluation holds 36 maps).  Must run BEFORE
    this process touches the GPU (the workers are forked)."""
    import multiprocessing as mp

    cores = os.cpu_count() or 1
    try:  # the CPUs this process may actually use, and the cgroup's share of them
        cores = min(cores, len(os.sched_getaffinity(0)))
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    per_worker_gb = 17.0 * (gridsize / 4096.0) ** 2 + 0.5
    try:
        import psutil

        avail_gb = psutil.virtual_memory().available / 1e9
    except Exception:  # noqa: BLE001
        avail_gb = 64.0
    try:  # a container's memory limit is not in /proc/meminfo
        limit = open("/sys/fs/cgroup/memory.max").read().strip()
        if limit != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read())
            avail_gb = min(avail_gb, (int(limit) - used) / 1e9)
    except Exception:  # noqa: BLE001
        pass
    by_memory = max(1, int(0.5 * avail_gb / per_worker_gb))
    # Never more than kMaxWorkers: a GPU box is a slice of a host whose /proc numbers describe the whole machine
    # (round 2 lost a box to 32 workers x 17 GB that the host-wide figures allowed); 8 is the measured-safe count.
    kMaxWorkers = 8
    workers = max(1, min(batch, cores, by_memory, kMaxWorkers))
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        each = pool.map(_oracle_seconds, [(64 * k, gridsize) for k in range(workers)])
    wall = time.perf_counter() - t0
    return {
        "value": workers / wall,
        "unit": "wavefronts/s",
        "cores": workers,
        "kind": "port",
        "sample": f"{workers} worker processes = min(batch {batch}, {cores} usable CPUs, 8, memory: {avail_gb:.0f} GB free at "
                  f"~{per_worker_gb:.0f} GB per worker) over wavelengths (pipeline.py:140), one SYN20 wavefront each at "
                  f"{gridsize}x{gridsize}: {wall:.1f} s wall ({min(each):.1f}-{max(each):.1f} s per worker)",
    }


def measure(dev, n, nb, precision, wavelengths, chains, steps, warmup, comm=None):
    """Timed region of the contract: W untimed steps, then exactly K steps between barriers; every pass launch
    is event-timed.  Returns a dict of raw numbers."""
    from paos_amd import _lib
    from paos_amd.run import run_batch

    stats = {}

    def step():
        return run_batch(1.0, wavelengths, n, 4, ON_AXIS, chains, precision=precision, outputs=(), dev=dev,
                         sync=False, stats=stats, keep_psf=True)

    def barrier():
        dev.sync()
        if comm is not None:
            comm.barrier()

    def release(results):  # power tickets of a step nobody will read (no synchronisation)
        if results is not None:
            for t in {rec["power_ticket"] for r in results for rec in r.values() if "power_ticket" in rec}:
                dev.norm2_release(t)

    res = None
    for _ in range(warmup):
        release(res)
        res = step()
    barrier()
    dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=64 * 1024)
    t0 = time.perf_counter()
    for _ in range(steps):
        release(res)
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ms, tags = dev.profile_end_launches()
    if comm is not None:
        elapsed = comm.max(elapsed)
    full = tags == 0
    return {"elapsed": elapsed, "launches": int(ms.size), "kern_ms": float(ms.sum()), "pruned": int((~full).sum()),
            "pruned_ms": float(ms[~full].sum()), "launch_ms": ms, "launch_tags": tags,
            "fused_passes": stats.get("fused_passes"), "res": res}


def measure_traffic(grid, batch, precision):
    """HBM bytes of every kernel of ONE chain step from the PMC counters, collected the way MI355X_MICROARCH.md
    prescribes: this script once under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and once under `... --pmc WRITE_SIZE`
    (separate passes, as child processes running one warm-up step and one counted step and nothing else: --traffic-child), FETCH_SIZE /
    WRITE_SIZE in KiB, FETCH_SIZE doubled (gfx950 reports half of the bytes of wide coalesced reads; the factor is
    calibrated for 16 B per lane, which is what the pass kernels issue -- for the 8-byte accesses of the reduction and
    PSF kernels it is an assumption).  Dispatches are matched between the two runs by dispatch order.
    Returns ({"pass": [(read, written)] in launch order, "other": {kernel: {calls, read, written, ms}}} or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    series, trace_ms = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="paos_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", tmp, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--grid", str(grid), "--batch", str(batch),
                   "--precision", precision, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
                   "--no-traffic", "--traffic-child"]
            run = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                 stderr=subprocess.PIPE, timeout=600)
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if run.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (exit {run.returncode})"
            per_dispatch = {}
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] != counter or row["Kernel_Name"].startswith("__amd_rocclr"):
                        continue  # the runtime's own fill / copy kernels (context creation, parameter uploads)
                    k = int(row["Dispatch_Id"])
                    name, val = per_dispatch.get(k, (row["Kernel_Name"], 0.0))
                    per_dispatch[k] = (name, val + float(row["Counter_Value"]))
            seq = [per_dispatch[k] for k in sorted(per_dispatch)]
            # the child runs a warm-up step first: the counted step is the steady state the timed region measures
            # (aperture records found in the context's sets, PSF zeros already in the buffer).  A step opens with the
            # start-of-chain power kernel: keep what follows the last one.
            starts = [i for i, (name, _) in enumerate(seq) if "start_power_kernel" in name]
            series[counter] = seq[starts[-1]:] if starts else seq
            if counter == "FETCH_SIZE":  # durations of the non-pass kernels (under the profiler: indicative)
                rows = []
                for path in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
                    with open(path) as fh:
                        for row in csv.DictReader(fh):
                            try:
                                t0, t1 = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
                            except (KeyError, ValueError):
                                continue
                            rows.append((t0, row.get("Kernel_Name", ""), (t1 - t0) * 1e-6))
                rows.sort()
                opens = [i for i, r in enumerate(rows) if "start_power_kernel" in r[1]]
                for _, name, dt in (rows[opens[-1]:] if opens else rows):  # the counted step, like the counters
                    trace_ms[name] = trace_ms.get(name, 0.0) + dt
        except Exception as exc:  # noqa: BLE001 -- the bench line must come out whatever the profiler does
            return None, f"rocprofv3 --pmc {counter}: {type(exc).__name__}: {exc}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    f, w = series["FETCH_SIZE"], series["WRITE_SIZE"]
    if not f or len(f) != len(w) or any(a[0] != b[0] for a, b in zip(f, w)):
        return None, f"dispatches seen: {len(f)} (FETCH_SIZE run) vs {len(w)} (WRITE_SIZE run), or in another order"
    out = {"pass": [], "other": {}}
    for (name, fetch), (_, write) in zip(f, w):
        rd, wr = 2.0 * fetch * 1024.0, write * 1024.0
        if "_pass_kernel" in name:
            out["pass"].append((rd, wr))
        else:
            short = name.split("<")[0].split("(")[0]
            rec = out["other"].setdefault(short, {"calls": 0, "read": 0.0, "written": 0.0, "ms": 0.0})
            rec["calls"] += 1
            rec["read"] += rd
            rec["written"] += wr
    for name, ms in trace_ms.items():
        short = name.split("<")[0].split("(")[0]
        if short in out["other"]:
            out["other"][short]["ms"] += ms
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE on one chain step of this workload behind a warm-up step "
            "(two child runs, counters in KiB, FETCH_SIZE x2 per MI355X_MICROARCH.md), every dispatch of that step kept")
    return out, note


CLASS_NAMES = {0: "full", 1: "skips tiles of dead lines", 2: "skips loads of dead positions", 4: "skips stores nobody reads",
               8: "stores the PSF instead of the field"}


def class_name(tag):
    return " + ".join(CLASS_NAMES[b] for b in (1, 2, 4, 8) if tag & b) if tag else CLASS_NAMES[0]


def roofline_block(m, n, nb, esz, dev, kernel_name, steps, traffic=None):
    """`achieved` = algorithmic bytes of one pass over the batch / mean HIP-event time of the launches that skip nothing.
    With ``traffic`` (measure_traffic): per class of launch and for all pass launches of a step the bytes the counters
    saw, next to the event times of the same launches -- no estimate anywhere."""
    pass_bytes = 2 * esz * n * n * nb  # one pass over the batch: every element read + written once
    ms, tags = m["launch_ms"], m["launch_tags"]
    full = tags == 0
    full_ms = float(ms[full].mean()) if full.any() else 0.0
    achieved = pass_bytes / (full_ms * 1e-3) / 1e9 if full.any() else 0.0
    y_ms, y_bytes = dev.copy_yardstick(10)
    per_step = ms.size // max(steps, 1)
    folded = ms[:per_step * steps].reshape(steps, per_step).mean(axis=0) if per_step and ms.size == per_step * steps else None
    step_tags = tags[:per_step] if folded is not None else None
    block = {
        "bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        "traffic_note": "not measured in this run (--no-traffic, an `extra` entry, or N > 1); per-kernel FETCH_SIZE / WRITE_SIZE "
                        "of the default command under rocprofv3: profiles/r03_pmc_hbm_traffic_bench.txt",
        "launches": int(full.sum()), "avg_launch_ms": full_ms, "algorithmic_bytes_per_launch": pass_bytes,
        "pruned": {"launches": int((~full).sum()), "avg_launch_ms": float(ms[~full].mean()) if (~full).any() else 0.0,
                   "what": "pass launches next to an aperture that skip the tiles / loads of rows or columns it has zeroed, "
                           "or the stores of rows it is about to zero, and the last pass that stores the PSF (they move "
                           "fewer bytes and are kept out of `achieved`)"},
        "all_pass_launches_avg_ms": float(ms.mean()) if ms.size else 0.0,
        "fused_passes_per_wavefront": m["fused_passes"],
        "copy_yardstick": {"ms_per_launch": y_ms, "GBps": y_bytes / (y_ms * 1e-3) / 1e9,
                           "what": "measured in this run (paos_copy_yardstick): in-place copy of the same batch buffer, "
                                   "16 B per lane, unit stride, no transform"},
    }
    if folded is not None:
        classes = {}
        for i in range(per_step):
            rec = classes.setdefault(class_name(int(step_tags[i])), {"launches_per_step": 0, "ms": 0.0, "bytes_measured": None})
            rec["launches_per_step"] += 1
            rec["ms"] += float(folded[i])
        measured = traffic is not None and len(traffic["pass"]) == per_step
        if measured:
            for i in range(per_step):
                rec = classes[class_name(int(step_tags[i]))]
                rec["bytes_measured"] = (rec["bytes_measured"] or 0.0) + sum(traffic["pass"][i])
        for rec in classes.values():
            k = rec["launches_per_step"]
            rec["avg_launch_ms"] = rec.pop("ms") / k
            if rec["bytes_measured"] is not None:
                rec["bytes_measured"] /= k
                rec["GBps_measured"] = rec["bytes_measured"] / (rec["avg_launch_ms"] * 1e-3) / 1e9
                rec["frac_measured"] = rec["GBps_measured"] / HBM_PEAK_GBS
        block["classes"] = classes
        block["all_launches"] = {"launches_per_step": per_step, "ms": float(folded.sum()), "bytes_measured": None, "frac": None,
                                 "what": "every pass launch of one step: HIP-event time (mean over the timed steps) and, when "
                                         "the counters were collected, the HBM bytes they saw for the same launches"}
        if measured:
            total = sum(r + w for r, w in traffic["pass"])
            block["all_launches"]["bytes_measured"] = total
            block["all_launches"]["frac"] = total / (float(folded.sum()) * 1e-3) / 1e9 / HBM_PEAK_GBS
            fulls = [sum(traffic["pass"][i]) for i in range(per_step) if step_tags[i] == 0]
            reads = [traffic["pass"][i][0] for i in range(per_step) if step_tags[i] == 0]
            if fulls:
                block["traffic"] = sum(fulls) / len(fulls)
                block["traffic_over_algorithmic"] = block["traffic"] / pass_bytes
                block["traffic_note"] = (f"mean over the {len(fulls)} of {per_step} pass launches of a step that skip nothing: read "
                                         f"{sum(reads) / len(reads) / 1e9:.3f} GB + written "
                                         f"{(sum(fulls) - sum(reads)) / len(fulls) / 1e9:.3f} GB per launch")
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child runs that measure roofline.traffic")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=0, help="wavefronts per GPU per step (default: 32 at 4096^2 = 8 GiB of fields + 4 GiB of PSFs of the 288 GB)")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the 2048^2 / 1024^2 entries")
    ap.add_argument("--allow-tcp", action="store_true",
                    help="with --gpus N > 1: accept the TCP transport when RCCL does not come up on every rank (default: exit 3)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # one chain step and nothing else (measure_traffic)
    args = ap.parse_args()
    if args.batch <= 0:
        # 8 -> 32 wavefronts per step is +3 % (launch tails and host work amortised; 64: +0.3 % more): 217 -> 224 at 4096^2
        args.batch = max(32, 32 * (4096 // args.grid) ** 2)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world

    # forked CPU workers: before anything initialises the GPU in this process
    cpu_parallel = cpu_single = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_parallel = cpu_baseline_parallel(args.grid, args.batch)
    if world > 1 and rank == 0 and not args.no_cpu_baseline:
        # N > 1: the one-core baseline on rank 0 before any GPU call (the other ranks wait at the rendezvous)
        cpu_single = cpu_baseline(args.grid)
    # the PMC child runs too: no process is started from one that holds a GPU context
    traffic_result = None
    if world == 1 and not args.no_traffic:
        traffic_result = measure_traffic(args.grid, args.batch, args.precision)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain
    from paos_amd.dist import broadcast_work, shard_bounds, syn20_work

    comm = None
    if world > 1 or os.environ.get("PAOS_BENCH_FORCE_COMM") == "1":
        from paos_amd.comm import Comm

        if os.environ.get("PAOS_BENCH_REHEARSAL") == "1":
            # several ranks sharing ONE GPU over the TCP transport: rehearses the N > 1 control flow (broadcast,
            # shards, barrier, MAX reduction) on a single-GPU box; the numbers mean nothing
            local_rank = 0
            comm = Comm.from_env(transport="socket")
        else:
            comm = Comm.from_env(transport="rccl", timeout=900.0)  # one process per GPU, RCCL over xGMI
            if comm.transport != "rccl" and world > 1 and not args.allow_tcp:
                # the agreement is collective (paos_comm_init_rank): every rank sees the same transport and leaves here
                if rank == 0:
                    print("bench.py: --gpus %d was asked for but RCCL did not come up on every rank (the ranks agreed on the "
                          "TCP transport); pass --allow-tcp to measure anyway" % world, file=sys.stderr)
                comm.close()
                sys.exit(3)

    n, nb = args.grid, args.batch
    total = nb * world
    work = syn20_work(total, "wavelengths") if rank == 0 else None
    work = broadcast_work(work, comm)  # the ONE broadcast
    lo, hi = shard_bounds(total, rank, world)
    wavelengths = work["wavelengths"][lo:hi]
    chains = [syn20_chain(coefficients=c) for c in work["coefficients"][lo:hi]]

    dev = _lib.DeviceFields(n, nb, args.precision, device=local_rank)
    m = measure(dev, n, nb, args.precision, wavelengths, chains, args.steps, args.warmup, comm)
    if args.traffic_child:  # measure_traffic's child: one step, every dispatch of which is counted
        dev.close()
        return
    # who took part, as the communicator saw it: rank r's device ordinal, gathered over the data plane
    ranks_seen = [int(p[0]) for p in comm.allgather_scalars([float(local_rank)])] if comm is not None else [local_rank]
    esz = 16 if args.precision == "fp64" else 8
    frugal = n >= (1024 if args.precision == "fp64" else 2048)
    kernel_name = ("frugal_pass_kernel" if frugal else "fused_pass_kernel") + " (every FFT pass launch, rows and columns)"

    if rank == 0:
        # The north star's headline step: one ptp (fft2 -> H -> ifft2, wfo.py:462-472) on the whole batch, timed
        # alone with the pass timer (3 launches: rows | cols x2 fused | rows).
        from paos_amd.planner import PilotBeam

        blk = [PilotBeam(1.0, wl, n, 4).ptp(2.5) for wl in wavelengths]
        for _ in range(2):
            dev.ptp(blk)
        dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=64)
        for _ in range(5):
            dev.ptp(blk)
        _, tot = dev.profile_end()
        ptp_ms = tot / 5.0

        value = total * args.steps / m["elapsed"]
        # the same steps with the compiler's ptp identities switched off (every ptp of the reference as three passes)
        plain = None
        if world == 1:
            import paos_amd.passes as ppasses

            ppasses.PTP_ALGEBRA = False
            try:
                mp = measure(dev, n, nb, args.precision, wavelengths, chains, min(args.steps, 3), 1, None)
                plain = {"value": nb * min(args.steps, 3) / mp["elapsed"], "unit": "wavefronts/s",
                         "fused_passes_per_wavefront": mp["fused_passes"],
                         "what": "PAOS_PTP_ALGEBRA=0: consecutive ptp neither share a middle pass nor cancel"}
            finally:
                ppasses.PTP_ALGEBRA = True
        n_ptp, n_stw, n_wts = chain_fft_counts(wavelengths[0], n)
        ffts = 2 * n_ptp + n_stw + n_wts
        survey_bytes = (ffts * 4 * esz + 8) * n * n  # 2 passes x (read + write) per 2-D FFT + the 8 B/px PSF write
        # bytes the fused path really moves: every dispatch of one step under the FETCH_SIZE / WRITE_SIZE counters
        traffic = traffic_result[0] if traffic_result is not None else None
        moved_bytes = None
        if traffic is not None:
            moved_bytes = (sum(r + w for r, w in traffic["pass"]) +
                           sum(k["read"] + k["written"] for k in traffic["other"].values())) / nb
        per_gpu = value / world
        dtype = "c128 (f64)" if args.precision == "fp64" else "c64 (f32, f64 phase arguments)"
        out = {
            "metric": f"wavefronts/sec ({n}^2 {'c128' if esz == 16 else 'c64'}, 20-surface chain) + achieved HBM GB/s",
            "value": value,
            "unit": "wavefronts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * m["elapsed"] / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": f"SYN20 20-surface chain, {n}x{n} {args.precision}, wavelength sweep 1um*(1+k/512), "
                                   f"{nb} wavefronts/GPU/step, {ffts} 2-D FFTs per wavefront ({n_ptp} ptp, {n_stw} stw, "
                                   f"{n_wts} wts as the reference executes them; the pass compiler runs them as "
                                   f"{m['fused_passes']} fused passes: at each of the chain's five foci the reference steps "
                                   f"ptp(+d), ptp(-d) with d = 1.6 nm, which cancel algebraically (H(-d) H(d) = 1, "
                                   f"fft2(ifft2(X)) = X; same results to 1e-15, `without_ptp_algebra` gives the rate with every "
                                   f"ptp run on its own); the final |u|^2 of every wavefront is written to HBM (8 B/px) and stays "
                                   f"there, powers of the saved surfaces are reduced on the GPU; aperture line records are "
                                   f"kept per context and found again by later steps (same optics)",
                       "grid": n, "batch_per_gpu": nb, "parallelism": f"wavefront-sharded x{world}",
                       "transport": comm.transport if comm is not None else "none (single process)",
                       "ranks_seen": len(ranks_seen), "devices_seen": ranks_seen},
            "roofline": roofline_block(m, n, nb, esz, dev, kernel_name, args.steps, traffic),
            "chain_vs_survey_model": {
                "survey_model_bytes_per_wavefront": survey_bytes,
                "frac_of_hbm_peak_vs_survey_model": survey_bytes * per_gpu / 1e9 / HBM_PEAK_GBS,
                "bytes_moved_per_wavefront": moved_bytes,
                "frac_bytes_moved": moved_bytes * per_gpu / 1e9 / HBM_PEAK_GBS if moved_bytes is not None else None,
                "other_kernels": traffic["other"] if traffic is not None else None,
                "note": "SURVEY 8d prices the chain UNFUSED (two HBM passes per 2-D FFT); the fused path moves fewer "
                        "bytes, so the first figure is a speed-up in model units, not a roofline fraction.  "
                        "bytes_moved_per_wavefront = every dispatch of one step under the FETCH_SIZE (x2) / WRITE_SIZE counters "
                        "(pass kernels and the start / Zernike / aperture-record / reduction kernels in `other_kernels`, whose "
                        "ms are rocprofv3 kernel-trace durations), null when the counters were not collected"},
            "ptp_step": {"what": "one ptp over the batch: 2 2-D FFTs + H, 3 fused passes",
                         "ms_per_wavefront": ptp_ms / nb,
                         "frac_of_hbm_peak_vs_survey_model": 8 * esz * n * n * nb / (ptp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_bytes_moved": 6 * esz * n * n * nb / (ptp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "SURVEY 8d prices a ptp at 128 B/px (4 passes); the fused path moves 96 B/px"},
            "without_ptp_algebra": plain,
            "power_check": float(dev.norm2_fetch(m["res"][0][20]["power_ticket"])[0]),
            "build": dev.build_info(),
        }
        dev.close()
        if world == 1 and not args.no_extras and n == 4096:
            extra = {}
            for n2, nb2 in ((2048, 64), (1024, 256)):
                w2 = syn20_work(nb2, "wavelengths")
                ch2 = [syn20_chain(coefficients=c) for c in w2["coefficients"]]
                dev2 = _lib.DeviceFields(n2, nb2, args.precision)
                m2 = measure(dev2, n2, nb2, args.precision, w2["wavelengths"], ch2, max(args.steps, 5), max(args.warmup, 2))
                fr2 = n2 >= (1024 if args.precision == "fp64" else 2048)
                extra[f"{n2}^2"] = {
                    "value": nb2 * max(args.steps, 5) / m2["elapsed"], "unit": "wavefronts/s", "batch": nb2,
                    "ms_per_step": 1e3 * m2["elapsed"] / max(args.steps, 5),
                    "roofline": roofline_block(m2, n2, nb2, esz, dev2,
                                               ("frugal_pass_kernel" if fr2 else "fused_pass_kernel") + " (every FFT pass launch)",
                                               max(args.steps, 5))}
                dev2.close()
            out["extra"] = extra
        if traffic_result is not None:
            out["roofline"]["counters_note"] = traffic_result[1]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
            out["cpu_baseline_parallel"] = cpu_parallel
        elif cpu_single is not None:
            out["cpu_baseline"] = cpu_single
        print(json.dumps(out), flush=True)
    else:
        dev.close()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
