#!/usr/bin/env python3
"""Headline benchmark: wavefronts/s through the 20-surface SYN20 chain.

    python bench.py --gpus 1 --steps 20 --warmup 3           # 4096^2 complex128 (default)
    python bench.py --gpus N --steps K --warmup W            # starts its own N ranks (one process per GPU, no torch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # ... or joins the ranks a launcher started

A "step" propagates one batch of ``--batch`` wavefronts per GPU through all 20 surfaces on the HIP path; fields are
created and stay in HBM.  The sweep is WALKED: step g (warm-up steps count) runs the wavelengths
lambda_k = 1 um (1 + k/512), k = g * batch * N ... (g + 1) * batch * N - 1 (mod 512), so what the library keeps
between batches (aperture line records, the shared start field) sees what a real 512-wavelength sweep gives it;
`sweep` in the line reports the records found / rendered per step.  With N GPUs every rank gets its own contiguous
block of each step's wavelengths (weak scaling) after ONE broadcast of the packed work description from rank 0
(paos_comm_bcast_blob: RCCL over xGMI, driven from libpaoship.so -- torch is never imported).  Without a launcher's
WORLD_SIZE in the environment `--gpus N` starts the N ranks itself, BEFORE anything touches a GPU in the parent, which
only relays rank 0's line and the exit codes.  With N > 1 the run FAILS (exit 3) when the ranks could not agree on
RCCL, unless --allow-tcp is given -- and still prints a JSON line then (`"value": null`, the transport, the ranks
seen and every rank's bring-up note).  Rank 0 prints ONE JSON line of at most 4 KB as the last line of stdout
(`contract_line`: the contract's keys, `roofline` for the dominant class of pass launch on the roofline that binds it,
`cpu_baseline`) and writes everything else to bench_detail.json (`--detail`), where the record holds these entries:

  roofline           the dominant kernel (the fused FFT pass): every pass launch of the timed region is bracketed by
                     HIP events on the context's stream (paos_profile_end_launches: time and class of each launch);
                     `achieved` = the bytes the library's pruning plan has those launches load + store (live lines x
                     (loaded + stored positions) x element size: paos_profile_planned_bytes -- for a launch that skips
                     nothing: one read and one write of every element of the batch) / their summed duration; `dense` =
                     the same kernel with the pruning off for one step (every launch moves the whole batch: what rounds
                     1-3 reported as `achieved`); `classes` lists every class of launch (skipping tiles of dead lines /
                     loads of dead positions / stores nobody reads, storing the PSF, running two or three passes of a
                     row / column chain) with its mean time, planned bytes and -- from the counters -- the bytes it
                     really moved; `all_launches` sums them over one step; `copy_yardstick` is measured in this run
                     (paos_copy_yardstick); `traffic` = HBM bytes per launch from FETCH_SIZE / WRITE_SIZE, collected
                     by two child runs of this script under rocprofv3 --pmc (measure_traffic; N = 1 only,
                     --no-traffic skips it).
  chain_vs_survey_model / ptp_step   the whole chain and one ptp priced with SURVEY 8d's UNFUSED byte model next to
                     the bytes the launches of one step really moved under the counters.
  without_ptp_algebra  the rate over the SAME number of steps with the pass compiler's ptp identities switched off
                     (PAOS_PTP_ALGEBRA=0: 44 passes per wavefront instead of 24).
  operator_by_operator  the rate over the same steps with the pass compiler of rounds 2-3 (PAOS_SEPARABLE=0: every operator's
                     transforms glued to its neighbours' instead of row factors first, column factors second).
  same_wavelengths_every_step  the rate when every step repeats the first block of the sweep (how rounds 1-3 quoted
                     the headline; the walked sweep costs ~3 %: records rendered, PSF zeros rewritten).
  extra              2048^2 / 1024^2 (the north star's sweep); fp32_4096 (SYN20 in fp32 mode with its roofline);
                     dense (SYN20 behind a white-noise grid-sag screen: rough fields everywhere, per-class times);
                     configs (BASELINE.json configs 3-5 on one GPU: AIRS-CH0 64 wavelengths @2048^2, FGS1 256
                     Monte-Carlo draws @2048^2 with on-device rEE90, Excite_TEL @4096^2 fp64 and fp32).
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
SWEEP = 512            # wavelengths of the sweep the steps walk through


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


# ---- CPU baseline (the ONLY place this script touches oracle/) ---------------------------------------------------
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


# ---- the timed region ---------------------------------------------------------------------------------------------
def sweep_block(step, per_step, lo, hi):
    """Sweep indices k of this rank's wavefronts in global step ``step`` (``per_step`` wavefronts over all ranks)."""
    return [(step * per_step + i) % SWEEP for i in range(lo, hi)]


def measure(dev, n, precision, wavelengths_of, chains, steps, warmup, comm=None, first_step=0, timer=True, keep_last=False):
    """Timed region of the contract: W untimed steps, then exactly K steps between barriers; every pass launch is
    event-timed.  ``wavelengths_of(g)`` lists this rank's wavelengths of global step g (warm-up steps are
    g = first_step ... first_step + W - 1).  Returns a dict of raw numbers."""
    from paos_amd import _lib
    from paos_amd.run import run_batch

    stats = {}

    def step(g):
        return run_batch(1.0, wavelengths_of(g), n, 4, ON_AXIS, chains, precision=precision, outputs=(), dev=dev,
                         sync=False, stats=stats, keep_psf=True)

    def barrier():
        dev.sync()
        if comm is not None:
            comm.barrier()

    def release(results):  # power tickets of a step nobody will read (no synchronisation)
        if results is not None:
            for t in {rec["power_ticket"] for r in results for rec in r.values() if "power_ticket" in rec}:
                dev.norm2_release(t)

    has_sets = hasattr(dev, "record_set_stats")
    res = None
    g = first_step
    # the launch timer's events are created here, in front of the warm-up: creating them behind it left the GPU idle for
    # the better part of a second and the first timed steps paid for it (20 steps measured 1.7 % under 400: round 5)
    max_launches = 64 * (steps + 1)
    if timer:
        dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=max_launches)
    for _ in range(warmup):
        release(res)
        res = step(g)
        g += 1
    barrier()
    if timer:
        dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=max_launches)  # (re-armed: the events exist)
    per_step_passes, per_step_sets, per_step_launches = [], [], []
    launched = 0
    seen = dev.record_set_stats() if has_sets else (0, 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        release(res)
        res = step(g)
        g += 1
        per_step_passes.append(stats.get("fused_passes"))
        if timer and hasattr(dev, "profile_planned_bytes"):
            # launches timed so far (a launch may run two passes of the program: frugal_pass.h, LONG builds)
            now_launched = int(dev.profile_planned_bytes().size)
            per_step_launches.append(now_launched - launched)
            launched = now_launched
        if has_sets:
            now = dev.record_set_stats()
            per_step_sets.append((now[0] - seen[0], now[1] - seen[1]))
            seen = now
    barrier()
    elapsed = time.perf_counter() - t0
    if not keep_last:  # the ticket ring is a ring: a ticket left outstanding blocks its slot when the ring comes round
        release(res)
        res = None
    import numpy as np

    if timer:
        planned = dev.profile_planned_bytes() if hasattr(dev, "profile_planned_bytes") else None
        lines = dev.profile_line_transforms() if hasattr(dev, "profile_line_transforms") else None
        ms, tags = dev.profile_end_launches()
        if planned is None or planned.size != ms.size:
            planned = np.zeros(ms.size)
        if lines is None or lines.size != ms.size:
            lines = np.zeros(ms.size)
    else:
        ms, tags, planned, lines = np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros(0), np.zeros(0)
    if comm is not None:
        elapsed = comm.max(elapsed)
    full = tags == 0
    return {"elapsed": elapsed, "launches": int(ms.size), "kern_ms": float(ms.sum()), "pruned": int((~full).sum()),
            "pruned_ms": float(ms[~full].sum()), "launch_ms": ms, "launch_tags": tags, "launch_bytes": planned,
            "launch_lines": lines,
            "fused_passes": stats.get("fused_passes"), "per_step_passes": per_step_passes,
            "per_step_launches": per_step_launches,
            "per_step_sets": per_step_sets, "first_timed_step": first_step + warmup, "res": res}


def counted_step_from(kernel_names):
    """Index of the first dispatch of the LAST chain step in a list of kernel names in dispatch order.  A step opens with the
    kernel that writes the start field -- behind the start's power kernel and its final reduction when the power sums are
    not found from the step before (round 5: they are kept across identical starts)."""
    writes = [i for i, name in enumerate(kernel_names) if "start_write_kernel" in name]
    first = writes[-1] if writes else 0
    if first >= 2 and "start_power_kernel" in kernel_names[first - 2] and "norm2_final_kernel" in kernel_names[first - 1]:
        first -= 2
    return first


def measure_traffic(grid, batch, precision):
    """HBM bytes of every kernel of ONE chain step from the PMC counters, collected the way MI355X_MICROARCH.md
    prescribes: this script once under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and once under `... --pmc WRITE_SIZE`
    (separate passes, as child processes running one warm-up step (g = 0) and one counted step (g = 1) and nothing
    else: --traffic-child), FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled (gfx950 reports half of the bytes of
    wide coalesced reads; the factor is calibrated for 16 B per lane, which is what the pass kernels issue -- for the
    8-byte accesses of the reduction and PSF kernels it is an assumption).  Dispatches are matched between the two
    runs by dispatch order.
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
    # (round 4: a third child counts the vector instructions of every dispatch -- the pass launches of the separable
    # programs are bound by their issue rate, not by bytes: roofline.issue)
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
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
                if counter == "SQ_INSTS_VALU":
                    continue  # (the byte counters stand on their own)
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
            # the child runs a warm-up step first: the counted step is the steady state the timed region measures.
            # A step opens with the kernel that writes the start field -- behind the start's power kernel and its final
            # reduction when the sums are not found from the step before (round 5): keep what follows the last opening.
            series[counter] = seq[counted_step_from([name for name, _ in seq]):]
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
                for _, name, dt in rows[counted_step_from([r[1] for r in rows]):]:  # the counted step, like the counters
                    trace_ms[name] = trace_ms.get(name, 0.0) + dt
        except Exception as exc:  # noqa: BLE001 -- the bench line must come out whatever the profiler does
            return None, f"rocprofv3 --pmc {counter}: {type(exc).__name__}: {exc}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    f, w = series["FETCH_SIZE"], series["WRITE_SIZE"]
    if not f or len(f) != len(w) or any(a[0] != b[0] for a, b in zip(f, w)):
        return None, f"dispatches seen: {len(f)} (FETCH_SIZE run) vs {len(w)} (WRITE_SIZE run), or in another order"
    out = {"pass": [], "other": {}, "pass_valu": None}
    v = series.get("SQ_INSTS_VALU")
    if v and len(v) == len(f) and all(a[0] == b[0] for a, b in zip(f, v)):
        out["pass_valu"] = [val for name, val in v if "_pass_kernel" in name]
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
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU on one chain step of this workload (global "
            "step 1 of the walked sweep) behind a warm-up step (three child runs, one counter each; byte counters in KiB, "
            "FETCH_SIZE x2 per MI355X_MICROARCH.md), every dispatch of that step kept")
    return out, note


CLASS_NAMES = {0: "full", 1: "skips tiles of dead lines", 2: "skips loads of dead positions", 4: "skips stores nobody reads",
               8: "stores the PSF instead of the field", 16: "runs two passes of a row / column chain",
               32: "runs three passes of a row / column chain"}


def class_name(tag):
    return " + ".join(CLASS_NAMES[b] for b in (1, 2, 4, 8, 16, 32) if tag & b) if tag else CLASS_NAMES[0]


def roofline_block(m, n, nb, esz, dev, kernel_name, steps, traffic=None, traffic_step=1, dense=None):
    """The pass kernel against the HBM roofline, over EVERY pass launch of the timed steps:
    `achieved` = the launches' algorithmic bytes / their HIP-event time, where a launch's algorithmic bytes are what the
    library's pruning plan has it load and store (live lines x (loaded + stored positions) x element size, summed over
    the batch: paos_profile_planned_bytes) -- for a launch that skips nothing that is the whole batch read and written
    once.  `traffic`: what the FETCH_SIZE / WRITE_SIZE counters saw per launch (``traffic``: measure_traffic, the launches
    of global step ``traffic_step``).  `dense` (``dense``: a measure() of one step with the pruning switched off): the
    same kernel when every line is alive -- the figure rounds 1-3 reported as `achieved`.  `classes`: per class of launch
    the mean event time, planned and counted bytes."""
    import numpy as np

    pass_bytes = 2 * esz * n * n * nb  # one pass over the batch: every element read + written once
    ms, tags, planned = m["launch_ms"], m["launch_tags"], m["launch_bytes"]
    known = ms.size > 0 and planned.size == ms.size and float(planned.sum()) > 0.0
    achieved = float(planned.sum()) / (float(ms.sum()) * 1e-3) / 1e9 if known else 0.0
    y_ms, y_bytes = dev.copy_yardstick(10)
    block = {
        "bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
        "traffic_note": "not measured in this run (--no-traffic, an `extra` entry, or N > 1); per-kernel FETCH_SIZE / WRITE_SIZE "
                        "of the default command under rocprofv3: profiles/r05_pmc_hbm_traffic_bench.txt",
        "launches": int(ms.size), "avg_launch_ms": float(ms.mean()) if ms.size else 0.0,
        "algorithmic_bytes_per_launch": float(planned.mean()) if known else None,
        "full_pass_bytes": pass_bytes,
        "what": "every pass launch of the timed steps: bytes the pruning plan has them load + store (their algorithmic bytes) "
                "over their HIP-event time.  Round 4: between two apertures the row factors of every operator run on the "
                "live rows and the column factors on the wanted columns only, so a launch moves about an eighth of the "
                "batch and is bound by the latency chain / fp64 issue of its butterflies, not by HBM (profiles/r05_fftbench_fused_variants.txt); "
                "`dense` is the same kernel with every line alive",
        "fused_passes_per_wavefront": m["fused_passes"],
        "launches_per_step": float(ms.size) / steps if steps else None,
        "copy_yardstick": {"ms_per_launch": y_ms, "GBps": y_bytes / (y_ms * 1e-3) / 1e9,
                           "what": "measured in this run (paos_copy_yardstick): in-place copy of the same batch buffer, "
                                   "16 B per lane, unit stride, no transform"},
    }
    if dense is not None and dense["launch_ms"].size:
        dms = dense["launch_ms"]
        full = dense["launch_tags"] == 0
        if full.any():
            block["dense"] = {"launches": int(full.sum()), "avg_launch_ms": float(dms[full].mean()),
                              "achieved": pass_bytes / (float(dms[full].mean()) * 1e-3) / 1e9,
                              "frac": pass_bytes / (float(dms[full].mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_launch": pass_bytes,
                              "what": "one step of the same chain with the pruning switched off (paos_ctx_set_pruning 0), "
                                      "measured in this run: every launch reads and writes the whole batch once"}
    counts = m.get("per_step_launches") or [c for c in m["per_step_passes"] if c is not None]
    if not ms.size or len(counts) != steps or sum(counts) != ms.size:
        return block  # (a generic-kernel pass that is not timed: the launches cannot be cut into steps)
    bounds = np.concatenate([[0], np.cumsum(counts)])
    classes = {}
    for t in sorted(set(int(x) for x in tags)):
        sel = tags == t
        classes[class_name(t)] = {"launches_per_step": float(sel.sum()) / steps, "avg_launch_ms": float(ms[sel].mean()),
                                  "bytes_planned": float(planned[sel].mean()) if known else None, "bytes_measured": None}
    block["classes"] = classes
    block["all_launches"] = {"launches_per_step": float(ms.size) / steps, "ms": float(ms.sum()) / steps,
                             "bytes_planned": float(planned.sum()) / steps if known else None, "bytes_measured": None,
                             "frac": None,
                             "what": "every pass launch of one step: HIP-event time (mean over the timed steps), planned bytes "
                                     "and, when the counters were collected, the HBM bytes they saw for the launches of one step"}
    # the counted step of the traffic children is global step `traffic_step`: its launches carry the tags of the same
    # step of this run when it lies in the timed region, of the first timed step otherwise (same chain, other wavelengths)
    s = traffic_step - m["first_timed_step"]
    s = s if 0 <= s < steps else 0
    step_tags = tags[bounds[s]:bounds[s + 1]]
    if traffic is not None and len(traffic["pass"]) == len(step_tags):
        per_class = {}
        for tag, (rd, wr) in zip(step_tags, traffic["pass"]):
            per_class.setdefault(class_name(int(tag)), []).append(rd + wr)
        for name, vals in per_class.items():
            rec = classes[name]
            rec["bytes_measured"] = sum(vals) / len(vals)
            rec["GBps_measured"] = rec["bytes_measured"] / (rec["avg_launch_ms"] * 1e-3) / 1e9
            rec["frac_measured"] = rec["GBps_measured"] / HBM_PEAK_GBS
        total = sum(r + w for r, w in traffic["pass"])
        reads = sum(r for r, w in traffic["pass"])
        block["all_launches"]["bytes_measured"] = total
        block["all_launches"]["frac"] = total / (block["all_launches"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        block["traffic"] = total / len(step_tags)
        if known:
            block["traffic_over_algorithmic"] = total / (float(planned[bounds[s]:bounds[s + 1]].sum()) or 1.0)
        block["traffic_note"] = (f"mean over the {len(step_tags)} pass launches of one step: read {reads / len(step_tags) / 1e9:.3f} GB "
                                 f"+ written {(total - reads) / len(step_tags) / 1e9:.3f} GB per launch")
        valu = traffic.get("pass_valu")
        if valu and len(valu) == len(step_tags):
            # vector instructions of the counted step's launches against the issue rate of the chip: a wave64 fp64
            # instruction holds its SIMD for 4 cycles (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak clock)
            insts = float(sum(valu))
            step_ms = float(ms.sum()) / steps  # (the counted step is another run's: the mean pass time of a timed step of this one)
            cycles = insts * 4.0 / (256 * 4)
            block["issue"] = {
                "valu_instructions_per_step": insts, "valu_instructions_per_launch": insts / len(step_tags),
                "pass_launch_ms_per_step": step_ms,
                "frac_of_issue_peak": cycles / (step_ms * 1e-3 * 2.4e9),
                "what": "SQ_INSTS_VALU of the pass launches of the counted step (a third rocprofv3 --pmc child) x 4 cycles per "
                        "wave instruction / 1024 SIMDs, over the mean HIP-event time of a step's pass launches in this run at the 2.4 GHz "
                        "peak clock: the fraction of the vector issue slots the launches fill.  About three quarters of the "
                        "instructions are fp64 (4 cycles); under the fused launches the chip holds 2.0-2.36 GHz (s_memtime / "
                        "s_memrealtime, profiles/r05_fftbench_fused_variants.txt): the launches are a latency chain per "
                        "workgroup at four waves per SIMD, not short of clock"}
    return block


FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X: 256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2 flop x 2.4 GHz (half the fp32 vector peak of MI355X_MICROARCH.md)
FP32_VECTOR_PEAK_TFLOPS = 157.3
CONTRACT_LINE_MAX = 4096         # bytes: the driver keeps an 8 KB tail of stdout; round 4's 25 KB line did not parse


def dominant_launch(m, n, precision, steps, traffic=None, traffic_step=1):
    """The class of pass launch the timed region spends most of its time in, priced on BOTH of its candidate rooflines:
    `hbm_frac` = the bytes the pruning plan has one such launch load + store / its mean HIP-event time / 8 TB/s;
    `flop_frac` = its 1-D line transforms (paos_profile_line_transforms) x 5 N log2 N nominal flops / the same time / the
    fp64 (fp32 mode: fp32) vector peak; `issue_frac` = SQ_INSTS_VALU of the class's launches in the counted step x 4 cycles
    / 1024 SIMDs / time at the 2.4 GHz peak clock; `traffic` = FETCH_SIZE x 2 + WRITE_SIZE per launch of the class.  The
    bound named is the resource with the larger fraction.  Returns None when nothing was timed."""
    import math

    import numpy as np

    ms, tags, planned = m["launch_ms"], m["launch_tags"], m["launch_bytes"]
    lines = m.get("launch_lines")
    if not ms.size:
        return None
    by_time = {}
    for t in set(int(x) for x in tags):
        by_time[t] = float(ms[tags == t].sum())
    tag = max(by_time, key=by_time.get)
    sel = tags == tag
    avg_ms = float(ms[sel].mean())
    alg_bytes = float(planned[sel].mean()) if planned.size == ms.size else 0.0
    peak_tf = FP64_VECTOR_PEAK_TFLOPS if precision == "fp64" else FP32_VECTOR_PEAK_TFLOPS
    flops = float(lines[sel].mean()) * 5.0 * n * math.log2(n) if lines is not None and lines.size == ms.size else 0.0
    out = {"tag": tag, "class": class_name(tag), "launches": int(sel.sum()), "avg_launch_ms": avg_ms,
           "share_of_pass_time": by_time[tag] / float(ms.sum()),
           "algorithmic_bytes_per_launch": alg_bytes or None,
           "line_transforms_per_launch": float(lines[sel].mean()) if flops else None,
           "hbm_GBps": alg_bytes / (avg_ms * 1e-3) / 1e9 if alg_bytes else None,
           "hbm_frac": alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if alg_bytes else None,
           "TFLOPs": flops / (avg_ms * 1e-3) / 1e12 if flops else None,
           "flop_frac": flops / (avg_ms * 1e-3) / 1e12 / peak_tf if flops else None,
           "flop_peak_TFLOPs": peak_tf, "traffic": None, "traffic_over_planned": None, "issue_frac": None,
           "valu_instructions_per_launch": None}
    counts = m.get("per_step_launches") or []
    if traffic is not None and len(counts) == steps and sum(counts) == ms.size:
        bounds = np.concatenate([[0], np.cumsum(counts)])
        sidx = traffic_step - m["first_timed_step"]
        sidx = sidx if 0 <= sidx < steps else 0
        step_tags = tags[bounds[sidx]:bounds[sidx + 1]]
        if len(traffic["pass"]) == len(step_tags):
            moved = [r + w for t, (r, w) in zip(step_tags, traffic["pass"]) if int(t) == tag]
            if moved:
                out["traffic"] = sum(moved) / len(moved)
                if alg_bytes:
                    out["traffic_over_planned"] = out["traffic"] / alg_bytes
            valu = traffic.get("pass_valu")
            if valu and len(valu) == len(step_tags):
                v = [x for t, x in zip(step_tags, valu) if int(t) == tag]
                if v:
                    out["valu_instructions_per_launch"] = sum(v) / len(v)
                    out["issue_frac"] = (out["valu_instructions_per_launch"] * 4.0 / 1024.0) / (avg_ms * 1e-3 * 2.4e9)
    cands = {"hbm": max(out["hbm_frac"] or 0.0, (out["traffic"] or 0.0) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
             "valu": max(out["flop_frac"] or 0.0, out["issue_frac"] or 0.0)}
    out["bound"] = "hbm" if cands["hbm"] >= cands["valu"] else ("valu_fp64" if precision == "fp64" else "valu_fp32")
    return out


def _r(x, digits=4):
    """A float rounded to ``digits`` significant digits (the contract line is read by people and a parser: 17 digits of a
    time help neither); None and non-finite values become None -- the line is strict JSON."""
    import math

    if x is None:
        return None
    if isinstance(x, (int, str, bool)):
        return x
    x = float(x)
    if not math.isfinite(x):
        return None
    if x == 0.0:
        return 0.0
    return round(x, digits - 1 - int(math.floor(math.log10(abs(x)))))


def contract_line(full):
    """The ONE line rank 0 prints last: the contract's keys and nothing else, <= CONTRACT_LINE_MAX bytes, strict JSON
    (``full`` is the detailed record that goes to bench_detail.json).  `roofline` describes the dominant class of pass
    launch on the roofline that binds it (`bound`; achieved / peak / unit / frac follow it) and carries the other
    figures a reader needs beside it: the same launch's fraction of the HBM roofline, its issue-slot fraction, counter
    traffic over planned bytes, the HBM fraction of all pass launches, of the dense (un-pruned) launches measured in the
    same run, of one dense `ptp` over the batch (the north star's "FFT-propagate step") and of the copy yardstick."""
    roof, dom = full.get("roofline") or {}, full.get("dominant_launch") or {}
    cfg = full.get("config") or {}
    hbm_bound = dom.get("bound", "hbm") == "hbm"
    if dom:
        achieved = dom.get("hbm_GBps") if hbm_bound else dom.get("TFLOPs")
        peak = HBM_PEAK_GBS if hbm_bound else dom.get("flop_peak_TFLOPs")
        unit = "GB/s" if hbm_bound else "TFLOP/s"
        frac = dom.get("hbm_frac") if hbm_bound else dom.get("flop_frac")
    else:  # (nothing was timed launch by launch: a generic-kernel grid)
        achieved, peak, unit, frac = roof.get("achieved"), HBM_PEAK_GBS, "GB/s", roof.get("frac")
    ptp = full.get("ptp_step") or {}
    yard = (roof.get("copy_yardstick") or {}).get("GBps")
    cpu = full.get("cpu_baseline")
    line = {
        "metric": full.get("metric"), "value": _r(full.get("value"), 6), "unit": full.get("unit"),
        "n_gpus": full.get("n_gpus"), "steps": full.get("steps"), "warmup": full.get("warmup"),
        "ms_per_step": _r(full.get("ms_per_step"), 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": full.get("dtype"), "data": "synthetic",
        "config": {"workload": str(cfg.get("workload", ""))[:300], "grid": cfg.get("grid"),
                   "batch_per_gpu": cfg.get("batch_per_gpu"), "parallelism": cfg.get("parallelism"),
                   "transport": cfg.get("transport"), "ranks_seen": cfg.get("ranks_seen"),
                   "devices_seen": cfg.get("devices_seen"), "launcher": cfg.get("launcher")},
        "roofline": {
            "bound": dom.get("bound", roof.get("bound", "hbm")),
            "kernel": (str(roof.get("kernel", "")).split(" (")[0] + ": " + str(dom.get("class", "every pass launch")))[:160],
            "achieved": _r(achieved), "peak": peak, "unit": unit, "frac": _r(frac),
            "traffic": _r(dom.get("traffic") if dom else roof.get("traffic")),
            "algorithmic_bytes_per_launch": _r(dom.get("algorithmic_bytes_per_launch") if dom else roof.get("algorithmic_bytes_per_launch")),
            "avg_launch_ms": _r(dom.get("avg_launch_ms") if dom else roof.get("avg_launch_ms")),
            "launches": dom.get("launches") if dom else roof.get("launches"),
            "share_of_pass_time": _r(dom.get("share_of_pass_time")),
            "hbm_frac": _r(dom.get("hbm_frac")), "flop_frac": _r(dom.get("flop_frac")), "issue_frac": _r(dom.get("issue_frac")),
            "traffic_over_planned": _r(dom.get("traffic_over_planned")),
            "all_launches_hbm_frac": _r(roof.get("frac")),
            "dense_frac": _r((roof.get("dense") or {}).get("frac")),
            "ptp_step_frac": _r(ptp.get("frac_bytes_moved")),
            "copy_yardstick_frac": _r(yard / HBM_PEAK_GBS if yard else None),
            "launches_per_step": _r(roof.get("launches_per_step")),
        },
        "cpu_baseline": ({"value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"),
                          "kind": cpu.get("kind"), "sample": str(cpu.get("sample", ""))[:260]} if cpu else None),
        "detail": full.get("detail_file"),
    }
    extra = full.get("extra") or {}
    if extra:  # the side measurements in one number each (wavefronts/s; everything else about them is in the detail file)
        brief = {k: _r(v.get("value")) for k, v in extra.items() if isinstance(v, dict) and v.get("unit") == "wavefronts/s"}
        psd = extra.get("psd_screen") or {}
        if psd.get("device_ms") is not None:
            brief["psd_screen_ms"] = {"device": _r(psd.get("device_ms")), "host": _r(psd.get("host_ms"))}
        line["extra_wavefronts_per_s"] = brief
    if full.get("error"):
        line["error"] = str(full["error"])[:800]
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > CONTRACT_LINE_MAX:  # cannot happen with the caps above; never print an unparseable line
        line["config"]["workload"] = line["config"]["workload"][:80]
        if line["cpu_baseline"]:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:80]
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) <= CONTRACT_LINE_MAX, len(text)
    return text


def _jsonable(x):
    """The detailed record as strict JSON: NumPy scalars / arrays to Python, non-finite floats to None."""
    import math

    import numpy as np

    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, np.ndarray):
        return _jsonable(x.tolist())
    if isinstance(x, (np.floating, float)):
        return float(x) if math.isfinite(float(x)) else None
    if isinstance(x, (np.integer,)):
        return int(x)
    return x


def write_detail(full, path):
    """Everything beyond the contract line (classes, sweep, other kernels, side measurements, `extra`) as one JSON file;
    returns the path written, or None (the line must come out whatever the file system says)."""
    try:
        with open(path, "w") as fh:
            json.dump(_jsonable(full), fh, allow_nan=False, indent=1)
        return path
    except Exception as exc:  # noqa: BLE001
        print(f"bench.py: could not write {path}: {exc}", file=sys.stderr)
        return None


def sweep_report(m):
    """What the walked sweep did to the context's kept aperture line records, per timed step."""
    sets = m["per_step_sets"]
    return {"walked": True, "wavelengths_in_sweep": SWEEP,
            "record_sets_found_per_step": [a for a, _ in sets], "record_sets_rendered_per_step": [b for _, b in sets],
            "what": "step g runs the next block of the 512-wavelength sweep (mod 512), so what the context keeps between "
                    "batches is hit or missed as in a real sweep: aperture line records found in / rendered into its kept "
                    "sets, per timed step (paos_record_set_stats)"}


# ---- self-launch: `bench.py --gpus N` with no launcher ------------------------------------------------------------
def self_launch(n_ranks, argv, child=None, timeout=3000.0, env_extra=None, grace=30.0):
    """Start ``n_ranks`` copies of this script (``child``: another command, for tests) as one job and relay rank 0's
    stdout.  The parent never touches a GPU -- no HIP call, no library load -- so its children are ordinary fresh
    processes (nothing is exec'ed from a process that holds a GPU context).  Every rank gets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT and a job key of its own (PAOS_COMM_KEY), like a launcher would set them.
    Returns the exit code: 0 when every rank returned 0, else the first non-zero one; when a rank fails the others are
    given ``grace`` seconds to finish and are then terminated (by PID)."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    key = f"bench_{os.getpid()}_{time.time_ns()}"
    one_gpu = os.environ.get("PAOS_BENCH_REHEARSAL") == "1"
    cmd = list(child) if child else [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if one_gpu else r), WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PAOS_COMM_KEY=key, PAOS_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_extra or {})
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + timeout
    first_bad, grace_until = None, None
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0) and first_bad is None:
                first_bad, grace_until = (r, p.returncode), now + grace
        if now > deadline or (grace_until is not None and now > grace_until):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            if first_bad is None:
                first_bad = (-1, 124)
                print(f"bench.py: the ranks did not finish within {timeout:.0f} s", file=sys.stderr)
            break
        time.sleep(0.05)
    reader.join(timeout=10.0)
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode(errors="replace"))
        sys.stdout.flush()
    codes = [p.returncode for p in procs]
    if first_bad is None:
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        first_bad = bad[0] if bad else None
    if first_bad is not None:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return first_bad[1] if first_bad[1] not in (None, 0) else 1
    return 0


def no_rccl_line(args, comm, world, local_rank):
    """The record of a scaling run that could not use RCCL (exit 3): how far the bring-up got, per rank."""
    notes = comm.gather_text(comm.bringup_note or "")
    seen = [int(p[0]) for p in comm.allgather_scalars([float(local_rank)])]
    if comm.rank != 0:
        return
    print(contract_line({
        "metric": f"wavefronts/sec ({args.grid}^2 {'c128' if args.precision == 'fp64' else 'c64'}, 20-surface chain) + achieved HBM GB/s",
        "value": None, "unit": "wavefronts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
        "dtype": "c128 (f64)" if args.precision == "fp64" else "c64 (f32, f64 phase arguments)",
        "error": "RCCL did not come up on every rank (exit 3, nothing measured; --allow-tcp accepts the TCP transport); "
                 "bring-up notes per rank: " + "; ".join(f"{r}: {t[:120]}" for r, t in enumerate(notes) if t),
        "config": {"workload": "not run", "grid": args.grid, "batch_per_gpu": args.batch, "parallelism": f"wavefront-sharded x{world}",
                   "transport": comm.transport, "ranks_seen": len(seen), "devices_seen": seen}}), flush=True)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child runs that measure roofline.traffic")
    # (defaults: a step is 22 ms since the separable programs; the first two or three steps of a context still allocate the
    # aperture record sets and the phase tables lazily, which `--warmup 1` left inside a five-step timed region)
    # (defaults, round 5: 100 timed steps = 1.4 s -- over 20 the edges of the timed region still weigh 0.5 %; the side
    # measurements below run min(steps, 20))
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=0, help="wavefronts per GPU per step (default: 32 at 4096^2 = 8 GiB of fields + 4 GiB of PSFs of the 288 GB)")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra entries (other grids, fp32, dense, BASELINE configs)")
    ap.add_argument("--allow-tcp", action="store_true",
                    help="with --gpus N > 1: accept the TCP transport when RCCL does not come up on every rank (default: exit 3)")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="self-launch: seconds the ranks may take")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # one chain step and nothing else (measure_traffic)
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the detailed record goes (classes of launch, sweep, other kernels, side measurements, extras); "
                         "stdout carries the contract line only")
    args = ap.parse_args(argv)
    if args.batch <= 0:
        # 8 -> 32 wavefronts per step is +3 % (launch tails and host work amortised; 64: +0.3 % more): 217 -> 224 at 4096^2
        args.batch = max(32, 32 * (4096 // args.grid) ** 2)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves, before anything here touches a GPU
        sys.exit(self_launch(args.gpus, sys.argv[1:] if argv is None else argv, timeout=args.launch_timeout))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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

    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength
    from paos_amd.dist import broadcast_work, shard_bounds, syn20_work

    comm = None
    if world > 1 or os.environ.get("PAOS_BENCH_FORCE_COMM") == "1":
        from paos_amd.comm import Comm  # (an RCCL Comm exports HSA_ENABLE_IPC_MODE_LEGACY=0 unless set: before the first HIP call)

        if os.environ.get("PAOS_BENCH_REHEARSAL") == "1":
            # several ranks sharing ONE GPU over the TCP transport: rehearses the N > 1 control flow (launch, broadcast,
            # shards, barrier, MAX reduction) on a single-GPU box; the numbers mean nothing
            local_rank = 0
            comm = Comm.from_env(transport="socket")
        else:
            comm = Comm.from_env(transport="rccl", timeout=900.0)  # one process per GPU, RCCL over xGMI
            if comm.transport != "rccl" and world > 1 and not args.allow_tcp:
                # the agreement is collective (paos_comm_init_rank): every rank sees the same transport and leaves here
                no_rccl_line(args, comm, world, local_rank)
                if rank == 0:
                    print("bench.py: --gpus %d was asked for but RCCL did not come up on every rank (the ranks agreed on the "
                          "TCP transport); pass --allow-tcp to measure anyway" % world, file=sys.stderr)
                comm.barrier()
                comm.close()
                sys.exit(3)

    n, nb = args.grid, args.batch
    total = nb * world
    # the ONE broadcast: the sweep's coefficient table (rank 0 -> everybody); wavelengths follow from the step number
    work = syn20_work(total, "wavelengths") if rank == 0 else None
    work = broadcast_work(work, comm)
    lo, hi = shard_bounds(total, rank, world)
    chains = [syn20_chain(coefficients=c) for c in work["coefficients"][lo:hi]]

    def wavelengths_of(g):
        return [syn20_wavelength(k) for k in sweep_block(g, total, lo, hi)]

    dev = _lib.DeviceFields(n, nb, args.precision, device=local_rank)
    m = measure(dev, n, args.precision, wavelengths_of, chains, args.steps, args.warmup, comm, keep_last=True)
    # the power behind the last surface of the last step (a sanity figure of the line); its tickets are given back here
    last = m.pop("res")
    power_check = float(dev.norm2_fetch(last[0][20]["power_ticket"])[0])
    for t in {rec["power_ticket"] for r in last for rec in r.values() if "power_ticket" in rec} - {last[0][20]["power_ticket"]}:
        dev.norm2_release(t)
    if args.traffic_child:  # measure_traffic's child: one step, every dispatch of which is counted
        dev.close()
        return
    # who took part, as the communicator saw it: rank r's device ordinal, gathered over the data plane
    ranks_seen = [int(p[0]) for p in comm.allgather_scalars([float(local_rank)])] if comm is not None else [local_rank]
    notes = comm.gather_text(comm.bringup_note or "") if comm is not None else []
    esz = 16 if args.precision == "fp64" else 8
    frugal = n >= (1024 if args.precision == "fp64" else 2048)
    kernel_name = ("frugal_pass_kernel" if frugal else "fused_pass_kernel") + " (every FFT pass launch, rows and columns)"

    if rank == 0:
        # The north star's headline step: one ptp (fft2 -> H -> ifft2, wfo.py:462-472) on the whole batch, timed
        # alone with the pass timer (3 launches: rows | cols x2 fused | rows).
        from paos_amd.planner import PilotBeam

        blk = [PilotBeam(1.0, wl, n, 4).ptp(2.5) for wl in wavelengths_of(0)]
        for _ in range(2):
            dev.ptp(blk)
        dev.profile_begin(_lib.KERNEL_PASS_ANY, max_launches=64)
        for _ in range(5):
            dev.ptp(blk)
        _, tot = dev.profile_end()
        ptp_ms = tot / 5.0

        value = total * args.steps / m["elapsed"]
        side_steps = max(1, min(args.steps, 20))  # the side measurements below (variants of the same steps)
        # the same steps with the compiler's ptp identities switched off (every ptp of the reference as three passes)
        plain = None
        if world == 1:
            import paos_amd.passes as ppasses

            ppasses.PTP_ALGEBRA = False
            try:
                mp = measure(dev, n, args.precision, wavelengths_of, chains, side_steps, 1, None, first_step=args.warmup - 1)
                plain = {"value": nb * side_steps / mp["elapsed"], "unit": "wavefronts/s", "steps": side_steps,
                         "fused_passes_per_wavefront": mp["fused_passes"],
                         "what": "PAOS_PTP_ALGEBRA=0 over the same steps of the walked sweep: consecutive ptp neither share a "
                                 "middle pass nor cancel, a wts and the stw that undoes it both run"}
            finally:
                ppasses.PTP_ALGEBRA = True
        # ... and operator by operator (rounds 2-3: every operator's transforms glued to its neighbours', all rows / columns
        # that are not known to be zero processed)
        by_operator = None
        if world == 1:
            import paos_amd.passes as ppasses

            ppasses.SEPARABLE = False
            try:
                mo = measure(dev, n, args.precision, wavelengths_of, chains, side_steps, 1, None, first_step=args.warmup - 1)
                by_operator = {"value": nb * side_steps / mo["elapsed"], "unit": "wavefronts/s", "steps": side_steps,
                               "fused_passes_per_wavefront": mo["fused_passes"],
                               "pass_launch_ms_per_step": float(mo["launch_ms"].sum()) / side_steps,
                               "what": "PAOS_SEPARABLE=0 over the same steps: the pass compiler of rounds 2-3 (a 2-D transform's "
                                       "second half glued to the next one's first; the column passes behind an aperture run on "
                                       "all 4096 columns)"}
            finally:
                ppasses.SEPARABLE = True
        # ... and one step with the pruning off: the pass kernel when every line is alive (roofline.dense)
        dense = None
        if hasattr(dev, "set_pruning"):
            dev.set_pruning(False)
            try:
                dense = measure(dev, n, args.precision, wavelengths_of, chains, 1, 1, None, first_step=args.warmup - 1)
            finally:
                dev.set_pruning(True)
        # ... and with the SAME wavelengths every step (rounds 1-3 measured this way): what the walk costs
        repeated = None
        if world == 1:
            mr = measure(dev, n, args.precision, lambda g: wavelengths_of(0), chains, side_steps, max(args.warmup, 1), None, timer=False)
            repeated = {"value": nb * side_steps / mr["elapsed"], "unit": "wavefronts/s", "steps": side_steps,
                        "what": "every step runs the first block of the sweep again (how rounds 1-3 quoted the headline): the "
                                "context's kept aperture records and PSF zeros are found every time"}
        n_ptp, n_stw, n_wts = chain_fft_counts(wavelengths_of(0)[0], n)
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
        passes_seen = sorted(set(c for c in m["per_step_passes"] if c is not None))
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
            "config": {"workload": f"SYN20 20-surface chain (SURVEY 8d: {ffts} 2-D FFTs per wavefront as the reference runs it), {n}x{n} "
                                   f"{args.precision}, walked 512-wavelength sweep, {nb} wavefronts/GPU/step, PSFs stay in HBM; "
                                   f"separable pass programs + ptp identities ({passes_seen} passes/wavefront, parity 1e-15..2e-14 vs oracle)",
                       "workload_notes": f"SYN20 20-surface chain, {n}x{n} {args.precision}, WALKED wavelength sweep 1um*(1+k/512): step g "
                                   f"runs k = g*{total} ... (mod 512), {nb} wavefronts/GPU/step, {ffts} 2-D FFTs per wavefront "
                                   f"({n_ptp} ptp, {n_stw} stw, {n_wts} wts as the reference executes them at the first "
                                   f"wavelength; the pass compiler runs a step as {passes_seen} fused passes: at each of the "
                                   f"chain's five foci the reference steps ptp(+d), ptp(-d) with d = 1.6 nm, which cancel "
                                   f"algebraically (H(-d) H(d) = 1, fft2(ifft2(X)) = X; same results to 1e-15, "
                                   f"`without_ptp_algebra` gives the rate with every ptp run on its own); since round 4 the row factors of every "
                                   f"operator between two apertures run first, on the live rows, then the column factors, on the "
                                   f"wanted columns, two or three consecutive passes of a chain per launch "
                                   f"(`roofline.launches_per_step`; `operator_by_operator` gives the rate of rounds 2-3's "
                                   f"order); the final |u|^2 of "
                                   f"every wavefront is written to HBM (8 B/px) and stays there, powers of the saved surfaces "
                                   f"are reduced on the GPU; aperture line records are kept per context and found again "
                                   f"where the next batch samples an aperture alike (`sweep`)",
                       "grid": n, "batch_per_gpu": nb, "parallelism": f"wavefront-sharded x{world}",
                       "transport": comm.transport if comm is not None else "none (single process)",
                       "launcher": "self (bench.py started the ranks)" if os.environ.get("PAOS_BENCH_SELF_LAUNCHED") == "1"
                                   else ("external (WORLD_SIZE from the environment)" if world > 1 else "none"),
                       "ranks_seen": len(ranks_seen), "devices_seen": ranks_seen,
                       "bringup_notes": {str(r): t for r, t in enumerate(notes) if t}},
            "sweep": sweep_report(m),
            "roofline": roofline_block(m, n, nb, esz, dev, kernel_name, args.steps, traffic, dense=dense),
            "dominant_launch": dominant_launch(m, n, args.precision, args.steps, traffic),
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
            "operator_by_operator": by_operator,
            "same_wavelengths_every_step": repeated,
            "power_check": power_check,
            "build": dev.build_info(),
        }
        dev.close()
        if world == 1 and not args.no_extras and n == 4096:
            from bench_extras import extras

            out["extra"] = extras(args, esz, measure, roofline_block, sweep_report)
        if traffic_result is not None:
            out["roofline"]["counters_note"] = traffic_result[1]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
            out["cpu_baseline_parallel"] = cpu_parallel
        elif cpu_single is not None:
            out["cpu_baseline"] = cpu_single
        # the detailed record to a file (and nothing but the contract line to stdout: <= 4 KB, strict JSON, last line)
        out["detail_file"] = os.path.basename(args.detail) if write_detail(out, args.detail) else None
        print(contract_line(out), flush=True)
    else:
        dev.close()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
