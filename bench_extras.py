"""The `extra` entries of bench.py's line (N = 1, default grid only): the same chain at other grids and in fp32 mode, a
rough-field variant, and BASELINE.json's configs 3-5 on one GPU.  Everything here runs on the HIP path through the same
`run_batch` the headline uses; nothing is timed with fields crossing PCIe unless an entry says so."""
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
LENS = os.path.join(ROOT, "data", "lens")
WFE = os.path.join(ROOT, "data", "wfe", "wfe_realization_SN20210914.csv")
ON_AXIS = {"us": 0.0, "ut": 0.0}


def _syn20(n2, nb2, precision, steps, warmup, esz, measure, roofline_block, sweep_report, chains=None, label=None):
    from paos_amd import _lib
    from paos_amd.chains import syn20_chain, syn20_wavelength

    chains = chains if chains is not None else [syn20_chain() for _ in range(nb2)]
    dev = _lib.DeviceFields(n2, nb2, precision)
    try:
        def wavelengths_of(g):
            return [syn20_wavelength((g * nb2 + i) % 512) for i in range(nb2)]

        m = measure(dev, n2, precision, wavelengths_of, chains, steps, warmup)
        frugal = n2 >= (1024 if precision == "fp64" else 2048)
        name = ("frugal_pass_kernel" if frugal else "fused_pass_kernel") + " (every FFT pass launch)"
        ms_step = 1e3 * m["elapsed"] / steps
        pass_ms = float(m["launch_ms"].sum()) / steps if m["launch_ms"].size else None
        return {"value": nb2 * steps / m["elapsed"], "unit": "wavefronts/s", "batch": nb2, "steps": steps,
                "ms_per_step": ms_step, "workload": label or f"SYN20, walked sweep, {n2}^2 {precision}",
                # (round 5: how much of a step's wall time the GPU spends in pass launches -- the host's share of a step at
                # 256 wavefronts was the bound at 1024^2 in round 4)
                "pass_launch_ms_per_step": pass_ms, "pass_share_of_step": pass_ms / ms_step if pass_ms else None,
                "sweep": sweep_report(m), "roofline": roofline_block(m, n2, nb2, esz, dev, name, steps)}
    finally:
        dev.close()


def dense_chain(n, rms_m=40.0e-9, seed=20260101):
    """SYN20 with a white-noise grid-sag screen (one map, `rms_m` metres rms, every item) on a surface of its own right
    behind the Zernike surface S02: the field that enters the relays is rough at every spatial frequency the grid
    carries, so every later plane is dense and noise-like -- what a chain with a PSD or measured-sag surface produces
    (wfo.py:656-871), and the data the fp64 pipe draws most power on."""
    from paos_amd.abcd import ABCD
    from paos_amd.chains import syn20_chain

    base = syn20_chain()
    sag = np.random.default_rng(seed).standard_normal((n, n)) * rms_m
    step = 4.0 / n  # pupil diameter 1 m x zoom 4 over n pixels: the map's samples are the grid's pixels
    out = {}
    for key, item in base.items():
        num = len(out) + 1
        out[num] = dict(item, num=num)
        if item["name"] == "Z1":
            num = len(out) + 1
            out[num] = {"num": num, "type": "Grid Sag", "name": "SCREEN", "is_stop": False, "save": False,
                        "grid_sag": sag, "nx": n, "ny": n, "delx": step, "dely": step, "xdec": 0, "ydec": 0,
                        "ABCDt": ABCD(thickness=0.0, curvature=0.0), "ABCDs": ABCD(thickness=0.0, curvature=0.0)}
    return out


def _dense(args, esz, measure, roofline_block):
    """SYN20 behind the screen at the headline grid, batch 8.  Round 5: the screen -- one array for every item, as a measured
    surface map is -- is turned into its map once (run.py: _sag_map_once) and crosses PCIe once (paos_phase_map_items keeps
    it on the device); rounds 2-4 rebuilt and uploaded 128 MiB per item and step (30 wavefronts/s end to end)."""
    from paos_amd import _lib
    from paos_amd.chains import syn20_wavelength

    n, nb, steps = args.grid, 8, 10
    chain = dense_chain(n)
    chains = [chain] * nb
    dev = _lib.DeviceFields(n, nb, args.precision)
    try:
        def wavelengths_of(g):
            return [syn20_wavelength((g * nb + i) % 512) for i in range(nb)]

        m = measure(dev, n, args.precision, wavelengths_of, chains, steps, 2)
        block = roofline_block(m, n, nb, esz, dev, "frugal_pass_kernel (every FFT pass launch)", steps)
        ms = m["launch_ms"]
        return {"value": nb * steps / m["elapsed"], "unit": "wavefronts/s", "batch": nb, "steps": steps,
                "ms_per_step": 1e3 * m["elapsed"] / steps,
                "pass_launch_ms_per_step": float(ms.sum()) / steps,
                "wavefronts_per_s_of_pass_time": nb * steps / (float(ms.sum()) * 1e-3) if ms.size else None,
                "workload": f"SYN20 + a white-noise grid-sag screen (40 nm rms) behind S02, {n}^2 {args.precision}, batch {nb}: "
                            f"the screen is one array shared by all items, built and uploaded once (round 5); "
                            f"`pass_launch_ms_per_step` / `roofline.classes` are the device times of the pass launches on "
                            f"these rough fields",
                "roofline": block}
    finally:
        dev.close()


def _psd_screen(args):
    """One ``WFO.psd`` screen at the headline grid: the host's two draws + ``paos_psd_screen`` (round 5) against the host
    path (NumPy's fft2 / ifft2: the reference's own arithmetic), same seed -- what each costs and how far apart the maps are."""
    from paos_amd import _lib, phase_maps

    n = args.grid
    kw = dict(A=12.0, B=1.0, C=2.2, fknee=3.0, fmin=None, fmax=None, SR=0.5, units="nm")
    dx = 4.0 / n
    dev = _lib.DeviceFields(n, 1, "fp64")
    try:
        state = np.random.get_state()
        np.random.seed(3)
        t0 = time.perf_counter()
        screen = phase_maps.PsdScreen((n, n), dx, dx, **kw)
        t1 = time.perf_counter()
        dev.psd_screen(screen.noise, screen.rough, screen.params, key=1)  # first call: allocations, LDS opt-ins
        t2 = time.perf_counter()
        got = dev.psd_screen(screen.noise, screen.rough, screen.params, key=2, want_map=True)
        t3 = time.perf_counter()
        want = np.ma.filled(screen.host_map(), 0.0)
        t4 = time.perf_counter()
        np.random.set_state(state)
        return {"grid": n, "draws_ms": 1e3 * (t1 - t0), "device_ms": 1e3 * (t3 - t2), "device_first_call_ms": 1e3 * (t2 - t1),
                "host_ms": 1e3 * (t4 - t3), "max_abs_diff_over_peak": float(np.abs(got - want).max() / np.abs(want).max()),
                "what": "draws: the two np.random.randn(n, n) of the call (host, the reference's generator); device: upload of both + "
                        "fft2 -> filter -> ifft2 -> roughness on the library's passes + the map's download; host: the same on NumPy"}
    finally:
        dev.close()


def _timed_batch(pup, wls, n, zoom, field, chains, precision="fp64", reps=3, **kw):
    from paos_amd import _lib
    from paos_amd.run import run_batch

    dev = _lib.DeviceFields(n, len(chains), precision)
    try:
        stats = {}
        run_batch(pup, wls, n, zoom, field, chains, precision=precision, outputs=(), dev=dev, sync=True, **kw)
        t0 = time.perf_counter()
        for _ in range(reps):
            res = run_batch(pup, wls, n, zoom, field, chains, precision=precision, outputs=(), dev=dev, sync=False,
                            stats=stats, **kw)
        dev.sync()
        return (time.perf_counter() - t0) / reps, res, stats
    finally:
        dev.close()


def _configs():
    """BASELINE.json configs 3-5 as one-GPU runs (tests/reports/run_configs.py is the long form with parity checks)."""
    from paos_amd import _lib
    from paos_amd.chains import inject_wfe, parse_config_variant, read_wfe_table
    from paos_amd.run import run_batch

    out = {}
    # 3. Ariel_AIRS-CH0, 64-wavelength batch, 2048^2
    sweep = np.linspace(1.95, 3.9, 64)
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Ariel_AIRS-CH0.ini"), sweep)
    w = [1e-6 * x for x in wls]
    dt, _, st = _timed_batch(pup, w, 2048, par["zoom"], fields[0], chains)
    light = [{key: dict(item, save=item["name"] == "IMAGE_PLANE") for key, item in c.items()} for c in chains]
    dtl, _, stl = _timed_batch(pup, w, 2048, par["zoom"], fields[0], light)
    out["airs_ch0_64wl_2048"] = {
        "config": "Ariel_AIRS-CH0.ini, 64-wavelength batch (1.95-3.9 um), 2048^2 fp64, one batch of 64, no arrays downloaded",
        "every_surface_saved": {"value": 64 / dt, "unit": "wavefronts/s", "fused_passes": st.get("fused_passes")},
        "light_output": {"value": 64 / dtl, "unit": "wavefronts/s", "fused_passes": stl.get("fused_passes"),
                         "what": "only the image plane saved (pipeline.py:111-114)"}}

    # 4. Ariel_FGS-FGS1 + WFE table, 256 Monte-Carlo draws, 2048^2, on-device rEE90 (8 batches of 32)
    _, _, _, table = read_wfe_table(WFE)
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Ariel_FGS-FGS1.ini"), unignore=("Z1",))
    base, wl = chains[0], 1e-6 * wls[0]
    radii = np.geomspace(2.0, 256.0, 16)
    entry = {"config": "Ariel_FGS-FGS1.ini + wfe_realization_SN20210914.csv, 256 Monte-Carlo WFE draws (Z1 un-ignored), 2048^2 fp64, "
                       "8 batches of 32, encircled energies on the device, rEE90 interpolated on the host"}
    for tag, chain in (("every_surface_saved", base),
                       ("light_output", {key: dict(item, save=item["name"] == "IMAGE_PLANE") for key, item in base.items()})):
        dev = _lib.DeviceFields(2048, 32)
        try:
            stats, ree = {}, []
            mc0 = [inject_wfe(chain, table[:, k]) for k in range(32)]
            run_batch(pup, [wl] * 32, 2048, par["zoom"], fields[0], mc0, outputs=(), dev=dev, sync=True, metrics_radii_px=radii)
            t0 = time.perf_counter()
            for lo in range(0, 256, 32):
                mc = [inject_wfe(chain, table[:, k]) for k in range(lo, lo + 32)]
                res = run_batch(pup, [wl] * 32, 2048, par["zoom"], fields[0], mc, outputs=(), dev=dev, sync=True,
                                metrics_radii_px=radii, stats=stats)
                for r in res:
                    rec = r[max(r)]
                    ee = rec["metrics"]["encircled"] / rec["metrics"]["power"]
                    ree.append(float(np.interp(0.9, ee, radii)) * rec["dx"])
            dt = time.perf_counter() - t0
        finally:
            dev.close()
        entry[tag] = {"value": 256 / dt, "unit": "wavefronts/s", "fused_passes": stats.get("fused_passes"),
                      "rEE90_um_min_median_max": [1e6 * min(ree), 1e6 * float(np.median(ree)), 1e6 * max(ree)]}
    out["fgs1_256_draws_2048"] = entry

    # 5. Excite_TEL, wavelength sweep at 4096^2, fp64 and fp32, PSFs left in HBM (32 of the 512 wavelengths per batch)
    sweep = np.linspace(1.0, 4.0, 512)[::16]
    pup, par, wls, fields, chains = parse_config_variant(os.path.join(LENS, "Excite_TEL.ini"), sweep)
    w = [1e-6 * x for x in wls]
    d64, _, s64 = _timed_batch(pup, w, 4096, par["zoom"], fields[0], chains, "fp64", keep_psf=True)
    d32, _, _ = _timed_batch(pup, w, 4096, par["zoom"], fields[0], chains, "fp32", keep_psf=True)
    out["excite_tel_4096"] = {
        "config": "Excite_TEL.ini, 32 of the 512 wavelengths of a 1-4 um sweep per batch, 4096^2, PSFs written to and left in HBM",
        "fp64": {"value": 32 / d64, "unit": "wavefronts/s", "fused_passes": s64.get("fused_passes")},
        "fp32": {"value": 32 / d32, "unit": "wavefronts/s", "ratio_to_fp64": d64 / d32,
                 "what": "complex64 field, fp64 phase arguments; PSF error vs the oracle 1e-6 at 4096^2 (tests/test_gpu_r3.py)"}}
    return out


def extras(args, esz, measure, roofline_block, sweep_report):
    steps, warmup = min(max(args.steps, 5), 20), max(args.warmup, 2)
    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001 -- an extra entry must not cost the headline line
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}

    for n2, nb2 in ((2048, 64), (1024, 256)):
        guarded(f"{n2}^2", lambda n2=n2, nb2=nb2: _syn20(n2, nb2, args.precision, steps, warmup, esz, measure, roofline_block,
                                                          sweep_report))
    if args.precision == "fp64":
        # the headline workload with twice the wavefronts per step: what the per-step costs that do not grow with the batch
        # (launch gaps, tails, the start) are worth -- the headline stays at 32 per step, as in every round
        guarded("4096^2 x 64", lambda: _syn20(4096, 64, "fp64", steps, warmup, esz, measure, roofline_block, sweep_report,
                                               label="SYN20, walked sweep, 4096^2 fp64, 64 wavefronts per step"))
        guarded("fp32_4096", lambda: _syn20(4096, 32, "fp32", steps, warmup, 8, measure, roofline_block, sweep_report,
                                            label="SYN20, walked sweep, 4096^2 in fp32 mode (complex64 field, fp64 phase arguments)"))
    guarded("dense", lambda: _dense(args, esz, measure, roofline_block))
    guarded("psd_screen", lambda: _psd_screen(args))
    guarded("configs", _configs)
    return out
