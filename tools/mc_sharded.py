#!/usr/bin/env python3
"""BASELINE.json configs[3]: Ariel_FGS-FGS1.ini with the shipped WFE table, 256 Monte-Carlo draws at
2048^2, sharded over the ranks of a process group (one GPU per rank), PSF metrics reduced on the GPUs.

    python tools/mc_sharded.py                                   # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port 29540 tools/mc_sharded.py                  # N GPUs (RCCL)
    PAOS_MC_REHEARSAL=1 ... --nproc-per-node 2 tools/mc_sharded.py --draws 16 --grid 512
                                                                 # 2 ranks on ONE GPU over the TCP transport
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=256)
    ap.add_argument("--grid", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    comm = None
    if world > 1:  # torch is only the launcher: the process group is paos_comm (RCCL from libpaoship.so)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from paos_amd.comm import Comm

        if os.environ.get("PAOS_MC_REHEARSAL") == "1":
            local_rank = 0
            comm = Comm.from_env(transport="socket")
        else:
            comm = Comm.from_env(transport="rccl")

    from paos_amd.chains import inject_wfe, parse_config_variant, read_wfe_table
    from paos_amd.dist import run_sharded

    wls = chains = None
    pup = zoom = field = None
    lens = os.path.join(ROOT, "data", "lens", "Ariel_FGS-FGS1.ini")
    pup, par, w, fields, ch = parse_config_variant(lens, unignore=("Z1",))
    zoom, field = par["zoom"], fields[0]
    if rank == 0:  # rank 0 alone reads the table; the work travels in run_sharded's one broadcast
        _, _, _, table = read_wfe_table(os.path.join(ROOT, "data", "wfe", "wfe_realization_SN20210914.csv"))
        chains = [inject_wfe(ch[0], table[:, k % table.shape[1]]) for k in range(args.draws)]
        wls = [1e-6 * w[0]] * args.draws
    radii = np.geomspace(2.0, 256.0, 16)
    # warm-up: library and kernel code load, first allocations (one batch per rank)
    warm = min(args.batch * world, args.draws)
    run_sharded(pup, wls[:warm] if rank == 0 else None, args.grid, zoom, field,
                chains[:warm] if rank == 0 else None, batch=args.batch, device=local_rank, outputs=(),
                metrics_radii_px=radii, gather=False, comm=comm)
    t0 = time.perf_counter()
    res = run_sharded(pup, wls, args.grid, zoom, field, chains, batch=args.batch, device=local_rank,
                      outputs=(), metrics_radii_px=radii, comm=comm)
    dt = time.perf_counter() - t0
    if rank == 0:
        ree90, power = [], []
        for _, r in res:
            rec = r[max(r)]
            ee = rec["metrics"]["encircled"] / rec["metrics"]["power"]
            ree90.append(float(np.interp(0.9, ee, radii)) * rec["dx"])
            power.append(rec["power"])
        print(f"{len(res)} draws on {world} rank(s), {args.grid}^2: {dt:.2f} s = {len(res) / dt:.1f} wavefronts/s "
              f"(including the gather); power {min(power):.6f}..{max(power):.6f}; "
              f"rEE90 {1e6 * min(ree90):.2f}..{1e6 * max(ree90):.2f} um (median {1e6 * float(np.median(ree90)):.2f})")
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
