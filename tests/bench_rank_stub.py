"""A rank of a self-launched job, without a GPU: what bench.py's children do around the measurement -- join the job
from the launcher's environment, take the ONE broadcast, barrier, reduce a time bracket with MAX, rank 0 prints one
JSON line -- on the library's TCP transport.  Started by tests/test_abi_and_dist.py through bench.self_launch."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from paos_amd.comm import Comm  # noqa: E402
from paos_amd.dist import broadcast_work, shard_bounds  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("STUB_FAIL_RANK") == str(rank):
        sys.exit(7)  # a rank that dies before the rendezvous
    comm = Comm.from_env(transport="socket", timeout=60.0)
    work = broadcast_work({"wavelengths": [1.0e-6 * (1 + k / 512.0) for k in range(4 * world)]} if rank == 0 else None, comm)
    lo, hi = shard_bounds(len(work["wavelengths"]), rank, world)
    comm.barrier()
    t = comm.max(0.001 * (rank + 1))
    seen = [int(p[0]) for p in comm.allgather_scalars([float(os.environ["LOCAL_RANK"])])]
    if rank == 0:
        print(json.dumps({"value": len(work["wavelengths"]) / t, "n_gpus": world, "shard": [lo, hi], "devices_seen": seen,
                          "key": os.environ["PAOS_COMM_KEY"], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                          "self_launched": os.environ.get("PAOS_BENCH_SELF_LAUNCHED")}), flush=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
