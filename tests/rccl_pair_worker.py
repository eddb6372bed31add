"""Worker of tests/test_gpu_r3.py::test_two_ranks_asking_for_rccl_on_one_gpu_agree (a module of its own so that the
fork server's children can import it without pytest)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(rank, key, out):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np

    from paos_amd.comm import Comm, CommError

    try:
        comm = Comm(2, rank, 0, "rccl", key=key, timeout=45)  # both ranks on device 0: the box has one GPU
    except CommError as exc:
        out.put((rank, "error", str(exc), None, None))
        return
    try:
        text = comm.bcast_blob(b"two ranks, one GPU" if rank == 0 else None, root=0)
        parts = comm.allgather_scalars(np.arange(rank + 2, dtype=float))
        out.put((rank, comm.transport, text, comm.max(float(rank)), [list(p) for p in parts]))
    finally:
        comm.close()


def run_bench(argv, env_extra, out):
    """Started from the fork server (a process that never touched the GPU): runs bench.py as a child program and
    hands back (exit code, stdout, tail of stderr)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "PAOS_COMM_KEY"):
        env.pop(k, None)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(root, "bench.py")] + list(argv)
    if env.pop("LAUNCH_WITH_TORCHRUN", "") == "1":  # the driver's way: an external launcher exports the job
        import socket

        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        n = argv[argv.index("--gpus") + 1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", n, "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py")] + list(argv)
    run = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out.put((run.returncode, run.stdout.decode(errors="replace"), run.stderr.decode(errors="replace")[-2000:]))
