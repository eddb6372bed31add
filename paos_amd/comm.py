"""ctypes face of include/paos_comm.h: the process group of the wavefront fan-out (one process per GPU).

``Comm.from_env()`` joins the job a launcher started (``python -m torch.distributed.run`` exports RANK,
WORLD_SIZE, LOCAL_RANK, MASTER_PORT, TORCHELASTIC_RUN_ID; any launcher that sets RANK / WORLD_SIZE and a
job key works) WITHOUT importing torch: RCCL is driven from libpaoship.so directly.  Transports: "rccl"
(device buffers over xGMI) and "socket" (TCP through rank 0; what the CPU tests use).
"""
import ctypes
import os

import numpy as np

from . import _lib



def _want_dmabuf_ipc():
    """RCCL's buffer exchange between processes needs dmabuf IPC on these hosts (HSA_ENABLE_IPC_MODE_LEGACY=0); the
    variable is read when the HSA runtime starts, i.e. it must be in the environment before the FIRST HIP call of the
    process.  Set only when an RCCL communicator is asked for (importing this module used to export it for
    every process, single-GPU and TCP-only ones included); a value the user exported is kept; a process that has already
    created a device context is told that the setting comes too late (paos_comm_init_rank sets it too, for callers of
    the C ABI)."""
    if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
        return
    if _lib.hip_initialised():
        import warnings

        warnings.warn("an RCCL communicator is being created after this process has already used the GPU: "
                      "HSA_ENABLE_IPC_MODE_LEGACY=0 was not in the environment when HIP started, so RCCL may fail with "
                      "'hipIpcGetMemHandle: invalid argument' -- create the Comm before the first DeviceFields, or export "
                      "the variable", RuntimeWarning, stacklevel=3)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"


SOCKET, RCCL = 0, 1
_c_comm = ctypes.c_void_p
_dbl_p = ctypes.POINTER(ctypes.c_double)

SYMBOLS = {
    "paos_comm_init_rank": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                           ctypes.c_char_p, ctypes.c_double, ctypes.POINTER(_c_comm)]),
    "paos_comm_destroy": (ctypes.c_int, [_c_comm]),
    "paos_comm_rank": (ctypes.c_int, [_c_comm]),
    "paos_comm_size": (ctypes.c_int, [_c_comm]),
    "paos_comm_transport": (ctypes.c_int, [_c_comm]),
    "paos_comm_last_error": (ctypes.c_char_p, []),
    "paos_comm_bringup_note": (ctypes.c_char_p, [_c_comm]),
    "paos_comm_bcast_size": (ctypes.c_int, [_c_comm, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]),
    "paos_comm_bcast_blob": (ctypes.c_int, [_c_comm, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_int]),
    "paos_comm_allgather_scalars": (ctypes.c_int, [_c_comm, _dbl_p, ctypes.c_int, _dbl_p]),
    "paos_comm_allgatherv_scalars": (ctypes.c_int, [_c_comm, _dbl_p, ctypes.c_int, _dbl_p, ctypes.c_ulonglong,
                                                    ctypes.POINTER(ctypes.c_int)]),
    "paos_comm_max": (ctypes.c_int, [_c_comm, _dbl_p]),
    "paos_comm_barrier": (ctypes.c_int, [_c_comm]),
}

_bound = None


def _load():
    global _bound
    if _bound is None:
        lib = _lib.load()
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bound = lib
    return _bound


class CommError(RuntimeError):
    pass


class Comm:
    def __init__(self, nranks=1, rank=0, device=0, transport="socket", key=None, rendezvous_dir=None, timeout=120.0):
        self._lib = _load()
        code = {"socket": SOCKET, "rccl": RCCL}[transport]
        if code == RCCL:
            _want_dmabuf_ipc()
        self._h = _c_comm()
        rc = self._lib.paos_comm_init_rank(int(nranks), int(rank), int(device), code,
                                           key.encode() if key else None,
                                           rendezvous_dir.encode() if rendezvous_dir else None, float(timeout),
                                           ctypes.byref(self._h))
        if rc != 0:
            self._h = _c_comm()
            raise CommError(f"paos_comm_init_rank failed ({rc}): {self._lib.paos_comm_last_error().decode()}")
        # the transport in use: "rccl" falls back to "socket" on every rank when RCCL cannot come up on all of them
        actual = {SOCKET: "socket", RCCL: "rccl"}[self._lib.paos_comm_transport(self._h)]
        self.rank, self.size, self.transport, self.device = int(rank), int(nranks), actual, int(device)
        self.requested = transport
        # why an RCCL request ended on TCP, as seen from this rank ("" otherwise)
        self.bringup_note = (self._lib.paos_comm_bringup_note(self._h) or b"").decode()

    @classmethod
    def from_env(cls, transport=None, timeout=300.0):
        """Join the job described by the launcher's environment.  ``transport`` defaults to "rccl" when
        PAOS_COMM_TRANSPORT is unset (one GPU per rank: LOCAL_RANK) -- set it to "socket" for ranks
        that share a GPU or have none."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        transport = transport or os.environ.get("PAOS_COMM_TRANSPORT", "rccl")
        key = os.environ.get("PAOS_COMM_KEY")
        if not key:
            # the launcher's rendezvous port is what makes the key unique on the host (two jobs cannot both listen
            # on it); without it concurrent jobs would meet in the same rendezvous file
            if world > 1 and "MASTER_PORT" not in os.environ:
                raise CommError("set PAOS_COMM_KEY (the same on every rank, unique per job on the host): the environment "
                                "has no MASTER_PORT to derive a job key from")
            key = "_".join([os.environ.get("TORCHELASTIC_RUN_ID", "job"), os.environ.get("MASTER_PORT", "0"),
                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")])
        return cls(world, rank, local, transport, key=key, timeout=timeout)

    def _check(self, rc, what):
        if rc != 0:
            raise CommError(f"{what} failed ({rc}): {self._lib.paos_comm_last_error().decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.paos_comm_destroy(self._h)
            self._h = _c_comm()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bcast_blob(self, blob, root=0):
        """``blob`` (bytes) from ``root`` to every rank; the others pass None."""
        size = ctypes.c_ulonglong(len(blob) if self.rank == root else 0)
        self._check(self._lib.paos_comm_bcast_size(self._h, ctypes.byref(size), int(root)), "paos_comm_bcast_size")
        buf = (ctypes.c_char * max(1, size.value))()
        if self.rank == root:
            ctypes.memmove(buf, blob, size.value)
        self._check(self._lib.paos_comm_bcast_blob(self._h, buf, size.value, int(root)), "paos_comm_bcast_blob")
        return bytes(buf[:size.value])

    def allgather_scalars(self, values):
        """Every rank contributes a 1-D float64 array (lengths may differ); returns the list of all of
        them, indexed by rank."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        counts = (ctypes.c_int * self.size)()
        # sizes first (so the receive buffer can be allocated), then the ragged gather
        mine = np.array([float(v.size)])
        sizes = np.empty(self.size)
        self._check(self._lib.paos_comm_allgather_scalars(self._h, mine.ctypes.data_as(_dbl_p), 1,
                                                          sizes.ctypes.data_as(_dbl_p)), "paos_comm_allgather_scalars")
        total = int(sizes.sum())
        out = np.empty(max(total, 1))
        self._check(self._lib.paos_comm_allgatherv_scalars(self._h, v.ctypes.data_as(_dbl_p) if v.size else None,
                                                           int(v.size), out.ctypes.data_as(_dbl_p), total, counts),
                    "paos_comm_allgatherv_scalars")
        parts, o = [], 0
        for r in range(self.size):
            parts.append(out[o:o + counts[r]].copy())
            o += counts[r]
        return parts

    def gather_text(self, text, limit=240):
        """Every rank's short ASCII note (e.g. ``bringup_note``), indexed by rank -- over the scalar gather, one
        double per character: diagnostics only."""
        codes = [float(b) for b in text.encode("ascii", "replace")[:limit]]
        return ["".join(chr(int(x)) for x in part) for part in self.allgather_scalars(codes)]

    def max(self, value):
        x = ctypes.c_double(float(value))
        self._check(self._lib.paos_comm_max(self._h, ctypes.byref(x)), "paos_comm_max")
        return x.value

    def barrier(self):
        self._check(self._lib.paos_comm_barrier(self._h), "paos_comm_barrier")
