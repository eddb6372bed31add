/* paos_comm.h -- the multi-GPU fan-out of the wavefront batch, C ABI (part of libpaoship.so).
 *
 * Stands in for the joblib fan-out of the reference, paos/core/pipeline.py:139-150
 * (`Parallel(n_jobs)(delayed(run)(...) for wavelength in ...)`): wavefronts (wavelengths, Monte-Carlo
 * WFE draws) are independent, so the only communication is ONE broadcast of the packed work
 * description from rank 0 and, if the caller wants them in one place, a gather of per-wavefront
 * scalars.  One process per GPU; no PyTorch.
 *
 * Two transports behind the same calls:
 *   PAOS_COMM_RCCL    device buffers over RCCL (xGMI between the GPUs of a node): ncclBroadcast /
 *                     ncclAllGather / ncclAllReduce on the communicator's own HIP stream.  librccl is
 *                     loaded with dlopen at init, so the library has no link-time dependency on it.
 *   PAOS_COMM_SOCKET  TCP over the loopback / node network through rank 0 (star): what the CPU-only
 *                     tests use, and what carries the RCCL unique id at start-up.
 * Bootstrap: rank 0 listens on an ephemeral port and publishes it in the rendezvous file
 * `<dir>/paos_comm_<key>` (written atomically); the other ranks poll the file and connect.  `key` must
 * be the same on every rank of one job and unique per job on the host (bench.py uses
 * MASTER_PORT + TORCHELASTIC_RUN_ID of the launcher, or a key of its own when it starts the ranks itself).
 * Single node, like the launch contract.  The two ends of a fresh connection exchange hello / reply / acknowledgement
 * (magic, hash of the key, world size, rank); rank 0 counts a rank only once the acknowledgement is in.
 * Environment: paos_comm_init_rank(PAOS_COMM_RCCL) exports HSA_ENABLE_IPC_MODE_LEGACY=0 unless the variable is
 * already set (dmabuf IPC, which RCCL needs on these hosts); it takes effect only when no HIP call has been made
 * in the process before -- create the communicator first, or export the variable yourself.
 * If paos_comm_init_rank fails with "ncclCommInitRank did not return" the process must exit (never re-exec it).
 *
 * Every function returns 0 on success, a PAOS_E* code of paos_hip.h otherwise; paos_comm_last_error()
 * gives the message of the calling thread's last failure.
 */
#ifndef PAOS_COMM_H
#define PAOS_COMM_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct paos_comm paos_comm;

enum { PAOS_COMM_SOCKET = 0, PAOS_COMM_RCCL = 1 };

/* Join the job: `nranks` processes call this with ranks 0..nranks-1.  `device` is the HIP device of
 * this rank (RCCL transport only).  `rendezvous_dir` may be NULL (= "/tmp"); timeout_s bounds the
 * wait for the other ranks.  nranks == 1 needs no peer and no file.
 * PAOS_COMM_RCCL is a request: the ranks agree over the control plane whether RCCL came up on ALL of them
 * (library found, device usable, ncclCommInitRank succeeded); if not, every rank continues on the TCP
 * transport, says so on stderr, and paos_comm_transport reports PAOS_COMM_SOCKET. */
int paos_comm_init_rank(int nranks, int rank, int device, int transport, const char* key,
                        const char* rendezvous_dir, double timeout_s, paos_comm** out);
int paos_comm_destroy(paos_comm* comm);
int paos_comm_rank(const paos_comm* comm);
int paos_comm_size(const paos_comm* comm);
int paos_comm_transport(const paos_comm* comm);
const char* paos_comm_last_error(void);
/* After a PAOS_COMM_RCCL request that ended on the TCP transport: what kept RCCL from coming up on THIS rank
 * (library not found, device error, the text of the ncclCommInitRank failure), or that it did come up here but not
 * on every rank; "" when RCCL is in use or was never asked for.  bench.py --gpus N gathers these into the record
 * it prints when a scaling run cannot use RCCL (a diagnosable failure instead of a silent one). */
const char* paos_comm_bringup_note(const paos_comm* comm);

/* Size first: root passes its length in *bytes, the others receive it (so that they can allocate). */
int paos_comm_bcast_size(paos_comm* comm, unsigned long long* bytes, int root);
/* The work description (packed optical chains, wavelengths, coefficients): `bytes` bytes of host memory,
 * valid on root on entry, on every rank on return.  RCCL: staged through a device buffer, ncclBroadcast. */
int paos_comm_bcast_blob(paos_comm* comm, void* host_buf, unsigned long long bytes, int root);
/* Per-wavefront scalars (power, encircled-energy radii, dx, fratio ...): every rank contributes `count`
 * doubles and receives all of them, ordered by rank: recv[nranks * count].  RCCL: ncclAllGather. */
int paos_comm_allgather_scalars(paos_comm* comm, const double* send, int count, double* recv);
/* Ragged variant: counts may differ per rank; recv is laid out rank after rank, counts_out[nranks]. */
int paos_comm_allgatherv_scalars(paos_comm* comm, const double* send, int count, double* recv,
                                 unsigned long long recv_capacity, int* counts_out);
/* MAX over ranks (the time bracket of a benchmark), and a barrier. */
int paos_comm_max(paos_comm* comm, double* value);
int paos_comm_barrier(paos_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* PAOS_COMM_H */
