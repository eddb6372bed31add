// paos_comm.cpp -- include/paos_comm.h: the one broadcast / one gather of the wavefront fan-out
// (reference: the joblib fan-out of paos/core/pipeline.py:139-150), over RCCL (xGMI) or plain TCP.
//
// Control plane: a TCP star through rank 0 on the loopback interface, set up through a rendezvous
// file.  It carries the RCCL unique id at start-up and is the whole transport of the socket backend.
// Data plane (PAOS_COMM_RCCL): ncclBroadcast / ncclAllGather / ncclAllReduce on device staging
// buffers; librccl is dlopen'ed so that libpaoship.so loads on machines without it.
#include "../../include/paos_comm.h"
#include "../../include/paos_hip.h"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/select.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_comm_err;

int cfail(int code, const std::string& msg) {
  g_comm_err = msg;
  return code;
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool send_all(int fd, const void* buf, size_t bytes) {
  const char* p = static_cast<const char*>(buf);
  while (bytes > 0) {
    const ssize_t k = ::send(fd, p, bytes, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += k;
    bytes -= (size_t)k;
  }
  return true;
}

bool recv_all(int fd, void* buf, size_t bytes) {
  char* p = static_cast<char*>(buf);
  while (bytes > 0) {
    const ssize_t k = ::recv(fd, p, bytes, 0);
    if (k == 0) return false;  // peer closed
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += k;
    bytes -= (size_t)k;
  }
  return true;
}

// receive timeout of a control-plane socket (0 = block for ever): a peer that never speaks must not stall a rank
void set_recv_timeout(int fd, double seconds) {
  timeval tv{};
  tv.tv_sec = (long)seconds;
  tv.tv_usec = (long)((seconds - (double)(long)seconds) * 1e6);
  ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
}

// What the two ends of a fresh control-plane connection tell each other, so that a stale rendezvous file whose
// port now belongs to an unrelated listener (or a stray connect to rank 0's port) is noticed instead of trusted.
constexpr unsigned kHelloMagic = 0x50414f53u;  // "PAOS"
struct Hello {
  unsigned magic, key_hash;
  int nranks, rank;
};
unsigned fnv1a(const std::string& s) {
  unsigned h = 2166136261u;
  for (unsigned char ch : s) { h ^= ch; h *= 16777619u; }
  return h;
}

// ---- the slice of the RCCL API this file uses, resolved at run time --------------------------
struct UniqueId { char internal[128]; };  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint8 = 1, kNcclFloat64 = 8, kNcclMax = 2;

bool load_rccl(Rccl& r, std::string& why) {
  for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) { why = std::string("librccl.so not found: ") + dlerror(); return false; }
  auto sym = [&](const char* n) { return dlsym(r.handle, n); };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Broadcast || !r.AllGather || !r.AllReduce) {
    why = "librccl.so lacks an expected symbol";
    return false;
  }
  return true;
}

}  // namespace

struct paos_comm {
  int nranks = 1, rank = 0, transport = PAOS_COMM_SOCKET, device = 0;
  int listen_fd = -1;
  std::vector<int> fds;  // rank 0: socket of every peer (index = rank); others: fds[0] = socket to rank 0
  Rccl rccl;
  void* nccl = nullptr;
  hipStream_t stream = nullptr;
  void* dbuf = nullptr;  // device staging
  size_t dcap = 0;
  std::string bringup_note;  // why RCCL is not in use on THIS rank ("" = it is, or it was never asked for)
};

namespace {

#define COMM_HIP(call)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) return cfail(PAOS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

int nccl_check(paos_comm* c, int rc, const char* what) {
  if (rc == 0) return PAOS_OK;
  return cfail(PAOS_EHIP, std::string(what) + ": " + (c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "RCCL error"));
}

int stage(paos_comm* c, size_t bytes) {
  if (bytes <= c->dcap) return PAOS_OK;
  if (c->dbuf) (void)hipFree(c->dbuf);
  c->dbuf = nullptr;
  c->dcap = 0;
  size_t cap = std::max<size_t>(bytes, 1 << 16);
  COMM_HIP(hipMalloc(&c->dbuf, cap));
  c->dcap = cap;
  return PAOS_OK;
}

// star primitives over TCP -------------------------------------------------------------------
int sock_bcast(paos_comm* c, void* buf, size_t bytes, int root) {
  if (c->nranks == 1 || bytes == 0) return PAOS_OK;
  if (c->rank == 0) {
    if (root != 0 && !recv_all(c->fds[root], buf, bytes)) return cfail(PAOS_EHIP, "broadcast: receiving from the root failed");
    for (int r = 1; r < c->nranks; ++r)
      if (r != root && !send_all(c->fds[r], buf, bytes)) return cfail(PAOS_EHIP, "broadcast: sending to a rank failed");
  } else if (c->rank == root) {
    if (!send_all(c->fds[0], buf, bytes)) return cfail(PAOS_EHIP, "broadcast: sending to rank 0 failed");
  } else {
    if (!recv_all(c->fds[0], buf, bytes)) return cfail(PAOS_EHIP, "broadcast: receiving failed (did rank 0 exit?)");
  }
  return PAOS_OK;
}

// every rank contributes counts[rank] doubles; everybody gets all of them, rank after rank
int sock_allgatherv(paos_comm* c, const double* send, int count, double* recv, const std::vector<int>& counts) {
  size_t total = 0, off = 0;
  for (int r = 0; r < c->nranks; ++r) total += (size_t)counts[r];
  for (int r = 0; r < c->rank; ++r) off += (size_t)counts[r];
  if (c->nranks == 1) {
    if (count) std::memcpy(recv, send, (size_t)count * sizeof(double));
    return PAOS_OK;
  }
  if (c->rank == 0) {
    if (count) std::memcpy(recv, send, (size_t)count * sizeof(double));
    size_t o = (size_t)counts[0];
    for (int r = 1; r < c->nranks; ++r) {
      if (counts[r] && !recv_all(c->fds[r], recv + o, (size_t)counts[r] * sizeof(double)))
        return cfail(PAOS_EHIP, "gather: receiving from a rank failed");
      o += (size_t)counts[r];
    }
    for (int r = 1; r < c->nranks; ++r)
      if (total && !send_all(c->fds[r], recv, total * sizeof(double))) return cfail(PAOS_EHIP, "gather: sending the result failed");
  } else {
    if (count && !send_all(c->fds[0], send, (size_t)count * sizeof(double))) return cfail(PAOS_EHIP, "gather: sending to rank 0 failed");
    if (total && !recv_all(c->fds[0], recv, total * sizeof(double))) return cfail(PAOS_EHIP, "gather: receiving the result failed");
  }
  (void)off;
  return PAOS_OK;
}

int sock_allgather(paos_comm* c, const double* send, int count, double* recv) {
  return sock_allgatherv(c, send, count, recv, std::vector<int>(c->nranks, count));
}

int connect_star(paos_comm* c, const std::string& path, unsigned key_hash, double timeout_s) {
  const double deadline = now_s() + timeout_s;
  if (c->rank == 0) {
    c->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (c->listen_fd < 0) return cfail(PAOS_EHIP, "socket() failed");
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    addr.sin_port = 0;
    if (::bind(c->listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0 || ::listen(c->listen_fd, c->nranks) != 0)
      return cfail(PAOS_EHIP, std::string("bind/listen on 127.0.0.1 failed: ") + strerror(errno));
    socklen_t len = sizeof(addr);
    ::getsockname(c->listen_fd, reinterpret_cast<sockaddr*>(&addr), &len);
    // the port goes into a file of our own making (O_EXCL | O_NOFOLLOW: no symlink somebody planted is followed, mode
    // 0600), published under the agreed name by rename
    const std::string tmp = path + "." + std::to_string((long)::getpid()) + ".tmp";
    ::unlink(tmp.c_str());
    const int tfd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    if (tfd < 0) return cfail(PAOS_EINVAL, "cannot create the rendezvous file " + tmp + ": " + strerror(errno));
    char text[32];
    const int tlen = std::snprintf(text, sizeof(text), "%d\n", (int)ntohs(addr.sin_port));
    const bool wrote = ::write(tfd, text, (size_t)tlen) == (ssize_t)tlen;
    ::close(tfd);
    if (!wrote) { ::unlink(tmp.c_str()); return cfail(PAOS_EINVAL, "cannot write the rendezvous file " + tmp); }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) { ::unlink(tmp.c_str()); return cfail(PAOS_EINVAL, "cannot publish the rendezvous file " + path); }
    c->fds.assign(c->nranks, -1);
    timeval tv{};
    for (int got = 1; got < c->nranks;) {
      const double left = deadline - now_s();
      if (left <= 0) { ::unlink(path.c_str()); return cfail(PAOS_EHIP, "timed out waiting for the other ranks to connect"); }
      tv.tv_sec = (long)left; tv.tv_usec = (long)((left - (long)left) * 1e6);
      fd_set set;
      FD_ZERO(&set);
      FD_SET(c->listen_fd, &set);
      if (::select(c->listen_fd + 1, &set, nullptr, nullptr, &tv) <= 0) continue;
      const int fd = ::accept(c->listen_fd, nullptr, nullptr);
      if (fd < 0) continue;
      int one = 1;
      ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      set_recv_timeout(fd, 2.0);  // a connection that does not introduce itself at once is not a rank of this job
      Hello hi{};
      if (!recv_all(fd, &hi, sizeof(hi)) || hi.magic != kHelloMagic || hi.key_hash != key_hash || hi.nranks != c->nranks) {
        ::close(fd);  // a stranger (port scan, a rank of another job reading a stale file): keep waiting
        continue;
      }
      if (hi.rank <= 0 || hi.rank >= c->nranks) {
        ::close(fd);
        ::unlink(path.c_str());
        return cfail(PAOS_EINVAL, "a peer announced an invalid rank");
      }
      // Three-way: the rank acknowledges the reply before its socket is committed.  A rank that had queued in the
      // backlog while a stranger was being waited for may have given up on this connection and come back on a new
      // one: its hello is still buffered here and the reply would "succeed" into a closed peer -- the missing
      // acknowledgement (EOF / timeout) tells, and the connection is dropped instead of being counted.
      const Hello reply{kHelloMagic, key_hash, c->nranks, 0};
      Hello ack{};
      if (!send_all(fd, &reply, sizeof(reply)) || !recv_all(fd, &ack, sizeof(ack)) || ack.magic != kHelloMagic ||
          ack.key_hash != key_hash || ack.nranks != c->nranks || ack.rank != hi.rank) {
        ::close(fd);
        continue;
      }
      set_recv_timeout(fd, 0.0);
      if (c->fds[hi.rank] != -1) {  // the same rank again on a fresh connection: the newer one is the live one
        ::close(c->fds[hi.rank]);
        c->fds[hi.rank] = fd;
        continue;
      }
      c->fds[hi.rank] = fd;
      ++got;
    }
    ::unlink(path.c_str());
    return PAOS_OK;
  }
  // The rendezvous file is re-read on every attempt: a file left behind by a job that died (same key) names
  // a port nobody listens on any more; rank 0 of THIS job replaces it as soon as it is up.
  for (;;) {
    int port = -1;
    if (FILE* fh = std::fopen(path.c_str(), "r")) {
      if (std::fscanf(fh, "%d", &port) != 1) port = -1;
      std::fclose(fh);
    }
    if (port > 0) {
      const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return cfail(PAOS_EHIP, "socket() failed");
      sockaddr_in addr{};
      addr.sin_family = AF_INET;
      addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
      addr.sin_port = htons((uint16_t)port);
      if (::connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) == 0) {
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        // introduce ourselves and insist on rank 0's answer: whoever listens on a port named by a stale file
        // does not know the job's key, and the connection is dropped and retried
        const Hello hi{kHelloMagic, key_hash, c->nranks, c->rank};
        Hello reply{};
        // (rank 0 serialises its accepts and gives a silent stranger 2 s: allow for a few of those ahead of us)
        set_recv_timeout(fd, 2.0 * c->nranks + 4.0);
        if (send_all(fd, &hi, sizeof(hi)) && recv_all(fd, &reply, sizeof(reply)) && reply.magic == kHelloMagic &&
            reply.key_hash == key_hash && reply.nranks == c->nranks && reply.rank == 0 && send_all(fd, &hi, sizeof(hi))) {
          set_recv_timeout(fd, 0.0);
          c->fds.assign(1, fd);
          return PAOS_OK;
        }
      }
      ::close(fd);
    }
    if (now_s() > deadline) return cfail(PAOS_EHIP, "timed out waiting for rank 0 (rendezvous file " + path + ")");
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
}

}  // namespace

extern "C" {

const char* paos_comm_last_error(void) { return g_comm_err.c_str(); }
const char* paos_comm_bringup_note(const paos_comm* c) { return c ? c->bringup_note.c_str() : ""; }
int paos_comm_rank(const paos_comm* c) { return c ? c->rank : -1; }
int paos_comm_size(const paos_comm* c) { return c ? c->nranks : -1; }
int paos_comm_transport(const paos_comm* c) { return c ? c->transport : -1; }

int paos_comm_destroy(paos_comm* c) {
  if (!c) return PAOS_OK;
  if (c->stream || c->dbuf || c->nccl) {  // whatever the transport ended up being: a fall-back keeps its stream
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl) (void)c->rccl.CommDestroy(c->nccl);
    if (c->dbuf) (void)hipFree(c->dbuf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    // the RCCL handle stays loaded: unloading a library that owns threads is not safe
  }
  for (int fd : c->fds)
    if (fd >= 0) ::close(fd);
  if (c->listen_fd >= 0) ::close(c->listen_fd);
  delete c;
  return PAOS_OK;
}

int paos_comm_init_rank(int nranks, int rank, int device, int transport, const char* key, const char* rendezvous_dir,
                        double timeout_s, paos_comm** out) {
  if (!out) return cfail(PAOS_EINVAL, "out is null");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return cfail(PAOS_EINVAL, "rank out of range");
  if (transport != PAOS_COMM_SOCKET && transport != PAOS_COMM_RCCL) return cfail(PAOS_EINVAL, "unknown transport");
  if (nranks > 1 && (!key || !key[0])) return cfail(PAOS_EINVAL, "a job key is required when nranks > 1");
  if (!(timeout_s > 0)) timeout_s = 120.0;
  paos_comm* c = new paos_comm();
  c->nranks = nranks; c->rank = rank; c->transport = transport; c->device = device;
  int rc = PAOS_OK;
  if (nranks > 1) {
    std::string safe;
    for (const char* p = key; *p; ++p) safe += (std::isalnum((unsigned char)*p) || *p == '-' || *p == '_' || *p == '.') ? *p : '_';
    const std::string path = std::string(rendezvous_dir && rendezvous_dir[0] ? rendezvous_dir : "/tmp") + "/paos_comm_" + safe;
    rc = connect_star(c, path, fnv1a(safe), timeout_s);
    if (rc) { paos_comm_destroy(c); return rc; }
  }
  if (transport == PAOS_COMM_RCCL) {
    // The host driver of these nodes only supports dmabuf IPC: without this RCCL's cross-process buffer exchange fails
    // with "hipIpcGetMemHandle: invalid argument".  It is read when the HSA runtime initialises, i.e. it only takes
    // effect if this is (before) the first HIP call of the process -- paos_amd.comm.Comm also sets it at import.  A
    // value the user has exported is left alone.
    ::setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    // RCCL is the data plane when every rank can bring it up; the ranks agree over the control plane after each
    // step that may fail on some of them only, and otherwise ALL continue on the TCP transport -- a rank that
    // skipped ncclCommInitRank would leave the others waiting in it.
    auto everyone_ok = [&](bool mine) {
      if (nranks == 1) return mine;
      const double flag = mine ? 0.0 : 1.0;
      std::vector<double> all((size_t)nranks, 1.0);
      if (sock_allgather(c, &flag, 1, all.data()) != PAOS_OK) return false;
      for (double v : all)
        if (v != 0.0) return false;
      return true;
    };
    std::string why;
    bool ok = load_rccl(c->rccl, why);
    if (ok) {
      hipError_t e = hipSetDevice(device);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
      if (e != hipSuccess) { ok = false; why = std::string("HIP device setup for RCCL: ") + hipGetErrorString(e); }
    }
    UniqueId id{};
    if (ok && rank == 0 && nccl_check(c, c->rccl.GetUniqueId(&id), "ncclGetUniqueId")) { ok = false; why = g_comm_err; }
    bool all_ok = everyone_ok(ok);
    if (all_ok && nranks > 1 && sock_bcast(c, &id, sizeof(id), 0) != PAOS_OK) {  // the id travels over the control plane
      const std::string keep = g_comm_err; paos_comm_destroy(c); return cfail(PAOS_EHIP, keep);
    }
    if (all_ok) {
      // ncclCommInitRank is a collective without a timeout: if one rank never enters it (it died after the id
      // broadcast, its device is wedged) the others would sit in it for ever and the vote below would never be
      // reached.  It runs on a helper thread under a watchdog; on expiry this rank gives up with an error (the
      // thread cannot be cancelled: it is left behind and the process is expected to exit).
      struct InitState {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        int rc = 0;
        void* comm = nullptr;
      };
      auto st = std::make_shared<InitState>();
      auto init_fn = c->rccl.CommInitRank;
      std::thread([st, init_fn, nranks, id, rank, device] {
        (void)hipSetDevice(device);
        void* comm = nullptr;
        const int rc_init = init_fn(&comm, nranks, id, rank);
        std::lock_guard<std::mutex> lock(st->mu);
        st->rc = rc_init; st->comm = comm; st->done = true;
        st->cv.notify_all();
      }).detach();
      bool finished;
      {
        std::unique_lock<std::mutex> lock(st->mu);
        finished = st->cv.wait_for(lock, std::chrono::duration<double>(timeout_s), [&] { return st->done; });
      }
      if (!finished) {
        // the communicator object stays allocated: the helper thread may still touch what it was given
        // ... but it never sees the sockets or the stream: those go.  The caller must treat this as fatal and let the
        // process EXIT (non-zero) -- never re-exec it: a thread is still inside RCCL and holds the GPU.
        for (int fd : c->fds)
          if (fd >= 0) ::close(fd);
        c->fds.clear();
        if (c->listen_fd >= 0) { ::close(c->listen_fd); c->listen_fd = -1; }
        if (c->stream) { (void)hipStreamDestroy(c->stream); c->stream = nullptr; }
        return cfail(PAOS_EHIP, "ncclCommInitRank did not return within " + std::to_string((int)timeout_s) +
                                    " s (a rank is missing or its device is stuck): giving up");
      }
      if (nccl_check(c, st->rc, "ncclCommInitRank")) { ok = false; why = g_comm_err; c->nccl = nullptr; }
      else c->nccl = st->comm;
      // the vote: bounded too, a rank whose watchdog fired will not take part
      for (int fd : c->fds)
        if (fd >= 0) set_recv_timeout(fd, timeout_s);
      all_ok = everyone_ok(ok);
      for (int fd : c->fds)
        if (fd >= 0) set_recv_timeout(fd, 0.0);
    }
    if (!all_ok) {
      if (c->nccl) { (void)c->rccl.CommDestroy(c->nccl); c->nccl = nullptr; }
      c->transport = PAOS_COMM_SOCKET;
      c->bringup_note = ok ? "RCCL came up here, but not on every rank" : why;
      std::fprintf(stderr, "paos_comm: rank %d of %d continues on the TCP transport, RCCL is not usable on every rank%s%s\n", rank, nranks,
                   ok ? "" : " -- here: ", ok ? "" : why.c_str());
    }
  }
  *out = c;
  return PAOS_OK;
}

int paos_comm_bcast_size(paos_comm* c, unsigned long long* bytes, int root) {
  if (!c || !bytes || root < 0 || root >= c->nranks) return cfail(PAOS_EINVAL, "bad broadcast request");
  return sock_bcast(c, bytes, sizeof(*bytes), root);  // eight bytes: the control plane is the right tool
}

int paos_comm_bcast_blob(paos_comm* c, void* host_buf, unsigned long long bytes, int root) {
  if (!c || (!host_buf && bytes) || root < 0 || root >= c->nranks) return cfail(PAOS_EINVAL, "bad broadcast request");
  if (bytes == 0) return PAOS_OK;
  if (c->transport == PAOS_COMM_SOCKET) return sock_bcast(c, host_buf, (size_t)bytes, root);
  // RCCL also with a single rank: the same calls, so that a one-GPU box exercises the transport
  COMM_HIP(hipSetDevice(c->device));
  int rc = stage(c, (size_t)bytes);
  if (rc) return rc;
  if (c->rank == root) COMM_HIP(hipMemcpyAsync(c->dbuf, host_buf, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  rc = nccl_check(c, c->rccl.Broadcast(c->dbuf, c->dbuf, (size_t)bytes, kNcclUint8, root, c->nccl, c->stream), "ncclBroadcast");
  if (rc) return rc;
  if (c->rank != root) COMM_HIP(hipMemcpyAsync(host_buf, c->dbuf, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  return PAOS_OK;
}

int paos_comm_allgather_scalars(paos_comm* c, const double* send, int count, double* recv) {
  if (!c || count < 0 || (count && (!send || !recv))) return cfail(PAOS_EINVAL, "bad gather request");
  if (count == 0) return PAOS_OK;
  if (c->transport == PAOS_COMM_SOCKET) return sock_allgather(c, send, count, recv);
  COMM_HIP(hipSetDevice(c->device));
  const size_t one = (size_t)count * sizeof(double), all = one * (size_t)c->nranks;
  int rc = stage(c, one + all);
  if (rc) return rc;
  char* d = static_cast<char*>(c->dbuf);
  COMM_HIP(hipMemcpyAsync(d, send, one, hipMemcpyHostToDevice, c->stream));
  rc = nccl_check(c, c->rccl.AllGather(d, d + one, (size_t)count, kNcclFloat64, c->nccl, c->stream), "ncclAllGather");
  if (rc) return rc;
  COMM_HIP(hipMemcpyAsync(recv, d + one, all, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  return PAOS_OK;
}

int paos_comm_allgatherv_scalars(paos_comm* c, const double* send, int count, double* recv,
                                 unsigned long long recv_capacity, int* counts_out) {
  if (!c || count < 0 || !counts_out || (count && !send)) return cfail(PAOS_EINVAL, "bad gather request");
  std::vector<double> mine(1, (double)count), cnt(c->nranks, 0.0);
  int rc = paos_comm_allgather_scalars(c, mine.data(), 1, cnt.data());
  if (rc) return rc;
  std::vector<int> counts(c->nranks);
  size_t total = 0;
  int widest = 0;
  for (int r = 0; r < c->nranks; ++r) { counts[r] = (int)cnt[r]; total += (size_t)counts[r]; widest = std::max(widest, counts[r]); counts_out[r] = counts[r]; }
  if (total > recv_capacity) return cfail(PAOS_EINVAL, "receive buffer too small for the gathered scalars");
  if (total == 0) return PAOS_OK;
  if (!recv) return cfail(PAOS_EINVAL, "null receive buffer");
  if (c->transport == PAOS_COMM_SOCKET) return sock_allgatherv(c, send, count, recv, counts);
  // RCCL: pad every contribution to the widest, one ncclAllGather, compact on the host
  std::vector<double> padded((size_t)widest, 0.0), wide((size_t)widest * c->nranks);
  if (count) std::memcpy(padded.data(), send, (size_t)count * sizeof(double));
  rc = paos_comm_allgather_scalars(c, padded.data(), widest, wide.data());
  if (rc) return rc;
  size_t o = 0;
  for (int r = 0; r < c->nranks; ++r) {
    std::memcpy(recv + o, wide.data() + (size_t)r * widest, (size_t)counts[r] * sizeof(double));
    o += (size_t)counts[r];
  }
  return PAOS_OK;
}

int paos_comm_max(paos_comm* c, double* value) {
  if (!c || !value) return cfail(PAOS_EINVAL, "null argument");
  if (c->transport == PAOS_COMM_SOCKET) {
    if (c->nranks == 1) return PAOS_OK;
    std::vector<double> all(c->nranks);
    int rc = sock_allgather(c, value, 1, all.data());
    if (rc) return rc;
    *value = *std::max_element(all.begin(), all.end());
    return PAOS_OK;
  }
  COMM_HIP(hipSetDevice(c->device));
  int rc = stage(c, 2 * sizeof(double));
  if (rc) return rc;
  double* d = static_cast<double*>(c->dbuf);
  COMM_HIP(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
  rc = nccl_check(c, c->rccl.AllReduce(d, d + 1, 1, kNcclFloat64, kNcclMax, c->nccl, c->stream), "ncclAllReduce");
  if (rc) return rc;
  COMM_HIP(hipMemcpyAsync(value, d + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  return PAOS_OK;
}

int paos_comm_barrier(paos_comm* c) {
  double x = 0.0;
  return paos_comm_max(c, &x);
}

}  // extern "C"
