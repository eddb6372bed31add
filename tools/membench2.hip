// membench2: what does the ACCESS PATTERN of a pass cost, apart from its arithmetic?  (round 3)
// A pass tile = 512 threads x 16 elements of 16 B: all loads, [a stand-in for the transforms], all stores, two
// workgroups per CU (LDS-limited like the real kernel).  Variants: which lines a tile touches (row tiles = 64-B
// halves shared with a sibling workgroup, column tiles = whole 128-B lines one block-row pitch apart, a
// contiguous chunk as the yardstick), nontemporal loads / stores, in place or ping-pong between two buffers,
// "super-block" layouts in which S consecutive block rows of a block column are contiguous (column tiles then
// read S x 128 B runs, row tiles halves S x 128 B apart), read-only / write-only rates, and a spin of
// dependent fp64 FMAs between loads and stores (how much arithmetic hides behind the pattern, and with how
// many workgroups per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/membench2.hip -o build/membench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kN = 4096, kBlocksPerRow = kN / 2, kBlockRows = kN / 4;

struct Args {
  const v2d* src;
  v2d* dst;
  int pad;        // blocks of padding per block row
  int spin;       // iterations of 16 dependent v2d FMAs between loads and stores
  int ntl, nts;   // nontemporal loads / stores
  int mode;       // 0 read + write, 1 read only, 2 write only
  double fa, fb;
  double* sink;
  unsigned* meet;  // != nullptr (row tiles): sibling workgroups (the two halves of a block row) meet before they store
  int meet_spins;  // give up after this many polls
};

// PAT 0: row tile (2 of the 4 rows of a block row, sibling 8 workgroups away), 1: column tile (one block column),
// 2: contiguous chunk of 8192 elements, 3: row tile of 4 rows = whole lines (1024 threads)
template <int PAT, int S>
__device__ __forceinline__ void elem_index(int tile, int tid, int pad, unsigned* base, unsigned* stride) {
  if (PAT == 2) { *base = (unsigned)tile * 8192u + tid; *stride = 512; return; }
  int brow, bcol, e;
  const unsigned rowblocks = (unsigned)(kBlocksPerRow + pad);
  if (PAT == 0) {
    const int grp = tile / 16, in = tile % 16;
    const int t2 = (grp * 8 + in % 8) * 2 + in / 8;
    const int bc = tid % 2, br = (tid / 2) % 2, q = tid / 4;
    brow = t2 / 2; bcol = q; e = ((t2 % 2) * 2 + br) * 2 + bc;
    *stride = 128u * S * 8u;
  } else if (PAT == 1) {
    const int bc = tid % 2, br = (tid / 2) % 4, q = tid / 8;
    brow = q; bcol = tile; e = br * 2 + bc;
    static_assert(64 % S == 0, "k advances whole super-blocks");
    *stride = (64u / S) * rowblocks * S * 8u;
  } else {
    const int bc = tid % 2, br = (tid / 2) % 4, q = tid / 8;
    brow = tile; bcol = q; e = br * 2 + bc;
    *stride = 128u * S * 8u;
  }
  const unsigned blk = ((unsigned)(brow / S) * rowblocks + bcol) * S + (brow % S);
  *base = blk * 8u + e;
}

template <int PAT, int S, int THREADS>
__global__ void __launch_bounds__(THREADS) tile_copy(Args a) {
  extern __shared__ unsigned char smem[];
  if (a.pad < 0) smem[threadIdx.x] = 0;
  const int tile = blockIdx.x, item = blockIdx.y, tid = threadIdx.x;
  const size_t item_off = PAT == 2 ? (size_t)item * kN * kN : (size_t)item * kBlockRows * (kBlocksPerRow + a.pad) * 8;
  const v2d* s = a.src + item_off;
  v2d* d = a.dst + item_off;
  unsigned base, stride;
  elem_index<PAT, S>(tile, tid, a.pad, &base, &stride);
#define idx_k(k) ((size_t)base + (size_t)(k) * stride)
  v2d v[16];
  if (a.mode == 2) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = v2d{(double)k, (double)tid};
  } else if (a.ntl) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(&s[idx_k(k)]);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = s[idx_k(k)];
  }
  for (int i = 0; i < a.spin; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = v[k] * a.fa + a.fb;
  }
  if (PAT == 0 && a.meet) {
    // the two workgroups that share every line of a block row wait for each other (bounded) so that their half-line
    // stores reach the L2 together
    __shared__ int dummy;
    if (tid == 0) {
      const int grp = tile / 16, in = tile % 16;
      unsigned* flag = a.meet + ((size_t)item * 1024 + (grp * 8 + in % 8));
      __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int tries = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u && tries < a.meet_spins) {
        __builtin_amdgcn_s_sleep(8);
        ++tries;
      }
      dummy = tries;
    }
    __syncthreads();
  }
  if (a.mode == 1) {
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y;
    if (acc == 1.2345e-300) a.sink[0] = acc;
  } else if (a.nts) {
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v[k], &d[idx_k(k)]);
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) d[idx_k(k)] = v[k];
  }
}

static v2d *gp, *gq;
static double* gsink;
static unsigned* gmeet;
static int gmeet_spins = 0;
static int greps = 10;
constexpr int kBatch = 8;

template <typename F>
static float timeit(F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < greps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / greps;
}

// out: 0 in place, 1 ping-pong (src/dst swap every launch)
template <int PAT, int S, int THREADS>
static void run(const char* name, int pad, int ntl, int nts, int out, int mode, int spin, int lds_kib) {
  auto k = tile_copy<PAT, S, THREADS>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024));
  const int tiles = PAT == 3 ? 1024 : 2048;
  int flip = 0;
  const float ms = timeit([&] {
    Args a;
    a.src = (out && flip) ? gq : gp;
    a.dst = out ? (flip ? gp : gq) : gp;
    flip ^= 1;
    a.pad = pad; a.spin = spin; a.ntl = ntl; a.nts = nts; a.mode = mode; a.fa = 1.0000001; a.fb = 1e-9; a.sink = gsink;
    a.meet = gmeet_spins > 0 ? gmeet : nullptr; a.meet_spins = gmeet_spins;
    if (a.meet) CK(hipMemsetAsync(gmeet, 0, (size_t)kBatch * 1024 * sizeof(unsigned), 0));
    hipLaunchKernelGGL(k, dim3(tiles, kBatch), dim3(THREADS), lds_kib * 1024, 0, a);
  });
  const double bytes = (mode == 0 ? 2.0 : 1.0) * kBatch * (double)kN * kN * 16;
  printf("%-34s S=%d pad=%2d ntl=%d nts=%d %s %s spin=%3d lds=%3dK  %7.3f ms  %7.1f GB/s\n", name, S, pad, ntl, nts,
         out ? "pingpong" : "inplace ", mode == 0 ? "rw" : (mode == 1 ? "r " : "w "), spin, lds_kib, ms, bytes / ms * 1e-6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  greps = argc > 1 ? atoi(argv[1]) : 10;
  const size_t n = (size_t)kBatch * kBlockRows * (kBlocksPerRow + 72) * 8;
  CK(hipMalloc(&gp, n * 16)); CK(hipMalloc(&gq, n * 16)); CK(hipMalloc(&gsink, 8));
  CK(hipMemset(gp, 0, n * 16)); CK(hipMemset(gq, 0, n * 16));
  CK(hipMalloc(&gmeet, (size_t)kBatch * 1024 * sizeof(unsigned)));
  if (argc > 2 && atoi(argv[2]) == 1) {  // only the sibling-rendezvous experiment
    printf("## row tiles whose sibling workgroups meet before storing (bounded spin), against free-running ones\n");
    for (int sp : {0, 20, 40, 60}) {
      for (int spins : {0, 50, 200, 1000}) {
        gmeet_spins = spins;
        char name[64];
        snprintf(name, sizeof(name), "rows half-lines, meet<=%d polls", spins);
        run<0, 1, 512>(name, 3, 0, 0, 0, 0, sp, 70);
      }
      gmeet_spins = 0;
      run<1, 1, 512>("cols lines nt (reference)", 3, 1, 1, 0, 0, sp, 70);
    }
    return 0;
  }

  printf("## yardstick: contiguous chunks\n");
  for (int out = 0; out < 2; ++out)
    for (int nt = 0; nt < 4; ++nt) run<2, 1, 512>("chunk", 0, nt & 1, nt >> 1, out, 0, 0, 70);
  run<2, 1, 512>("chunk", 0, 0, 0, 0, 1, 0, 70);
  run<2, 1, 512>("chunk", 0, 1, 0, 0, 1, 0, 70);
  run<2, 1, 512>("chunk", 0, 0, 0, 0, 2, 0, 70);
  run<2, 1, 512>("chunk", 0, 0, 1, 0, 2, 0, 70);

  printf("## row tiles, shipped layout (64-B halves, siblings)\n");
  for (int out = 0; out < 2; ++out)
    for (int nt = 0; nt < 4; ++nt) run<0, 1, 512>("rows half-lines", 3, nt & 1, nt >> 1, out, 0, 0, 70);
  run<0, 1, 512>("rows half-lines", 3, 0, 0, 0, 1, 0, 70);
  run<0, 1, 512>("rows half-lines", 3, 0, 0, 0, 2, 0, 70);
  run<0, 1, 512>("rows half-lines", 3, 0, 1, 0, 2, 0, 70);
  printf("## row tiles of 4 rows = whole lines (1024 threads, one workgroup per CU)\n");
  for (int out = 0; out < 2; ++out)
    for (int nt = 0; nt < 4; nt += 3) run<3, 1, 1024>("rows whole lines 1024thr", 3, nt & 1, nt >> 1, out, 0, 0, 140);

  printf("## column tiles, shipped layout (whole lines, stride = pitch)\n");
  for (int out = 0; out < 2; ++out)
    for (int nt = 0; nt < 4; ++nt) run<1, 1, 512>("cols lines", 3, nt & 1, nt >> 1, out, 0, 0, 70);
  run<1, 1, 512>("cols lines", 3, 1, 0, 0, 1, 0, 70);
  run<1, 1, 512>("cols lines", 3, 0, 0, 0, 1, 0, 70);
  run<1, 1, 512>("cols lines", 3, 0, 1, 0, 2, 0, 70);
  run<1, 1, 512>("cols lines", 3, 0, 0, 0, 2, 0, 70);

  printf("## super-block layouts: S block rows of a block column contiguous\n");
  const int pads[] = {0, 1, 3, 5};
  for (int pad : pads) {
    run<1, 2, 512>("cols S=2", pad, 1, 1, 0, 0, 0, 70);
    run<0, 2, 512>("rows S=2", pad, 0, 0, 0, 0, 0, 70);
    run<1, 4, 512>("cols S=4", pad, 1, 1, 0, 0, 0, 70);
    run<0, 4, 512>("rows S=4", pad, 0, 0, 0, 0, 0, 70);
    run<1, 8, 512>("cols S=8", pad, 1, 1, 0, 0, 0, 70);
    run<0, 8, 512>("rows S=8", pad, 0, 0, 0, 0, 0, 70);
  }
  run<1, 8, 512>("cols S=8 pingpong", 1, 1, 1, 1, 0, 0, 70);
  run<0, 8, 512>("rows S=8 pingpong", 1, 0, 0, 1, 0, 0, 70);
  run<1, 4, 512>("cols S=4 pingpong", 1, 1, 1, 1, 0, 0, 70);
  run<0, 4, 512>("rows S=4 pingpong", 1, 0, 0, 1, 0, 0, 70);

  printf("## arithmetic behind the pattern: spin of fp64 FMAs (32 instructions per iteration and thread), workgroups per CU\n");
  const int spins[] = {0, 10, 20, 40, 60};
  for (int lds : {140, 70, 50}) {
    for (int sp : spins) {
      run<2, 1, 512>("chunk nt", 0, 1, 1, 0, 0, sp, lds);
      run<0, 1, 512>("rows half-lines", 3, 0, 0, 0, 0, sp, lds);
      run<1, 1, 512>("cols lines nt", 3, 1, 1, 0, 0, sp, lds);
    }
  }
  printf("## again: yardstick\n");
  run<2, 1, 512>("chunk", 0, 0, 0, 0, 0, 0, 70);
  run<2, 1, 512>("chunk", 0, 1, 1, 0, 0, 0, 70);
  return 0;
}
