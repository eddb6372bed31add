// membench6 (round 4): does a CU run faster when its two resident workgroups are of DIFFERENT kinds?
// The pass launches of a SYN20 step alternate between memory-bound ones (every element read and written, 3.45 ms) and
// fp64-bound ones (three quarters of the loads skipped, 2.87 ms).  A launch fills every CU with two workgroups of ONE
// kind; two half-batches one pass apart could put one workgroup of each kind on every CU (one "fat" launch whose
// workgroups alternate kinds).  Model: the pass's column tile (512 threads x 16 elements of 16 B, whole 128-byte lines a
// block-row pitch apart, two workgroups per CU), all loads -> a spin of fp64 FMAs -> all stores; kind V loads only
// elements 6..9 (the live quarter) of its 16.  Compared: a launch of kind M only, one of kind V only, and one whose
// workgroups alternate M, V, M, V ... (same tiles, same total work as half of each).
//   hipcc -O3 --offload-arch=gfx950 tools/membench6.hip -o build/membench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kN = 4096, kBlocksPerRow = kN / 2, kBlockRows = kN / 4, kPad = 3, kBatch = 16;

struct Args {
  v2d* buf;
  int spin_m, spin_v;
  int mode;  // 0: every workgroup kind M, 1: every workgroup kind V, 2: alternate by workgroup, 3: alternate by item
  double fa, fb;
};

__global__ void __launch_bounds__(512) tile_kernel(Args a) {
  extern __shared__ unsigned char smem[];
  if (a.spin_m < 0) smem[threadIdx.x] = 0;
  const int tile = blockIdx.x, item = blockIdx.y, tid = threadIdx.x;
  const bool kind_v = a.mode == 1 || (a.mode == 2 && (tile & 1)) || (a.mode == 3 && (item & 1));
  v2d* d = a.buf + (size_t)item * kBlockRows * (kBlocksPerRow + kPad) * 8;
  const int bc = tid % 2, br = (tid / 2) % 4, q = tid / 8;
  const unsigned rowblocks = (unsigned)(kBlocksPerRow + kPad);
  const unsigned base = ((unsigned)q * rowblocks + tile) * 8u + br * 2 + bc;
  const unsigned stride = 64u * rowblocks * 8u;
  v2d v[16];
  if (kind_v) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (k >= 6 && k < 10) ? __builtin_nontemporal_load(&d[(size_t)base + (size_t)k * stride]) : v2d{0.0, 0.0};
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(&d[(size_t)base + (size_t)k * stride]);
  }
  const int spin = kind_v ? a.spin_v : a.spin_m;
  for (int i = 0; i < spin; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = v[k] * a.fa + a.fb;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v[k], &d[(size_t)base + (size_t)k * stride]);
}

static v2d* gbuf;
static int greps = 10;

static float run(int mode, int spin_m, int spin_v) {
  CK(hipFuncSetAttribute((const void*)tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
  Args a{gbuf, spin_m, spin_v, mode, 1.0000001, 1e-9};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto f = [&] { hipLaunchKernelGGL(tile_kernel, dim3(2048, kBatch), dim3(512), 70 * 1024, 0, a); };
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < greps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms / greps;
}

int main(int argc, char** argv) {
  greps = argc > 1 ? atoi(argv[1]) : 10;
  const size_t n = (size_t)kBatch * kBlockRows * (kBlocksPerRow + kPad) * 8;
  CK(hipMalloc(&gbuf, n * 16));
  CK(hipMemset(gbuf, 0, n * 16));
  printf("# %d items of 4096^2 complex128 (%.2f GB read + written by a launch of kind M); ms per launch\n", kBatch, 2.0 * kBatch * kN * kN * 16 * 1e-9);
  printf("# spin: iterations of 32 fp64 FMAs per thread between loads and stores (55 ~ a two-transform pass)\n");
  for (int rep = 0; rep < 2; ++rep)
    for (int spin : {0, 30, 55, 70}) {
      const float tm = run(0, spin, spin), tv = run(1, spin, spin), tx = run(2, spin, spin), ti = run(3, spin, spin);
      printf("spin %2d: all M %7.3f | all V %7.3f | mean of the two %7.3f | alternating by workgroup %7.3f (%+.1f %%) | by item %7.3f (%+.1f %%)\n", spin, tm, tv,
             0.5f * (tm + tv), tx, 100.0 * (tx / (0.5 * (tm + tv)) - 1.0), ti, 100.0 * (ti / (0.5 * (tm + tv)) - 1.0));
      fflush(stdout);
    }
  // a memory-only kind next to an arithmetic-heavy one: the most a mix could give
  for (int spin_v : {60, 100}) {
    const float tm = run(0, 0, 0), tv = run(1, spin_v, spin_v), tx = run(2, 0, spin_v);
    printf("M without arithmetic, V spin %3d: all M %7.3f | all V %7.3f | mean %7.3f | alternating %7.3f (%+.1f %%)\n", spin_v, tm, tv, 0.5f * (tm + tv), tx,
           100.0 * (tx / (0.5 * (tm + tv)) - 1.0));
  }
  return 0;
}
