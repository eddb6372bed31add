// membench4: does HBM pay for MIXING reads and writes?  membench2 measured read-only sweeps at 6.4-6.5 TB/s, write-only at 6.2,
// and the in-place read-then-write tile pattern of a pass at 5.5-5.7.  Here 512 resident workgroups (two per CU, 512 threads x
// 16 x 16 B = a 128 KiB tile each, like a pass) walk the buffer in rounds, and in the PHASED variant a device-wide barrier
// separates "everybody loads" from "everybody stores": the memory system then sees 64 MiB of reads, then 64 MiB of writes.
// If the mixing itself costs the ~12 %, the phased walk should approach 2 / (1 / 6.45 + 1 / 6.2) = 6.3 TB/s (minus the
// barrier's own cost, which the third variant measures: same barriers, loads and stores NOT separated by them).
// Every wait is bounded (a workgroup that is not co-resident would otherwise hang the grid): a timed-out barrier is counted
// and reported, and the walk goes on.
//   hipcc -O3 --offload-arch=gfx950 tools/membench4.hip -o build/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target, unsigned* timeouts) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int bound = 200000;  // ~0.1 s at worst
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && --bound > 0) __builtin_amdgcn_s_sleep(2);
    if (bound <= 0) atomicAdd(timeouts, 1u);
  }
  __syncthreads();
}

// MODE 0: free-running walk; 1: barrier between loads and stores and between stores and the next loads (phased);
// 2: the same two barriers per round, but both in front of the loads (cost of the barriers alone)
template <int MODE, int SPIN>
__global__ void __launch_bounds__(512, 4) walk(v2* buf, unsigned ntiles, double a, unsigned* ctr, unsigned* timeouts) {
  extern __shared__ double smem[];
  const unsigned rounds = ntiles / gridDim.x;
  unsigned phase = 0;
  for (unsigned r = 0; r < rounds; ++r) {
    v2* p = buf + ((size_t)r * gridDim.x + blockIdx.x) * 8192 + threadIdx.x;
    if (MODE == 2) { grid_barrier(ctr, ++phase * gridDim.x, timeouts); grid_barrier(ctr, ++phase * gridDim.x, timeouts); }
    if (MODE == 1) grid_barrier(ctr, ++phase * gridDim.x, timeouts);
    v2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = p[k * 512];
    for (int s = 0; s < SPIN; ++s) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { v[k].x = fma(v[k].x, a, v[k].y); v[k].y = fma(v[k].y, a, -v[k].x); }
    }
    if (MODE == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loads have landed before the barrier says so
      grid_barrier(ctr, ++phase * gridDim.x, timeouts);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k * 512] = v[k];
  }
  if (a == 12345.0) smem[threadIdx.x] = 1.0;
}

template <int MODE, int SPIN>
static void run(v2* d, size_t mib, unsigned* dctr, const char* what) {
  const unsigned ntiles = (unsigned)(mib * 8);
  auto k = walk<MODE, SPIN>;
  const size_t lds = 70 * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  unsigned timeouts = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(dctr, 0, 2 * sizeof(unsigned)));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(512), dim3(512), lds, 0, d, ntiles, 1.0000001, dctr, dctr + 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    unsigned h[2]; CK(hipMemcpy(h, dctr, sizeof h, hipMemcpyDeviceToHost));
    timeouts += h[1];
    if (rep && ms < best) best = ms;
  }
  const double gb = 2.0 * mib * 1048576.0 / 1e9;
  printf("%-58s spin %2d  %5zu MiB  %8.4f ms  %8.1f GB/s  (%.3f of 8 TB/s)  barrier timeouts %u\n", what, SPIN, mib, best, gb / best * 1e3,
         gb / best * 1e3 / 8000.0, timeouts);
}

int main() {
  const size_t mib = 4096;  // 4.29 GB read + 4.29 GB written: 64 rounds of 512 tiles
  v2* d; CK(hipMalloc(&d, mib << 20)); CK(hipMemset(d, 0, mib << 20));
  unsigned* dctr; CK(hipMalloc(&dctr, 2 * sizeof(unsigned)));
  run<0, 1>(d, mib, dctr, "512 resident workgroups walk the tiles, free-running");
  run<2, 1>(d, mib, dctr, "  + two device-wide barriers per round (not separating)");
  run<1, 1>(d, mib, dctr, "  phased: all load | barrier | all store | barrier");
  run<0, 24>(d, mib, dctr, "free-running");
  run<2, 24>(d, mib, dctr, "  + two barriers per round");
  run<1, 24>(d, mib, dctr, "  phased");
  run<0, 48>(d, mib, dctr, "free-running");
  run<1, 48>(d, mib, dctr, "  phased");
  return 0;
}
