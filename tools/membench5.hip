// membench5: would workgroups that WALK the tiles beat one workgroup per tile if the walk were software-pipelined?
// Round 2 found persistent workgroups 7-10 % slower than the hardware's dispatch (they fall into lockstep).  Two things that
// experiment did not have: (1) the stores of tile i interleaved with the loads of tile i + 1 (a register is reloaded as soon as
// it has been stored, so a tile's life has ONE memory phase instead of a store phase, a ~5 us dispatch gap and a load phase),
// (2) the two workgroups of a CU started half a period apart (the second one -- LDS base != 0 -- sleeps first).
// Model: 128 KiB tiles (512 threads x 16 x 16 B, contiguous), `SPIN` x 32 dependent fp64 FMAs per thread per tile (48 = the
// instruction count of a two-transform pass), 4 GiB buffer, in place.
//   hipcc -O3 --offload-arch=gfx950 tools/membench5.hip -o build/membench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));

template <int SPIN>
__device__ __forceinline__ void work(v2* v, double a) {
  for (int s = 0; s < SPIN; ++s) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k].x = fma(v[k].x, a, v[k].y); v[k].y = fma(v[k].y, a, -v[k].x); }
  }
}

// MODE 0: one workgroup per tile.  1: walk, plain loop.  2: walk, stores of tile i interleaved with loads of tile i + 1.
// 3: as 2, and the workgroup that sits in the upper half of its CU's LDS starts `delay` x s_sleep(127) later.
template <int MODE, int SPIN>
__global__ void __launch_bounds__(512, 4) tiles(v2* buf, unsigned ntiles, double a, int delay) {
  extern __shared__ double smem[];
  v2 v[16];
  // wave-uniform tile base (scalar registers) + one 32-bit byte offset per thread, like the pass kernels
  const unsigned voff = threadIdx.x * (unsigned)sizeof(v2);
  auto at = [&](unsigned tile, int k) {
    char* tb = reinterpret_cast<char*>(buf) + (size_t)tile * 131072 + (size_t)k * 8192;
    return reinterpret_cast<v2*>(tb + voff);
  };
  if (MODE == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *at(blockIdx.x, k);
    work<SPIN>(v, a);
#pragma unroll
    for (int k = 0; k < 16; ++k) *at(blockIdx.x, k) = v[k];
  } else {
    if (MODE == 3) {
      const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6);  // HW_REG_LDS_ALLOC.LDS_BASE
      if (lds_base != 0)
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(127);
    }
    const unsigned rounds = ntiles / gridDim.x;
    unsigned tile = blockIdx.x;
    if (MODE >= 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = *at(tile, k);
    }
    for (unsigned r = 0; r < rounds; ++r) {
      const unsigned next = tile + gridDim.x;  // this workgroup's next tile
      if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *at(tile, k);
      }
      work<SPIN>(v, a);
      if (MODE == 1 || r + 1 == rounds) {
#pragma unroll
        for (int k = 0; k < 16; ++k) *at(tile, k) = v[k];
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) { *at(tile, k) = v[k]; v[k] = *at(next, k); __builtin_amdgcn_sched_barrier(0); }
      }
      tile = next;
    }
  }
  if (a == 12345.0) smem[threadIdx.x] = v[0].x;
}

template <int MODE, int SPIN>
static void run(v2* d, size_t mib, const char* what, int delay = 0) {
  const unsigned ntiles = (unsigned)(mib * 8);
  auto k = tiles<MODE, SPIN>;
  const size_t lds = 70 * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(MODE == 0 ? ntiles : 512), dim3(512), lds, 0, d, ntiles, 1.0000001, delay);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    if (rep && ms < best) best = ms;
  }
  const double gb = 2.0 * mib * 1048576.0 / 1e9;
  printf("%-72s spin %2d  %8.4f ms  %8.1f GB/s  (%.3f of 8 TB/s)\n", what, SPIN, best, gb / best * 1e3, gb / best * 1e3 / 8000.0);
  fflush(stdout);
}

template <int SPIN>
static void all(v2* d, size_t mib) {
  run<0, SPIN>(d, mib, "one workgroup per tile (hardware dispatch)");
  run<1, SPIN>(d, mib, "512 workgroups walk the tiles: load | work | store");
  run<2, SPIN>(d, mib, "  walk, stores of tile i interleaved with loads of tile i + 1");
  run<3, SPIN>(d, mib, "  + second workgroup of a CU starts 1 x s_sleep(127) late", 1);
  run<3, SPIN>(d, mib, "  + second workgroup of a CU starts 2 x s_sleep(127) late", 2);
  run<3, SPIN>(d, mib, "  + second workgroup of a CU starts 4 x s_sleep(127) late", 4);
}

int main() {
  const size_t mib = 4096;
  v2* d; CK(hipMalloc(&d, mib << 20)); CK(hipMemset(d, 0, mib << 20));
  all<1>(d, mib);
  all<24>(d, mib);
  all<48>(d, mib);
  all<64>(d, mib);
  return 0;
}
