// xposebench: the exchange INSIDE a group of 16 lanes that the 16 x [16 x 16] decomposition of a 4096-point
// transform needs between its second and third radix-16 stage (fft_core.h: fft4096_nat_to_swapped, X2) -- a
// 16 x 16 transposition between the register index and the low four lane bits of complex128 values -- done
//   (a) through LDS without a barrier (what the library's digit-swapped variant does: 17-padded rows), and
//   (b) with wavefront shuffles only (north star: "wavefront __shfl butterflies for the inner radices"):
//       four xor-butterfly steps, each swapping half of the registers with the lane 1 << s away.
// Count per thread and transposition: (a) 32 ds_write_b64 + 32 ds_read_b64 (re and im in turn), a dozen VALU;
// (b) 4 steps x 8 register pairs x 4 dwords = 128 cross-lane moves (DPP where the compiler finds one, else
// ds_bpermute_b32 / ds_swizzle_b32, which run on the LDS pipe as well) + 256 v_cndmask.  An FFT stage is
// O(n) data movement; a transposition by butterflies is O(n log n).  This bench measures both at the occupancy of
// the pass kernels (512-thread workgroups, two per CU) with a radix-16-sized block of fp64 FMAs in between so
// that the exchange competes with arithmetic as it does in the transform.
//   hipcc -O3 --offload-arch=gfx950 tools/xposebench.hip -o build/xposebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct cx { double x, y; };
constexpr int kRow = 272;  // 16 * 17 doubles per group of 16 lanes

__device__ __forceinline__ void fma_block(cx* v, double a, double b, int reps) {
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[r].x = fma(v[r].x, a, b); v[r].y = fma(v[r].y, a, -b); }
  }
}

// (a) slot r of lane t0 <-> slot t0 of lane r, through this group's private LDS row
__device__ __forceinline__ void xpose_lds(cx* v, double* row, int t0) {
#pragma unroll
  for (int part = 0; part < 2; ++part) {
#pragma unroll
    for (int r = 0; r < 16; ++r) row[t0 + r * 17] = part ? v[r].y : v[r].x;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const double val = row[t0 * 17 + r];
      if (part) v[r].y = val; else v[r].x = val;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// (b) the same permutation by four xor steps
__device__ __forceinline__ void xpose_shfl(cx* v, int t0) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int m = 1 << s;
    const bool up = (t0 & m) != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r & m) continue;
      const cx lo = v[r], hi = v[r | m];
      const double gx = up ? lo.x : hi.x, gy = up ? lo.y : hi.y;  // what goes to the partner
      const double rx = __shfl_xor(gx, m, 64), ry = __shfl_xor(gy, m, 64);
      v[r] = {up ? rx : lo.x, up ? ry : lo.y};
      v[r | m] = {up ? hi.x : rx, up ? hi.y : ry};
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(512, 4) bench(cx* out, double a, double b, int iters, int fma_reps) {
  extern __shared__ double smem[];
  const int t = threadIdx.x, t0 = t & 15, grp = t >> 4;
  double* row = smem + grp * kRow;
  cx v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = {(double)(t * 16 + r), (double)(r - t)};
  for (int i = 0; i < iters; ++i) {
    fma_block(v, a, b, fma_reps);
    if (MODE == 0) xpose_lds(v, row, t0);
    if (MODE == 1) xpose_shfl(v, t0);
  }
  cx acc = {0, 0};
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc.x += v[r].x * (r + 1); acc.y += v[r].y; }
  out[(size_t)blockIdx.x * blockDim.x + t] = acc;
}

template <int MODE>
static float run(cx* d, int blocks, int iters, int fma_reps, double a, double b) {
  auto k = bench<MODE>;
  const size_t lds = 70 * 1024;  // two workgroups per CU, like the pass kernels
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, d, a, b, iters, fma_reps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, d, a, b, iters, fma_reps);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms;
}

int main() {
  const int blocks = 512 * 8, iters = 64;
  cx* d; CK(hipMalloc(&d, (size_t)blocks * 512 * sizeof(cx)));
  // correctness: both modes must produce the same values with a = 1, b = 0 (pure permutation, applied an even number
  // of times = identity; compare the two modes after an odd count)
  cx *h0 = (cx*)malloc((size_t)512 * sizeof(cx)), *h1 = (cx*)malloc((size_t)512 * sizeof(cx));
  run<0>(d, 1, 3, 0, 1.0, 0.0); CK(hipMemcpy(h0, d, 512 * sizeof(cx), hipMemcpyDeviceToHost));
  run<1>(d, 1, 3, 0, 1.0, 0.0); CK(hipMemcpy(h1, d, 512 * sizeof(cx), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 512; ++i) bad += (h0[i].x != h1[i].x) || (h0[i].y != h1[i].y);
  printf("permutations agree: %s\n", bad ? "NO" : "yes");
  printf("# %d workgroups of 512 threads (2 per CU), %d iterations of [fma block x reps | 16x16 in-group transposition of 16 complex128 per lane]\n", blocks, iters);
  for (int reps : {0, 2, 4}) {
    const float none = run<2>(d, blocks, iters, reps, 1.0000001, 1e-9);
    const float lds = run<0>(d, blocks, iters, reps, 1.0000001, 1e-9);
    const float shf = run<1>(d, blocks, iters, reps, 1.0000001, 1e-9);
    const double per = 1e3 / iters / (blocks / 512.0);  // us per transposition and resident wave set
    printf("fma reps %d (%3d fp64 instructions per thread): arithmetic alone %7.3f ms | + LDS exchange %7.3f ms (+%.3f us each) | + shuffle butterflies %7.3f ms (+%.3f us each)  ratio %.2f\n",
           reps, reps * 32, none, lds, (lds - none) * per, shf, (shf - none) * per, (shf - none) / (lds - none));
  }
  return 0;
}
