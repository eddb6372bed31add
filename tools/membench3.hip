// membench3: how fast is an in-place read-modify-write sweep when the buffer fits the 256 MiB Infinity Cache, and how does the
// rate fall off around that size?  One workgroup of 512 threads moves one 128 KiB chunk (16 x 16 B per thread, all loads, then all
// stores -- the shape of a pass tile); a "pass" sweeps the whole buffer; even passes walk the chunks in address order, odd passes
// in a strided order (chunk c -> (c * 257) mod nchunks, the way a column pass follows a row pass).  Time = mean over 20 passes
// launched back to back on one stream.
//   hipcc -O3 --offload-arch=gfx950 tools/membench3.hip -o build/membench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));

template <int SPIN>
__global__ void __launch_bounds__(512, 4) sweep(v2* buf, unsigned nchunks, unsigned mul, double a, unsigned first = 0) {
  extern __shared__ double smem[];
  const unsigned c = first + (unsigned)(((unsigned long long)blockIdx.x * mul) % nchunks);
  v2* p = buf + (size_t)c * 8192 + threadIdx.x;
  v2 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = p[k * 512];
  for (int s = 0; s < SPIN; ++s) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k].x = fma(v[k].x, a, v[k].y); v[k].y = fma(v[k].y, a, -v[k].x); }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) p[k * 512] = v[k];
  if (a == 12345.0) smem[threadIdx.x] = v[0].x;
}

template <int SPIN>
static void run(v2* d, size_t mib, int alternate, int lds_k) {
  const unsigned nchunks = (unsigned)(mib * 8);  // 128 KiB chunks
  auto k = sweep<SPIN>;
  const size_t lds = (size_t)lds_k * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int passes = 20;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(nchunks), dim3(512), lds, 0, d, nchunks, 1u, 1.0, 0u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < passes; ++i)
    hipLaunchKernelGGL(k, dim3(nchunks), dim3(512), lds, 0, d, nchunks, (alternate && (i & 1)) ? 257u : 1u, 1.0, 0u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  const double per = ms / passes, gb = 2.0 * mib * 1048576.0 / 1e9;
  printf("%5zu MiB  %s  spin %3d  lds %3dK   %8.4f ms per pass   %8.1f GB/s  (%.3f of 8 TB/s)   %.1f us per MiB\n", mib,
         alternate ? "alternating order" : "address order    ", SPIN, lds_k, per, gb / per * 1e3, gb / per * 1e3 / 8000.0, per * 1e3 / mib);
}

// the same launches (grid of `mib` MiB worth of chunks), but every launch takes the NEXT window of a 2 GiB buffer: same tails and
// launch gaps as the resident case, nothing found on-die
template <int SPIN>
static void run_windows(v2* d, size_t mib, size_t total_mib) {
  const unsigned nchunks = (unsigned)(mib * 8), nwin = (unsigned)(total_mib / mib);
  auto k = sweep<SPIN>;
  const size_t lds = 70 * 1024;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int passes = 40;
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < passes; ++i)
    hipLaunchKernelGGL(k, dim3(nchunks), dim3(512), lds, 0, d, nchunks, (i & 1) ? 257u : 1u, 1.0, (unsigned)(i % nwin) * nchunks);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  const double per = ms / passes, gb = 2.0 * mib * 1048576.0 / 1e9;
  printf("%5zu MiB windows of %zu MiB  spin %3d   %8.4f ms per launch   %8.1f GB/s  (%.3f of 8 TB/s)\n", mib, total_mib, SPIN, per, gb / per * 1e3,
         gb / per * 1e3 / 8000.0);
}

int main() {
  const size_t max_mib = 2048;
  v2* d; CK(hipMalloc(&d, max_mib << 20)); CK(hipMemset(d, 0, max_mib << 20));
  printf("# in-place sweep of a buffer of the given size, 20 passes back to back (MI355X Infinity Cache: 256 MiB)\n");
  for (int alt = 0; alt < 2; ++alt)
    for (size_t mib : {16, 32, 64, 128, 192, 224, 240, 256, 258, 272, 320, 384, 512, 1024, 2048}) run<1>(d, mib, alt, 70);
  printf("# with arithmetic between the loads and the stores (spin x 32 fp64 FMAs per thread; a two-transform pass issues ~1500)\n");
  for (size_t mib : {64, 128, 256, 2048}) { run<20>(d, mib, 1, 70); run<40>(d, mib, 1, 70); }
  printf("# around the capacity, spin 48 (the instruction count of a two-transform pass)\n");
  for (size_t mib : {128, 192, 224, 240, 248, 252, 256, 257, 258, 260, 264, 272, 288, 320, 384, 512, 2048}) run<48>(d, mib, 1, 70);
  printf("# same launch size, resident or not\n");
  for (size_t mib : {64, 128, 256}) { run<48>(d, mib, 1, 70); run_windows<48>(d, mib, 2048); }
  return 0;
}
