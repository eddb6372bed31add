// dispatchbench (round 4): how long does a CU slot stay empty between two workgroups?  tools/timeline.hip sees one
// workgroup resident on a CU for 24 % of its busy span during a pass launch; the pass's own prologue is not it
// (profiles/r04_ab_variants_bench.txt).  Here every workgroup does nothing but wait T microseconds on the real-time
// counter and leave; R rounds of 2 workgroups per CU: what a launch takes beyond R x T, per round, is the cost of
// replacing a workgroup -- as a function of its size (threads), its LDS allocation and its registers.
//   hipcc -O3 --offload-arch=gfx950 tools/dispatchbench.hip -o build/dispatchbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS, int MINW>
__global__ void __launch_bounds__(THREADS, MINW) wait_kernel(unsigned long long ticks, double* sink, int touch_lds, int stores, double* out) {
  extern __shared__ unsigned char smem[];
  if (touch_lds) smem[threadIdx.x] = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (stores) {  // a store phase like the pass's: 16 x 16 B per thread
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d* o = reinterpret_cast<v2d*>(out) + (size_t)(blockIdx.x % 65536) * THREADS * 16 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v2d{(double)k, 1.0}, o + (size_t)k * THREADS);
  }
  if (ticks == 0xffffffffffffffffull) sink[0] = (double)smem[0];
}

template <int THREADS, int MINW>
static void run(const char* name, int lds_kib, double t_us, int rounds, int stores, double* out) {
  auto k = wait_kernel<THREADS, MINW>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, THREADS, lds_kib * 1024));
  const int wgs = 256 * per_cu * rounds;
  const unsigned long long ticks = (unsigned long long)(t_us * 100.0);  // s_memrealtime: 100 MHz
  double* sink;
  CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(THREADS), lds_kib * 1024, 0, ticks, sink, 1, stores, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(THREADS), lds_kib * 1024, 0, ticks, sink, 1, stores, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double ideal = rounds * t_us * 1e-3;
  const double store_us = stores ? 256.0 * per_cu * THREADS * 256.0 / 6.0e6 : 0.0;  // the round's stores at 6 TB/s
  printf("%-28s thr=%4d lds=%3dK occ=%d/CU wait=%5.1f us x %3d rounds: %7.3f ms, ideal %7.3f -> %6.2f us per round beyond the wait%s\n", name, THREADS, lds_kib,
         per_cu, t_us, rounds, ms, ideal, (ms - ideal) * 1e3 / rounds, stores ? " (the stores themselves need ~this at 6 TB/s:" : "");
  if (stores) printf("%92s %6.2f us)\n", "", store_us);
  fflush(stdout);
  CK(hipFree(sink)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  double* out;
  CK(hipMalloc(&out, (size_t)65536 * 1024 * 16 * 16));
  for (int stores = 0; stores < 2; ++stores) {
    printf("## %s\n", stores ? "with a store phase (16 x 16 B per thread) before leaving" : "wait only");
    for (double t : {5.0, 20.0}) {
      run<512, 4>("512 thr, 128 VGPR, 76K LDS", 76, t, 64, stores, out);
      run<512, 4>("512 thr, 128 VGPR, 38K LDS", 38, t, 64, stores, out);
      run<512, 4>("512 thr, 128 VGPR,  1K LDS", 1, t, 64, stores, out);
      run<256, 4>("256 thr, 128 VGPR, 38K LDS", 38, t, 64, stores, out);
      run<256, 4>("256 thr, 128 VGPR,  1K LDS", 1, t, 64, stores, out);
      run<1024, 4>("1024 thr, 128 VGPR, 76K LDS", 76, t, 64, stores, out);
      run<64, 4>("64 thr,  1K LDS", 1, t, 64, stores, out);
    }
  }
  return 0;
}
