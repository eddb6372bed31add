// fftbench -- standalone micro-benchmark of the FFT passes (tile-shape exploration).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I paos_amd/csrc tools/fftbench.hip -o gpurun_out/fftbench
// Prints one line per variant: time per launch, algorithmic GB/s (read + write of
// every element once), and a forward->inverse round-trip error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_kernels.h"

using namespace paos;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <typename T>
__global__ void copy_kernel(cx<T>* f, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    cx<T> v = f[i];
    v.x += (T)1e-30;
    f[i] = v;
  }
}

template <typename T, int N>
std::vector<std::complex<T>> make_twiddles() {
  std::vector<std::complex<T>> tw(N);
  for (int m = 0; m < N; ++m) {
    long double a = -2.0L * 3.14159265358979323846264338327950288L * m / N;
    tw[m] = std::complex<T>((T)cosl(a), (T)sinl(a));
  }
  return tw;
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <typename F>
  float run(F&& f, int reps) {
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
  }
};

template <typename T, int N, int E, int LINES, int AXIS, int BR, int BC, bool SPLIT, int MINW = 1,
          int TILES = 1>
void bench_variant(const char* name, int batch, int reps, int pad_blocks = 0) {
  const unsigned pitch = (unsigned)N * BR + (unsigned)pad_blocks * BR * BC;
  const unsigned item_stride = pitch * (N / BR);
  const size_t elems = (size_t)N * N * batch;
  cx<T>* d;
  CK(hipMalloc(&d, (size_t)item_stride * batch * sizeof(cx<T>)));
  CK(hipMemset(d, 0, (size_t)item_stride * batch * sizeof(cx<T>)));
  std::vector<std::complex<T>> h((size_t)N * N);
  srand(1);
  for (auto& z : h) z = std::complex<T>((T)(rand() / (double)RAND_MAX - 0.5), (T)(rand() / (double)RAND_MAX - 0.5));
  // the layout is a permutation: a dense copy into the first rows is as good as any data
  for (int b = 0; b < batch; ++b)
    for (int r = 0; r < N / BR; ++r)
      CK(hipMemcpy(d + (size_t)b * item_stride + (size_t)r * pitch, h.data() + (size_t)r * N * BR,
                   (size_t)N * BR * sizeof(cx<T>), hipMemcpyHostToDevice));
  auto tw = make_twiddles<T, N>();
  cx<T>* dtw;
  CK(hipMalloc(&dtw, N * sizeof(cx<T>)));
  CK(hipMemcpy(dtw, tw.data(), N * sizeof(cx<T>), hipMemcpyHostToDevice));

  FftPassArgs a{};
  a.field = d; a.tw = dtw; a.params = nullptr; a.pre_mode = 0; a.post_mode = 0; a.scale = 1.0;
  a.pitch = pitch; a.item_stride = item_stride;
  const dim3 grid(N / LINES / TILES, batch), block(TILES * LINES * N / E);
  const size_t lds = TILES * LINES * line_lds_bytes<T, N, SPLIT>();
  auto kf = fft_pass_kernel<T, N, E, LINES, TILES, AXIS, BR, BC, SPLIT, +1, MINW>;
  auto ki = fft_pass_kernel<T, N, E, LINES, TILES, AXIS, BR, BC, SPLIT, -1, MINW>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

  // round trip on item 0
  a.scale = 1.0;
  hipLaunchKernelGGL(kf, dim3(N / LINES / TILES, 1), block, lds, 0, a);
  a.scale = 1.0 / N;
  hipLaunchKernelGGL(ki, dim3(N / LINES / TILES, 1), block, lds, 0, a);
  CK(hipDeviceSynchronize());
  std::vector<std::complex<T>> back((size_t)N * N);
  for (int r = 0; r < N / BR; ++r)
    CK(hipMemcpy(back.data() + (size_t)r * N * BR, d + (size_t)r * pitch, (size_t)N * BR * sizeof(cx<T>),
                 hipMemcpyDeviceToHost));
  double err = 0;
  for (size_t i = 0; i < back.size(); ++i) err = fmax(err, (double)std::abs(back[i] - h[i]));

  a.scale = 1.0 / 64;  // keep magnitudes bounded over repeated launches
  Timer tm;
  float ms = tm.run([&] { hipLaunchKernelGGL(kf, grid, block, lds, 0, a); }, reps);
  const double bytes = 2.0 * elems * sizeof(cx<T>);
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kf, block.x, lds));
  printf("%-30s N=%d b=%d E=%d L=%dx%d ax=%d blk=%dx%d pad=%d split=%d minw=%d thr=%d lds=%zuK occ=%d  %8.3f ms  %7.1f GB/s  rt_err=%.2e\n",
         name, N, batch, E, LINES, TILES, AXIS, BR, BC, pad_blocks, (int)SPLIT, MINW, block.x, lds / 1024, nb, ms,
         bytes / ms * 1e-6, err);
  fflush(stdout);

  if (AXIS == 1) {  // fused ptp middle pass on the same shape
    std::vector<double> hp(batch * FP_STRIDE);
    for (int b = 0; b < batch; ++b) {
      hp[b * FP_STRIDE + FP_ENABLE] = 1; hp[b * FP_STRIDE + FP_SX] = 0.01; hp[b * FP_STRIDE + FP_SY] = 0.01;
      hp[b * FP_STRIDE + FP_COEF] = 0.37; hp[b * FP_STRIDE + FP_SGN] = -1;
    }
    double* dp;
    CK(hipMalloc(&dp, hp.size() * sizeof(double)));
    CK(hipMemcpy(dp, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
    auto km = fft_ptp_mid_kernel<T, N, E, LINES, TILES, AXIS, BR, BC, SPLIT, MINW>;
    CK(hipFuncSetAttribute((const void*)km, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    a.params = dp; a.scale = 1.0 / N;
    float ms2 = tm.run([&] { hipLaunchKernelGGL(km, grid, block, lds, 0, a); }, reps);
    printf("%-34s   fused fwd*H*inv            %8.3f ms  %7.1f GB/s\n", name, ms2, bytes / ms2 * 1e-6);
    fflush(stdout);
    CK(hipFree(dp));
  }
  CK(hipFree(d));
  CK(hipFree(dtw));
}

template <typename T>
void bench_copy(int n, int batch, int reps) {
  const size_t elems = (size_t)n * n * batch;
  cx<T>* d;
  CK(hipMalloc(&d, elems * sizeof(cx<T>)));
  CK(hipMemset(d, 0, elems * sizeof(cx<T>)));
  Timer tm;
  float ms = tm.run([&] { hipLaunchKernelGGL(copy_kernel<T>, dim3(2048), dim3(256), 0, 0, d, elems); }, reps);
  printf("%-34s N=%d b=%d  %8.3f ms  %7.1f GB/s\n", "inplace copy (yardstick)", n, batch, ms,
         2.0 * elems * sizeof(cx<T>) / ms * 1e-6);
  fflush(stdout);
  CK(hipFree(d));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const int b4 = 4, b2 = 16, b1 = 64;
  bench_copy<double>(4096, b4, reps);
  // ---- 4096 rows
  bench_variant<double, 4096, 16, 1, 0, 1, 1, false>("4096 rows canon", b4, reps);
  bench_variant<double, 4096, 16, 2, 0, 2, 4, true>("4096 rows blk2x4", b4, reps, 0);
  bench_variant<double, 4096, 16, 2, 0, 2, 4, true>("4096 rows blk2x4", b4, reps, 1);
  bench_variant<double, 4096, 16, 2, 0, 2, 4, true, 4>("4096 rows blk2x4", b4, reps, 1);
  bench_variant<double, 4096, 16, 4, 0, 4, 2, true, 4>("4096 rows blk4x2", b4, reps, 1);
  // ---- 4096 cols, pad sweep
  for (int pad : {0, 1, 2, 3, 8, 33}) {
    bench_variant<double, 4096, 32, 4, 1, 2, 4, true>("4096 cols blk2x4 E32", b4, reps, pad);
  }
  for (int pad : {0, 1, 3, 33}) {
    bench_variant<double, 4096, 16, 2, 1, 4, 2, true>("4096 cols blk4x2 E16", b4, reps, pad);
    bench_variant<double, 4096, 16, 2, 1, 4, 2, true, 4>("4096 cols blk4x2 E16", b4, reps, pad);
  }
  bench_variant<double, 4096, 16, 4, 1, 2, 4, true, 4>("4096 cols blk2x4 E16", b4, reps, 1);
  bench_variant<double, 4096, 16, 2, 1, 1, 1, true, 4>("4096 cols canon W2", b4, reps, 0);
  // ---- 2048
  bench_variant<double, 2048, 16, 2, 0, 2, 4, false>("2048 rows blk2x4", b2, reps, 1);
  for (int pad : {0, 1, 3, 33}) {
    bench_variant<double, 2048, 16, 4, 1, 2, 4, false>("2048 cols blk2x4", b2, reps, pad);
  }
  bench_variant<double, 2048, 16, 4, 1, 2, 4, true, 4>("2048 cols blk2x4", b2, reps, 1);
  // ---- 1024
  bench_variant<double, 1024, 16, 2, 0, 2, 4, false>("1024 rows blk2x4", b1, reps, 1);
  bench_variant<double, 1024, 16, 2, 0, 2, 4, false, 1, 2>("1024 rows blk2x4 T2", b1, reps, 1);
  for (int pad : {0, 1, 3}) {
    bench_variant<double, 1024, 16, 4, 1, 2, 4, false>("1024 cols blk2x4", b1, reps, pad);
  }
  bench_variant<double, 1024, 16, 4, 1, 2, 4, false, 4>("1024 cols blk2x4", b1, reps, 1);
  // ---- fp32
  bench_copy<float>(4096, b4, reps);
  bench_variant<float, 4096, 16, 2, 0, 2, 4, false>("4096 f32 rows blk2x4", b4, reps, 1);
  bench_variant<float, 4096, 16, 4, 1, 2, 4, false>("4096 f32 cols blk2x4", b4, reps, 1);
  bench_variant<float, 4096, 16, 4, 1, 2, 4, true, 4>("4096 f32 cols blk2x4", b4, reps, 1);
  return 0;
}
