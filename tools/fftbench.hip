// fftbench -- standalone micro-benchmark of the FFT passes (tile-shape exploration).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I paos_amd/csrc tools/fftbench.hip -o gpurun_out/fftbench
// Prints one line per variant: time per launch, algorithmic GB/s (read + write of
// every element once), and a forward->inverse round-trip error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "frugal_pass.h"

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
          int TILES = 1, int PW = 0, int SEQ = 1>
void bench_variant(const char* name, int batch, int reps, int pad_blocks = 0) {
  if (batch == 0) return;
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

  // block sets: 0 = forward control, 1 = inverse control, 2 = scale 1/N, 3 = scale 1/64, 4 = H phase
  const int NB = 5;
  std::vector<double> hb((size_t)NB * batch * FP_STRIDE, 0.0);
  for (int i = 0; i < batch; ++i) {
    double* p0 = &hb[((size_t)0 * batch + i) * FP_STRIDE]; p0[0] = 1; p0[1] = 0;
    double* p1 = &hb[((size_t)1 * batch + i) * FP_STRIDE]; p1[0] = 1; p1[1] = 1;
    double* p2 = &hb[((size_t)2 * batch + i) * FP_STRIDE]; p2[0] = 1; p2[3] = 1.0 / N;
    double* p3 = &hb[((size_t)3 * batch + i) * FP_STRIDE]; p3[0] = 1; p3[3] = 1.0 / 64;
    double* p4 = &hb[((size_t)4 * batch + i) * FP_STRIDE]; p4[0] = 1; p4[1] = 0.01; p4[2] = 0.01; p4[3] = 0.37; p4[4] = -1;
  }
  double* dblk;
  CK(hipMalloc(&dblk, hb.size() * sizeof(double)));
  CK(hipMemcpy(dblk, hb.data(), hb.size() * sizeof(double), hipMemcpyHostToDevice));
  PassArgs a{};
  a.field = d; a.tw = dtw; a.blocks = dblk; a.batch = batch; a.fft1 = 0; a.fft2 = -1;
  a.pitch = pitch; a.item_stride = item_stride;
  const dim3 grid(N / LINES / TILES, batch), block(TILES * LINES * N / E / SEQ);
  const size_t lds = TILES * LINES / SEQ * line_lds_bytes<T, N, SPLIT>();
  auto kf = fused_pass_kernel<T, N, E, LINES, TILES, AXIS, BR, BC, SPLIT, MINW, SEQ>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

  // round trip on item 0: forward, then inverse with the 1/N scale
  hipLaunchKernelGGL(kf, dim3(N / LINES / TILES, 1), block, lds, 0, a);
  { PassArgs bk = a; bk.fft1 = 1; bk.n_mid = 1; bk.mid[0] = {PWK_SCALE, 0, 2};
    hipLaunchKernelGGL(kf, dim3(N / LINES / TILES, 1), block, lds, 0, bk); }
  CK(hipDeviceSynchronize());
  std::vector<std::complex<T>> back((size_t)N * N);
  for (int r = 0; r < N / BR; ++r)
    CK(hipMemcpy(back.data() + (size_t)r * N * BR, d + (size_t)r * pitch, (size_t)N * BR * sizeof(cx<T>),
                 hipMemcpyDeviceToHost));
  double err = 0;
  for (size_t i = 0; i < back.size(); ++i) err = fmax(err, (double)std::abs(back[i] - h[i]));

  a.n_mid = 1; a.mid[0] = {PWK_SCALE, 0, 3};  // keep magnitudes bounded over repeated launches
  Timer tm;
  float ms = tm.run([&] { hipLaunchKernelGGL(kf, grid, block, lds, 0, a); }, reps);
  const double bytes = 2.0 * elems * sizeof(cx<T>);
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kf, block.x, lds));
  printf("%-30s N=%d b=%d E=%d SEQ=%d L=%dx%d ax=%d blk=%dx%d pad=%d split=%d minw=%d pw=%d thr=%d lds=%zuK occ=%d  %8.3f ms  %7.1f GB/s  rt_err=%.2e\n",
         name, N, batch, E, SEQ, LINES, TILES, AXIS, BR, BC, pad_blocks, (int)SPLIT, MINW, PW, block.x, lds / 1024, nb, ms,
         bytes / ms * 1e-6, err);
  fflush(stdout);

  if (PW) {  // double pass: fwd | H, 1/N | inv  (the ptp middle pass)
    PassArgs m2 = a;
    m2.fft1 = 0; m2.fft2 = 1; m2.n_mid = 2; m2.mid[0] = {PWK_QPHASE_N, 0, 4}; m2.mid[1] = {PWK_SCALE, 0, 2};
    float ms2 = tm.run([&] { hipLaunchKernelGGL(kf, grid, block, lds, 0, m2); }, reps);
    printf("%-30s     double pass fwd|H|inv      %8.3f ms  %7.1f GB/s\n", name, ms2, bytes / ms2 * 1e-6);
    fflush(stdout);
  }
  CK(hipFree(dblk));
  CK(hipFree(d));
  CK(hipFree(dtw));
}

template <typename T, int N, int AXIS, int KPRE, int KMID, int NFFT, int LINES = 2, int TILES = 1, int E = 16, int BC = 2>
void bench_frugal(const char* name, int batch, int reps, int pad_blocks) {
#ifndef PAOS_BENCH_BR
#define PAOS_BENCH_BR 4
#endif
  constexpr int BR = PAOS_BENCH_BR;
  if (getenv("PAOS_BENCH_BATCH")) batch = atoi(getenv("PAOS_BENCH_BATCH"));
  const unsigned pitch = (unsigned)N * BR + (unsigned)pad_blocks * BR * BC;
  const unsigned item_stride = pitch * (N / BR);
  cx<T>* d;
  CK(hipMalloc(&d, (size_t)item_stride * batch * sizeof(cx<T>)));
  CK(hipMemset(d, 0, (size_t)item_stride * batch * sizeof(cx<T>)));
  std::vector<std::complex<T>> h((size_t)N * N);
  srand(1);
  for (auto& z : h) z = std::complex<T>((T)(rand() / (double)RAND_MAX - 0.5), (T)(rand() / (double)RAND_MAX - 0.5));
  // PAOS_BENCH_FIELD: what the field holds (default: uniform random everywhere).  "beam": a smooth Gaussian beam with a
  // smooth phase, nonzero everywhere; "pupil": random inside the central N/4 x N/4 box, exact zeros outside (a pupil plane
  // of the chain at zoom 4); "zeros".  Same bytes, same instructions: what differs is what the lanes toggle.
  if (const char* fld = getenv("PAOS_BENCH_FIELD")) {
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) {
        std::complex<T>& z = h[(size_t)r * N + c];
        const double y = (r - N / 2) / (double)N, x = (c - N / 2) / (double)N;
        if (!strcmp(fld, "beam")) {
          const double amp = exp(-(x * x + y * y) * 64.0), ph = 40.0 * (x * x + y * y) + 3.0 * x;
          z = std::complex<T>((T)(amp * cos(ph)), (T)(amp * sin(ph)));
        } else if (!strcmp(fld, "pupil")) {
          if (fabs(x) > 0.125 || fabs(y) > 0.125) z = std::complex<T>(0, 0);
        } else if (!strcmp(fld, "zeros")) {
          z = std::complex<T>(0, 0);
        }
      }
  }
  for (int b = 0; b < batch; ++b)
    for (int r = 0; r < N / BR; ++r)
      CK(hipMemcpy(d + (size_t)b * item_stride + (size_t)r * pitch, h.data() + (size_t)r * N * BR,
                   (size_t)N * BR * sizeof(cx<T>), hipMemcpyHostToDevice));
  auto tw = make_twiddles<T, N>();
  cx<T>* dtw;
  CK(hipMalloc(&dtw, N * sizeof(cx<T>)));
  CK(hipMemcpy(dtw, tw.data(), N * sizeof(cx<T>), hipMemcpyHostToDevice));
  std::vector<FrugalItem> items(batch);
  for (auto& it : items) {
    std::memset(&it, 0, sizeof(it));
    it.active = 1; it.fft1_on = 1; it.fft1_inv = 0; it.fft2_on = 1; it.fft2_inv = 1;
    if (getenv("PAOS_BENCH_NOFFT")) it.fft1_on = it.fft2_on = 0;  // tile yardstick: the pass's loads and stores, no transform
    it.pre.scale = 1.0; it.mid.scale = 1.0 / N; it.mid.sign_on = 0;
    it.line_lo = 0; it.line_hi = N; it.pos_lo = 0; it.pos_hi = N; it.spos_lo = 0; it.spos_hi = N;  // no pruning
    // PAOS_BENCH_LIVE=mid|low: only a quarter of the lines is live (a tile-skipping launch without an aperture on it;
    // the GB/s column still counts the whole field)
    if (const char* lv = getenv("PAOS_BENCH_LIVE")) {  // "mid": the central quarter, "low": the first quarter
      if (!strcmp(lv, "mid")) { it.line_lo = N * 3 / 8; it.line_hi = N * 5 / 8; }
      if (!strcmp(lv, "low")) { it.line_lo = 0; it.line_hi = N / 4; }
    }
    for (int j = 0; j < kFrugalMaxPre; ++j) it.pre_ph[j] = {0.01, 0.01, 0.21, 1.0, 1.0, 0.0};
    for (int j = 0; j < kFrugalMaxMid; ++j) it.mid_ph[j] = {0.01, 0.01, 0.37, -1.0, 1.0, 1.0};
  }
  FrugalItem* ditems;
  CK(hipMalloc(&ditems, items.size() * sizeof(FrugalItem)));
  CK(hipMemcpy(ditems, items.data(), items.size() * sizeof(FrugalItem), hipMemcpyHostToDevice));
  std::vector<double> ones((size_t)batch, 1.0);
  double* dones;
  CK(hipMalloc(&dones, ones.size() * sizeof(double)));
  CK(hipMemcpy(dones, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
  FrugalArgs a{d, dtw, ditems, pitch, item_stride};
  a.dyn_scale = dones;
  const dim3 grid(N / LINES / TILES, batch), block(TILES * LINES * N / E);
#ifndef PAOS_F32_SPLIT
#define PAOS_F32_SPLIT 0
#endif
  constexpr bool SPLIT = sizeof(T) == 8 || PAOS_F32_SPLIT || E == 32;
  const size_t lds = frugal_lds_bytes<T, N, LINES, TILES, SPLIT, KPRE, KMID, E>();
  auto kf = frugal_pass_kernel<T, N, E, LINES, TILES, AXIS, BR, BC, SPLIT, KPRE, KMID, NFFT>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  Timer tm;
  float ms = tm.run([&] { hipLaunchKernelGGL(kf, grid, block, lds, 0, PAOS_FRUGAL_PASS(a)); }, reps);
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kf, block.x, lds));
  const double bytes = 2.0 * (double)N * N * batch * sizeof(cx<T>);
  printf("%-40s %s N=%d b=%d ax=%d k=%d,%d nfft=%d E=%d thr=%d lds=%zuK occ=%d  %8.3f ms  %7.1f GB/s\n", name, sizeof(T) == 8 ? "c128" : "c64 ",
         N, batch, AXIS, KPRE, KMID, NFFT, E, block.x, lds / 1024, nb, ms, bytes / ms * 1e-6);
  fflush(stdout);
  CK(hipFree(d)); CK(hipFree(dtw)); CK(hipFree(ditems));
}

// Round 5: the fused launches of the separable pass programs -- load | slot F slot F | slot F slot F | store on a quarter of the
// lines, a quarter of the positions loaded and stored, every slot reading a phase table (TAB = 1, LONG = 2) -- with the
// library's two-line workgroups (512 threads, two per CU) against one-line workgroups (256 threads, four per CU).
template <int AXIS, int LINES, int LONG = 2, int NFFT = 2, int E = 16>
void bench_fused(const char* name, int batch, int reps, int pad_blocks) {
  using T = double;
  constexpr int N = 4096, BR = 4, BC = 2, TILES = 1;
  if (getenv("PAOS_BENCH_BATCH")) batch = atoi(getenv("PAOS_BENCH_BATCH"));
  const unsigned pitch = (unsigned)N * BR + (unsigned)pad_blocks * BR * BC;
  const unsigned item_stride = pitch * (N / BR);
  cx<T>* d;
  CK(hipMalloc(&d, (size_t)item_stride * batch * sizeof(cx<T>)));
  std::vector<std::complex<T>> h((size_t)N * N);
  srand(1);
  for (auto& z : h) z = std::complex<T>((T)(rand() / (double)RAND_MAX - 0.5), (T)(rand() / (double)RAND_MAX - 0.5));
  for (int b = 0; b < batch; ++b)
    for (int r = 0; r < N / BR; ++r)
      CK(hipMemcpy(d + (size_t)b * item_stride + (size_t)r * pitch, h.data() + (size_t)r * N * BR,
                   (size_t)N * BR * sizeof(cx<T>), hipMemcpyHostToDevice));
  auto tw = make_twiddles<T, N>();
  cx<T>* dtw;
  CK(hipMalloc(&dtw, N * sizeof(cx<T>)));
  CK(hipMemcpy(dtw, tw.data(), N * sizeof(cx<T>), hipMemcpyHostToDevice));
  // one table of unit factors per slot and item (2 MiB per slot of a 32-item launch, like the library's)
  constexpr int kPasses = 1 + (LONG + 1) / 2;
  std::vector<std::complex<double>> tab((size_t)2 * kPasses * batch * N);
  for (size_t i = 0; i < tab.size(); ++i) { const double a = 0.001 * (double)(i % 6283); tab[i] = {cos(a), sin(a)}; }
  cx<double>* dtab;
  CK(hipMalloc(&dtab, tab.size() * sizeof(cx<double>)));
  CK(hipMemcpy(dtab, tab.data(), tab.size() * sizeof(cx<double>), hipMemcpyHostToDevice));
  const int lo = 1536, hi = 2560;  // the live quarter (whole groups of 32 workgroups either way)
  std::vector<FrugalItem> items((size_t)kPasses * batch);
  for (int p = 0; p < kPasses; ++p)
    for (int b = 0; b < batch; ++b) {
      FrugalItem& it = items[(size_t)p * batch + b];
      std::memset(&it, 0, sizeof(it));
      it.active = 1; it.fft1_on = 1; it.fft1_inv = 0; it.fft2_on = 1; it.fft2_inv = 1;
      it.pre.scale = 1.0; it.mid.scale = 1.0 / N;
      it.line_lo = lo; it.line_hi = hi; it.pos_lo = lo; it.pos_hi = hi; it.spos_lo = lo; it.spos_hi = hi;
      it.pre.table = dtab + ((size_t)(2 * p) * batch + b) * N;
      it.mid.table = dtab + ((size_t)(2 * p + 1) * batch + b) * N;
      for (int j = 0; j < kFrugalMaxPre; ++j) it.pre_ph[j] = {0.01, 0.01, 0.21, 1.0, 1.0, 0.0};
      for (int j = 0; j < kFrugalMaxMid; ++j) it.mid_ph[j] = {0.01, 0.01, 0.37, -1.0, 1.0, 1.0};
    }
  FrugalItem* ditems;
  CK(hipMalloc(&ditems, items.size() * sizeof(FrugalItem)));
  CK(hipMemcpy(ditems, items.data(), items.size() * sizeof(FrugalItem), hipMemcpyHostToDevice));
  std::vector<double> ones((size_t)batch, 1.0);
  double* dones;
  CK(hipMalloc(&dones, ones.size() * sizeof(double)));
  CK(hipMemcpy(dones, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
  FrugalArgs a{d, dtw, ditems, pitch, item_stride};
  a.dyn_scale = dones;
  a.wg0 = lo / (LINES * TILES);
  const dim3 grid((hi - lo) / (LINES * TILES), batch), block(TILES * LINES * N / E);
#if PAOS_STAMPS
  unsigned long long* dstamps;
  const size_t nstamps = (size_t)grid.x * grid.y * kStampSlots;
  CK(hipMalloc(&dstamps, nstamps * sizeof(unsigned long long)));
  CK(hipMemset(dstamps, 0, nstamps * sizeof(unsigned long long)));
  a.stamps = dstamps;
#endif
  constexpr bool SPLIT = true;
  const size_t lds = frugal_lds_bytes<T, N, LINES, TILES, SPLIT, 1, 1, E>();
  auto kf = frugal_pass_kernel<T, N, E, LINES, TILES, AXIS, BR, BC, SPLIT, 1, 1, NFFT, 0, 1, LONG>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // checksum of one launch on a fresh copy (the variants must agree bit for bit)
  hipLaunchKernelGGL(kf, grid, block, lds, 0, PAOS_FRUGAL_PASS(a));
  CK(hipDeviceSynchronize());
  std::vector<std::complex<T>> row(N);
  double sum = 0.0;
  for (int r = lo; r < hi; r += 97) {
    for (int c0 = 0; c0 < N; c0 += 2) {
      std::complex<T> two[2];
      CK(hipMemcpy(two, d + layout_index<BR, BC>(r, c0, pitch), sizeof(two), hipMemcpyDeviceToHost));
      sum += two[0].real() * (1 + (c0 % 7)) + two[1].imag() * (1 + (r % 5));
    }
  }
  Timer tm;
  float ms = tm.run([&] { hipLaunchKernelGGL(kf, grid, block, lds, 0, PAOS_FRUGAL_PASS(a)); }, reps);
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kf, block.x, lds));
  const double lines = (double)(hi - lo) * batch, transforms = lines * 2 * kPasses;
  printf("%-40s ax=%d lines/wg=%d thr=%d lds=%zuK occ=%d  %8.3f ms  %6.2f TFLOP/s (5 N log2 N)  checksum %.17g\n", name, AXIS, LINES, block.x,
         lds / 1024, nb, ms, transforms * 5.0 * N * 12.0 / (ms * 1e-3) / 1e12, sum);
  fflush(stdout);
#if PAOS_STAMPS
  {  // the clock the chip held in the last launch, and where a workgroup's time went (wave 0 of every workgroup)
    std::vector<unsigned long long> st(nstamps);
    CK(hipMemcpy(st.data(), dstamps, nstamps * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> clk, seg[7], life;
    for (size_t w = 0; w < (size_t)grid.x * grid.y; ++w) {
      const unsigned long long* q = &st[w * kStampSlots];
      if (!q[7] || q[10] <= q[8]) continue;
      const double cyc = (double)(q[7] - q[0]), real = (double)(q[10] - q[8]);  // s_memrealtime: 100 MHz
      clk.push_back(cyc / real * 0.1);
      life.push_back(cyc);
      for (int k = 0; k < 7; ++k) seg[k].push_back((double)(q[k + 1] - q[k]));
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("   stamps: %zu workgroups, shader clock (median) %.3f GHz, workgroup life %.0f cyc; load wait %.0f | pre slot %.0f | fft1 %.0f | mid slot %.0f | fft2 + later passes %.0f | store issue %.0f | drain %.0f\n",
           clk.size(), med(clk), med(life), med(seg[0]), med(seg[1]), med(seg[2]), med(seg[3]), med(seg[4]), med(seg[5]), med(seg[6]));
    CK(hipFree(dstamps));
  }
#endif
  CK(hipFree(d)); CK(hipFree(dtw)); CK(hipFree(ditems)); CK(hipFree(dtab)); CK(hipFree(dones));
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
  const int b4 = 8, pad = getenv("PAOS_BENCH_PAD") ? atoi(getenv("PAOS_BENCH_PAD")) : 3;  // pitch padding in blocks
  // NOTE: only instantiate 512-or-fewer-thread shapes: a 1024-thread instantiation in the same translation unit
  // changes the register allocation of the others (measured in round 2).
  bench_copy<double>(4096, b4, reps);
  if (getenv("PAOS_BENCH_FUSED")) {  // the fused launches of the separable programs: two-line against one-line workgroups
    for (int round = 0; round < 2; ++round) {
      bench_fused<0, 2>("fused rows, 2 lines per workgroup", 32, reps, pad);
      bench_fused<0, 1>("fused rows, 1 line per workgroup", 32, reps, pad);
      bench_fused<1, 2>("fused cols, 2 lines per workgroup", 32, reps, pad);
      bench_fused<1, 1>("fused cols, 1 line per workgroup", 32, reps, pad);
#ifdef PAOS_E8_MINW
      bench_fused<0, 1, 2, 2, 8>("fused rows, 1 line, 8 points per thread", 32, reps, pad);
      bench_fused<1, 1, 2, 2, 8>("fused cols, 1 line, 8 points per thread", 32, reps, pad);
#endif
    }
    return 0;
  }
  if (getenv("PAOS_BENCH_DENSE1")) {  // byte-bound (dense) passes of 32 wavefronts: two-line workgroups against one-line ones
    for (int round = 0; round < 2; ++round) {
      bench_copy<double>(4096, 32, reps);
      bench_frugal<double, 4096, 0, 0, 0, 1>("rows single, 2 lines", 32, reps, pad);
      bench_frugal<double, 4096, 0, 0, 0, 1, 1>("rows single, 1 line", 32, reps, pad);
      bench_frugal<double, 4096, 1, 0, 0, 1>("cols single, 2 lines", 32, reps, pad);
      bench_frugal<double, 4096, 1, 0, 0, 1, 1>("cols single, 1 line", 32, reps, pad);
      bench_frugal<double, 4096, 0, 0, 1, 2>("rows double 1 phase, 2 lines", 32, reps, pad);
      bench_frugal<double, 4096, 0, 0, 1, 2, 1>("rows double 1 phase, 1 line", 32, reps, pad);
      bench_frugal<double, 4096, 1, 0, 1, 2>("cols double 1 phase, 2 lines", 32, reps, pad);
      bench_frugal<double, 4096, 1, 0, 1, 2, 1>("cols double 1 phase, 1 line", 32, reps, pad);
    }
    return 0;
  }
  bench_frugal<double, 4096, 0, 0, 0, 1>("rows single", b4, reps, pad);
  bench_frugal<double, 4096, 1, 0, 0, 1>("cols single", b4, reps, pad);
  bench_frugal<double, 4096, 0, 0, 1, 2>("rows double 1 phase", b4, reps, pad);
  bench_frugal<double, 4096, 1, 0, 1, 2>("cols double 1 phase", b4, reps, pad);
  bench_frugal<double, 4096, 0, 0, 1, 3>("rows double 1 phase, digit-swapped", b4, reps, pad);
  bench_frugal<double, 4096, 1, 0, 1, 3>("cols double 1 phase, digit-swapped", b4, reps, pad);
  bench_frugal<double, 4096, 1, 0, 0, 3>("cols double 0 phases, digit-swapped", b4, reps, pad);
  bench_frugal<double, 4096, 1, 0, 0, 2>("cols double 0 phases", b4, reps, pad);
  if (getenv("PAOS_BENCH_CORE")) return 0;  // the 4096^2 complex128 shapes only (counter runs)
  // N = 2048: 256-thread workgroups (library, round 2a) against 512-thread ones (two tiles per workgroup)
  bench_frugal<double, 2048, 0, 0, 1, 2, 2, 1>("rows double 1 phase 256 thr", 32, reps, pad);
  bench_frugal<double, 2048, 0, 0, 1, 2, 2, 2>("rows double 1 phase 512 thr (2 tiles)", 32, reps, pad);
  bench_frugal<double, 2048, 1, 0, 1, 2, 2, 1>("cols double 1 phase 256 thr", 32, reps, pad);
  bench_frugal<double, 2048, 1, 0, 1, 2, 2, 2>("cols double 1 phase 512 thr (2 tiles)", 32, reps, pad);
  bench_frugal<double, 2048, 0, 0, 0, 1, 2, 1>("rows single 256 thr", 32, reps, pad);
  bench_frugal<double, 2048, 0, 0, 0, 1, 2, 2>("rows single 512 thr (2 tiles)", 32, reps, pad);
  bench_frugal<double, 2048, 1, 0, 0, 1, 2, 1>("cols single 256 thr", 32, reps, pad);
  bench_frugal<double, 2048, 1, 0, 0, 1, 2, 2>("cols single 512 thr (2 tiles)", 32, reps, pad);
  // N = 1024
  bench_frugal<double, 1024, 0, 0, 1, 2, 4, 1>("rows double 1 phase 256 thr", 128, reps, pad);
  bench_frugal<double, 1024, 0, 0, 1, 2, 4, 2>("rows double 1 phase 512 thr (2 tiles)", 128, reps, pad);
  bench_frugal<double, 1024, 1, 0, 1, 2, 2, 1>("cols double 1 phase 128 thr", 128, reps, pad);
  bench_frugal<double, 1024, 1, 0, 1, 2, 2, 2>("cols double 1 phase 256 thr (2 tiles)", 128, reps, pad);
  bench_frugal<double, 1024, 1, 0, 1, 2, 2, 4>("cols double 1 phase 512 thr (4 tiles)", 128, reps, pad);
  bench_frugal<double, 1024, 0, 0, 0, 1, 4, 1>("rows single 256 thr", 128, reps, pad);
  bench_frugal<double, 1024, 0, 0, 0, 1, 4, 2>("rows single 512 thr (2 tiles)", 128, reps, pad);
  bench_frugal<double, 1024, 1, 0, 0, 1, 2, 1>("cols single 128 thr", 128, reps, pad);
  bench_frugal<double, 1024, 1, 0, 0, 1, 2, 4>("cols single 512 thr (4 tiles)", 128, reps, pad);
  // complex64 (fp32 mode): 8 B elements, 4 x 2 blocks of 64 B
  bench_frugal<float, 4096, 0, 0, 0, 1>("rows single", 16, reps, pad);
  bench_frugal<float, 4096, 1, 0, 0, 1>("cols single", 16, reps, pad);
  bench_frugal<float, 4096, 0, 0, 1, 2>("rows double 1 phase", 16, reps, pad);
  bench_frugal<float, 4096, 1, 0, 1, 2>("cols double 1 phase", 16, reps, pad);
#ifdef PAOS_BENCH_ROWS4_F32
  // complex64 row tiles of 4 rows (1024 threads, 64-byte pieces of 8 x 2 blocks when PAOS_BENCH_BR=8)
  bench_frugal<float, 4096, 0, 0, 0, 1, 4, 1>("rows single 4-row tiles", 16, reps, pad);
  bench_frugal<float, 4096, 0, 0, 1, 2, 4, 1>("rows double 1 phase 4-row tiles", 16, reps, pad);
#endif
#ifdef PAOS_BENCH_E32
  // complex64 with 32 points per thread over 4 x 4 blocks (128 B): 4-line tiles, 512 threads, whole lines on both axes
  bench_frugal<float, 4096, 0, 0, 0, 1, 4, 1, 32, 4>("rows single E=32 4x4 blocks", 16, reps, pad);
  bench_frugal<float, 4096, 1, 0, 0, 1, 4, 1, 32, 4>("cols single E=32 4x4 blocks", 16, reps, pad);
  bench_frugal<float, 4096, 0, 0, 1, 2, 4, 1, 32, 4>("rows double 1 phase E=32 4x4 blocks", 16, reps, pad);
  bench_frugal<float, 4096, 1, 0, 1, 2, 4, 1, 32, 4>("cols double 1 phase E=32 4x4 blocks", 16, reps, pad);
#endif
  return 0;
}
