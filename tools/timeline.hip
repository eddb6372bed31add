// timeline -- where a workgroup of the frugal pass kernel spends its time (diagnostic build:
// -DPAOS_STAMPS=1 makes wave 0 of every workgroup record s_memtime at the phase boundaries).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DPAOS_STAMPS=1 -I paos_amd/csrc tools/timeline.hip -o build/timeline
// Prints, per kernel shape: the launch time, the mean / median shader cycles of each phase of a
// workgroup (load wait | pre slot | FFT 1 | mid slot | FFT 2 | store issue | store drain), the clock
// (shader cycles per 10 ns real-time tick) and how many workgroups were resident per CU over time.
#define PAOS_STAMPS 1
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

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

template <typename T, int N, int AXIS, int KPRE, int KMID, int NFFT, int LINES = 2>
void timeline(const char* name, int batch, int pad_blocks) {
  constexpr int E = 16, BR = 4, BC = 2;
  const unsigned pitch = (unsigned)N * BR + (unsigned)pad_blocks * BR * BC;
  const unsigned item_stride = pitch * (N / BR);
  cx<T>* d;
  CK(hipMalloc(&d, (size_t)item_stride * batch * sizeof(cx<T>)));
  std::vector<std::complex<T>> h((size_t)item_stride);
  srand(1);
  for (auto& z : h) z = std::complex<T>((T)(rand() / (double)RAND_MAX - 0.5), (T)(rand() / (double)RAND_MAX - 0.5));
  for (int b = 0; b < batch; ++b)
    CK(hipMemcpy(d + (size_t)b * item_stride, h.data(), (size_t)item_stride * sizeof(cx<T>), hipMemcpyHostToDevice));
  std::vector<std::complex<T>> tw(N);
  for (int m = 0; m < N; ++m) {
    long double a = -2.0L * 3.14159265358979323846264338327950288L * m / N;
    tw[m] = std::complex<T>((T)cosl(a), (T)sinl(a));
  }
  cx<T>* dtw;
  CK(hipMalloc(&dtw, N * sizeof(cx<T>)));
  CK(hipMemcpy(dtw, tw.data(), N * sizeof(cx<T>), hipMemcpyHostToDevice));
  std::vector<FrugalItem> items(batch);
  for (auto& it : items) {
    std::memset(&it, 0, sizeof(it));
    it.active = 1; it.fft1_on = 1; it.fft1_inv = 0; it.fft2_on = 1; it.fft2_inv = 1;
    it.pre.scale = 1.0; it.mid.scale = 1.0 / N; it.mid.sign_on = 0;
    it.line_lo = 0; it.line_hi = N; it.pos_lo = 0; it.pos_hi = N; it.spos_lo = 0; it.spos_hi = N;  // no pruning
    for (int j = 0; j < kFrugalMaxPre; ++j) it.pre_ph[j] = {0.01, 0.01, 0.21, 1.0, 1.0, 0.0};
    for (int j = 0; j < kFrugalMaxMid; ++j) it.mid_ph[j] = {0.01, 0.01, 0.37, -1.0, 1.0, 1.0};
  }
  FrugalItem* ditems;
  CK(hipMalloc(&ditems, items.size() * sizeof(FrugalItem)));
  CK(hipMemcpy(ditems, items.data(), items.size() * sizeof(FrugalItem), hipMemcpyHostToDevice));
  const dim3 grid(N / LINES, batch), block(LINES * N / E);
  const size_t nwg = (size_t)grid.x * grid.y;
  unsigned long long* dst;
  CK(hipMalloc(&dst, nwg * kStampSlots * sizeof(unsigned long long)));
  CK(hipMemset(dst, 0, nwg * kStampSlots * sizeof(unsigned long long)));
  std::vector<double> ones((size_t)batch, 1.0);
  double* dones;
  CK(hipMalloc(&dones, ones.size() * sizeof(double)));
  CK(hipMemcpy(dones, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
  FrugalArgs a{d, dtw, ditems, pitch, item_stride, nullptr, nullptr, nullptr, dones, dst};
  constexpr bool SPLIT = sizeof(T) == 8;
  const size_t lds = frugal_lds_bytes<T, N, LINES, 1, SPLIT, KPRE, KMID, E>();
  auto kf = frugal_pass_kernel<T, N, E, LINES, 1, AXIS, BR, BC, SPLIT, KPRE, KMID, NFFT>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kf, grid, block, lds, 0, PAOS_FRUGAL_PASS(a));  // warm, clocks up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kf, grid, block, lds, 0, PAOS_FRUGAL_PASS(a));
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st(nwg * kStampSlots);
  CK(hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));

  const char* phase[7] = {"load wait", "pre slot", "fft 1", "mid slot", "fft 2", "store issue", "store drain"};
  std::vector<double> dur[8];
  double clk_sum = 0;
  size_t clk_n = 0;
  unsigned long long rt0 = ~0ull, rt1 = 0;
  std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> per_cu;  // realtime events +1 / -1
  for (size_t w = 0; w < nwg; ++w) {
    const unsigned long long* s = &st[w * kStampSlots];
    for (int p = 0; p < 7; ++p) dur[p].push_back((double)(s[p + 1] - s[p]));
    dur[7].push_back((double)(s[7] - s[0]));
    const double ticks = (double)(s[10] - s[8]);
    if (ticks > 0) { clk_sum += (double)(s[7] - s[0]) / ticks; ++clk_n; }
    rt0 = std::min(rt0, s[8]); rt1 = std::max(rt1, s[10]);
    const unsigned hw = (unsigned)(s[9] & 0xffffffffu), xcc = (unsigned)(s[9] >> 32) & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    per_cu[key].push_back({s[8], +1});
    per_cu[key].push_back({s[10], -1});
  }
  auto stat = [](std::vector<double>& v, double& mean, double& med, double& p90) {
    std::sort(v.begin(), v.end());
    mean = 0; for (double x : v) mean += x; mean /= v.size();
    med = v[v.size() / 2]; p90 = v[(size_t)(v.size() * 0.9)];
  };
  printf("== %s: N=%d batch=%d axis=%d kpre=%d kmid=%d nfft=%d  launch %.3f ms (%.0f GB/s), %zu workgroups, %zu CUs seen\n",
         name, N, batch, AXIS, KPRE, KMID, NFFT, ms, 2.0 * N * N * batch * sizeof(cx<T>) / ms * 1e-6, nwg, per_cu.size());
  const double clk = clk_n ? clk_sum / clk_n : 0.0;  // cycles per 10 ns
  printf("   shader clock ~ %.2f GHz (s_memtime / s_memrealtime); real-time span of the grid %.3f ms\n", clk * 0.1,
         (double)(rt1 - rt0) * 1e-5);
  double mean, med, p90, tot_mean = 0;
  for (int p = 0; p < 8; ++p) {
    stat(dur[p], mean, med, p90);
    if (p == 7) tot_mean = mean;
    printf("   %-12s mean %8.0f cyc (%6.2f us)  median %8.0f  p90 %8.0f\n", p < 7 ? phase[p] : "WORKGROUP", mean,
           mean / (clk * 100.0), med, p90);
  }
  // residency: time-weighted distribution of the number of resident workgroups per CU
  double w_res[8] = {0};
  double span_total = 0;
  for (auto& kv : per_cu) {
    auto& ev = kv.second;
    std::sort(ev.begin(), ev.end());
    int cur = 0;
    for (size_t i = 0; i + 1 < ev.size(); ++i) {
      cur += ev[i].second;
      const double dt = (double)(ev[i + 1].first - ev[i].first);
      w_res[std::min(std::max(cur, 0), 7)] += dt;
      span_total += dt;
    }
  }
  printf("   resident workgroups per CU (share of the CU's busy span): ");
  for (int k = 0; k < 5; ++k) printf("%d: %.1f%%  ", k, 100.0 * w_res[k] / span_total);
  printf("\n   per workgroup %.2f us x %zu workgroups / (%zu CUs x 2) = %.3f ms if perfectly packed\n",
         tot_mean / (clk * 100.0), nwg, per_cu.size(), tot_mean / (clk * 100.0) * nwg / (per_cu.size() * 2) * 1e-3);
  fflush(stdout);
  CK(hipFree(d)); CK(hipFree(dtw)); CK(hipFree(ditems)); CK(hipFree(dst));
}

int main() {
  const int b = 8, pad = 3;
  // One-transform shapes only: the stamped build of a two-transform pass runs 1.8x slower than the real kernel (the
  // stamps' s_waitcnt / s_memtime pairs serialise the second transform: round 2 measured 1.98 ms against 1.10), so its
  // phase split says nothing about the real kernel.  These two run within 4 % of their un-instrumented time.
  timeline<double, 4096, 0, 0, 0, 1>("rows single", b, pad);
  timeline<double, 4096, 1, 0, 0, 1>("cols single", b, pad);
  return 0;
}
