// ldsprobe.hip -- how many LDS cycles do the access patterns of the FFT exchanges cost on gfx950?
//
// One wave issues REPS back-to-back DS instructions whose per-lane addresses follow a given pattern and
// times them with s_memtime (shader clock).  A conflict-free 64-lane b64 access moves 512 B = 4 cycles
// of the 128 B/clk LDS, a b128 one 8 cycles; whatever is measured above that is bank conflicts.
// The patterns are the index maps of fft_core.h (fft_stages with SPLIT, N = 4096, E = 16) and of the
// circle-table reads; the result decides how the exchange slots / tables should be laid out.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ldsprobe.hip -o build/ldsprobe && ./build/ldsprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int REPS = 64;
enum Op { RD64, WR64, WR2_64, RD128, WR128, RD2_64 };

// addr[lane][rep]: byte addresses prepared on the host, so that the kernel body is pure DS traffic
template <int OP, int OFF1 = 1>
__global__ void probe(const unsigned* addr, unsigned long long* out, double* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 160 * 1024 / 8; i += 64) reinterpret_cast<double*>(smem)[i] = i;
  __syncthreads();
  unsigned a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = addr[r * 64 + lane];
  double acc = 0;
  double d0 = lane, d1 = lane + 1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < REPS / 16; ++it) {
    if constexpr (OP == RD64) {
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("ds_read_b64 %0, %1" : "=v"(x[r]) : "v"(a[r]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += x[r];
    } else if constexpr (OP == RD2_64) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 x[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(x[r]) : "v"(a[2 * r]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 8; ++r) acc += x[r].x + x[r].y;
    } else if constexpr (OP == WR64) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("ds_write_b64 %0, %1" ::"v"(a[r]), "v"(d0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (OP == WR2_64) {
      // two adjacent slots per instruction, like the compiler's merge of v[r], v[r+1]
#pragma unroll
      for (int r = 0; r < 16; r += 2) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:%3" ::"v"(a[r]), "v"(d0), "v"(d1), "n"(OFF1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (OP == RD128) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(x[r]) : "v"(a[r]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += x[r].x + x[r].y;
    } else {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 w = {d0, d1};
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("ds_write_b128 %0, %1" ::"v"(a[r]), "v"(w));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[0] = t1 - t0;
  sink[lane] = acc;
}

static unsigned* daddr;
static unsigned long long* dout;
static double* dsink;

template <int OP, int OFF1 = 1>
int run(const char* name, const std::function<unsigned(int lane, int r)>& slot_bytes, int instr_per_16) {
  std::vector<unsigned> h(16 * 64);
  for (int r = 0; r < 16; ++r)
    for (int l = 0; l < 64; ++l) h[r * 64 + l] = slot_bytes(l, r) % (159 * 1024);
  CK(hipMemcpy(daddr, h.data(), h.size() * sizeof(unsigned), hipMemcpyHostToDevice));
  auto kf = probe<OP, OFF1>;
  CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  unsigned long long best = ~0ull;
  for (int k = 0; k < 5; ++k) {
    hipLaunchKernelGGL(kf, dim3(1), dim3(64), 160 * 1024, 0, daddr, dout, dsink);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, dout, sizeof(c), hipMemcpyDeviceToHost));
    if (c < best) best = c;
  }
  const int n = REPS / 16 * instr_per_16;
  printf("%-64s %6.1f cycles / instruction (%d instr)\n", name, (double)best / n, n);
  fflush(stdout);
  return 0;
}

int main() {
  CK(hipMalloc(&daddr, 16 * 64 * sizeof(unsigned)));
  CK(hipMalloc(&dout, sizeof(unsigned long long)));
  CK(hipMalloc(&dsink, 64 * sizeof(double)));
  auto pad = [](int i) { return i + (i >> 4); };
  // ---- calibration: strides in 8-byte slots ---------------------------------------------------
  for (int s : {1, 2, 3, 4, 8, 16, 17, 32, 33}) {
    char nm[96];
    snprintf(nm, sizeof nm, "ds_read_b64   lane stride %2d slots", s);
    run<RD64>(nm, [=](int l, int r) { return (unsigned)(l * s + r * 2048) * 8u; }, 16);
  }
  for (int s : {1, 2, 3, 16, 17}) {
    char nm[96];
    snprintf(nm, sizeof nm, "ds_write_b64  lane stride %2d slots", s);
    run<WR64>(nm, [=](int l, int r) { return (unsigned)(l * s + r * 2048) * 8u; }, 16);
  }
  for (int s : {1, 2, 3, 16, 17, 18}) {
    char nm[96];
    snprintf(nm, sizeof nm, "ds_write2_b64 lane stride %2d slots (slots +0, +1)", s);
    run<WR2_64>(nm, [=](int l, int r) { return (unsigned)(l * s + r * 2048) * 8u; }, 8);
  }
  for (int s : {1, 2, 3, 16, 17, 18}) {
    char nm[96];
    snprintf(nm, sizeof nm, "ds_read2_b64  lane stride %2d slots (slots +0, +1)", s);
    run<RD2_64>(nm, [=](int l, int r) { return (unsigned)(l * s + r * 2048) * 8u; }, 8);
  }
  for (int s : {1, 2, 3, 4, 8, 9}) {
    char nm[96];
    snprintf(nm, sizeof nm, "ds_read_b128  lane stride %2d x 16 B", s);
    run<RD128>(nm, [=](int l, int r) { return (unsigned)(l * s + r * 1024) * 16u; }, 16);
  }
  for (int s : {1, 2, 9}) {
    char nm[96];
    snprintf(nm, sizeof nm, "ds_write_b128 lane stride %2d x 16 B", s);
    run<WR128>(nm, [=](int l, int r) { return (unsigned)(l * s + r * 1024) * 16u; }, 16);
  }
  // ---- the exchanges of fft_stages<double, 4096, 16, SPLIT> (wave w = 0: t = lane) ---------------
  run<WR2_64>("exchange 1 write2 (slot 17 t + r, r even)", [=](int l, int r) { return (unsigned)(17 * l + r) * 8u; }, 8);
  run<WR64>("exchange 1 write, one slot per instruction", [=](int l, int r) { return (unsigned)(17 * l + r) * 8u; }, 16);
  run<RD64>("exchange 1 read  (slot pad(t) + 272 r)", [=](int l, int r) { return (unsigned)(pad(l) + 272 * r) * 8u; }, 16);
  run<WR2_64, 17>("exchange 2 write2 (slot 272 (t / 16) + t % 16 + 17 r; offsets +0, +17)",
              [=](int l, int r) { return (unsigned)(272 * (l / 16) + (l % 16) + 17 * r) * 8u; }, 8);
  run<WR64>("exchange 2 write  (slot 272 (t / 16) + t % 16 + 17 r)",
            [=](int l, int r) { return (unsigned)(272 * (l / 16) + (l % 16) + 17 * r) * 8u; }, 16);
  run<RD64>("exchange 2 read  (slot pad(t) + 272 r)", [=](int l, int r) { return (unsigned)(pad(l) + 272 * r) * 8u; }, 16);
  // ---- table reads -------------------------------------------------------------------------------
  run<RD128>("circle[t0 * m], m = r + 1 (stage-2 twiddles, 16 B entries)", [=](int l, int r) { return (unsigned)(((l & 15) * (r + 1)) & 255) * 16u; }, 16);
  run<RD128>("circle[(t0 m) swizzled by + (e >> 3)]", [=](int l, int r) {
    const int e = ((l & 15) * (r + 1)) & 255;
    return (unsigned)((e & ~7) | ((e + (e >> 3)) & 7)) * 16u; }, 16);
  run<RD128>("table[m][t0] (transposed twiddle table)", [=](int l, int r) { return (unsigned)(r * 16 + (l & 15)) * 16u; }, 16);
  run<RD128>("circle[random j] (sincos_tab)", [=](int l, int r) { return (unsigned)((l * 37 + r * 101 + (l * l) % 13) & 255) * 16u; }, 16);
  run<RD128>("tw[t] (stage-3 base twiddle, unit stride)", [=](int l, int r) { return (unsigned)(l) * 16u; }, 16);
  return 0;
}
