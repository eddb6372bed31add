// frugal_pass.h -- the register-frugal pass kernel (<= 128 VGPRs: two 512-thread workgroups
// per CU, so the load / compute / store phases of neighbouring tiles overlap).
//
// Same contract as fused_pass_kernel, but the shape of the work is a compile-time constant:
//   load -> slot(KPRE) -> FFT -> slot(KMID) -> [FFT] -> store
// where a slot multiplies every pixel by  s * scale * prod_{j<K} exp(i sgn_j q_j):
//   s      = (-1)^(row+col) if the item's sign flag is set, else 1 (fftshift folding)
//   scale  = exact power of two (ortho 1/N), 1 when unused
//   q_j    = m2_j * fl(coef_j * fl(x^2 + y^2)), centred or natural-order coordinates,
//            m2_j in {1, 2 pi}  (fl(1 * q) == q, so the reference's rounding is kept)
// Everything that varies per batch item is DATA (a disabled phase has coef = 0 -> factor 1,
// exactly), so the instruction stream has no branches on operator kinds -- those branches are
// what pushes the generic interpreter to ~230 VGPRs (profiles/r01_vgpr_experiments.txt).
#pragma once
#include "fft_kernels.h"

#ifndef PAOS_FENCE_EVERY
#define PAOS_FENCE_EVERY 2
#endif

namespace paos {

constexpr int kFrugalMaxPre = 2, kFrugalMaxMid = 3;

struct FrugalPhase {
  double sx, sy, coef, sgn, m2, natural;  // natural != 0: np.fft.fftfreq index order
};
// A convex aperture along one line (a row for row passes, a column for column passes):
//   pos < p0: w_out | p0 <= pos < p1: vals[pos-p0] | p1 <= pos < p2: w_in |
//   p2 <= pos < p3: vals[kMaskW + pos-p2] | pos >= p3: w_out          (times the line factor lm)
// rendered by mask_lines_kernel (pointwise.h) with the same per-pixel functions the stand-alone
// aperture kernel uses, so values and the {0, partial, 1} classification are identical.
constexpr int kMaskW = 192;  // longest partial run per side the records can hold
struct MaskLine {
  int p0, p1, p2, p3;
  double lm;  // line multiplier (rectangles: the separable count of the other axis / 32)
  double pad;
};
struct FrugalSlot {
  double sign_on, scale;
  double mask_on, w_in, w_out;   // aperture riding on this slot (0/1), interior / exterior weight
  const MaskLine* lines;          // [N] records of THIS item for the pass axis
  const double* vals;             // [N][2 * kMaskW]
};
static_assert(sizeof(FrugalSlot) == 7 * sizeof(double), "record of 8-byte fields");
// per-item record, doubles: [fft1_on, fft1_inv, fft2_on, fft2_inv, pre slot, pre phases[2],
// mid slot, mid phases[3]]
struct FrugalItem {
  double active;  // 0: the item takes no part in this pass (no load, no store)
  double fft1_on, fft1_inv, fft2_on, fft2_inv;
  FrugalSlot pre;
  FrugalPhase pre_ph[kFrugalMaxPre];
  FrugalSlot mid;
  FrugalPhase mid_ph[kFrugalMaxMid];
};

constexpr int kTwiddleLds = 256;  // table entries the stage twiddles can address: k N / (NS R) < N / R <= 256

// PAOS_STAMPS (tools/fftbench.hip timeline builds only): wave 0 of every workgroup records
// s_memtime at the phase boundaries plus where it ran (HW_ID, XCC_ID) into FrugalArgs::stamps.
#ifndef PAOS_STAMPS
#define PAOS_STAMPS 0
#endif
constexpr int kStampSlots = 12;

struct FrugalArgs {
  void* field;
  const void* tw;
  const FrugalItem* items;  // [batch]
  unsigned pitch, item_stride;
#if PAOS_STAMPS
  unsigned long long* stamps;  // [gridDim.y][gridDim.x][kStampSlots]
#endif
};
#if PAOS_STAMPS
#define PAOS_STAMP(i)                                                                              \
  do {                                                                                             \
    if (threadIdx.x == 0)                                                                          \
      a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kStampSlots + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define PAOS_STAMP_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PAOS_STAMP(i) do { } while (0)
#define PAOS_STAMP_WAIT_VM() do { } while (0)
#endif

template <typename T, int N, int E, int K, typename Map>
__device__ __forceinline__ void frugal_slot(cx<T>* v, const FrugalSlot& sl, const FrugalPhase* ph,
                                            const Map& m) {
  const double sc = sl.scale;
  const bool sign_on = sl.sign_on != 0.0;
  if (sl.mask_on != 0.0) {  // wave-uniform: an aperture rides on this slot
    const int line = Map::kAxis == 0 ? m.row(0) : m.col(0);  // constant per thread
    const MaskLine ml = sl.lines[line];
    const double* vals = sl.vals + (size_t)line * (2 * kMaskW);
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = Map::kAxis == 0 ? m.col(k) : m.row(k);
      double w = (pos >= ml.p1 && pos < ml.p2) ? sl.w_in : sl.w_out;
      if (pos >= ml.p0 && pos < ml.p1) w = vals[pos - ml.p0];
      if (pos >= ml.p2 && pos < ml.p3) w = vals[kMaskW + pos - ml.p2];
      w *= ml.lm;
      v[k] = {(T)__dmul_rn((double)v[k].x, w), (T)__dmul_rn((double)v[k].y, w)};
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int row = m.row(k), col = m.col(k);
    cx<double> vd = {(double)v[k].x, (double)v[k].y};
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool nat = ph[j].natural != 0.0;
      const int gx = nat ? ((col < N / 2) ? col : col - N) : col - N / 2;
      const int gy = nat ? ((row < N / 2) ? row : row - N) : row - N / 2;
      const double x = (double)gx * ph[j].sx, y = (double)gy * ph[j].sy;
      const double s = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
      const double q = __dmul_rn(ph[j].m2, __dmul_rn(ph[j].coef, s));
      if constexpr (sizeof(T) == 4) {
        // fp32 mode: the argument is still formed in fp64 like the reference's, reduced to a
        // fraction of a turn in fp64, and only then handed to the hardware sin / cos (inputs in
        // revolutions, ~1e-7 absolute) -- the field itself carries no more than that
        const double turns = q * 0.15915494309189535;  // 1 / (2 pi)
        const float frac = (float)(turns - rint(turns));
        const float snf = __builtin_amdgcn_sinf(frac) * (float)ph[j].sgn, csf = __builtin_amdgcn_cosf(frac);
        const float xr = (float)vd.x, xi = (float)vd.y;
        vd = {(double)fmaf(xr, csf, -(xi * snf)), (double)fmaf(xr, snf, xi * csf)};
        continue;
      }
      double sn, cs;
      sincos_fast(q, &sn, &cs);
      sn *= ph[j].sgn;
      // only the phase ARGUMENT is rounded like the reference's; the product itself may use FMA
      vd = {fma(vd.x, cs, -(vd.y * sn)), fma(vd.x, sn, vd.y * cs)};
    }
    const double f = (sign_on && ((row + col) & 1)) ? -sc : sc;
    v[k] = {(T)(vd.x * f), (T)(vd.y * f)};
    if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// Column tiles move whole 128-byte lines that no other workgroup touches during the pass, so
// they go around the caches (nontemporal: +7 % on the in-place copy yardstick,
// profiles/r01_membench_rmw_patterns.txt).  Row tiles share each line with their XCD sibling and
// need the L2 to merge the halves, so they use ordinary accesses (nontemporal rows: -1.5 %).
template <bool NT, typename T>
__device__ __forceinline__ cx<T> stream_load(const cx<T>* p) {
  typedef T vec2 __attribute__((ext_vector_type(2)));
  if constexpr (NT) {
    const vec2 t = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(p));
    return {t.x, t.y};
  } else {
    return *p;
  }
}
template <bool NT, typename T>
__device__ __forceinline__ void stream_store(cx<T>* p, cx<T> v) {
  typedef T vec2 __attribute__((ext_vector_type(2)));
  if constexpr (NT) __builtin_nontemporal_store(vec2{v.x, v.y}, reinterpret_cast<vec2*>(p));
  else *p = v;
}

// direction as data: conj(FFT(conj x)) with the conjugations as multiplications by +-1
template <typename T, int N, int E, bool SPLIT>
__device__ __forceinline__ void frugal_fft(cx<T>* v, void* lds, int t, const cx<T>* tw, double inv) {
  const T s = inv != 0.0 ? (T)-1 : (T)1;
#pragma unroll
  for (int k = 0; k < E; ++k) v[k].y *= s;
  __builtin_amdgcn_sched_barrier(0);
  fft_stages<T, N, E, +1, SPLIT, 1, 1>(v, lds, t, tw);
  unpermute_slots<N, E>(v);
#pragma unroll
  for (int k = 0; k < E; ++k) v[k].y *= s;
  __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int N, int E, int LINES, int TILES, int AXIS, int BR, int BC, bool SPLIT,
          int KPRE, int KMID, int NFFT>
__global__ void __launch_bounds__(TILES* LINES* N / E, 4)
    frugal_pass_kernel(FrugalArgs a) {
  const int item = blockIdx.y;
  // constant address space: the per-item records are invariant during the kernel, so the scalar
  // loads of their fields may be kept or merged across the workgroup barriers instead of re-issued
  typedef const __attribute__((address_space(4))) FrugalItem* ConstItemPtr;
  const FrugalItem& it = *(const FrugalItem*)((ConstItemPtr)a.items + item);
  if (it.active == 0.0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const TileMap<N, E, LINES, TILES, AXIS, BR, BC> m(blockIdx.x, threadIdx.x, a.pitch);
  cx<T>* f = reinterpret_cast<cx<T>*>(a.field) + (size_t)item * a.item_stride;
  void* lds = smem + (size_t)m.lds_line * line_lds_bytes<T, N, SPLIT>();
  // The stage twiddles (indices < 256 of the table for every supported N) sit in LDS behind the
  // exchange areas: the load that follows each exchange barrier is then a ~100-cycle ds_read
  // instead of a dependent global load.  Published by the first exchange's barriers.
  cx<T>* tw_lds = reinterpret_cast<cx<T>*>(smem + (size_t)TILES * LINES * line_lds_bytes<T, N, SPLIT>());
  for (int i = threadIdx.x; i < kTwiddleLds; i += TILES * LINES * N / E)
    tw_lds[i] = reinterpret_cast<const cx<T>*>(a.tw)[i];
  const cx<T>* tw = tw_lds;
#if PAOS_STAMPS
  if (threadIdx.x == 0) {
    unsigned long long* st = a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kStampSlots;
    st[8] = __builtin_amdgcn_s_memrealtime();
    st[9] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
  PAOS_STAMP(0);

  cx<T> v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = stream_load<AXIS == 1>(&f[m.base + (unsigned)k * m.stride]);
  __builtin_amdgcn_sched_barrier(0);
  PAOS_STAMP_WAIT_VM();
  PAOS_STAMP(1);

  frugal_slot<T, N, E, KPRE>(v, it.pre, it.pre_ph, m);
  PAOS_STAMP(2);
  const bool ran1 = it.fft1_on != 0.0;
  if (ran1) frugal_fft<T, N, E, SPLIT>(v, lds, m.t, tw, it.fft1_inv);
  PAOS_STAMP(3);
  frugal_slot<T, N, E, KMID>(v, it.mid, it.mid_ph, m);
  PAOS_STAMP(4);
  if constexpr (NFFT == 2) {
    if (it.fft2_on != 0.0) {
      if (ran1) __syncthreads();
      frugal_fft<T, N, E, SPLIT>(v, lds, m.t, tw, it.fft2_inv);
    }
  }
  PAOS_STAMP(5);
#pragma unroll
  for (int k = 0; k < E; ++k) stream_store<AXIS == 1>(&f[m.base + (unsigned)k * m.stride], v[k]);
  PAOS_STAMP(6);
  PAOS_STAMP_WAIT_VM();
  PAOS_STAMP(7);
#if PAOS_STAMPS
  if (threadIdx.x == 0)
    a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kStampSlots + 10] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace paos
