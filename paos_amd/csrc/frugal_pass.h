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
// Round-4 scheduling experiments (tools/build_variant.sh; profiles/r04_ab_variants_bench.txt) -- all three measured
// and left OFF: priority -2.0 / -2.5 % wavefronts/s, prefetch -9 % (full launches 3.43 -> 4.04 ms):
//   PAOS_PRIO_LOAD  p > 0: the waves of a fresh workgroup run at priority p until their tile loads are issued (they
//                   compete for the issue ports with the fp64 stream of the OLDER workgroup of the CU, and the
//                   arbiter prefers the older wave)
//   PAOS_PRIO_STORE p > 0: ... and again from their last butterflies to the end of their stores
//   PAOS_PREFETCH   d > 0: a workgroup touches the tile of workgroup wg + d (one dword per 64 bytes), so that the
//                   tile comes from the Infinity Cache when its own workgroup asks for it about one round later:
//                   bytes in flight are otherwise bounded by the register files (2 x 128 KiB per CU)
#ifndef PAOS_PRIO_LOAD
#define PAOS_PRIO_LOAD 0
#endif
#ifndef PAOS_PRIO_STORE
#define PAOS_PRIO_STORE 0
#endif
#ifndef PAOS_PREFETCH
#define PAOS_PREFETCH 0
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
  // Round 4 (complex128): when every phase of the slot varies ALONG the line only -- the row / column factors of the
  // separable pass programs -- its factor is the same on every line of the pass: [N] factors of THIS item by position,
  // filled by phase_table_kernel right before the pass with the arithmetic of frugal_slot (slot_factor) and read back
  // instead of being evaluated 16 times per thread on each of a thousand lines.  nullptr: the slot evaluates.
  const cx<double>* table;
};
static_assert(sizeof(FrugalSlot) == 8 * sizeof(double), "record of 8-byte fields");
// per-item record, doubles: [fft1_on, fft1_inv, fft2_on, fft2_inv, pruning ranges, pre slot, pre phases[2],
// mid slot, mid phases[3]]
struct FrugalItem {
  double active;  // 0: the item takes no part in this pass (no load, no store)
  double fft1_on, fft1_inv, fft2_on, fft2_inv;
  // Pruning (zero lines of a field stay zero under every operator of a pass, and a transform of a
  // zero line is a zero line): the host tracks which rows / columns an aperture has just zeroed.
  //   lines outside [line_lo, line_hi) come out zero: tiles made of such lines only are not
  //     processed -- written with zeros when line_fill != 0, otherwise left alone because the next
  //     pass (along the other axis) will not read them;
  //   positions along a line outside [pos_lo, pos_hi) are known to be zero (and may hold stale
  //     data): they are not loaded.
  // Full ranges [0, N) switch all of it off.  Bounds are multiples of the block height.
  double line_lo, line_hi, line_fill, pos_lo, pos_hi;
  // positions along a line outside [spos_lo, spos_hi) need not be STORED: the next pass (along the other axis)
  // does not process the tiles of those lines -- an aperture in it is about to zero them
  double spos_lo, spos_hi;
  FrugalSlot pre;
  FrugalPhase pre_ph[kFrugalMaxPre];
  FrugalSlot mid;
  FrugalPhase mid_ph[kFrugalMaxMid];
};

// table entries the stage twiddles can address: the base twiddle of a stage is tw[k N / (NS R)] with
// k < NS, i.e. an index below N / R <= N / 4; with 16 points per thread every supported N stays below 256,
// with 32 points per thread (complex64) below 1024
template <int N, int E>
constexpr int twiddle_lds_entries() { return E == 8 ? N / 8 : (E <= 16 ? 256 : (N / 4 < 1024 ? N / 4 : 1024)); }  // (E = 8: round-5 experiment)

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
  // STORE = 1 (the last pass of a program whose caller wants the PSF, not the field -- plot.py:125-130): |u|^2 goes
  // to `psf` (doubles, the field's own blocked layout and item stride) and its sum over the workgroup's tile to
  // `psf_partial[item * (workgroups per item) + workgroup]`; the field itself is not written
  double* psf;
  double* psf_partial;
  // STORE = 2 builds (round 4; the shapes without phases in front of their first transform -- what a program that ends on a
  // saved surface ends with; as a run-time switch in every build it cost the whole chain 0.4 %, profiles/r04_ab_variants_bench.txt):
  // the pass stores the field as usual AND the sum of |u|^2 over the workgroup's tile
  // to pow_partial[item * (workgroups per item) + workgroup] -- the power of a saved surface (run.py:218-223 callers'
  // sum |wfo|^2) without a separate sweep that reads the field back.  Dead tiles write nothing: the host zeroes the array.
  double* pow_partial;
  // [batch] factors the slot between the transforms multiplies its scale by: ones, except right behind a stop whose
  // 1 / sqrt(power) has been left for this pass to apply (paos_stop_defer_last_power: the stop's own sweep over the field
  // -- a read and a write of every element -- is gone; make_stop, wfo.py:195-201).  Never null.
  const double* dyn_scale;
#if PAOS_STAMPS
  unsigned long long* stamps;  // [gridDim.y][gridDim.x][kStampSlots]
#endif
  // Workgroups of dead lines that have nothing to write need not be launched: when every item's live lines lie in
  // [live_lo, live_hi) and no item wants its dead tiles written (host: launch_lowered; 0, 0 = all lines), the grid
  // covers the workgroups of these lines only and workgroup blockIdx.x stands for wg0 + blockIdx.x (frugal_launch
  // sets wg0, a multiple of 16: TileMap renumbers the tiles inside aligned groups of 16 workgroups).
  unsigned live_lo, live_hi, wg0;
  // host only: launch the TAB build of the shape (every slot with phases reads FrugalSlot::table)
  int tab = 0;
  // host only: the LONG value of the build to launch -- the transforms of the NEXT one or two passes of the program that the
  // launch runs as well (same axis, same lines; their item records follow this pass's in `items`: [2 or 3][batch])
  int fuse = 0;
  // host only (round 5): a single table pass that moves few bytes -- every item loads and stores at most half of its positions --
  // runs on the one-line workgroups of the fused launches (paos_hip.hip: frugal_launch, ONE)
  int one_line = 0;
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

// sin and cos of a phase argument from a 256-entry table of the unit circle in LDS plus two
// short Taylor kernels:  a = n (2 pi / 256) + r, |r| <= pi / 256,  exp(i a) = C[n mod 256] (cos r + i sin r).
// Two-term Cody-Waite reduction with FMA (the host bounds |a| < 1e12, so n < 2^46 and n * lo(pi/128)
// is known to 1e-20), Taylor remainders r^7/5040 < 1e-17 and r^8/40320 < 2e-20: ~1.5e-16 absolute,
// like sincos_fast, in 17 fp64 instructions and one ds_read_b128 instead of ~40 -- the phase factors
// were 40 % of the vector instructions of a two-transform pass, and these passes run at the rate
// the fp64 pipe (and the power cap that throttles it) allows (profiles/r02_timeline_*.txt).
constexpr int kCircleLds = 256;
__device__ __forceinline__ void sincos_tab(double a, const cx<double>* circle, double* sn, double* cs) {
  const double kMagic = 6755399441055744.0;  // 1.5 * 2^52: the integer lands in the low mantissa bits
  const double nb = fma(a, 40.743665431525205956834243423363677, kMagic);  // 128 / pi
  const unsigned j = (unsigned)__double2loint(nb) & (unsigned)(kCircleLds - 1);
  const double n = nb - kMagic;
  double r = fma(-n, 2.45436926061702596754894014318e-02, a);   // hi(pi / 128)
  r = fma(-n, 9.56755311833869697380e-19, r);                   // lo(pi / 128)
  const cx<double> c = circle[j];
  const double z = r * r;
  const double ps = fma(z, 8.33333333333333333333e-03, -1.66666666666666666667e-01);
  const double s1 = fma(r * z, ps, r);
  double pc = fma(z, -1.38888888888888888889e-03, 4.16666666666666666667e-02);
  pc = fma(z, pc, -0.5);
  const double c1 = fma(z, pc, 1.0);
  *cs = fma(c.x, c1, -(c.y * s1));
  *sn = fma(c.y, c1, c.x * s1);
}

// sign flip as an XOR on the sign bit (an integer instruction instead of an fp64 multiply)
template <typename T>
__device__ __forceinline__ T flip_sign(T x, unsigned mask_hi) {
  if constexpr (sizeof(T) == 8)
    return __hiloint2double(__double2hiint(x) ^ (int)mask_hi, __double2loint(x));
  else
    return __uint_as_float(__float_as_uint(x) ^ mask_hi);
}

// PLAIN (the slot in front of the first transform when the pass has no phase there): the host guarantees that
// the slot carries no sign, no scale and no aperture for any item (lower_frugal routes the rare pass that does
// to the KPRE = 1 shape with a null phase), so all that is left is the conjugation in front of an inverse
// transform -- 16 sign flips instead of 32 multiplications by +-1, in 43 of the 44 passes of the SYN20 chain.
// which form the empty slot takes per axis (0: the general slot): tuning knobs, defaults = measured best
#ifndef PAOS_ROW_PRE
#define PAOS_ROW_PRE 2
#endif
#ifndef PAOS_COL_PRE
#define PAOS_COL_PRE 2  // round 3 (24-pass chain, shared phase factors): the paced form is +0.8 % in column passes too (profiles/r03_ab_variants_bench.txt)
#endif
// SHARE (round 3): a quadratic phase is even along the line -- position p and its mirror N - p (mod N; both in the
// centred and in the natural-order coordinates) have the SAME rounded argument, hence the same factor bit for bit.
// Element k of thread t sits at p = t + k TL; its mirror belongs to thread TL - t, element E - 1 - k (thread 0:
// itself, element E - k).  So every thread evaluates the factors of its first E / 2 elements only, leaves them in
// the line's exchange area (idle between the transforms) and fetches the other half from its mirror thread:
// per phase slot 8 x (argument + sincos) = 184 fp64 instructions less per thread for 8 ds_write_b128 + 8
// ds_read_b128 and two workgroup barriers (`area_busy`: the area may still be read by the transform in front).
// The self-mirrored position N / 2 (thread 0, element E / 2) is evaluated by its owner.
#ifndef PAOS_SHARE_PHASES
#define PAOS_SHARE_PHASES 1
#endif
#ifndef PAOS_MERGE_PHASES
#define PAOS_MERGE_PHASES 1  // two phases of one slot through one sincos of their exactly summed arguments
#endif
#ifndef PAOS_LDS_SWIZZLE
#define PAOS_LDS_SWIZZLE 0   // 1: the single-pass shapes of the pass kernel take the bank swizzle of fft_kernels.h (TileMap: SWZ).  Measured and
                             // left OFF (profiles/r05_ab_variants_bench.txt): it removes every LDS bank conflict of the exchanges and buys dense
                             // launches +0.8 %, but a lane pair of an odd line then sits in another 128-byte block than its even neighbour,
                             // and the launches that load or store a quarter of their positions take 2-13 % longer
#endif
#ifndef PAOS_TABLE_FENCE
#define PAOS_TABLE_FENCE 8   // table slots (TAB builds): factors fetched and applied in groups of this many elements.  Measured
                             // (profiles/r04_ab_variants_bench.txt, 11c-f): 2 / 4 / 16 are 12 / 5 / 4 % slower; the next group's loads issued
                             // before the current one is applied (groups of 2 or 4): -2 %; the first 8-12 factors of the slot in front
                             // of the first transform fetched behind the tile's loads: the main shapes spill, -6 %
#endif

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The factor prod_j exp(i q_j) of a slot's K phases at one position: g[j] the signed pixel index along the line
// (centred or natural order, exact in a double), step[j] the sampling along the line, across2[j] the squared coordinate
// across it, coefq[j] the coefficient (its sign flipped behind a conjugated transform), m2[j] = 1 or 2 pi.  Each argument
// is rounded like the reference's (x = g dx; s = x^2 + y^2; q = coef s).  Shared by frugal_slot and phase_table_kernel:
// a table entry IS the value the slot would have computed.
template <int K>
__device__ __forceinline__ cx<double> slot_factor(const double* g, const double* step, const double* across2,
                                                  const double* coefq, const double* m2, const cx<double>* circle) {
  cx<double> p = {1.0, 0.0};
  if constexpr (K == 2 && PAOS_MERGE_PHASES != 0) {
    // Two phases of one slot through ONE sincos: exp(i q0) exp(i q1) = exp(i (q0 + q1)).  The sum of the two rounded
    // arguments is taken exactly (TwoSum: a = fl(q0 + q1), e = q0 + q1 - a, |e| <= ulp(a) / 2 ~ 1e-10 at 1e6 rad)
    // and the tail applied to first order, exp(i (a + e)) = exp(i a) (1 + i e) + O(e^2 ~ 1e-20): 12 + 6 + 17 + 2
    // instructions instead of 2 x 23 + 4.  (Each argument is still the reference's rounded one.)
    const double x0 = __dmul_rn(g[0], step[0]);
    const double q0 = __dmul_rn(m2[0], __dmul_rn(coefq[0], __dadd_rn(__dmul_rn(x0, x0), across2[0])));
    const double x1 = __dmul_rn(g[1], step[1]);
    const double q1 = __dmul_rn(m2[1], __dmul_rn(coefq[1], __dadd_rn(__dmul_rn(x1, x1), across2[1])));
    const double a = __dadd_rn(q0, q1);
    const double bb = __dsub_rn(a, q0);
    const double e = __dadd_rn(__dsub_rn(q0, __dsub_rn(a, bb)), __dsub_rn(q1, bb));
    double sn, cs;
    sincos_tab(a, circle, &sn, &cs);
    return cx<double>{fma(-e, sn, cs), fma(e, cs, sn)};
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const double x = __dmul_rn(g[j], step[j]);
    // x^2 + y^2 in the reference's order (addition commutes, so which of the two is "x" is moot)
    const double s = __dadd_rn(__dmul_rn(x, x), across2[j]);
    // the host stores coef * sgn (lower_frugal): fl(-c s) = -fl(c s), so the argument is the reference's up to
    // its sign, and sin is odd -- no multiply by sgn here
    const double q = __dmul_rn(m2[j], __dmul_rn(coefq[j], s));
    double sn, cs;
    sincos_tab(q, circle, &sn, &cs);
    // only the phase ARGUMENT is rounded like the reference's; the products themselves may use FMA
    p = j == 0 ? cx<double>{cs, sn} : cx<double>{fma(p.x, cs, -(p.y * sn)), fma(p.x, sn, p.y * cs)};
  }
  return p;
}

// The same factor in fp32 mode (complex64 fields): the argument in TURNS, still formed in fp64 (it reaches 1e5 turns), one
// fraction, then the hardware sin / cos (inputs in revolutions).  turn_coef[j] = m2 coef / 2 pi (the sign rides on coef).
// Shared by frugal_slot and phase_table_kernel like slot_factor.
template <int K>
__device__ __forceinline__ cx<float> slot_factor32(const double* g, const double* step, const double* across2,
                                                   const double* turn_coef) {
  cx<float> p = {1.0f, 0.0f};
  if constexpr (K == 2 && PAOS_MERGE_PHASES != 0) {
    // two phases of one slot: their turns add in fp64 (1e5 turns to 1e-11, the field carries 1e-7), then ONE
    // fraction and ONE hardware sin / cos instead of two of each and a rotation
    const double x0 = g[0] * step[0];
    const double x1 = g[1] * step[1];
    const double turns = fma(fma(x0, x0, across2[0]), turn_coef[0], fma(x1, x1, across2[1]) * turn_coef[1]);
    const float frac = (float)(turns - floor(turns));
    return cx<float>{__builtin_amdgcn_cosf(frac), __builtin_amdgcn_sinf(frac)};
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const double x = g[j] * step[j];
    const double turns = fma(x, x, across2[j]) * turn_coef[j];
    const float frac = (float)(turns - floor(turns));
    const float snf = __builtin_amdgcn_sinf(frac), csf = __builtin_amdgcn_cosf(frac);
    p = j == 0 ? cx<float>{csf, snf} : cmul_after_trans(p, cx<float>{csf, snf});
  }
  return p;
}

// RECS > 0: the aperture line records of the workgroup's RECS lines (first line ``lbase``) were fetched by the kernel
// before the tile's loads (wave-uniform: scalar registers); RECS < 0: the kernel staged them in LDS (``recs``);
// RECS = 0: the slot loads its line's record itself.
template <typename T, int N, int E, int K, typename Map, int PLAIN = 0, bool SHARE = false, int RECS = 0, int TAB = 0>
__device__ __forceinline__ void frugal_slot(cx<T>* v, const FrugalSlot& sl, const FrugalPhase* ph,
                                            const Map& m, const cx<double>* circle, bool conj_in, bool conj_out, int tpos,
                                            void* area = nullptr, bool area_busy = false, const MaskLine* recs = nullptr,
                                            int lbase = 0, double extra_scale = 1.0) {
  if constexpr (PLAIN == 1) {  // column passes: the conjugation, nothing else
    static_assert(K == 0, "a plain slot has no phases");
    const unsigned mask = (conj_out != conj_in) ? 0x80000000u : 0u;
#pragma unroll
    for (int k = 0; k < E; ++k) v[k].y = flip_sign(v[k].y, mask);
    return;
  }
  if constexpr (PLAIN == 2) {
    // Row passes.  Measured (profiles/r02_fftbench_plain_slot.txt): with the 16 flips alone they run 2 % SLOWER
    // than with the general slot's 32 multiplications, and so do the multiplications without the aperture branch
    // around them; 32 integer instructions consumed in load order, two elements per scheduling fence like the
    // general slot, are 1 % faster than it.  Their tiles share every 128-byte line with a sibling workgroup and
    // the pace at which the prologue drains the loads evidently matters; the flip of the real part is by a mask
    // that is zero under the host's guarantee (sign_on == 0) but not known to the compiler.
    static_assert(K == 0, "a plain slot has no phases");
    const unsigned mask = (conj_out != conj_in) ? 0x80000000u : 0u;
    const unsigned none = (sl.sign_on != 0.0) ? 0x80000000u : 0u;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      v[k].x = flip_sign(v[k].x, none);
      v[k].y = flip_sign(v[k].y, mask);
      if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  // tpos: position along the line of the thread's element 0 (its elements are TL apart): m.t in natural
  // order, swap_nibbles(m.t) between the two transforms of a digit-swapped pass
  constexpr int TL = N / E;
  static_assert(TL % 2 == 0, "the checkerboard sign is constant along a thread's elements");
  const double sc = __dmul_rn(sl.scale, extra_scale);  // (extra_scale: 1 exactly, except behind a deferred stop)
  const int line = Map::kAxis == 0 ? m.row(0) : m.col(0);  // this thread's row (column): constant
  if (sl.mask_on != 0.0) {  // wave-uniform: an aperture rides on this slot
    MaskLine ml;
    if constexpr (RECS > 0) {
      ml = recs[0];
#pragma unroll
      for (int j = 1; j < RECS; ++j)
        if (line - lbase == j) ml = recs[j];
    } else if constexpr (RECS < 0) {
      ml = recs[line - lbase];  // staged in LDS by the kernel
    } else {
      ml = sl.lines[line];
    }
    // The two partial runs [p0, p1) and [p2, p3) are at most kMaskW positions long, so a thread (positions
    // tpos + k TL) meets each of them at most R = ceil(kMaskW / TL) times, at consecutive k: which k, and which
    // recorded weights, is known before the loop.  The weights are fetched up front, unconditionally (a clamped
    // index where the thread misses the run) -- as conditional loads inside the loop they cost every wave that met a
    // run a full memory latency in the middle of the tile's life, and the pass 40 % of its time
    // (tools/per_launch.py with PAOS_NO_PRUNE=1: 4.9 against 3.5 ms).
    constexpr int R = (kMaskW + TL - 1) / TL;
    const char* vb = reinterpret_cast<const char*>(sl.vals);
    const unsigned vlo = (unsigned)line * (unsigned)(2 * kMaskW * sizeof(double));
    auto val = [&](int idx) { return *reinterpret_cast<const double*>(vb + (vlo + (unsigned)idx * (unsigned)sizeof(double))); };
    // first k at or behind the start of each run: ceil((p - tpos) / TL), the numerator + TL - 1 is never negative
    const int ka = (int)((unsigned)(ml.p0 - tpos + TL - 1) / (unsigned)TL);
    const int kb = (int)((unsigned)(ml.p2 - tpos + TL - 1) / (unsigned)TL);
    double wa[R], wb[R];
    int sel_a[R], sel_b[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int pa = tpos + (ka + j) * TL, pb = tpos + (kb + j) * TL;
      const bool hit_a = ka + j < E && pa < ml.p1, hit_b = kb + j < E && pb < ml.p3;
      wa[j] = val(hit_a ? pa - ml.p0 : 0);
      wb[j] = val(hit_b ? kMaskW + pb - ml.p2 : kMaskW);
      sel_a[j] = hit_a ? ka + j : -1;
      sel_b[j] = hit_b ? kb + j : -1;
    }
    const double w_in = __dmul_rn(sl.w_in, ml.lm), w_out = __dmul_rn(sl.w_out, ml.lm);
#pragma unroll
    for (int j = 0; j < R; ++j) { wa[j] = __dmul_rn(wa[j], ml.lm); wb[j] = __dmul_rn(wb[j], ml.lm); }
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = tpos + k * TL;
      double w = (pos >= ml.p1 && pos < ml.p2) ? w_in : w_out;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        w = (k == sel_a[j]) ? wa[j] : w;
        w = (k == sel_b[j]) ? wb[j] : w;
      }
      if constexpr (sizeof(T) == 4) {
        // fp32 mode: the weight rounded to the field's type and two fp32 products (the fp64 round trip -- two
        // conversions up, two products, two conversions down at the fp64 rate -- bought nothing at 1e-7)
        const float wf = (float)w;
        v[k] = {(T)(v[k].x * wf), (T)(v[k].y * wf)};
      } else {
        v[k] = {(T)__dmul_rn((double)v[k].x, w), (T)__dmul_rn((double)v[k].y, w)};
      }
      if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // (-1)^(row+col): the position along the line advances by TL (even) from element to element
  // sign_on: 1 the checkerboard, 2 its half along the line (-1)^position, 3 its half across (-1)^line (the separable
  // pass programs: passes.py, SeparableCompiler)
  const int parity = sl.sign_on == 1.0 ? line + tpos : (sl.sign_on == 2.0 ? tpos : line);
  const double f = (sl.sign_on != 0.0 && (parity & 1)) ? -sc : sc;
  // The conjugations around an inverse transform (IFFT x = conj FFT conj x) ride on this slot: the one in
  // front of the transform that FOLLOWS flips the sign of the imaginary scale; the one behind the transform
  // that PRECEDES (conj_in: the values in ``v`` are the conjugates of the true ones) does too, since
  //   conj(v) w e^{iq} = conj(v w e^{-iq})        (w: real mask weight)
  // -- and the phases run with -q, a sign flip of the (wave-uniform) coefficient.
  const double fy = (conj_out != conj_in) ? -f : f;
  const int qflip = conj_in ? (int)0x80000000 : 0;
  // Per phase, what does not depend on the element: the squared coordinate ACROSS the line and
  // the scale ALONG it.  Coordinates are exact integers times the sampling step, formed like the
  // reference's (g * dx, one rounding); centred index g = i - N/2, natural order g = i or i - N.
  constexpr int KK = K > 0 ? K : 1;
  // Along the line, element k sits at position t + k TL, which is < N/2 exactly when k < E/2 (t < TL):
  // its signed index is (t + base) + k TL with base = -N/2 (centred), or 0 / -N for the two halves of
  // the natural order -- two per-thread doubles per phase, picked per k at compile time.
  double across2[KK], step[KK], g_lo[KK], g_hi[KK], coefq[KK];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    coefq[j] = __hiloint2double(__double2hiint(ph[j].coef) ^ qflip, __double2loint(ph[j].coef));
    const bool nat = ph[j].natural != 0.0;
    const int ga = nat ? ((line < N / 2) ? line : line - N) : line - N / 2;
    const double a = (double)ga * (Map::kAxis == 0 ? ph[j].sy : ph[j].sx);
    across2[j] = __dmul_rn(a, a);
    step[j] = Map::kAxis == 0 ? ph[j].sx : ph[j].sy;
    g_lo[j] = (double)(nat ? tpos : tpos - N / 2);
    g_hi[j] = (double)(nat ? tpos - N : tpos - N / 2);
  }
  const int mt = tpos == 0 ? 0 : TL - tpos;             // the thread that holds the mirror positions
  const int mbase = (tpos == 0 ? TL : 0) + mt;          // + (E - 1 - k) TL: where element k's factor was left
  if constexpr (sizeof(T) == 4) {
    // fp32 mode: the field carries ~1e-7, so the phase needs no more than that -- but its ARGUMENT
    // reaches 1e6 rad and is still formed in fp64: in turns, s * (m2 coef / 2 pi), one fraction
    // instruction, then the hardware sin / cos (inputs in revolutions) and an fp32 rotation.  No
    // fp64 copy of the element: 8 temporaries instead of 14, which is what lets the kernel keep to
    // the register budget of three workgroups per CU.
    double turn_coef[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) turn_coef[j] = ph[j].m2 * coefq[j] * 0.15915494309189535;  // / 2 pi; coef carries the sign
    const float ff = (float)f, ffy = (float)fy;
    auto factor32 = [&](int k) __attribute__((always_inline)) {
      double gk[KK];
#pragma unroll
      for (int j = 0; j < K; ++j) gk[j] = (k < E / 2 ? g_lo[j] : g_hi[j]) + (double)(k * TL);
      return slot_factor32<K>(gk, step, across2, turn_coef);
    };
    if constexpr (K > 0 && TAB != 0) {  // TAB builds: the factors by position, from phase_table_kernel's table (cx<float> entries)
      static_assert(!SHARE, "a table slot does not stand for a barrier");
      const cx<float>* tb = reinterpret_cast<const cx<float>*>(sl.table) + tpos;
#pragma unroll
      for (int k = 0; k < E; ++k) {
        const cx<float> zs = scale2(cmul(cx<float>{(float)v[k].x, (float)v[k].y}, tb[k * TL]), ff, ffy);
        v[k] = {(T)zs.x, (T)zs.y};
        if ((k + 1) % PAOS_TABLE_FENCE == 0) __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    if constexpr (SHARE && K > 0) {
      cx<float>* fac = reinterpret_cast<cx<float>*>(area);
      if (area_busy) lds_barrier();
#pragma unroll
      for (int k = 0; k < E / 2; ++k) {
        const cx<float> p = factor32(k);
        fac[k * TL + tpos] = p;
        constexpr bool kFresh = K == 1 || (K == 2 && PAOS_MERGE_PHASES != 0);  // p comes straight from the transcendental unit
        const cx<float> zs = scale2(kFresh ? cmul_after_trans(cx<float>{(float)v[k].x, (float)v[k].y}, p)
                                           : cmul(cx<float>{(float)v[k].x, (float)v[k].y}, p), ff, ffy);
        v[k] = {(T)zs.x, (T)zs.y};
        if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
      }
      if (tpos == 0) fac[(E / 2) * TL] = factor32(E / 2);
      lds_barrier();
#pragma unroll
      for (int k = E / 2; k < E; ++k) {
        const cx<float> p = fac[(E - 1 - k) * TL + mbase];
        const cx<float> zs = scale2(cmul(cx<float>{(float)v[k].x, (float)v[k].y}, p), ff, ffy);
        v[k] = {(T)zs.x, (T)zs.y};
      }
      lds_barrier();  // the transform behind the slot writes the area next
      return;
    }
    // (round 4: the slot's ONE factor per element -- two phases through one sin / cos of their summed turns, as the sharing
    // loop above and the tables of the fused launches have it -- instead of one rotation per phase: every path that puts a
    // slot's phases on an element now multiplies by the same number, bit for bit)
#pragma unroll
    for (int k = 0; k < E; ++k) {
      cx<float> z = {(float)v[k].x, (float)v[k].y};
      if constexpr (K > 0) z = cmul_after_trans(z, factor32(k));  // two packed instructions (fft_core.h)
      const cx<float> zs = scale2(z, ff, ffy);
      v[k] = {(T)zs.x, (T)zs.y};
      if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  // the factor prod_j exp(i q_j) of element k: each argument rounded like the reference's (slot_factor)
  double m2s[KK];
#pragma unroll
  for (int j = 0; j < K; ++j) m2s[j] = ph[j].m2;
  auto factor = [&](int k) __attribute__((always_inline)) {
    double gk[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) gk[j] = (k < E / 2 ? g_lo[j] : g_hi[j]) + (double)(k * TL);
    return slot_factor<K>(gk, step, across2, coefq, m2s, circle);
  };
  auto apply = [&](int k, cx<double> p) __attribute__((always_inline)) {
    const cx<double> vd = {(double)v[k].x, (double)v[k].y};
#if defined(PAOS_DIAG_NOSCALE)  // (timing diagnostic: what folding sign and scale into the table would save -- results wrong)
    v[k] = {(T)(fma(vd.x, p.x, -(vd.y * p.y))), (T)(fma(vd.x, p.y, vd.y * p.x))};
#else
    v[k] = {(T)(fma(vd.x, p.x, -(vd.y * p.y)) * f), (T)(fma(vd.x, p.y, vd.y * p.x) * fy)};
#endif
  };
  if constexpr (K > 0 && TAB != 0) {
    // TAB builds (round 4): the factors of this item's slot by position, from phase_table_kernel's table -- a compile-time
    // variant, not a branch: as a wave-uniform branch next to the evaluating code it cost most shapes their spill-free
    // register allocation (60-100 bytes of scratch; both variants 15 % slower, profiles/r04_ab_variants_bench.txt)
    static_assert(!SHARE, "a table slot does not stand for a barrier");
    const cx<double>* tb = sl.table + tpos;
    // (one-line workgroups -- four per CU, one wave each per SIMD -- do best with groups of 4: -2 ... -3 % against 8, which
    // the two-line workgroups keep; 16: +14 %.  profiles/r05_fftbench_fused_variants.txt)
    constexpr int kFence = Map::kLinesPerWorkgroup == 1 ? 4 : PAOS_TABLE_FENCE;
#pragma unroll
    for (int k = 0; k < E; ++k) {
#if defined(PAOS_DIAG_NOTABLE)  // (timing diagnostic: what the table loads cost -- results wrong)
      apply(k, cx<double>{0.6, 0.8});
#else
      apply(k, tb[k * TL]);
#endif
      if ((k + 1) % kFence == 0) __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if constexpr (SHARE && K > 0) {
    cx<double>* fac = reinterpret_cast<cx<double>*>(area);
    if (area_busy) lds_barrier();
#pragma unroll
    for (int k = 0; k < E / 2; ++k) {
      const cx<double> p = factor(k);
      fac[k * TL + tpos] = p;
      apply(k, p);
      if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
    }
    if (tpos == 0) fac[(E / 2) * TL] = factor(E / 2);
    lds_barrier();
#pragma unroll
    for (int k = E / 2; k < E; ++k) {
      apply(k, fac[(E - 1 - k) * TL + mbase]);
      if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();  // the transform behind the slot writes the area next
    return;
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    if constexpr (K > 0) apply(k, factor(k));
    else v[k] = {(T)((double)v[k].x * f), (T)((double)v[k].y * fy)};
    if ((k + 1) % PAOS_FENCE_EVERY == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// The factors of the slots whose phases vary along the line only (FrugalSlot::table), for every item and position:
// blockIdx = (position / 256, item, table) -- the tables of every launch of a pass program in one go --, complex128 passes.  The unit circle of sincos_tab is rebuilt from
// the context's twiddle table exactly as frugal_pass_kernel builds it, the arguments are formed by slot_factor: an entry
// is bit for bit what the slot would have evaluated at that position on any line.
// One table: the slot (`mid` = 0: in front of the first transform, 1: behind it) of the pass whose item records start at
// items[item_base], with `k` phases, along `axis`.
struct PhaseSlotDesc { int item_base, mid, k, axis; };
struct PhaseTableArgs {
  const FrugalItem* items;     // the staged records of every launch the tables are for (paos_hip.hip: stage_groups)
  const cx<double>* tw;        // the context's twiddle table for n
  const PhaseSlotDesc* desc;   // [gridDim.z]
  int n;
  int f32;                     // complex64 fields: cx<float> entries from slot_factor32, no circle table
};
template <int UNIT = 0>  // (a template: the header is compiled into several translation units)
__global__ void __launch_bounds__(256) phase_table_kernel(PhaseTableArgs a) {
  __shared__ cx<double> circle[kCircleLds];
  if (!a.f32) {
    const cx<double> w = a.tw[threadIdx.x * (a.n / kCircleLds)];
    circle[threadIdx.x] = {w.x, -w.y};
  }
  __syncthreads();
  const PhaseSlotDesc d = a.desc[blockIdx.z];
  const FrugalItem& it = a.items[d.item_base + blockIdx.y];
  const bool mid = d.mid != 0;
  const FrugalSlot& sl = mid ? it.mid : it.pre;
  if (it.active == 0.0 || sl.table == nullptr) return;
  const FrugalPhase* ph = mid ? it.mid_ph : it.pre_ph;
  const int K = d.k;
  // behind a conjugated transform the slot runs its phases with -q (frugal_slot: qflip)
  const int qflip = (mid && it.fft1_on != 0.0 && it.fft1_inv != 0.0) ? (int)0x80000000 : 0;
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  double g[kFrugalMaxMid], step[kFrugalMaxMid], across2[kFrugalMaxMid], coefq[kFrugalMaxMid], m2[kFrugalMaxMid];
#pragma unroll
  for (int j = 0; j < kFrugalMaxMid; ++j) {
    const FrugalPhase& q = ph[j < K ? j : 0];
    g[j] = (double)(q.natural != 0.0 ? (pos < a.n / 2 ? pos : pos - a.n) : pos - a.n / 2);
    step[j] = d.axis == 0 ? q.sx : q.sy;
    across2[j] = 0.0;
    coefq[j] = __hiloint2double(__double2hiint(q.coef) ^ qflip, __double2loint(q.coef));
    m2[j] = q.m2;
  }
  if (a.f32) {
    double turn_coef[kFrugalMaxMid];
#pragma unroll
    for (int j = 0; j < kFrugalMaxMid; ++j) turn_coef[j] = m2[j] * coefq[j] * 0.15915494309189535;
    cx<float> pf = {1.0f, 0.0f};
    if (K <= 0) {}
    else if (K == 1) pf = slot_factor32<1>(g, step, across2, turn_coef);
    else if (K == 2) pf = slot_factor32<2>(g, step, across2, turn_coef);
    else pf = slot_factor32<3>(g, step, across2, turn_coef);
    reinterpret_cast<cx<float>*>(const_cast<cx<double>*>(sl.table))[pos] = pf;
    return;
  }
  cx<double> p = {1.0, 0.0};  // (K = 0: a slot without phases that rides in a fused pair)
  if (K <= 0) {}
  else if (K == 1) p = slot_factor<1>(g, step, across2, coefq, m2, circle);
  else if (K == 2) p = slot_factor<2>(g, step, across2, coefq, m2, circle);
  else p = slot_factor<3>(g, step, across2, coefq, m2, circle);
  const_cast<cx<double>*>(sl.table)[pos] = p;
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

// Direction as data: IFFT(x) = conj(FFT(conj x)).  A scalar branch between a forward and an inverse
// instantiation would spare the conjugations but costs ~60 spilled VGPRs at the join (measured), so
// the direction stays data and the conjugations are made cheap instead: the one in FRONT of the
// transform is folded into the slot that precedes it (frugal_slot multiplies the imaginary part by
// -f instead of f), the one BEHIND it is an XOR on the sign bit (half the issue cost of an fp64 multiply).
// FLIP = false: the conjugation behind the transform is left to the slot that follows (frugal_slot: conj_in)
template <typename T, int N, int E, bool SPLIT, bool FLIP = true>
__device__ __forceinline__ void frugal_fft(cx<T>* v, void* lds, int t, const cx<T>* tw, const cx<double>* circle,
                                           double inv, const cx<T>* w1_last = nullptr) {
  fft_stages<T, N, E, +1, SPLIT, 1, 1>(v, lds, t, tw, circle, w1_last);
  unpermute_slots<N, E>(v);
  if constexpr (FLIP) {
    const unsigned mask = inv != 0.0 ? 0x80000000u : 0u;
#pragma unroll
    for (int k = 0; k < E; ++k) v[k].y = flip_sign(v[k].y, mask);
  }
  if constexpr (!(FLIP && PAOS_TAIL_FENCE == 0)) __builtin_amdgcn_sched_barrier(0);
}

// dynamic LDS of one workgroup: exchange areas | stage twiddles | (c128 with phases) circle table
constexpr size_t kStoreScratch = 16 * sizeof(double);  // one partial sum per wave (<= 16 waves) of a STORE = 1 workgroup
#ifndef PAOS_HOIST_RECORDS
#define PAOS_HOIST_RECORDS 1  // experiment knob: 0 = the slot between the transforms loads its aperture record itself
#endif
// How the slot between the transforms gets the aperture line records of the workgroup's lines: 2 = fetched in front of
// the tile's loads into scalar registers (two-line workgroups with an empty first slot), 1 = fetched behind the tile's
// loads into a few bytes of LDS (the others: more lines, or a first slot whose register budget has no room),
// 0 = loaded by the slot when it gets there.
template <int LINES, int TILES, int KPRE>
constexpr int frugal_record_mode() {
  return PAOS_HOIST_RECORDS == 0 ? 0 : ((TILES * LINES == 2 && KPRE == 0) ? 2 : (TILES * LINES <= 16 ? 1 : 0));
}
// One-line workgroups (round 5: 4096-point complex128 lines, 256 threads): the only stage twiddle that is not a 256th root of
// unity is the last stage's tw[t] -- one value per thread for the whole kernel, kept in registers -- so the 4 KiB stage table
// stays out of LDS and FOUR workgroups (4 x 38.2 KiB) fit a CU.
// OCC = 1 (round 5, N = 2048 / 1024 complex128, the launches that are bound by their latency chain): the same workgroup of 256
// threads (two lines of 2048 points, four of 1024), but FOUR of them per CU instead of three -- 128 VGPRs and, like the
// one-line shapes of 4096, no stage table in LDS (the last stage's two / four twiddles per thread, tw[t + s N / 16], live in
// registers): 4 x 38.6 KiB.
template <typename T, int N, int E, int LINES, int TILES, int OCC = 0>
constexpr bool frugal_tw_in_regs() {
  return sizeof(T) == 8 && E == 16 && ((N == 4096 && TILES * LINES == 1) || (N == 2048 && OCC != 0 && TILES * LINES == 2) ||
                                       (N == 1024 && OCC != 0 && TILES * LINES == 4));
}
template <typename T, int N, int LINES, int TILES, bool SPLIT, int KPRE, int KMID, int E = 16, int STORE = 0, int OCC = 0>
constexpr size_t frugal_lds_bytes() {
  return (size_t)TILES * LINES * line_lds_bytes<T, N, SPLIT, LINES>() +
         (frugal_tw_in_regs<T, N, E, LINES, TILES, OCC>() ? 0 : twiddle_lds_entries<N, E>() * sizeof(cx<T>)) +
         (sizeof(T) == 8 ? kCircleLds * sizeof(cx<double>) : 0) + kStoreScratch +
         (frugal_record_mode<LINES, TILES, KPRE>() == 1 ? (size_t)TILES * LINES * sizeof(MaskLine) : 0);
}

// Waves per SIMD the kernel is compiled for.  N = 4096: 512-thread workgroups, two per CU (their
// tiles fill the register file) = 4 waves per SIMD = 128 VGPRs.  Smaller N: 256-thread workgroups
// whose number per CU is bounded by LDS -- three of ~43 KiB -- so the compiler may as well have the 168
// VGPRs three waves per SIMD leave it (at 128 these shapes spilled 50-180 B per lane).
#ifndef PAOS_NT_FULL_LINES
#define PAOS_NT_FULL_LINES 1
#endif
#ifndef PAOS_SHARED_NT_LOADS
#define PAOS_SHARED_NT_LOADS 0
#endif
#ifndef PAOS_MINW_SMALL
#define PAOS_MINW_SMALL 3
#endif
// complex64 tiles hold half the registers of complex128 ones: with the exchange split into real and
// imaginary halves (35 KiB per 512-thread workgroup) THREE workgroups fit a CU if the kernel keeps to
// 80 VGPRs (6 waves per SIMD) -- a third actor to overlap the memory phases with (PAOS_F32_MINW).
#ifndef PAOS_F32_MINW
#define PAOS_F32_MINW 4
#endif
template <typename T, int N, int THREADS, int OCC = 0>
constexpr int frugal_min_waves() {
  if (OCC != 0) return 4;  // (four 256-thread workgroups per CU: 128 VGPRs)
  if (sizeof(T) == 4 && THREADS >= 1024) return 8;  // 4-row tiles of complex64: two 1024-thread workgroups per CU
  if (sizeof(T) == 4 && THREADS >= 512) return PAOS_F32_MINW;
  if (sizeof(T) == 8 && N == 4096 && THREADS == 256) return 4;  // one-line workgroups: four per CU, one wave each per SIMD
#ifdef PAOS_E8_MINW
  if (sizeof(T) == 8 && N == 4096 && THREADS == 512) return PAOS_E8_MINW;  // (experiment: 8 points per thread, one line per workgroup)
#endif
  return THREADS >= 512 ? 4 : PAOS_MINW_SMALL;
}

// sum |u|^2 of the tile, one value per workgroup (STORE = 1): lanes by shuffles, waves through the scratch doubles
template <int THREADS>
__device__ __forceinline__ void tile_power_out(double acc, double* scratch, double* out) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) s += scratch[w];
    *out = s;
  }
}

// The kernel takes the fields of FrugalArgs as separate scalar arguments, the ones the prologue needs first: built with
// -mllvm -amdgpu-kernarg-preload-count=N (Makefile) they arrive in SGPRs with the wave, instead of by a scalar load
// that every workgroup waits for before it can even ask for its item's record.  Launch with PAOS_FRUGAL_PASS(args).
#if PAOS_STAMPS
#define PAOS_FRUGAL_PARAMS const FrugalItem *k_items, void *k_field, unsigned k_pitch, unsigned k_item_stride, unsigned k_wg0, \
                           const void *k_tw, double *k_psf, double *k_psf_partial, double *k_pow_partial, const double *k_dyn_scale, \
                           unsigned long long *k_stamps
#define PAOS_FRUGAL_PASS(a) (a).items, (a).field, (a).pitch, (a).item_stride, (a).wg0, (a).tw, (a).psf, (a).psf_partial, (a).pow_partial, (a).dyn_scale, (a).stamps
#else
#define PAOS_FRUGAL_PARAMS const FrugalItem *k_items, void *k_field, unsigned k_pitch, unsigned k_item_stride, unsigned k_wg0, \
                           const void *k_tw, double *k_psf, double *k_psf_partial, double *k_pow_partial, const double *k_dyn_scale
#define PAOS_FRUGAL_PASS(a) (a).items, (a).field, (a).pitch, (a).item_stride, (a).wg0, (a).tw, (a).psf, (a).psf_partial, (a).pow_partial, (a).dyn_scale
#endif
// TAB != 0 (complex128): every slot that has phases (KPRE / KMID = 1: however many) reads its factors from the item's
// table by position (FrugalSlot::table) instead of evaluating them.
// LONG = 1 ... 4 (TAB builds with phases in both slots): the launch goes on with the NEXT pass -- or the next two: LONG counts
// their transforms, [1], [2], [2][1], [2][2] -- of the program: same axis,
// same lines, one / two transforms, phases in both of its slots -- whose item records follow this pass's
// (items[batch + item], items[2 batch + item]): load | slot F slot F | slot F slot [F] | [slot F slot [F]] | store.  The tile never leaves the registers between the
// two passes: one load, one store, one prologue and one launch less per pair; results are bit-identical.
template <typename T, int N, int E, int LINES, int TILES, int AXIS, int BR, int BC, bool SPLIT,
          int KPRE, int KMID, int NFFT, int STORE = 0, int TAB = 0, int LONG = 0, int OCC = 0>
__global__ void __launch_bounds__(TILES* LINES* N / E, (frugal_min_waves<T, N, TILES * LINES * N / E, OCC>()))
    frugal_pass_kernel(PAOS_FRUGAL_PARAMS) {
  static_assert(LONG == 0 || (TAB != 0 && (NFFT == 2 || NFFT == 3) && KPRE == 1 && KMID == 1), "LONG builds: see above");
  FrugalArgs a;
  a.items = k_items; a.field = k_field; a.pitch = k_pitch; a.item_stride = k_item_stride; a.wg0 = k_wg0;
  a.tw = k_tw; a.psf = k_psf; a.psf_partial = k_psf_partial; a.pow_partial = k_pow_partial; a.dyn_scale = k_dyn_scale; a.live_lo = a.live_hi = 0;
#if PAOS_STAMPS
  a.stamps = k_stamps;
#endif
  const int item = blockIdx.y;
  // constant address space: the per-item records are invariant during the kernel, so the scalar
  // loads of their fields may be kept or merged across the workgroup barriers instead of re-issued
  typedef const __attribute__((address_space(4))) FrugalItem* ConstItemPtr;
  const FrugalItem& it = *(const FrugalItem*)((ConstItemPtr)a.items + item);
  // The prologue is what a tile pays before its first load is in flight, with a CU slot already taken.  As the
  // compiler laid it out it was a chain of five dependent scalar loads (one item field, wait, branch, the next
  // field, ...) plus -- for the waves that fill the LDS tables -- a full global round trip (table load, wait,
  // ds_write) in front of the tile loads.  Now the item's pruning header is fetched in one go (the empty asm needs
  // all of it in scalar registers at this point, so the loads are merged into wide ones and waited for once), and
  // the table fill has moved behind the tile loads (below).
  constexpr int kThreads = TILES * LINES * N / E;
  constexpr int kTwiddleLds = twiddle_lds_entries<N, E>();
  constexpr int kTwIt = (kTwiddleLds + kThreads - 1) / kThreads;
  constexpr int kClIt = sizeof(T) == 8 ? (kCircleLds + kThreads - 1) / kThreads : 1;
  // (LONG builds: what is written, and what becomes of dead tiles, is the second pass's business)
  constexpr int kExtra = (LONG + 1) / 2;  // passes behind the first one that this launch runs too
  const FrugalItem& it_last = LONG == 0 ? it : *(const FrugalItem*)((ConstItemPtr)a.items + item + kExtra * gridDim.y);
  const double h_active = it.active, h_line_lo = it.line_lo, h_line_hi = it.line_hi, h_line_fill = it_last.line_fill,
               h_pos_lo = it.pos_lo, h_pos_hi = it.pos_hi, h_spos_lo = it_last.spos_lo, h_spos_hi = it_last.spos_hi;
  // (shapes that fetch their aperture line records in front of the tile: the switch and the pointer ride along
  // instead of costing two more dependent scalar loads behind the header)
  const double h_mask_on = it.mid.mask_on;
  const MaskLine* const h_lines = it.mid.lines;
  // (... and the item's dynamic scale factor: one more scalar load in the same round trip)
#if defined(PAOS_NO_DYN)  // (A/B build: what the dynamic factor costs the passes that never see one)
  const double h_dyn = 1.0;
#else
  const double h_dyn = *((const __attribute__((address_space(4))) double*)a.dyn_scale + item);
#endif
  if constexpr (frugal_record_mode<LINES, TILES, KPRE>() != 0)
    asm volatile("" ::"s"(h_active), "s"(h_line_lo), "s"(h_line_hi), "s"(h_line_fill), "s"(h_pos_lo), "s"(h_pos_hi), "s"(h_spos_lo),
                 "s"(h_spos_hi), "s"(h_mask_on), "s"(h_lines), "s"(h_dyn));
  else
    asm volatile("" ::"s"(h_active), "s"(h_line_lo), "s"(h_line_hi), "s"(h_line_fill), "s"(h_pos_lo), "s"(h_pos_hi), "s"(h_spos_lo),
                 "s"(h_spos_hi), "s"(h_dyn));
  if (h_active == 0.0) return;
  if constexpr (PAOS_PRIO_LOAD > 0) __builtin_amdgcn_s_setprio(PAOS_PRIO_LOAD);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // a block of BR x BC elements is a whole 128-byte line for complex128, half of one for complex64
  constexpr bool kBlockIsLine = BR * BC * sizeof(cx<T>) >= 128;
  // (column tiles of fewer columns than a block has share the block's lines with BC / LINES - 1 siblings too)
  constexpr int COLSIB = (kBlockIsLine ? 1 : (int)(128 / (BR * BC * sizeof(cx<T>)))) * (AXIS == 1 && LINES < BC ? BC / LINES : 1);
  const unsigned wg = blockIdx.x + a.wg0;  // compact grids (FrugalArgs::wg0)
  constexpr int kSwz = (PAOS_LDS_SWIZZLE != 0 && LONG == 0 && sizeof(T) == 8) ? 1 : 0;
  const TileMap<N, E, LINES, TILES, AXIS, BR, BC, 1, COLSIB, kSwz> m(wg, threadIdx.x, a.pitch);
  cx<T>* f = reinterpret_cast<cx<T>*>(a.field) + (size_t)item * a.item_stride;
  // tiles that own whole 128-byte lines (c128 column tiles; row tiles that span a full block row) stream
  // around the caches; tiles that share lines with a sibling need the L2 to merge the halves
  constexpr bool NT = kBlockIsLine && (PAOS_NT_FULL_LINES ? ((AXIS == 1 && LINES >= BC) || (AXIS == 0 && LINES == BR)) : (AXIS == 1 && LINES >= BC));
  // experiment knob (tools/build_variant.sh): nontemporal LOADS for tiles that share their lines with a sibling
  // (stores stay ordinary so that the L2 can merge the halves)
  constexpr bool NTL = NT || (PAOS_SHARED_NT_LOADS != 0);
  {  // a workgroup of dead lines only: nothing to transform (its tiles are consecutive lines)
    const int l0 = TILES == 1 ? (AXIS == 0 ? m.row0 : m.col0) : (int)wg * (TILES * LINES);
    if (l0 + TILES * LINES <= (int)h_line_lo || l0 >= (int)h_line_hi) {
      if constexpr (STORE == 1) {  // the PSF of a dead tile is zero (and so is its share of the power)
        if (h_line_fill == 0.0) return;  // ... and the buffer is known to hold those zeros already (host: psf_zero_*)
        // ... or holds something only on the lines [spos_lo, spos_hi) the previous storing pass left live
        if (l0 + TILES * LINES <= (int)h_spos_lo || l0 >= (int)h_spos_hi) return;
        double* ps = a.psf + (size_t)item * a.item_stride;
#pragma unroll
        for (int k = 0; k < E; ++k) ps[m.base + (unsigned)k * m.stride] = 0.0;
        if (threadIdx.x == 0) a.psf_partial[(size_t)item * (N / LINES / TILES) + wg] = 0.0;
        return;
      }
      if (h_line_fill != 0.0) {
#pragma unroll
        for (int k = 0; k < E; ++k)
          stream_store<NT>(reinterpret_cast<cx<T>*>(reinterpret_cast<char*>(f) + (m.base + (unsigned)k * m.stride) * (unsigned)sizeof(cx<T>)),
                           cx<T>{(T)0, (T)0});
      }
      return;
    }
  }
  void* lds = smem + (size_t)m.lds_line * line_lds_bytes<T, N, SPLIT, LINES>();
  // The stage twiddles (indices < 256 of the table for every supported N) sit in LDS behind the
  // exchange areas: the load that follows each exchange barrier is then a ~100-cycle ds_read
  // instead of a dependent global load.  Published by the barrier behind the tile's loads.
  cx<T>* tw_lds = reinterpret_cast<cx<T>*>(smem + (size_t)TILES * LINES * line_lds_bytes<T, N, SPLIT, LINES>());
  constexpr bool kTwRegs = frugal_tw_in_regs<T, N, E, LINES, TILES, OCC>();  // no stage table in LDS: tw[t] in registers (below)
  const cx<T>* tw = kTwRegs ? reinterpret_cast<const cx<T>*>(a.tw) : tw_lds;
  // the unit circle in 256 steps for the phase factors (sincos_tab) and the twiddles of the second
  // stage: conj of every (N/256)-th entry of the twiddle table
  cx<double>* cl = reinterpret_cast<cx<double>*>(tw_lds + (kTwRegs ? 0 : kTwiddleLds));
  const cx<double>* circle = sizeof(T) == 8 ? cl : nullptr;
#if PAOS_STAMPS
  if (threadIdx.x == 0) {
    unsigned long long* st = a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kStampSlots;
    st[8] = __builtin_amdgcn_s_memrealtime();
    st[9] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
  PAOS_STAMP(0);

  // tiles that own whole 128-byte lines (column tiles; row tiles that span a full block row) stream
  // around the caches (NT); row tiles that share lines with a sibling need the L2 to merge the halves
  // Addresses as a wave-uniform base (SGPRs) plus a 32-bit byte offset per element (an item is < 4 GiB):
  // 64-bit pointers for the sixteen elements would sit in 32 VGPRs from the loads to the stores.
  const char* fb = reinterpret_cast<const char*>(f);
  // element k sits k * stride further on: that part is wave-uniform and is added to the base on the scalar unit
  const unsigned boff = m.base * (unsigned)sizeof(cx<T>);
  // (readfirstlane pins the sum to a scalar register pair: otherwise it is reassociated into a 64-bit vector add
  // per access; with it the access is "scalar base + 32-bit vector offset" and costs no vector instruction)
  const size_t bstride = (size_t)(m.stride * (unsigned)sizeof(cx<T>));
  auto at = [&](int k) {
    const unsigned long long u = (unsigned long long)(fb + (size_t)k * bstride);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    typedef __attribute__((address_space(1))) char* GlobalBytes;  // keeps the access a global_*, not a flat_* one
    // the zero-extension of the offset has to sit in the basic block of the access for the instruction selector to
    // see "scalar base + 32-bit offset"; hoisted to the entry block (it is common to all 48 accesses) it is a 64-bit
    // register pair and every access pays a 64-bit vector add.  The empty asm keeps it local.
    unsigned bo = boff;
    asm volatile("" : "+v"(bo));
    return (cx<T>*)((GlobalBytes)(((unsigned long long)hi << 32) | lo) + bo);
  };
  // The records of an aperture riding BETWEEN the transforms are wanted in the middle of the tile's life; fetched
  // there they are a dependent global load (and the recorded weights a second one) with nothing to hide behind.
  // The lines of a workgroup are consecutive, so their records are read here, wave-uniformly, in front of the tile.
  // (two-line workgroups without phases in front of the first transform: with four records, or with a busy first
  // slot, the early fetch costs the shape its spill-free register allocation)
  constexpr int kRecMode = frugal_record_mode<LINES, TILES, KPRE>();
  constexpr int kRecs = kRecMode == 2 ? 2 : 0;
  MaskLine mrec[kRecs > 0 ? kRecs : 1] = {};
  const int lbase = __builtin_amdgcn_readfirstlane(TILES == 1 ? (AXIS == 0 ? m.row0 : m.col0) : (int)wg * (TILES * LINES));
  if constexpr (kRecs > 0) {
    if (h_mask_on != 0.0) {
#pragma unroll
      for (int j = 0; j < kRecs; ++j) mrec[j] = h_lines[lbase + j];
    }
  }
  // The empty slot in front of the first transform without its 32 multiplications by +-1 (frugal_slot: PLAIN):
  // 16 sign flips in column passes, a paced variant in row passes.
  constexpr int kPlainPre = KPRE != 0 ? 0 : (AXIS == 1 ? PAOS_COL_PRE : PAOS_ROW_PRE);
  cx<T> v[E];
  const int plo = (int)h_pos_lo, phi = (int)h_pos_hi;
  // PAOS_PREFETCH: one dword out of every 64 bytes of the tile of workgroup wg + d, issued IN FRONT of the own tile's
  // loads.  Loads return in order, so once any tile element has arrived these have too: their (single, never read)
  // destination register stays reserved until the tile's stores (the empty asm in front of them).
  unsigned pf_sink = 0;
  if constexpr (PAOS_PREFETCH > 0) {
    const unsigned wgp = wg + (unsigned)PAOS_PREFETCH;
    if (wgp < a.wg0 + gridDim.x) {
      const TileMap<N, E, LINES, TILES, AXIS, BR, BC, 1, COLSIB, kSwz> mp(wgp, threadIdx.x & ~3u, a.pitch);
      // threads 4 q .. 4 q + 3 own one 64-byte piece per element index k: lane (tid & 3) touches the pieces k = 4 j + (tid & 3)
      typedef const __attribute__((address_space(1))) char* GlobalBytes;
      const GlobalBytes pbase = (GlobalBytes)fb;
#pragma unroll
      for (int j = 0; j < E / 4; ++j) {
        const unsigned off = (mp.base + (unsigned)(4 * j + (threadIdx.x & 3)) * mp.stride) * (unsigned)sizeof(cx<T>);
        asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_sink) : "v"(off), "s"(pbase) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (plo <= 0 && phi >= N) {  // wave-uniform: the whole line is live
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = stream_load<NTL>(at(k));
  } else {
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = m.t + k * (N / E);
      v[k] = cx<T>{(T)0, (T)0};
      if (pos >= plo && pos < phi) v[k] = stream_load<NTL>(at(k));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // The LDS tables are fetched BEHIND the tile's loads (they are L2 hits and return in order right after the
  // tile) and written once everything has arrived: nothing stands between the prologue and the first tile load.
  cx<T> tw_fetch[kTwIt], cl_fetch[kClIt];
  // (the aperture records of the workgroup's lines ride along: one dword per thread of the first wave)
  constexpr int kRecDwords = kRecMode == 1 ? TILES * LINES * (int)(sizeof(MaskLine) / 4) : 0;
  static_assert(kRecDwords <= 128, "the record area is filled by the first 128 threads");
  const bool stage_recs = kRecMode == 1 && h_mask_on != 0.0;
  unsigned rec_fetch = 0;
  if constexpr (kRecMode == 1) {
    if (stage_recs && (int)threadIdx.x < kRecDwords) rec_fetch = reinterpret_cast<const unsigned*>(h_lines + lbase)[threadIdx.x];
  }
  // (one-line workgroups: the thread's one stage twiddle rides behind the tile's loads like the table fetches do)
  constexpr int kLastTpt = E / last_radix<N, E>();  // butterflies per thread in the last stage: their twiddles tw[t + s TL]
  cx<T> w1_keep[kLastTpt];
#pragma unroll
  for (int q = 0; q < kLastTpt; ++q) w1_keep[q] = cx<T>{(T)1, (T)0};
  if constexpr (kTwRegs) {
#pragma unroll
    for (int q = 0; q < kLastTpt; ++q) w1_keep[q] = reinterpret_cast<const cx<T>*>(a.tw)[m.t + q * (N / E)];
  }
  const cx<T>* const w1_last = kTwRegs ? w1_keep : nullptr;
#pragma unroll
  for (int j = 0; j < kTwIt; ++j) {
    const int i = (int)threadIdx.x + j * kThreads;
    tw_fetch[j] = cx<T>{(T)0, (T)0};
    if (!kTwRegs && i < kTwiddleLds) tw_fetch[j] = reinterpret_cast<const cx<T>*>(a.tw)[i];
  }
  if constexpr (sizeof(T) == 8) {
#pragma unroll
    for (int j = 0; j < kClIt; ++j) {
      const int i = (int)threadIdx.x + j * kThreads;
      cl_fetch[j] = cx<T>{(T)0, (T)0};
      if (i < kCircleLds) cl_fetch[j] = reinterpret_cast<const cx<T>*>(a.tw)[i * (N / kCircleLds)];
    }
  }
#pragma unroll
  for (int j = 0; j < kTwIt; ++j) {
    const int i = (int)threadIdx.x + j * kThreads;
    if (!kTwRegs && i < kTwiddleLds) tw_lds[i] = tw_fetch[j];
  }
  if constexpr (sizeof(T) == 8) {
#pragma unroll
    for (int j = 0; j < kClIt; ++j) {
      const int i = (int)threadIdx.x + j * kThreads;
      if (i < kCircleLds) cl[i] = {(double)cl_fetch[j].x, -(double)cl_fetch[j].y};
    }
  }
  MaskLine* rec_lds = reinterpret_cast<MaskLine*>(smem + frugal_lds_bytes<T, N, LINES, TILES, SPLIT, KPRE, KMID, E, 0, OCC>() -
                                                  (kRecMode == 1 ? (size_t)TILES * LINES * sizeof(MaskLine) : 0));
  if constexpr (kRecMode == 1) {
    if (stage_recs && (int)threadIdx.x < kRecDwords) reinterpret_cast<unsigned*>(rec_lds)[threadIdx.x] = rec_fetch;
  }
  // The LDS tables above are read before the first exchange barrier when a phase sits in front of the first
  // transform (sincos_tab in the pre slot), or when the first transform is switched off: publish them here,
  // with the tile's loads already in flight.  LDS only -- a __syncthreads() would also wait for those loads.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (PAOS_PRIO_LOAD > 0) __builtin_amdgcn_s_setprio(0);
  PAOS_STAMP_WAIT_VM();
  PAOS_STAMP(1);

  if constexpr (NFFT == 3) {
    // two transforms with a digit-swapped layout in between (fft_core.h: fft4096_nat_to_swapped): the host
    // launches this shape only when both transforms run for every active item of the batch
    static_assert(N == 4096 && E == 16 && sizeof(T) == 8 && SPLIT, "digit-swapped passes: 4096-point complex128 lines");
    T* area = reinterpret_cast<T*>(lds);
    // (round 5 experiment: LONG = 2 runs the next pass of the program the same way -- two digit-swapped pairs per launch, three
    // workgroup barriers per transform instead of seven; every transform of both passes must be on for every item)
#define PAOS_SWAPPED_PASS(ix, kpre, plain, idle, dyn)                                                                                        \
  {                                                                                                                                          \
    frugal_slot<T, N, E, kpre, decltype(m), plain, false, 0, TAB>(v, (ix).pre, (ix).pre_ph, m, circle, false, (ix).fft1_inv != 0.0, m.t);    \
    fft4096_nat_to_swapped<T>(v, area, m.t, tw, circle, idle);                                                                               \
    {                                                                                                                                        \
      const unsigned mask = (ix).fft1_inv != 0.0 ? 0x80000000u : 0u;                                                                         \
      _Pragma("unroll") for (int k = 0; k < E; ++k) v[k].y = flip_sign(v[k].y, mask);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
    }                                                                                                                                        \
    /* (dyn: a deferred stop's 1 / sqrt(power) rides on this slot like on the NFFT <= 2 shapes' -- ADVICE r04) */                            \
    frugal_slot<T, N, E, KMID, decltype(m), 0, false, 0, TAB>(v, (ix).mid, (ix).mid_ph, m, circle, false, (ix).fft2_inv != 0.0,              \
                                                              swap_nibbles(m.t), nullptr, false, nullptr, 0, dyn);                           \
    fft4096_swapped_to_nat<T>(v, area, m.t, tw, circle);                                                                                     \
    {                                                                                                                                        \
      const unsigned mask = (ix).fft2_inv != 0.0 ? 0x80000000u : 0u;                                                                         \
      _Pragma("unroll") for (int k = 0; k < E; ++k) v[k].y = flip_sign(v[k].y, mask);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                                     \
    }                                                                                                                                        \
  }
    PAOS_SWAPPED_PASS(it, KPRE, kPlainPre, true, h_dyn)
    if constexpr (LONG == 2) {
      const FrugalItem& it2 = *(const FrugalItem*)((ConstItemPtr)a.items + item + gridDim.y);
      __syncthreads();  // the exchange area is still being read by the transform in front
      PAOS_SWAPPED_PASS(it2, 1, 0, true, 1.0)
    }
#undef PAOS_SWAPPED_PASS
    PAOS_STAMP(5);
    {
      const int slo3 = (int)h_spos_lo, shi3 = (int)h_spos_hi;
      if (slo3 <= 0 && shi3 >= N) {
#pragma unroll
        for (int k = 0; k < E; ++k) stream_store<NT>(at(k), v[k]);
      } else {
#pragma unroll
        for (int k = 0; k < E; ++k) {
          const int pos = m.t + k * (N / E);
          if (pos >= slo3 && pos < shi3) stream_store<NT>(at(k), v[k]);
        }
      }
    }
    return;
  }
  const bool ran1 = it.fft1_on != 0.0;
  const bool ran2 = NFFT == 2 && it.fft2_on != 0.0;
  const bool inv1 = ran1 && it.fft1_inv != 0.0;
  constexpr bool kShare = PAOS_SHARE_PHASES != 0 && TAB == 0;
  // (with phases in BOTH slots the second sharing loop costs the shape its spill-free register allocation: there
  // only the slot between the transforms shares)
  frugal_slot<T, N, E, KPRE, decltype(m), kPlainPre, kShare && KMID == 0, 0, TAB>(v, it.pre, it.pre_ph, m, circle, false, inv1, m.t, lds, false);
  PAOS_STAMP(2);
  if (ran1) frugal_fft<T, N, E, SPLIT, false>(v, lds, m.t, tw, circle, it.fft1_inv, w1_last);
  PAOS_STAMP(3);
  constexpr bool kShareMid = kShare && KPRE == 0 && KMID < 3;
  frugal_slot<T, N, E, KMID, decltype(m), 0, kShareMid, (kRecMode == 1 ? -1 : kRecs), TAB>(
      v, it.mid, it.mid_ph, m, circle, inv1, ran2 && it.fft2_inv != 0.0, m.t, lds, ran1, kRecMode == 1 ? rec_lds : mrec, lbase, h_dyn);
  PAOS_STAMP(4);
  if constexpr (NFFT == 2) {
    if (ran2) {
      if (ran1 && !(kShareMid && KMID > 0)) __syncthreads();  // (a sharing slot ends on a barrier of its own)
      frugal_fft<T, N, E, SPLIT>(v, lds, m.t, tw, circle, it.fft2_inv, w1_last);
    }
  }
  if constexpr (LONG != 0) {
    // (a macro, not a lambda: a generic lambda that captures the tile by reference -- even inside a discarded
    // `if constexpr` branch -- cost EVERY shape of the family its register allocation: 52-72 B of scratch, make spillcheck)
    bool busy = ran1 || ran2;  // the exchange area may still be read by the transform in front
#define PAOS_EXTRA_PASS(ix, kTwo)                                                                                                    \
  {                                                                                                                                  \
    const bool ran3 = (ix).fft1_on != 0.0;                                                                                           \
    const bool ran4 = (kTwo) && (ix).fft2_on != 0.0;                                                                                 \
    const bool inv3 = ran3 && (ix).fft1_inv != 0.0;                                                                                  \
    frugal_slot<T, N, E, 1, decltype(m), 0, false, 0, TAB>(v, (ix).pre, (ix).pre_ph, m, circle, false, inv3, m.t, lds, false);       \
    if (ran3) {                                                                                                                      \
      if (busy) __syncthreads();                                                                                                     \
      frugal_fft<T, N, E, SPLIT, false>(v, lds, m.t, tw, circle, (ix).fft1_inv, w1_last);                                            \
      busy = true;                                                                                                                   \
    }                                                                                                                                \
    frugal_slot<T, N, E, 1, decltype(m), 0, false, 0, TAB>(v, (ix).mid, (ix).mid_ph, m, circle, inv3, ran4 && (ix).fft2_inv != 0.0,  \
                                                           m.t, lds, ran3);                                                          \
    if constexpr (kTwo) {                                                                                                            \
      if (ran4) {                                                                                                                    \
        if (busy) __syncthreads();                                                                                                   \
        frugal_fft<T, N, E, SPLIT>(v, lds, m.t, tw, circle, (ix).fft2_inv, w1_last);                                                 \
        busy = true;                                                                                                                 \
      }                                                                                                                              \
    }                                                                                                                                \
  }
    // LONG = extra transforms: 1 -> [1], 2 -> [2], 3 -> [2][1], 4 -> [2][2]
    const FrugalItem& it2 = *(const FrugalItem*)((ConstItemPtr)a.items + item + gridDim.y);
    if constexpr (LONG == 1) PAOS_EXTRA_PASS(it2, false)
    else PAOS_EXTRA_PASS(it2, true)
    if constexpr (LONG == 3) PAOS_EXTRA_PASS(it_last, false)
    if constexpr (LONG == 4) PAOS_EXTRA_PASS(it_last, true)
#undef PAOS_EXTRA_PASS
  }
  PAOS_STAMP(5);
  if constexpr (STORE == 1) {
    // |u|^2 as export_kernel forms it (x x + y y, unfused), 8 B per element at the element's own offset; the sum in
    // a fixed order: element by element per thread, lanes, waves
    double* ps = a.psf + (size_t)item * a.item_stride;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const double x = (double)v[k].x, y = (double)v[k].y;
      const double w = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
      ps[m.base + (unsigned)k * m.stride] = w;
      acc += w;
    }
    double* scratch = reinterpret_cast<double*>(smem + frugal_lds_bytes<T, N, LINES, TILES, SPLIT, KPRE, KMID, E, 0, OCC>() - kStoreScratch -
                                                (kRecMode == 1 ? (size_t)TILES * LINES * sizeof(MaskLine) : 0));
    tile_power_out<TILES * LINES * N / E>(acc, scratch, a.psf_partial + (size_t)item * (N / LINES / TILES) + wg);
    return;
  }
  const int slo = (int)h_spos_lo, shi = (int)h_spos_hi;
  if constexpr (PAOS_PREFETCH > 0) asm volatile("" ::"v"(pf_sink));
  if constexpr (PAOS_PRIO_STORE > 0) __builtin_amdgcn_s_setprio(PAOS_PRIO_STORE);
  if (slo <= 0 && shi >= N) {  // wave-uniform: everything is stored
#pragma unroll
    for (int k = 0; k < E; ++k) stream_store<NT>(at(k), v[k]);
  } else {
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = m.t + k * (N / E);
      if (pos >= slo && pos < shi) stream_store<NT>(at(k), v[k]);
    }
  }
  if constexpr (STORE == 2) {  // the power of the field just stored, summed like the PSF-storing builds do
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const double x = (double)v[k].x, y = (double)v[k].y;
      acc += __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
    }
    double* scratch = reinterpret_cast<double*>(smem + frugal_lds_bytes<T, N, LINES, TILES, SPLIT, KPRE, KMID, E, 0, OCC>() - kStoreScratch -
                                                (kRecMode == 1 ? (size_t)TILES * LINES * sizeof(MaskLine) : 0));
    tile_power_out<TILES * LINES * N / E>(acc, scratch, a.pow_partial + (size_t)item * (N / LINES / TILES) + wg);
  }
  PAOS_STAMP(6);
  PAOS_STAMP_WAIT_VM();
  PAOS_STAMP(7);
#if PAOS_STAMPS
  if (threadIdx.x == 0)
    a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kStampSlots + 10] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace paos
