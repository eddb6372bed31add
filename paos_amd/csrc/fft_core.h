// fft_core.h -- register-radix Stockham FFT building blocks for gfx950 (CDNA4).
//
// One "line" (a row or a column of the N x N field) is transformed by N/E
// threads that hold E complex elements each in VGPRs.  Every stage is a radix-E
// (last stage: the remaining radix) DFT done entirely in registers; between
// stages the line is exchanged through LDS with the Stockham auto-sort index
// map, so the result comes out in natural order and the first load / last store
// hit global memory with unit stride across lanes.  No MFMA: a c128 FFT has
// ~2 flop/byte, far below the machine balance, so the pass is HBM-bound and the
// job of this file is to keep every global access a full 128-byte line.
//
// Replaces, on the GPU, numpy.fft.fft2 / ifft2 as called by the reference at
// paos/classes/wfo.py:462-472 (ptp), :493-509 (stw), :535-545 (wts).
#pragma once
#include <hip/hip_runtime.h>

namespace paos {

template <typename T>
struct cx {
  T x, y;
};

template <typename T>
__device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
  return {fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)};
}
template <typename T>
__device__ __forceinline__ cx<T> cconj(cx<T> a) { return {a.x, -a.y}; }
// a - i b, a + i b, and (re, im) -> (p.x re + q.x im_or_re ...) helpers used by the butterflies:
template <typename T>
__device__ __forceinline__ cx<T> sub_ib(cx<T> a, cx<T> b) { return {a.x + b.y, a.y - b.x}; }
template <typename T>
__device__ __forceinline__ cx<T> add_ib(cx<T> a, cx<T> b) { return {a.x - b.y, a.y + b.x}; }
// a + c b and a - c b for a real c (FMA)
template <typename T>
__device__ __forceinline__ cx<T> fma_real(T c, cx<T> b, cx<T> a) { return {fma(c, b.x, a.x), fma(c, b.y, a.y)}; }
// (b.x - tau b.y, b.y + tau b.x) = b (1 + i tau)
template <typename T>
__device__ __forceinline__ cx<T> rot_tau(cx<T> b, T tau) { return {fma(-tau, b.y, b.x), fma(tau, b.x, b.y)}; }
// a - i c b and a + i c b for a real c
template <typename T>
__device__ __forceinline__ cx<T> fma_mic(T c, cx<T> b, cx<T> a) { return {fma(c, b.y, a.x), fma(-c, b.x, a.y)}; }
template <typename T>
__device__ __forceinline__ cx<T> fma_pic(T c, cx<T> b, cx<T> a) { return {fma(-c, b.y, a.x), fma(c, b.x, a.y)}; }
// v (c + i s) for compile-time c, s
template <typename T>
__device__ __forceinline__ cx<T> cmul_const(cx<T> v, T c, T s) { return {fma(v.x, c, -(v.y * s)), fma(v.x, s, v.y * c)}; }
// (v.x p, v.y q)
template <typename T>
__device__ __forceinline__ cx<T> scale2(cx<T> v, T p, T q) { return {v.x * p, v.y * q}; }

// ---- complex64 on the packed fp32 pipe --------------------------------------------------------
// A complex64 value is a (re, im) register pair, and gfx950 has packed fp32 arithmetic on such pairs
// (v_pk_add / mul / fma_f32) whose operand modifiers pick halves (op_sel) and flip signs (neg_lo / neg_hi) per
// lane: a complex sum is ONE instruction, a complex product two, a +- i b one.  Left to itself the compiler
// vectorises across unrelated values instead and spends a quarter of the vector instructions of a complex64
// pass on register moves (546 v_mov_b32 of 2316, round 2).  These overloads spell the operations out; the two
// with lane-wise negation of a variable operand -- which the compiler does not fold into the modifiers -- are
// two lines of assembly.  Per byte moved a complex64 pass has twice the points of a complex128 one, so this is
// what lets the fp32 mode approach 2x instead of 1.5x.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk(cx<float> a) { return f2{a.x, a.y}; }
__device__ __forceinline__ cx<float> unpk(f2 v) { return {v.x, v.y}; }
__device__ __forceinline__ cx<float> cadd(cx<float> a, cx<float> b) { return unpk(pk(a) + pk(b)); }
__device__ __forceinline__ cx<float> csub(cx<float> a, cx<float> b) { return unpk(pk(a) - pk(b)); }
__device__ __forceinline__ cx<float> cmul(cx<float> a, cx<float> b) {
  f2 t, r;
  const f2 av = pk(a), bv = pk(b);
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));                       // (a.x b.x, a.x b.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                    // (-a.y b.y, a.y b.x) + t
      : "=v"(r) : "v"(av), "v"(bv), "v"(t));
  return unpk(r);
}
// The same product where ``b`` may come straight out of the transcendental unit (v_sin_f32 / v_cos_f32): gfx950
// wants one wait state between such an instruction and a consumer of its result.  The compiler inserts it for the
// instructions it emits itself but does not look into inline assembly, so the wait state is part of the text.
__device__ __forceinline__ cx<float> cmul_after_trans(cx<float> a, cx<float> b) {
  f2 t, r;
  const f2 av = pk(a), bv = pk(b);
  asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
  return unpk(r);
}
__device__ __forceinline__ cx<float> sub_ib(cx<float> a, cx<float> b) {  // (a.x + b.y, a.y - b.x)
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(pk(a)), "v"(pk(b)));
  return unpk(r);
}
__device__ __forceinline__ cx<float> add_ib(cx<float> a, cx<float> b) {  // (a.x - b.y, a.y + b.x)
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(pk(a)), "v"(pk(b)));
  return unpk(r);
}
__device__ __forceinline__ cx<float> fma_real(float c, cx<float> b, cx<float> a) {
  return unpk(__builtin_elementwise_fma(f2{c, c}, pk(b), pk(a)));
}
__device__ __forceinline__ cx<float> rot_tau(cx<float> b, float tau) {
  const f2 bv = pk(b);
  return unpk(__builtin_elementwise_fma(bv.yx, f2{-tau, tau}, bv));
}
__device__ __forceinline__ cx<float> fma_mic(float c, cx<float> b, cx<float> a) {
  return unpk(__builtin_elementwise_fma(pk(b).yx, f2{c, -c}, pk(a)));
}
__device__ __forceinline__ cx<float> fma_pic(float c, cx<float> b, cx<float> a) {
  return unpk(__builtin_elementwise_fma(pk(b).yx, f2{-c, c}, pk(a)));
}
__device__ __forceinline__ cx<float> cmul_const(cx<float> v, float c, float s) {
  const f2 vv = pk(v);
  return unpk(__builtin_elementwise_fma(vv.yy, f2{-s, c}, vv.xx * f2{c, s}));
}
__device__ __forceinline__ cx<float> scale2(cx<float> v, float p, float q) { return unpk(pk(v) * f2{p, q}); }

// sin and cos of a double: three-term Cody-Waite reduction by pi/2 + the classic minimax
// kernels on [-pi/4, pi/4].  With FMA the first reduction step a - n*pio2_1 is exact for any
// n (the difference needs <= 34 bits), the later steps round at 1e-16 of |r| <= 1, and the
// three-term pi/2 is good to 8.5e-32 * n, so the result is ~1 ulp for |a| up to 2^40
// (the rounding trick below needs |a * 2/pi| < 2^51); the
// library validates on the host that no phase argument exceeds kMaxPhaseArg (paos_hip.hip)
// so the kernels carry no Payne-Hanek fallback.  ~45 instructions, branch-free.
constexpr double kMaxPhaseArg = 1.0e12;
__device__ __forceinline__ void sincos_fast(double a, double* sn, double* cs) {
  // round(a * 2/pi) by the 1.5 * 2^52 trick: the integer lands in the low mantissa bits, so its
  // two low bits (the quadrant) are read straight from the register, no conversion
  const double kMagic = 6755399441055744.0;
  const double nb = fma(a, 0.63661977236758134308, kMagic);  // 2/pi
  const unsigned q = (unsigned)__double2loint(nb);
  const double n = nb - kMagic;
  double r = fma(-n, 1.57079632673412561417e+00, a);
  r = fma(-n, 6.07710050630396597660e-11, r);
  r = fma(-n, 2.02226624879595063154e-21, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double s = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
  const bool odd = (q & 1u) != 0;
  const double s1 = odd ? c : s;
  const double c1 = odd ? s : c;
  // signs: sin flips in quadrants 2, 3; cos in 1, 2 -- an XOR on the sign bit
  const int sflip = (int)((q & 2u) << 30), cflip = (int)(((q + 1u) & 2u) << 30);
  *sn = __hiloint2double(__double2hiint(s1) ^ sflip, __double2loint(s1));
  *cs = __hiloint2double(__double2hiint(c1) ^ cflip, __double2loint(c1));
}

// cos(2 pi m / 32), m = 0..8 (first octant pair); everything else by symmetry.
__device__ constexpr double kCos32[9] = {
    1.0,
    0.98078528040323044913,
    0.92387953251128675613,
    0.83146961230254523708,
    0.70710678118654752440,
    0.55557023301960222474,
    0.38268343236508977173,
    0.19509032201612826785,
    0.0};

__host__ __device__ constexpr double cos32(int m) {
  m &= 31;
  if (m > 16) m = 32 - m;
  return (m > 8) ? -kCos32[16 - m] : kCos32[m];
}
__host__ __device__ constexpr double sin32(int m) { return cos32(m - 8); }

// v * W_R^m with W_R = exp(-2 pi i / R) for DIR = +1 (forward) and its
// conjugate for DIR = -1.  m is a compile-time constant after unrolling, so
// the trivial cases fold away.
template <int R, int DIR, typename T>
__device__ __forceinline__ cx<T> mulw(cx<T> v, int m) {
  const int m32 = (m * (32 / R)) & 31;
  if (m32 == 0) return v;
  if (m32 == 16) return {-v.x, -v.y};
  if (m32 == 8) return (DIR > 0) ? cx<T>{v.y, -v.x} : cx<T>{-v.y, v.x};    // * -/+ i
  if (m32 == 24) return (DIR > 0) ? cx<T>{-v.y, v.x} : cx<T>{v.y, -v.x};   // * +/- i
  const T c = (T)cos32(m32);
  const T s = (T)((DIR > 0) ? -sin32(m32) : sin32(m32));
  return cmul_const(v, c, s);
}

// ---- 16-point DFT with the inner twiddles folded into the butterflies (Linzer-Feig) ---------------
// The 4 x 4 decomposition below multiplies nine of the sixteen intermediate values by W16^m between its two
// layers: 8 non-trivial complex products, 32 of the 160 fp64 instructions.  With w = c (1 + i tau), tau = tan,
//   w b = c u,  u = (b.x - tau b.y, b.y + tau b.x)           2 FMA instead of 4 instructions,
// and the factor c rides on the FMA that replaces the butterfly's addition:  a +- w b = fma(+-c, u, a).
// For the pair  w1 b1 +- w3 b3 = c1 (u1 +- (c3 / c1) u3)  the factor c1 moves one level down the same way.
// A radix-4 butterfly with three twiddled inputs costs 22 FMA instead of 28 instructions; the DFT 144 instead
// of 160 -- 16 fewer per transform stage, six stages per two-transform pass, in kernels that run at the rate of
// the fp64 pipe (profiles/r02_valu_mix_fftbench.txt).  All constants are compile-time; cos(2 pi m / 16) is never
// zero for the m used here (m = 4, the multiplication by -+i, is kept exact and free).
#ifndef PAOS_LF16
#define PAOS_LF16 1
#endif
template <int M16, int DIR>
struct W16 {  // W16^M16 for DIR = +1 (exp(-2 pi i M16 / 16)), its conjugate for DIR = -1
  static constexpr int m32 = (M16 * 2) & 31;
  static constexpr bool quarter = m32 == 8 || m32 == 24;  // * -+i: exact, no arithmetic
  static constexpr bool times_minus_i = (m32 == 8) == (DIR > 0);
  static constexpr double c = cos32(m32);
  static constexpr double s = (DIR > 0) ? -sin32(m32) : sin32(m32);
  static constexpr double tau = quarter ? 0.0 : s / c;
};
template <typename T>
__device__ __forceinline__ cx<T> lf_u(cx<T> b, double tau) { return rot_tau(b, (T)tau); }
// out[k2], k2 = 0..3, of the radix-4 butterfly over (b0, W^K1 b1, W^2K1 b2, W^3K1 b3), W = W16
template <int K1, int DIR, typename T>
__device__ __forceinline__ void lf_twiddled_dft4(const cx<T>* b, cx<T>* out) {
  using W1 = W16<K1, DIR>;
  using W2 = W16<2 * K1, DIR>;
  using W3 = W16<3 * K1, DIR>;
  static_assert(!W1::quarter && !W3::quarter, "K1 and 3 K1 are never multiples of 4 here");
  cx<T> t0, t1;
  if constexpr (W2::quarter) {
    const cx<T> wb = W2::times_minus_i ? cx<T>{b[2].y, -b[2].x} : cx<T>{-b[2].y, b[2].x};
    t0 = cadd(b[0], wb);
    t1 = csub(b[0], wb);
  } else {
    const cx<T> u2 = lf_u(b[2], W2::tau);
    t0 = fma_real((T)W2::c, u2, b[0]);
    t1 = fma_real((T)-W2::c, u2, b[0]);
  }
  const cx<T> u1 = lf_u(b[1], W1::tau), u3 = lf_u(b[3], W3::tau);
  constexpr double rho = W3::c / W1::c;
  const cx<T> t2 = fma_real((T)rho, u3, u1);   // (w1 b1 + w3 b3) / c1
  const cx<T> t3 = fma_real((T)-rho, u3, u1);  // (w1 b1 - w3 b3) / c1
  constexpr T c1 = (T)W1::c;
  out[0] = fma_real(c1, t2, t0);
  out[2] = fma_real(-c1, t2, t0);
  if constexpr (DIR > 0) {  // t1 -+ i c1 t3
    out[1] = fma_mic(c1, t3, t1);
    out[3] = fma_pic(c1, t3, t1);
  } else {
    out[1] = fma_pic(c1, t3, t1);
    out[3] = fma_mic(c1, t3, t1);
  }
}

template <int R, int DIR, typename T>
__device__ __forceinline__ void dft(cx<T>* v);

template <int DIR, typename T>
__device__ __forceinline__ void dft16_lf(cx<T>* v) {
  cx<T> y[16];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {  // first layer: n = 4 n1 + n2 -> a[k1], untwiddled
    cx<T> a[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
    dft<4, DIR>(a);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) y[k1 * 4 + n2] = a[k1];
  }
  cx<T> o[4];
  {
    cx<T> b[4] = {y[0], y[1], y[2], y[3]};
    dft<4, DIR>(b);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[4 * k2] = b[k2];
  }
  lf_twiddled_dft4<1, DIR>(y + 4, o);
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) v[1 + 4 * k2] = o[k2];
  lf_twiddled_dft4<2, DIR>(y + 8, o);
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) v[2 + 4 * k2] = o[k2];
  lf_twiddled_dft4<3, DIR>(y + 12, o);
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) v[3 + 4 * k2] = o[k2];
}

// In-place DFT of R points held in registers, natural order in and out.
template <int R, int DIR, typename T>
__device__ __forceinline__ void dft(cx<T>* v) {
  if constexpr (R == 16 && PAOS_LF16 != 0) {
    dft16_lf<DIR>(v);
  } else if constexpr (R == 2) {
    cx<T> a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  } else if constexpr (R == 4) {
    cx<T> t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    cx<T> t2 = cadd(v[1], v[3]), t3 = csub(v[1], v[3]);
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    if constexpr (DIR > 0) {
      v[1] = sub_ib(t1, t3);
      v[3] = add_ib(t1, t3);
    } else {
      v[1] = add_ib(t1, t3);
      v[3] = sub_ib(t1, t3);
    }
  } else if constexpr (R > 4) {
    // R = A * B, n = B n1 + n2, k = k1 + A k2:
    //   W_R^{nk} = W_A^{n1 k1} W_R^{n2 k1} W_B^{n2 k2}
    constexpr int A = 4, B = R / 4;
    cx<T> y[R];
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) {
      cx<T> a[A];
#pragma unroll
      for (int n1 = 0; n1 < A; ++n1) a[n1] = v[B * n1 + n2];
      dft<A, DIR>(a);
#pragma unroll
      for (int k1 = 0; k1 < A; ++k1) y[k1 * B + n2] = mulw<R, DIR>(a[k1], n2 * k1);
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
      cx<T> b[B];
#pragma unroll
      for (int n2 = 0; n2 < B; ++n2) b[n2] = y[k1 * B + n2];
      dft<B, DIR>(b);
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = b[k2];
    }
  }
}

// LDS padding: one extra slot every 16 so that the stride-R scatter of the
// first exchange and the stride-1 gather both stay conflict-free.
__device__ __forceinline__ constexpr int lds_pad(int i) { return i + (i >> 4); }
template <int N>
constexpr int lds_line_slots() { return N + (N >> 4); }

template <int N, int E, int NS>
struct StageInfo {
  static constexpr int REM = N / NS;               // points still to combine
  static constexpr int R = (REM >= E) ? E : REM;   // radix of this stage
  static constexpr int TPT = E / R;                // butterflies per thread
  static constexpr bool LAST = (NS * R == N);
};

// radix of the final stage / butterflies per thread in it
template <int N, int E, int NS = 1>
constexpr int last_radix() {
  if constexpr (StageInfo<N, E, NS>::LAST) return StageInfo<N, E, NS>::R;
  else return last_radix<N, E, NS * StageInfo<N, E, NS>::R>();
}

// Slot k of the register file after the transform holds X[t + outslot(k) * TL]
// (TL = N / E threads per line); before it, slot k holds x[t + k * TL].
template <int N, int E>
__device__ __forceinline__ constexpr int outslot(int k) {
  constexpr int RL = last_radix<N, E>();
  constexpr int TPTL = E / RL;
  return (k / RL) + (k % RL) * TPTL;
}

// Twiddle powers v[r] *= w^r.  w^r = (w^4)^a * w^b keeps only five powers live
// (w, w^2, w^3, w^4 and the running (w^4)^a) instead of R of them, and bounds
// the product depth at R/4 + 1.
template <int R, typename T>
__device__ __forceinline__ void apply_twiddle_powers(cx<T>* v, cx<T> w1) {
  v[1] = cmul(v[1], w1);
  if constexpr (R > 2) {
    const cx<T> w2 = cmul(w1, w1);
    const cx<T> w3 = cmul(w2, w1);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    if constexpr (R > 4) {
      const cx<T> w4 = cmul(w2, w2);
      cx<T> base = w4;
#pragma unroll
      for (int a = 1; a < R / 4; ++a) {
        v[4 * a] = cmul(v[4 * a], base);
        v[4 * a + 1] = cmul(v[4 * a + 1], cmul(base, w1));
        v[4 * a + 2] = cmul(v[4 * a + 2], cmul(base, w2));
        v[4 * a + 3] = cmul(v[4 * a + 3], cmul(base, w3));
        if (a + 1 < R / 4) base = cmul(base, w4);
      }
    }
  }
}

// Register-frugal twiddle powers: a running product, two twiddles live (vs five above).
// Error grows linearly (<= R roundings, ~2e-15 for R = 16): well inside the parity budget.
template <int R, typename T>
__device__ __forceinline__ void apply_twiddle_chain(cx<T>* v, cx<T> w1) {
  cx<T> w = w1;
#pragma unroll
  for (int r = 1; r < R; ++r) {
    v[r] = cmul(v[r], w);
    if (r + 1 < R) w = cmul(w, w1);
  }
}

// PAOS_DIAG (microbench timing builds only; results are wrong): bit 0 drops the barriers,
// bit 1 drops the LDS traffic of the exchanges.
#ifndef PAOS_DIAG
#define PAOS_DIAG 0
#endif
#define PAOS_SYNC() do { if (!(PAOS_DIAG & 1)) __syncthreads(); } while (0)
// inside fft_stages; bit 2 (timing only): only the first exchange keeps its barriers
#define PAOS_SYNC_X() do { if (!((PAOS_DIAG & 4) && NS > 1)) PAOS_SYNC(); } while (0)

// All stages of one line.  ``lds`` is this line's exchange area: cx<T> slots
// when !SPLIT, T slots (real and imaginary parts exchanged one after the other,
// halving the LDS footprint) when SPLIT.  ``tw`` = exp(-2 pi i m / N), m < N.
// FR (frugal): sequential twiddle chain and scheduling fences between the phases of a stage,
// which keeps a 4096-point c128 pass at ~118 VGPRs -- four waves per SIMD, i.e. two 512-thread
// workgroups per CU whose load / compute / store phases overlap (profiles/r01_vgpr_experiments.txt).
#define PAOS_FENCE() do { if constexpr (FR != 0) __builtin_amdgcn_sched_barrier(0); } while (0)
// (round-5 scheduling experiments, tools/fftbench.hip builds only: bit 0 drops the fence behind an exchange -- the next stage's
// twiddle loads and first products may then start while the exchange's last reads are in flight --, bit 1 the fence behind the
// twiddle loop, bit 2 the fences inside the circle-twiddle loop)
#ifndef PAOS_NOFENCE
#define PAOS_NOFENCE 0
#endif
#define PAOS_FENCE_IF(bit) do { if constexpr (FR != 0 && !(PAOS_NOFENCE & (bit))) __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef PAOS_TAIL_FENCE
#define PAOS_TAIL_FENCE 1
#endif
#ifndef PAOS_TABLE_TWIDDLES
#define PAOS_TABLE_TWIDDLES 1
#endif
// ``circle`` (frugal complex128 kernels, else nullptr): exp(+2 pi i m / 256), m < 256, in LDS.  The
// stage whose twiddles are 256th roots of unity (NS * R == 256: w^r = exp(-2 pi i k r / 256) with
// k r < 256) reads its 15 powers from it instead of multiplying them up -- 15 ds_read_b128 for
// 14 complex products.
// ``w1_last`` (optional): the base twiddles tw[t + s TL] of the butterflies s = 0 .. TPT - 1 a thread runs in the LAST stage
// (N = 4096: one, k = t; N = 2048: two) -- per-thread constants the caller keeps in registers, so that the stage needs no
// table in LDS (the four-per-CU shapes of frugal_pass.h only fit without the 4 KiB table).
template <typename T, int N, int E, int DIR, bool SPLIT, int NS = 1, int FR = 0>
__device__ __forceinline__ void fft_stages(cx<T>* v, void* lds, int t,
                                           const cx<T>* __restrict__ tw, const cx<double>* circle = nullptr,
                                           const cx<T>* w1_last = nullptr) {
  using S = StageInfo<N, E, NS>;
  constexpr int TL = N / E;
  constexpr int R = S::R;
  constexpr int TPT = S::TPT;

#pragma unroll
  for (int s = 0; s < TPT; ++s) {
    if constexpr (NS > 1) {
      const int j = t + s * TL;
      const int k = j & (NS - 1);
      if constexpr (FR != 0 && PAOS_TABLE_TWIDDLES && sizeof(T) == 8 && NS * R == 256) {
        // two reads in flight ahead of the product that consumes them; the fences keep the scheduler
        // from hoisting all fifteen (60 VGPRs) above the first product
        cx<double> wa = circle[k], wb = circle[2 * k];
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const cx<double> w = wa;
          wa = wb;
          if (r + 2 < R) wb = circle[k * (r + 2)];
          v[s * R + r] = cmul(v[s * R + r], cx<T>{(T)w.x, (T)(DIR > 0 ? -w.y : w.y)});
          PAOS_FENCE_IF(4);
        }
      } else {
        cx<T> w1 = (w1_last != nullptr && NS * R == N) ? w1_last[s] : tw[k * (N / (NS * R))];
        if constexpr (DIR < 0) w1.y = -w1.y;
        if constexpr (FR != 0) apply_twiddle_chain<R>(v + s * R, w1);
        else apply_twiddle_powers<R>(v + s * R, w1);
      }
      PAOS_FENCE_IF(2);
    }
    // (timing diagnostics, results wrong: PAOS_DIAG bit 3 drops the butterflies of every FIRST stage, bit 4 of every
    // LAST stage -- what a zero-aware first / last stage could save at the very most, profiles/r04_zero_aware_stages.txt)
    if constexpr (!(((PAOS_DIAG & 8) && NS == 1) || ((PAOS_DIAG & 16) && S::LAST))) dft<R, DIR>(v + s * R);
    // (PAOS_TAIL_FENCE = 0, experiment: no fence behind the LAST stage's butterflies, so that the scheduler may start
    // the tile's stores while the remaining outputs are still being computed)
    if constexpr (!(S::LAST && PAOS_TAIL_FENCE == 0)) PAOS_FENCE();
  }

  if constexpr (!S::LAST) {
    using SN = StageInfo<N, E, NS * R>;
    constexpr int R2 = SN::R;
    constexpr int TPT2 = SN::TPT;
    // Scatter/gather slots are pad(base) + a compile-time offset (no carry out of
    // the low four bits for power-of-two radices), so each thread needs one LDS
    // address VGPR per exchange and the ds_* immediates carry the rest.
    constexpr bool LIN_R = (TL % 16 == 0) && ((N / R2) % 16 == 0);
    int wb[TPT];
#pragma unroll
    for (int s = 0; s < TPT; ++s) {
      const int j = t + s * TL;
      wb[s] = lds_pad((j / NS) * (NS * R) + (j & (NS - 1)));
    }
    const int rb = lds_pad(t);
    auto ridx = [&](int s, int r) -> int {
      const int off = s * TL + r * (N / R2);
      return LIN_R ? rb + lds_pad(off) : lds_pad(t + off);
    };
    if constexpr (!SPLIT) {
      cx<T>* l = reinterpret_cast<cx<T>*>(lds);
#pragma unroll
      for (int s = 0; s < TPT; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r) if (!(PAOS_DIAG & 2)) l[wb[s] + lds_pad(r * NS)] = v[s * R + r];
      PAOS_SYNC_X();
#pragma unroll
      for (int s = 0; s < TPT2; ++s)
#pragma unroll
        for (int r = 0; r < R2; ++r) if (!(PAOS_DIAG & 2)) v[s * R2 + r] = l[ridx(s, r)];
      if constexpr (!SN::LAST) PAOS_SYNC_X();
    } else {
      T* l = reinterpret_cast<T*>(lds);
#pragma unroll
      for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int s = 0; s < TPT; ++s)
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (!(PAOS_DIAG & 2)) l[wb[s] + lds_pad(r * NS)] = part ? v[s * R + r].y : v[s * R + r].x;
        PAOS_SYNC_X();
#pragma unroll
        for (int s = 0; s < TPT2; ++s)
#pragma unroll
          for (int r = 0; r < R2; ++r) {
            if (!(PAOS_DIAG & 2)) {
              const T val = l[ridx(s, r)];
              if (part) v[s * R2 + r].y = val; else v[s * R2 + r].x = val;
            }
          }
        if (part == 0 || !SN::LAST) PAOS_SYNC_X();
      }
    }
    PAOS_FENCE_IF(1);
    fft_stages<T, N, E, DIR, SPLIT, NS * R, FR>(v, lds, t, tw, circle, w1_last);
  }
}

// ---- 4096-point transforms with a digit-swapped side (the two transforms of a fused pass) ----------
// A pass that runs FFT | pointwise | FFT needs natural order only at its two ends (global memory).  In
// between, the 256 threads of a line may hold the 4096 points in any order as long as each thread knows
// which positions it owns.  That freedom removes the barriers of one exchange per transform:
//
//   index digits  n = t0 + 16 t1 + 256 s  (thread t = t0 + 16 t1 holds slots s = 0..15)
//   NAT -> SWP (decimation in frequency):
//     DFT16 over s -> a | * W4096^(t a) | X1: (t0, t1; a) -> (t0, a; t1)   cross-wave: LDS + barriers
//     DFT16 over slots  -> b | * W256^(t0 b) | X2: (t0, a; b) -> (b, a; t0)    inside the 16 lanes of a group:
//     DFT16 over slots  -> c                                                 LDS, no barrier
//     thread (t0 = b, t1 = a), slot c holds X[a + 16 b + 256 c]: position sigma(t) + 256 c, sigma = nibble swap
//   SWP -> NAT is the transposed flow (the DFT matrix is symmetric): the same three DFT16s, the same two
//   diagonal twiddles at the same (thread, slot) places, the exchanges inverted and in reverse order.
//
// X1 moves slot a of thread t to row a of a 16 x 272 table (272 = 16 * 17: the second digit of a row is
// padded to 17 so that the in-group transposition X2 reads and writes it without bank conflicts; rows 272
// apart put the two groups of a 32-lane LDS cycle on the two halves of the banks).  Row a is read -- and
// later reused for X2 -- only by the 16 threads of group a, which sit in one wave: X2 needs no barrier.
// Real and imaginary parts take turns (SPLIT) like in fft_stages.  Seven barriers per transform become three.
constexpr int kSwapRow = 272;

__device__ __forceinline__ int swap_nibbles(int t) { return ((t & 15) << 4) | (t >> 4); }

// scatter ``v[r]`` to ``lds[wbase + r * wstride]``, [barrier], gather ``v[r]`` from ``lds[rbase + r * rstride]``
template <typename T, bool BARRIER>
__device__ __forceinline__ void swap_exchange(cx<T>* v, T* lds, int wbase, int wstride, int rbase, int rstride, bool first) {
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    if (BARRIER && (part == 1 || !first)) __syncthreads();  // the area may still be read (previous round / transform)
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[wbase + r * wstride] = part ? v[r].y : v[r].x;
    if (BARRIER) __syncthreads();
    else __builtin_amdgcn_wave_barrier();  // same wave: LDS operations execute in order; keep the compiler from reordering
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const T val = lds[rbase + r * rstride];
      if (part) v[r].y = val; else v[r].x = val;
    }
    if (!BARRIER) __builtin_amdgcn_wave_barrier();
  }
}

// v[b] *= W256^(t0 b) = conj(circle[t0 b]), two table reads in flight
template <typename T>
__device__ __forceinline__ void swap_twiddle256(cx<T>* v, const cx<double>* circle, int t0) {
  cx<double> wa = circle[t0], wb = circle[2 * t0];
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    const cx<double> w = wa;
    wa = wb;
    if (r + 2 < 16) wb = circle[t0 * (r + 2)];
    v[r] = cmul(v[r], cx<T>{(T)w.x, (T)-w.y});
    __builtin_amdgcn_sched_barrier(0);
  }
}

// natural order in (slot s = x[t + 256 s]) -> digit-swapped out (slot c = X[swap_nibbles(t) + 256 c])
template <typename T>
__device__ __forceinline__ void fft4096_nat_to_swapped(cx<T>* v, T* lds, int t, const cx<T>* tw, const cx<double>* circle,
                                                       bool area_idle) {
  const int t0 = t & 15, t1 = t >> 4;
  dft<16, +1>(v);
  __builtin_amdgcn_sched_barrier(0);
  apply_twiddle_chain<16>(v, tw[t]);
  __builtin_amdgcn_sched_barrier(0);
  swap_exchange<T, true>(v, lds, t, kSwapRow, t1 * kSwapRow + t0, 16, area_idle);
  __builtin_amdgcn_sched_barrier(0);
  dft<16, +1>(v);
  __builtin_amdgcn_sched_barrier(0);
  swap_twiddle256(v, circle, t0);
  swap_exchange<T, false>(v, lds, t1 * kSwapRow + t0, 17, t1 * kSwapRow + t0 * 17, 1, false);
  __builtin_amdgcn_sched_barrier(0);
  dft<16, +1>(v);
  __builtin_amdgcn_sched_barrier(0);
}

// digit-swapped in -> natural order out: the transposed flow
template <typename T>
__device__ __forceinline__ void fft4096_swapped_to_nat(cx<T>* v, T* lds, int t, const cx<T>* tw, const cx<double>* circle) {
  const int t0 = t & 15, t1 = t >> 4;
  dft<16, +1>(v);
  __builtin_amdgcn_sched_barrier(0);
  // X2 transposed: (b = t0, a = t1; r) -> (r, a; b); row t1 is private to this group
  swap_exchange<T, false>(v, lds, t1 * kSwapRow + t0, 17, t1 * kSwapRow + t0 * 17, 1, false);
  swap_twiddle256(v, circle, t0);
  dft<16, +1>(v);
  __builtin_amdgcn_sched_barrier(0);
  // X1 transposed: (t0, a = t1; s') -> (t0 + 16 s'; a): writes stay in the group's own row
  swap_exchange<T, true>(v, lds, t1 * kSwapRow + t0, 16, t, kSwapRow, true);
  __builtin_amdgcn_sched_barrier(0);
  apply_twiddle_chain<16>(v, tw[t]);
  __builtin_amdgcn_sched_barrier(0);
  dft<16, +1>(v);
  __builtin_amdgcn_sched_barrier(0);
}

// Undo the output slot permutation in registers (compile-time renaming), so
// that slot k holds X[t + k * TL] again -- lets a forward transform feed an
// inverse one with no LDS trip in between (fused ptp column pass).
template <int N, int E, typename T>
__device__ __forceinline__ void unpermute_slots(cx<T>* v) {
  cx<T> tmp[E];
#pragma unroll
  for (int k = 0; k < E; ++k) tmp[outslot<N, E>(k)] = v[k];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = tmp[k];
}

// --------------------------------------------------------------------------
// HBM layout of a field.  Blocks of BR rows x BC columns are contiguous, blocks
// are ordered row-major.  BR = 1 is plain row-major.  The library uses BR x BC = 4 x 2: with
// c128 a block is exactly one 128-byte line, which the column pass (2 columns per workgroup)
// moves whole and the row pass moves whole (4 rows per workgroup) or in halves (2 rows, N >= 2048).
// ``pitch`` = elements from one block-row to the next: n * BR when dense; the
// context pads it by a few blocks so that the column pass (stride = pitch) does
// not march through HBM channels with a power-of-two stride.
template <int BR, int BC>
__device__ __host__ __forceinline__ constexpr size_t layout_index(int row, int col, size_t pitch) {
  return (size_t)(row / BR) * pitch + (size_t)(col / BC) * (BR * BC) + (row % BR) * BC + (col % BC);
}

}  // namespace paos
