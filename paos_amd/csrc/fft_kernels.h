// fft_kernels.h -- the HBM passes of the 2-D transforms.
//
// A pass = every workgroup loads a tile of LINES lines (rows: AXIS 0, columns:
// AXIS 1) of one batch item into registers, optionally multiplies by a
// checkerboard sign and/or a quadratic phase, transforms the lines, applies
// the output factors and stores the tile back IN PLACE (tiles are disjoint).
//
//   stw  (wfo.py:474-509): fftshift(Q FFT(ifftshift u)) == S Qc FFT(S u)
//   wts  (wfo.py:511-545): fftshift(FFT(ifftshift(P u))) == S FFT(S P u)
//   ptp  (wfo.py:445-472): the shifts cancel: ifft2(H fft2(u)); the forward and
//        the inverse COLUMN transforms meet in registers around the H multiply,
//        so a ptp costs three HBM passes (rows, columns x2 fused, rows) not four.
//
// S = (-1)^(row+col); P, Qc, H are quadratic phases whose argument is formed in
// fp64 with the reference's operation order (no FMA contraction) so that the
// rounded argument -- up to 1e6 rad -- is bit-identical to NumPy's.
#pragma once
#include "fft_core.h"

namespace paos {

// ---- pointwise operators that ride on an FFT pass ----------------------------------
// A pass applies up to three short lists of diagonal operators -- before the first
// transform, between the two transforms, after the second -- each operator with a
// per-item parameter block [enable, sx, sy, coef, sgn] (doubles, device memory).
enum : int {
  PWK_SIGN = 1,       // (-1)^(row+col): the fftshift / ifftshift pair folded into the data
  PWK_QPHASE_C = 2,   // exp(i sgn coef ((gx sx)^2 + (gy sy)^2)), g = index - N/2 (centred)
  PWK_QPHASE_N = 3,   // same with g = natural-order signed frequency index (np.fft.fftfreq)
  PWK_SCALE = 4,      // multiply by block[FP_COEF] (exact power of two: the ortho 1/N)
  PWK_MASK = 5,       // multiply by the aperture weight map rendered just before the pass
};
enum : int { PWF_MUL2PI = 1,   // argument gets an extra factor 2 pi (lens form, wfo.py:363-366)
             PWF_X_ONLY = 2, PWF_Y_ONLY = 4 };  // SIGN: (-1)^column / (-1)^row, the two halves of the checkerboard
// Bits 8.. of PwOp::flags: 1 + index of this operator's separable phase table (0 = no table:
// the phase is evaluated per pixel with sincos).
constexpr int kTableShift = 8;
enum : int { FP_ENABLE = 0, FP_SX = 1, FP_SY = 2, FP_COEF = 3, FP_SGN = 4, FP_STRIDE = 5 };
// FFT control block: [enable, inverse, -, -, -]
enum : int { FC_ENABLE = 0, FC_INVERSE = 1 };
constexpr int kMaxPw = 6;

struct PwOp {
  int kind;
  int flags;
  int block;  // index of the [batch][FP_STRIDE] parameter block set
};

struct PassArgs {
  void* field;            // batch of fields, complex<T>, blocked layout
  const void* tw;         // exp(-2 pi i m / N), m < N, complex<T>
  const double* blocks;   // parameter block sets: [block][item][FP_STRIDE]
  const cx<double>* tables;  // phase tables: [table][item][2][N] = exp(i sgn A_x(col)), exp(i sgn A_y(row))
  const double* mask;     // aperture weights in the field's own layout: [item][item_stride]
  int batch;
  int fft1, fft2;         // control block index of the first / second transform, -1 = none
  int n_pre, n_mid, n_post;
  PwOp pre[kMaxPw], mid[kMaxPw], post[kMaxPw];
  unsigned pitch;         // elements between block rows of the layout
  unsigned item_stride;   // elements between batch items
};

// exp(i * sgn * arg) with arg = coef * ((gx*sx)^2 + (gy*sy)^2) [* 2 pi]
// Operation order follows the reference: x = g*dx; s = x*x + y*y; q = coef*s.
__device__ __forceinline__ cx<double> quad_phase(int gx, int gy, double sx, double sy, double coef,
                                                 double sgn, bool mul2pi) {
  const double x = (double)gx * sx;
  const double y = (double)gy * sy;
  const double s = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
  double q = __dmul_rn(coef, s);
  if (mul2pi) q = __dmul_rn(6.283185307179586, q);
  double sn, cs;
  sincos_fast(q, &sn, &cs);
  return {cs, sgn * sn};
}

// The quadratic phase of one pixel from the separable tables plus an exact correction.
// Reference value: exp(i sgn a), a = fl([2 pi] fl(coef fl(X + Y))), X = fl(x^2), Y = fl(y^2).
// The tables hold exp(i sgn [2 pi] coef X) and exp(i sgn [2 pi] coef Y) for the EXACT real
// products, so   a = [2 pi] coef (X + Y) + delta   with a delta of a few ulp(a) that is
// recovered exactly with error-free transformations (TwoSum, FMA residuals):
//   s + e1 = X + Y,  q + e2 = coef s,  [a + e3 = 2 pi q]
//   delta = -(e2 + coef e1)            [delta' = 2 pi delta - e3]
// and exp(i a) = Ex Ey (1 + i sgn delta) to 1e-20.  ~25 fp64 instructions instead of ~65.
__device__ __forceinline__ cx<double> table_phase(double x, double y, double coef, double sgn,
                                                  bool mul2pi, cx<double> ex, cx<double> ey) {
  const double X = __dmul_rn(x, x), Y = __dmul_rn(y, y);
  const double s = __dadd_rn(X, Y);
  const double bb = __dsub_rn(s, X);
  const double e1 = __dadd_rn(__dsub_rn(X, __dsub_rn(s, bb)), __dsub_rn(Y, bb));
  const double q = __dmul_rn(coef, s);
  const double e2 = fma(coef, s, -q);
  double delta = -fma(coef, e1, e2);
  if (mul2pi) {
    const double a = __dmul_rn(6.283185307179586, q);
    const double e3 = fma(6.283185307179586, q, -a);
    delta = fma(6.283185307179586, delta, -e3);
  }
  delta *= sgn;
  const cx<double> w = {fma(ex.x, ey.x, -(ex.y * ey.y)), fma(ex.x, ey.y, ex.y * ey.x)};
  return {fma(-delta, w.y, w.x), fma(delta, w.x, w.y)};
}

// FEAT selects the optional operators compiled into a kernel: bit 0 = separable phase tables,
// bit 1 = aperture weight maps.  The default kernels carry neither (they cost registers even
// when idle); the dispatcher picks the FEAT = 3 build only for programs that use them.
//
// Every operator is expressed as ONE complex factor per pixel (sign: +-1, scale: c, mask: w,
// phase: cos + i sgn sin) and a single multiply, so the branches on the operator kind merge
// four registers instead of the whole register file of the line (which cost 2x the VGPRs).
// v * (c + 0i) computed as (x c - y 0, x 0 + y c) is exact, so sign/scale/mask keep their
// plain-multiplication results.
template <int FEAT>
__device__ __forceinline__ cx<double> pw_factor(const PwOp& op, const double* p, int row, int col,
                                                int n, const cx<double>* tab, const double* mask_at) {
  if ((FEAT & 2) && op.kind == PWK_MASK) return {*mask_at, 0.0};
  if (op.kind == PWK_SIGN)
    return {((((op.flags & PWF_Y_ONLY) ? 0 : col) + ((op.flags & PWF_X_ONLY) ? 0 : row)) & 1) ? -1.0 : 1.0, 0.0};
  if (op.kind == PWK_SCALE) return {p[FP_COEF], 0.0};
  int gx, gy;
  if (op.kind == PWK_QPHASE_C) {
    gx = col - n / 2; gy = row - n / 2;
  } else {
    gx = (col < n / 2) ? col : col - n;
    gy = (row < n / 2) ? row : row - n;
  }
  if ((FEAT & 1) && tab)
    return table_phase((double)gx * p[FP_SX], (double)gy * p[FP_SY], p[FP_COEF], p[FP_SGN],
                       (op.flags & PWF_MUL2PI) != 0, tab[col], tab[n + row]);
  return quad_phase(gx, gy, p[FP_SX], p[FP_SY], p[FP_COEF], p[FP_SGN], (op.flags & PWF_MUL2PI) != 0);
}

template <typename T, int FEAT>
__device__ __forceinline__ cx<T> apply_pw(cx<T> v, const PwOp& op, const double* p, int row, int col,
                                          int n, const cx<double>* tab, const double* mask_at) {
  const cx<double> f = pw_factor<FEAT>(op, p, row, col, n, tab, mask_at);
  const cx<double> vd = {(double)v.x, (double)v.y};
  return {(T)__dsub_rn(__dmul_rn(vd.x, f.x), __dmul_rn(vd.y, f.y)),
          (T)__dadd_rn(__dmul_rn(vd.x, f.y), __dmul_rn(vd.y, f.x))};
}

// Fills the tables of one pass program: grid = (ceil(2N / 256), items, tables).  Entry j < N is
// the column factor exp(i sgn A_x(j)), entry N + i the row factor; A = [2 pi] coef fl((g s)^2) as
// an exact double-double product, reduced by the library sincos on the head plus a first-order
// correction for the tail (|tail| <= ulp(head), second order 1e-20).
struct TableJob {
  int block;   // parameter block set
  int kind;    // PWK_QPHASE_C or PWK_QPHASE_N
  int flags;
};
constexpr int kMaxTables = 32;
struct TableArgs {
  const double* blocks;
  cx<double>* tables;
  int batch, n, count;
  TableJob jobs[kMaxTables];
};

static __global__ void phase_table_kernel(TableArgs a) {  // static: the header is compiled into several translation units
  const int item = blockIdx.y, tb = blockIdx.z;
  const TableJob job = a.jobs[tb];
  const double* p = a.blocks + ((size_t)job.block * a.batch + item) * FP_STRIDE;
  if (p[FP_ENABLE] == 0.0) return;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * a.n) return;
  const int idx = j < a.n ? j : j - a.n;
  const int g = job.kind == PWK_QPHASE_C ? idx - a.n / 2 : (idx < a.n / 2 ? idx : idx - a.n);
  const double w = (double)g * (j < a.n ? p[FP_SX] : p[FP_SY]);
  const double X = __dmul_rn(w, w);
  double hi = __dmul_rn(p[FP_COEF], X);
  double lo = fma(p[FP_COEF], X, -hi);
  if (job.flags & PWF_MUL2PI) {
    const double h2 = __dmul_rn(6.283185307179586, hi);
    lo = fma(6.283185307179586, hi, -h2) + 6.283185307179586 * lo;
    hi = h2;
  }
  double sn, cs;
  sincos(hi, &sn, &cs);
  const double c2 = fma(-lo, sn, cs), s2 = fma(lo, cs, sn);
  a.tables[((size_t)tb * a.batch + item) * 2 * a.n + j] = {c2, p[FP_SGN] * s2};
}

// Which elements a thread owns.  A workgroup handles TILES tiles of LINES lines;
// lanes are ordered so that consecutive lanes touch consecutive bytes of a block.
// Row tiles may cover only LINES < BR rows of a block row (N = 4096: two of the four
// rows, 64 contiguous bytes per block); the sibling tile that owns the other rows
// is then placed 8 workgroups away so that both land on the same XCD/L2 under the
// round-robin dispatch (a speed hint only -- any placement is correct).
// SEQ > 1: the tile still spans LINES lines, but only LINES / SEQ of them are processed at a
// time -- each thread owns the same position of SEQ lines and walks them one after the other
// (smaller workgroups, two of them resident per CU, and the loads of line s+1 are in flight
// while line s is transformed).
// COLSIB > 1 (complex64: a block is 64 B, so two column tiles share every 128-byte line): column tiles
// are renumbered like the half-block row tiles, siblings 8 workgroups apart = on one XCD.
template <int N, int E, int LINES, int TILES, int AXIS, int BR, int BC, int SEQ = 1, int COLSIB = 1, int SWZ = 0>
struct TileMap {
  static constexpr int kLinesPerWorkgroup = LINES * TILES;
  static constexpr int kAxis = AXIS;
  static constexpr int TL = N / E;
  static constexpr int PAR = LINES / SEQ;            // lines processed concurrently
  static constexpr int TILE_THREADS = PAR * TL;
  unsigned seq_stride;  // element offset from line s to line s+1 of the same thread
  int line, t;      // which line of the tile / position inside the line
  int lds_line;     // line slot inside the workgroup's LDS
  int row0, col0;   // first row (AXIS 0) or column (AXIS 1) of the tile
  unsigned base;    // element offset (inside the item) of register slot 0
  unsigned stride;  // element offset between slot positions kp and kp+1

  __device__ __forceinline__ TileMap(int wg, int tid_wg, unsigned pitch) {
    int tile = wg * TILES + tid_wg / TILE_THREADS;
    const int tid = tid_wg % TILE_THREADS;
    if constexpr (AXIS == 0) {
      col0 = 0;
      if constexpr (BR == 1) {
        row0 = tile * LINES;
        line = tid / TL; t = tid % TL;
        static_assert(SEQ == 1 || PAR == 1, "canonical rows: sequential lines need PAR == 1");
      } else {
        static_assert(BR % LINES == 0, "row tiles cover whole or 1/2^k block rows");
        static_assert((N / E) % BC == 0, "threads per line must cover whole blocks");
        constexpr int SUB = BR / LINES;
        if constexpr (SUB > 1 && TILES == 1 && (N / LINES) % (8 * SUB) == 0) {
          const int grp = tile / (8 * SUB), in = tile % (8 * SUB);
          tile = (grp * 8 + in % 8) * SUB + in / 8;
        }
        row0 = tile * LINES;
        const int bc = tid % BC, br = (tid / BC) % PAR;
        int q = tid / (BC * PAR);
        // SWZ (round 5): the lanes of a wave alternate between the PAR lines of the tile, so the 16 lanes of one
        // ds_write_b64 group hold 16 / PAR consecutive positions of every line -- and the scatter slots 17 t + r of two lines
        // (whose areas start a multiple of 128 B apart, as the gathers need) fall on the same banks: a 2-way conflict on every
        // exchange write, all of SQ_LDS_BANK_CONFLICT (profiles/r04_sq_counters.txt: 1.458e8 = 32 writes x 4 array cycles x
        // 8 exchanges x 139 300 waves).  Odd lines therefore take their positions 8 further on (t ^ 8: the other half of the
        // 16 slot residues); the set of addresses a wave touches in HBM is the same, only which lane holds which changes.
        // Measured (profiles/r05_fftbench_fused_variants.txt, r05_ab_variants_bench.txt): stand-alone dense two-transform passes
        // -1 ... -2.5 %, the fused launches of the separable programs 0 ... +4 % (bound by the latency of their chain, not by
        // LDS cycles), and in the chain the launches that load / store a quarter of their positions +2 ... +13 % -- the lane
        // pairs of a quad then come from two 128-byte blocks.  Off in the library (frugal_pass.h: PAOS_LDS_SWIZZLE).
        if constexpr (SWZ != 0 && PAR == 2 && BC == 2) q ^= (br & 1) ? 4 : 0;
        line = br; t = q * BC + bc;
      }
      base = (unsigned)layout_index<BR, BC>(row0 + line, t, pitch);
      stride = (unsigned)TL * BR;
      seq_stride = (BR == 1) ? (unsigned)PAR * pitch : (unsigned)PAR * BC;  // next rows of the block
    } else {
      if constexpr (COLSIB > 1 && TILES == 1 && (N / LINES) % (8 * COLSIB) == 0) {
        const int grp = tile / (8 * COLSIB), in = tile % (8 * COLSIB);
        tile = (grp * 8 + in % 8) * COLSIB + in / 8;
      }
      row0 = 0; col0 = tile * LINES;
      if constexpr (BR == 1) {
        line = tid % LINES; t = tid / LINES;
      } else {
        static_assert(BC % LINES == 0, "column tiles span one block column, or one column of it (COLSIB siblings share its lines)");
        static_assert((N / E) % BR == 0, "threads per line must cover whole blocks");
        const int bc = tid % PAR, br = (tid / PAR) % BR;
        int q = tid / (PAR * BR);
        if constexpr (SWZ != 0 && PAR == 2 && BR == 4) q ^= (bc & 1) ? 2 : 0;  // (t ^ 8 for the odd column, as above)
        line = bc; t = q * BR + br;
      }
      base = (unsigned)layout_index<BR, BC>(t, col0 + line, pitch);
      stride = (unsigned)(TL / BR) * pitch;
      seq_stride = (unsigned)PAR;  // next columns of the block
    }
    lds_line = (tid_wg / TILE_THREADS) * PAR + line;
  }
  // (row, col) of the element at position t + kp*TL along the line (sequential line s)
  __device__ __forceinline__ int row(int kp, int s = 0) const { return AXIS == 0 ? row0 + line + s * PAR : t + kp * TL; }
  __device__ __forceinline__ int col(int kp, int s = 0) const { return AXIS == 0 ? t + kp * TL : col0 + line + s * PAR; }
};

// Exchange area of one line.  Lanes of a wave alternate between the lines of a tile, so two
// lines whose areas start a multiple of 256 B apart hit the same banks on every read
// (SQ_LDS_BANK_CONFLICT = half of the LDS cycles, profiles/r01_sq_counters_frugal.txt);
// a 128-B skew puts the second line on the other half of the 64 read banks.
// With LINES lines interleaved among the lanes of a wave (LINES = 4: N = 1024 row tiles), each line's
// lanes read 256 / LINES consecutive bytes per 32-lane group, so the areas must start 256 / LINES bytes
// apart (mod 256) to cover all 64 banks: measured at N = 1024, rows (4 lines, 128-B skew: lines 0/2 and
// 1/3 collided) ran a two-transform pass in 1.31 ms against 1.09 ms for columns (2 lines).
template <typename T, int N, bool SPLIT, int LINES = 2>
constexpr size_t line_lds_bytes() {
  const size_t b = (size_t)lds_line_slots<N>() * (SPLIT ? sizeof(T) : 2 * sizeof(T));
  if (LINES == 2) return (b % 256 == 0) ? b + 128 : b;  // the round-1 rule (kept bit for bit: generic kernels)
  const size_t want = LINES >= 4 ? 64 : 0;
  return b + (want + 256 - b % 256) % 256;
}

template <typename T, int E, int FR, int FEAT, typename Map>
__device__ __forceinline__ void apply_list(cx<T>* v, const PwOp* list, int count, const PassArgs& a,
                                           int item, const Map& m, int n, int sq = 0) {
  for (int o = 0; o < count; ++o) {
    const PwOp op = list[o];
    const double* p = a.blocks + ((size_t)op.block * a.batch + item) * FP_STRIDE;
    if (p[FP_ENABLE] == 0.0) continue;
    const int ti = (FEAT & 1) ? (op.flags >> kTableShift) - 1 : -1;
    const cx<double>* tab = ti >= 0 ? a.tables + ((size_t)ti * a.batch + item) * 2 * n : nullptr;
    const double* mk = (FEAT & 2) ? a.mask + (size_t)item * a.item_stride + m.base + (unsigned)sq * m.seq_stride : nullptr;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      v[k] = apply_pw<T, FEAT>(v[k], op, p, m.row(k, sq), m.col(k, sq), n, tab, (FEAT & 2) ? mk + (unsigned)k * m.stride : nullptr);
      if constexpr (FR != 0) __builtin_amdgcn_sched_barrier(0);  // one phase factor at a time
    }
  }
}

// Forward transform of the thread's line slots, natural slot order in and out; the
// inverse is conj(FFT(conj x)) so the direction is a per-item runtime flag.
template <typename T, int N, int E, bool SPLIT, int FR>
__device__ __forceinline__ void line_fft(cx<T>* v, void* lds, int t, const cx<T>* tw, bool inverse) {
  if (inverse) {
#pragma unroll
    for (int k = 0; k < E; ++k) v[k].y = -v[k].y;
  }
  fft_stages<T, N, E, +1, SPLIT, 1, FR>(v, lds, t, tw);
  unpermute_slots<N, E>(v);
  if (inverse) {
#pragma unroll
    for (int k = 0; k < E; ++k) v[k].y = -v[k].y;
  }
}

// THE pass kernel: load tile -> pre ops -> [FFT] -> mid ops -> [FFT] -> post ops -> store,
// in place.  Which transforms run, their direction and every operator are per-item
// runtime data, so one launch serves a batch whose items disagree (wavelengths whose
// planners skip a step, Monte-Carlo draws).  Examples (wfo.py lines in the header):
//   stw   = rows[pre S | FFT] , cols[FFT | mid S Qc 1/N]
//   wts   = rows[pre S P | FFT] , cols[FFT | mid S 1/N]
//   ptp   = rows[FFT] , cols[FFT | mid H 1/N | IFFT] , rows[IFFT | mid 1/N]
// and, because a 2-D transform may equally run columns first, the last pass of one
// operator and the first pass of the next share an axis and are emitted as ONE pass:
//   ... ptp | lens | ptp ... = rows[IFFT | mid 1/N, lens | FFT].
template <typename T, int N, int E, int LINES, int TILES, int AXIS, int BR, int BC, bool SPLIT,
          int MINW, int SEQ = 1, int FR = 0, int FEAT = 0>
__global__ void __launch_bounds__(TILES* LINES* N / E / SEQ, MINW)
    fused_pass_kernel(PassArgs a) {
  const int item = blockIdx.y;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const TileMap<N, E, LINES, TILES, AXIS, BR, BC, SEQ> m(blockIdx.x, threadIdx.x, a.pitch);
  const double* c1 = a.fft1 >= 0 ? a.blocks + ((size_t)a.fft1 * a.batch + item) * FP_STRIDE : nullptr;
  const double* c2 = a.fft2 >= 0 ? a.blocks + ((size_t)a.fft2 * a.batch + item) * FP_STRIDE : nullptr;
  const bool do1 = c1 && c1[FC_ENABLE] != 0.0, do2 = c2 && c2[FC_ENABLE] != 0.0;
  // an item with nothing enabled in this pass is left untouched (no load, no store)
  bool any = do1 || do2;
  for (int o = 0; o < a.n_pre && !any; ++o) any = a.blocks[((size_t)a.pre[o].block * a.batch + item) * FP_STRIDE] != 0.0;
  for (int o = 0; o < a.n_mid && !any; ++o) any = a.blocks[((size_t)a.mid[o].block * a.batch + item) * FP_STRIDE] != 0.0;
  for (int o = 0; o < a.n_post && !any; ++o) any = a.blocks[((size_t)a.post[o].block * a.batch + item) * FP_STRIDE] != 0.0;
  if (!any) return;

  cx<T>* f = reinterpret_cast<cx<T>*>(a.field) + (size_t)item * a.item_stride;  // wave-uniform
  void* lds = smem + (size_t)m.lds_line * line_lds_bytes<T, N, SPLIT>();
  const cx<T>* tw = reinterpret_cast<const cx<T>*>(a.tw);

  static_assert(SEQ == 1 || SEQ == 2, "one or two sequential lines");
  // Two separately named register files (a 2-D array would be demoted to scratch).
  cx<T> va[E], vb[E];
#pragma unroll
  for (int k = 0; k < E; ++k) va[k] = f[m.base + (unsigned)k * m.stride];
  if constexpr (SEQ == 2) {
#pragma unroll
    for (int k = 0; k < E; ++k) vb[k] = f[m.base + m.seq_stride + (unsigned)k * m.stride];
  }

  auto process = [&](cx<T>* v, int sq) __attribute__((always_inline)) {
    apply_list<T, E, FR, FEAT>(v, a.pre, a.n_pre, a, item, m, N, sq);
    if (do1) line_fft<T, N, E, SPLIT, FR>(v, lds, m.t, tw, c1[FC_INVERSE] != 0.0);
    apply_list<T, E, FR, FEAT>(v, a.mid, a.n_mid, a, item, m, N, sq);
    if (do2) {
      if (do1) PAOS_SYNC();  // the first transform's last LDS reads precede new writes
      line_fft<T, N, E, SPLIT, FR>(v, lds, m.t, tw, c2[FC_INVERSE] != 0.0);
    }
    apply_list<T, E, FR, FEAT>(v, a.post, a.n_post, a, item, m, N, sq);
#pragma unroll
    for (int k = 0; k < E; ++k) f[m.base + (unsigned)sq * m.seq_stride + (unsigned)k * m.stride] = v[k];
  };
  process(va, 0);
  if constexpr (SEQ == 2) {
    PAOS_SYNC();  // line 0 is done with the exchange area
    process(vb, 1);
  }
}

}  // namespace paos
