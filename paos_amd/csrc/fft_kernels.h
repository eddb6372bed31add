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

enum : int {
  PW_SIGN = 1,     // multiply by (-1)^(row+col)
  PW_PHASE = 2,    // multiply by exp(i sgn coef ((gx sx)^2 + (gy sy)^2)), centred coords
  PW_MUL2PI = 4,   // argument gets an extra factor 2 pi (lens form, wfo.py:363-366)
};

// per-item parameter block of an FFT pass (doubles, device memory)
enum : int { FP_ENABLE = 0, FP_SX = 1, FP_SY = 2, FP_COEF = 3, FP_SGN = 4, FP_STRIDE = 5 };

struct FftPassArgs {
  void* field;            // batch of N*N complex<T>
  const void* tw;         // exp(-2 pi i m / N), m < N, complex<T>
  const double* params;   // [item][FP_STRIDE] or nullptr (all items enabled, no phase)
  int pre_mode;           // PW_* flags applied on load
  int post_mode;          // PW_* flags applied on store
  double scale;           // applied on store (exact power of two)
  unsigned pitch;         // elements between block rows of the layout (>= N * BR)
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

template <typename T>
__device__ __forceinline__ cx<T> pointwise(cx<T> v, int mode, int row, int col, int n,
                                           const double* p) {
  if (mode & PW_SIGN) {
    if ((row + col) & 1) { v.x = -v.x; v.y = -v.y; }
  }
  if (mode & PW_PHASE) {
    const cx<double> f = quad_phase(col - n / 2, row - n / 2, p[FP_SX], p[FP_SY], p[FP_COEF],
                                    p[FP_SGN], (mode & PW_MUL2PI) != 0);
    const cx<double> vd = {(double)v.x, (double)v.y};
    const cx<double> r = {__dsub_rn(__dmul_rn(vd.x, f.x), __dmul_rn(vd.y, f.y)),
                          __dadd_rn(__dmul_rn(vd.x, f.y), __dmul_rn(vd.y, f.x))};
    v.x = (T)r.x;
    v.y = (T)r.y;
  }
  return v;
}

// Which elements a thread owns.  A workgroup handles TILES tiles of LINES lines;
// lanes are ordered so that consecutive lanes touch consecutive bytes of a block.
// Row tiles may cover only LINES < BR rows of a block row (N = 4096: two of the four
// rows, 64 contiguous bytes per block); the sibling tile that owns the other rows
// is then placed 8 workgroups away so that both land on the same XCD/L2 under the
// round-robin dispatch (a speed hint only -- any placement is correct).
template <int N, int E, int LINES, int TILES, int AXIS, int BR, int BC>
struct TileMap {
  static constexpr int TL = N / E;
  static constexpr int TILE_THREADS = LINES * TL;
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
      } else {
        static_assert(BR % LINES == 0, "row tiles cover whole or 1/2^k block rows");
        static_assert((N / E) % BC == 0, "threads per line must cover whole blocks");
        constexpr int SUB = BR / LINES;
        if constexpr (SUB > 1 && TILES == 1 && (N / LINES) % (8 * SUB) == 0) {
          const int grp = tile / (8 * SUB), in = tile % (8 * SUB);
          tile = (grp * 8 + in % 8) * SUB + in / 8;
        }
        row0 = tile * LINES;
        const int bc = tid % BC, br = (tid / BC) % LINES, q = tid / (BC * LINES);
        line = br; t = q * BC + bc;
      }
      base = (unsigned)layout_index<BR, BC>(row0 + line, t, pitch);
      stride = (unsigned)TL * BR;
    } else {
      row0 = 0; col0 = tile * LINES;
      if constexpr (BR == 1) {
        line = tid % LINES; t = tid / LINES;
      } else {
        static_assert(LINES == BC, "column tiles span exactly one block column");
        static_assert((N / E) % BR == 0, "threads per line must cover whole blocks");
        const int bc = tid % BC, br = (tid / BC) % BR, q = tid / (BC * BR);
        line = bc; t = q * BR + br;
      }
      base = (unsigned)layout_index<BR, BC>(t, col0 + line, pitch);
      stride = (unsigned)(TL / BR) * pitch;
    }
    lds_line = (tid_wg / TILE_THREADS) * LINES + line;
  }
  // (row, col) of the element at position t + kp*TL along the line
  __device__ __forceinline__ int row(int kp) const { return AXIS == 0 ? row0 + line : t + kp * TL; }
  __device__ __forceinline__ int col(int kp) const { return AXIS == 0 ? t + kp * TL : col0 + line; }
};

template <typename T, int N, bool SPLIT>
constexpr size_t line_lds_bytes() {
  return (size_t)lds_line_slots<N>() * (SPLIT ? sizeof(T) : 2 * sizeof(T));
}

// One FFT per line.  DIR = +1 forward, -1 inverse (unnormalised; ``scale`` carries 1/N).
template <typename T, int N, int E, int LINES, int TILES, int AXIS, int BR, int BC, bool SPLIT,
          int DIR, int MINW>
__global__ void __launch_bounds__(TILES* LINES* N / E, MINW)
    fft_pass_kernel(FftPassArgs a) {
  const int item = blockIdx.y;
  const double* p = a.params ? a.params + (size_t)item * FP_STRIDE : nullptr;
  if (p && p[FP_ENABLE] == 0.0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const TileMap<N, E, LINES, TILES, AXIS, BR, BC> m(blockIdx.x, threadIdx.x, a.pitch);
  cx<T>* f = reinterpret_cast<cx<T>*>(a.field) + (size_t)item * a.item_stride;  // wave-uniform
  void* lds = smem + (size_t)m.lds_line * line_lds_bytes<T, N, SPLIT>();

  cx<T> v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = f[m.base + (unsigned)k * m.stride];
  if (a.pre_mode) {
#pragma unroll
    for (int k = 0; k < E; ++k) v[k] = pointwise(v[k], a.pre_mode, m.row(k), m.col(k), N, p);
  }

  fft_stages<T, N, E, DIR, SPLIT>(v, lds, m.t, reinterpret_cast<const cx<T>*>(a.tw));

  const T sc = (T)a.scale;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int kp = outslot<N, E>(k);
    cx<T> o = v[k];
    if (a.post_mode) o = pointwise(o, a.post_mode, m.row(kp), m.col(kp), N, p);
    o.x *= sc; o.y *= sc;
    f[m.base + (unsigned)kp * m.stride] = o;
  }
}

// Fused middle pass of ptp: forward FFT along the line, multiply by
// H = exp(-i coef (fx^2 + fy^2)) (natural-order signed frequencies,
// wfo.py:464-468), inverse FFT along the line.
template <typename T, int N, int E, int LINES, int TILES, int AXIS, int BR, int BC, bool SPLIT,
          int MINW>
__global__ void __launch_bounds__(TILES* LINES* N / E, MINW)
    fft_ptp_mid_kernel(FftPassArgs a) {
  const int item = blockIdx.y;
  const double* p = a.params + (size_t)item * FP_STRIDE;
  if (p[FP_ENABLE] == 0.0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const TileMap<N, E, LINES, TILES, AXIS, BR, BC> m(blockIdx.x, threadIdx.x, a.pitch);
  cx<T>* f = reinterpret_cast<cx<T>*>(a.field) + (size_t)item * a.item_stride;
  void* lds = smem + (size_t)m.lds_line * line_lds_bytes<T, N, SPLIT>();
  const cx<T>* tw = reinterpret_cast<const cx<T>*>(a.tw);

  cx<T> v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = f[m.base + (unsigned)k * m.stride];

  fft_stages<T, N, E, +1, SPLIT>(v, lds, m.t, tw);
  unpermute_slots<N, E>(v);

  const double sx = p[FP_SX], sy = p[FP_SY], coef = p[FP_COEF];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int r = m.row(k), c = m.col(k);
    const int gy = (r < N / 2) ? r : r - N;
    const int gx = (c < N / 2) ? c : c - N;
    const cx<double> h = quad_phase(gx, gy, sx, sy, coef, -1.0, false);
    const cx<double> vd = {(double)v[k].x, (double)v[k].y};
    v[k].x = (T)__dsub_rn(__dmul_rn(vd.x, h.x), __dmul_rn(vd.y, h.y));
    v[k].y = (T)__dadd_rn(__dmul_rn(vd.x, h.y), __dmul_rn(vd.y, h.x));
  }

  __syncthreads();  // the forward transform's last LDS reads precede the next writes
  fft_stages<T, N, E, -1, SPLIT>(v, lds, m.t, tw);

  const T sc = (T)a.scale;
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int kp = outslot<N, E>(k);
    f[m.base + (unsigned)kp * m.stride] = {v[k].x * sc, v[k].y * sc};
  }
}

}  // namespace paos
