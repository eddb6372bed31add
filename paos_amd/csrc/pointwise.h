// pointwise.h -- element-wise and reduction kernels of the wavefront operators.
//
// Every kernel walks the field in MEMORY order (16 B per lane, unit stride) and
// recovers (row, col) from the blocked layout, so the access pattern is a plain
// stream regardless of the layout chosen for the FFT passes.
//
//   aperture  -> paos/classes/wfo.py:203-278 (mask values: photutils semantics,
//                restated in oracle/aperture_np.py -- same operations, same order)
//   make_stop -> wfo.py:195-201
//   lens      -> wfo.py:359-366 (the scalar pilot-beam part stays on the host)
//   zernikes  -> wfo.py:620-652 + paos/classes/zernike.py:85-109,245-247
//   amplitude / phase / intensity -> wfo.py:166-172, paos/core/plot.py:125-130
#pragma once
#include "frugal_pass.h"

namespace paos {

// memory index (inside one item) -> (row, col); false for pitch-padding slots
template <int BR, int BC>
__device__ __forceinline__ bool layout_unmap(size_t m, int n, unsigned pitch, int& row, int& col) {
  const unsigned brow = (unsigned)(m / pitch), rem = (unsigned)(m % pitch);
  if (rem >= (unsigned)n * BR) return false;
  const unsigned in = rem % (BR * BC);
  row = (int)(brow * BR + in / BC);
  col = (int)((rem / (BR * BC)) * BC + in % BC);
  return true;
}

constexpr int kPwThreads = 256;

template <typename T>
__global__ void fill_kernel(cx<T>* f, size_t total, T re, T im) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < total; i += (size_t)gridDim.x * blockDim.x) f[i] = {re, im};
}

// ---- in-place copy yardstick (measurement aid, bench.py): read every element of the batch and
// write it back unchanged, 16 B per lane, unit stride -- what the chip's HBM path reaches on an
// in-place read-modify-write of this buffer, without any transform
template <typename T>
__global__ void rmw_copy_kernel(cx<T>* f, size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < total; i += (size_t)gridDim.x * blockDim.x) {
    cx<T> v = f[i];
    asm volatile("" : "+v"(v.x), "+v"(v.y));  // keep the store: the value is unchanged, the compiler must not know
    f[i] = v;
  }
}

// ---- host <-> device layout conversion (staging buffer is row-major complex128)
template <typename T, int BR, int BC>
__global__ void import_kernel(cx<T>* f, const cx<double>* staged, int n, unsigned pitch) {
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const cx<double> v = staged[(size_t)r * n + c];
    f[m] = {(T)v.x, (T)v.y};
  }
}

// what: 0 complex field, 1 amplitude |u| (hypot), 2 phase atan2(im, re), 3 intensity |u|^2
template <typename T, int BR, int BC>
__global__ void export_kernel(const cx<T>* f, double* out, int n, unsigned pitch, int what) {
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const size_t o = (size_t)r * n + c;
    const double x = (double)f[m].x, y = (double)f[m].y;
    if (what == 0) {
      out[2 * o] = x;
      out[2 * o + 1] = y;
    } else if (what == 1) {
      out[o] = hypot(x, y);
    } else if (what == 2) {
      out[o] = atan2(y, x);
    } else {
      out[o] = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
    }
  }
}

// |u|^2 of every batch item into a device buffer, in the FIELD'S OWN blocked layout and item stride (doubles): the PSF
// of plot.py:125-130 kept in HBM (24 B/px: 16 read + 8 written).  psf_unblock_kernel turns one item row-major.
template <typename T, int BR, int BC>
__global__ void intensity_kernel(const cx<T>* field, double* out, int n, unsigned pitch, unsigned item_stride) {
  const int item = blockIdx.y;
  const cx<T>* f = field + (size_t)item * item_stride;
  double* o = out + (size_t)item * item_stride;
  const size_t total = item_stride;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    if ((unsigned)(m % pitch) >= (unsigned)n * BR) continue;  // pitch padding
    const double x = (double)f[m].x, y = (double)f[m].y;
    o[m] = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
  }
}

template <int BR, int BC>
__global__ void psf_unblock_kernel(const double* psf_item, double* out, int n, unsigned pitch) {
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    out[(size_t)r * n + c] = psf_item[m];
  }
}

// ---- tabulated phase screen: the field part shared by WFO.grid_sag and WFO.psd ---------------
// u *= exp(2 pi i wfe / wl) (wfo.py:869-871, 945-949) for a host-supplied map (row-major doubles in
// the staging buffer).  NumPy forms the argument as fl(fl(2 pi w) / wl) and multiplies the complex
// numbers without FMA contraction; both are kept.
template <typename T, int BR, int BC>
__global__ void phase_map_kernel(cx<T>* f, const double* staged, int n, unsigned pitch, double wl) {
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const double w = staged[(size_t)r * n + c];
    const double arg = __ddiv_rn(__dmul_rn(6.283185307179586, w), wl);
    double sn, cs;
    sincos(arg, &sn, &cs);
    const double x = (double)f[m].x, y = (double)f[m].y;
    f[m] = {(T)__dsub_rn(__dmul_rn(x, cs), __dmul_rn(y, sn)), (T)__dadd_rn(__dmul_rn(x, sn), __dmul_rn(y, cs))};
  }
}

// The same for several items that share ONE map (round 5: a measured surface map is the same for every wavelength of a sweep
// and every draw of a Monte-Carlo study -- uploaded once, applied to each listed item with its own wavelength):
// blockIdx.y walks the list.
template <typename T, int BR, int BC>
__global__ void phase_map_items_kernel(cx<T>* field, unsigned item_stride, const double* map, int n, unsigned pitch,
                                       const double* items, const double* wls) {
  cx<T>* f = field + (size_t)(int)items[blockIdx.y] * item_stride;
  const double wl = wls[blockIdx.y];
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const double w = map[(size_t)r * n + c];
    const double arg = __ddiv_rn(__dmul_rn(6.283185307179586, w), wl);  // (phase_map_kernel's own expressions)
    double sn, cs;
    sincos(arg, &sn, &cs);
    const double x = (double)f[m].x, y = (double)f[m].y;
    f[m] = {(T)__dsub_rn(__dmul_rn(x, cs), __dmul_rn(y, sn)), (T)__dadd_rn(__dmul_rn(x, sn), __dmul_rn(y, cs))};
  }
}

// ---- PSD screen built on the device (round 5: wfo.py:908-943 + psd.py:100-160; the host restatement is
// paos_amd/phase_maps.py: psd_map) ----------------------------------------------------------------------------
// The white noise comes from the host (NumPy's generator is the reference's); fft2 -> filter -> ifft2 run on the library's
// own passes over one scratch item in the field's layout.
struct PsdParams {
  double fx, fy;         // 1 / (n dx), 1 / (n dy): np.fft.fftfreq's step
  double A, B, C, fknee;
  double fmin, fmax;
  double cell;           // frequency-bin area (rho[0,2] - rho[0,1]) * (rho[2,0] - rho[1,0])
  double gain;           // sqrt(n0 * n1)
  double SR, unit;
};

template <int BR, int BC>
__global__ void psd_load_kernel(cx<double>* f, const double* noise, int n, unsigned pitch) {
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) { f[m] = {0.0, 0.0}; continue; }
    f[m] = {noise[(size_t)r * n + c], 0.0};
  }
}

// spectrum *= sqrt(A / (B + (rho / fknee)^C) / (2 pi rho) * cell) * sqrt(n0 n1), zero outside [fmin, fmax] -- psd.py:113-142
// in the reference's order of operations (the pow is the device library's: a few ulp from glibc's)
template <int BR, int BC>
__global__ void psd_filter_kernel(cx<double>* f, int n, unsigned pitch, PsdParams p) {
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const double gx = __dmul_rn((double)(c < n / 2 ? c : c - n), p.fx), gy = __dmul_rn((double)(r < n / 2 ? r : r - n), p.fy);
    double rho = sqrt(__dadd_rn(__dmul_rn(gx, gx), __dmul_rn(gy, gy)));
    if (rho == 0.0) rho = 1e-100;
    if (rho < p.fmin || rho > p.fmax) { f[m] = {0.0, 0.0}; continue; }
    double d = __ddiv_rn(p.A, __dadd_rn(p.B, pow(__ddiv_rn(rho, p.fknee), p.C)));
    d = __ddiv_rn(d, __dmul_rn(6.283185307179586, rho));
    d = __dmul_rn(d, p.cell);
    const double g = __dmul_rn(sqrt(d), p.gain);
    f[m] = {__dmul_rn(f[m].x, g), __dmul_rn(f[m].y, g)};
  }
}

// screen = Re(ifft2) + SR * roughness; *= 2; *= unit -- written row-major; `bad` counts non-finite values
template <int BR, int BC>
__global__ void psd_finish_kernel(double* map, const cx<double>* f, const double* rough, int n, unsigned pitch, PsdParams p, int* bad) {
  const size_t total = (size_t)pitch * (n / BR);
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    double v = f[m].x;
    if (rough) v = __dadd_rn(v, __dmul_rn(p.SR, rough[(size_t)r * n + c]));
    v = __dmul_rn(__dmul_rn(v, 2.0), p.unit);
    if (!isfinite(v)) atomicAdd(bad, 1);
    map[(size_t)r * n + c] = v;
  }
}

// ---- stand-alone pointwise pass (op list without a transform) --------------------
// Used when a lens (wfo.py:359-366) is not adjacent to an FFT pass it could ride on.
template <typename T, int BR, int BC>
__global__ void pointwise_kernel(PassArgs a, int n) {
  const int item = blockIdx.y;
  bool any = false;
  for (int o = 0; o < a.n_pre && !any; ++o) any = a.blocks[((size_t)a.pre[o].block * a.batch + item) * FP_STRIDE] != 0.0;
  if (!any) return;
  cx<T>* f = reinterpret_cast<cx<T>*>(a.field) + (size_t)item * a.item_stride;
  const size_t total = a.item_stride;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, a.pitch, r, c)) continue;
    cx<T> v = f[m];
    for (int o = 0; o < a.n_pre; ++o) {
      const double* p = a.blocks + ((size_t)a.pre[o].block * a.batch + item) * FP_STRIDE;
      if (p[FP_ENABLE] != 0.0) {
        const int ti = (a.pre[o].flags >> kTableShift) - 1;
        v = apply_pw<T, 3>(v, a.pre[o], p, r, c, n,
                     ti >= 0 ? a.tables + ((size_t)ti * a.batch + item) * 2 * n : nullptr,
                     a.mask + (size_t)item * a.item_stride + m);
      }
    }
    f[m] = v;
  }
}

// ---- make_stop: sum |u|^2 (two deterministic stages), then scale ---------------
__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;  // valid in thread 0
}

// ``rows`` (optional, [batch][2]): rows of the item outside [lo, hi) are exact zeros (or hold stale data standing for
// zeros) and are not read.  Block rows are contiguous in memory, so that is a window of the walk; every thread keeps
// the elements it would have had, in the same order, and a zero adds nothing: the sum is bit-identical.
template <typename T, int BR, int BC>
__global__ void norm2_partial_kernel(const cx<T>* field, double* partial, int n, unsigned pitch,
                                     unsigned item_stride, const double* enable, int enable_stride,
                                     const double* rows = nullptr, const double* cols = nullptr) {
  const int item = blockIdx.y;
  if (enable && enable[(size_t)item * enable_stride] == 0.0) return;
  __shared__ double sh[kPwThreads / 64];
  const cx<T>* f = field + (size_t)item * item_stride;
  size_t total = item_stride, first = 0;
  if (rows) {
    first = (size_t)((int)rows[2 * item] / BR) * pitch;
    total = (size_t)(((int)rows[2 * item + 1] + BR - 1) / BR) * pitch;
    if (total > item_stride) total = item_stride;
  }
  double acc = 0.0;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m < first) m += (first - m + step - 1) / step * step;
  // (round 5) ``cols``: columns outside [lo, hi) -- rounded outward to blocks -- are zero or stand for zeros: not read
  unsigned in_lo = 0, in_hi = (unsigned)n * BR;  // element offsets inside a block row
  if (cols) {
    in_lo = (unsigned)((int)cols[2 * item] / BC) * (BR * BC);
    const unsigned h = (unsigned)(((int)cols[2 * item + 1] + BC - 1) / BC) * (BR * BC);
    in_hi = h < in_hi ? h : in_hi;
  }
  for (; m < total; m += step) {
    const unsigned off = (unsigned)(m % pitch);
    if (off >= in_hi || off < in_lo) continue;  // pitch padding / outside the column window
    const double x = (double)f[m].x, y = (double)f[m].y;
    acc += __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
  }
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[(size_t)item * gridDim.x + blockIdx.x] = s;
}

// intensity_kernel that also leaves the partial sums of sum |u|^2 norm2_partial_kernel would (same grid,
// same walk, same order of additions: the power of a saved last surface comes out bit-identical without
// reading the field a second time)
template <typename T, int BR, int BC>
__global__ void intensity_power_kernel(const cx<T>* field, double* out, double* partial, int n, unsigned pitch,
                                       unsigned item_stride) {
  const int item = blockIdx.y;
  __shared__ double sh[kPwThreads / 64];
  const cx<T>* f = field + (size_t)item * item_stride;
  double* o = out + (size_t)item * item_stride;  // blocked like the field (intensity_kernel)
  const size_t total = item_stride;
  double acc = 0.0;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    if ((unsigned)(m % pitch) >= (unsigned)n * BR) continue;  // pitch padding
    const double x = (double)f[m].x, y = (double)f[m].y;
    const double v = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
    o[m] = v;
    acc += v;
  }
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[(size_t)item * gridDim.x + blockIdx.x] = s;
}

// ``source`` (optional): item whose partial sums make up this item's result (identical inputs, summed once)
static __global__ void norm2_final_kernel(const double* partial, double* norm2, int nparts,
                                   const double* enable, int enable_stride, const double* source = nullptr) {
  const int item = blockIdx.x;
  if (enable && enable[(size_t)item * enable_stride] == 0.0) return;
  __shared__ double sh[kPwThreads / 64];
  double acc = 0.0;
  const int src = source ? (int)source[item] : item;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += partial[(size_t)src * nparts + i];
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) norm2[item] = s;
}

// u *= 1/sqrt(norm2[item])  (NumPy's complex /= real multiplies by the reciprocal)
template <typename T>
__global__ void stop_scale_kernel(cx<T>* field, const double* norm2, unsigned item_stride,
                                  const double* enable, int enable_stride) {
  const int item = blockIdx.y;
  if (enable && enable[(size_t)item * enable_stride] == 0.0) return;
  const double s = 1.0 / sqrt(norm2[item]);
  cx<T>* f = field + (size_t)item * item_stride;
  const size_t total = item_stride;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x)
    f[m] = {(T)__dmul_rn((double)f[m].x, s), (T)__dmul_rn((double)f[m].y, s)};
}

// ---- PSF metrics for Monte-Carlo studies ------------------------------------
// One pass over |u|^2 per item: total power, first moments (centroid), peak, and the power
// inside circles of given radii (pixel units) about a given centre -- the ingredients of the
// encircled-energy radii the reference's Monte-Carlo workflow derives from each PSF
// (docs/source/user/montecarlo/index.rst:26-66) -- so thousands of draws never leave the GPU.
constexpr int kMaxRadii = 16;
struct MetricArgs {
  const void* field;
  double* partial;       // [item][block][4 + nr]
  int n, nr;
  unsigned pitch, item_stride;
  double cx, cy;         // circle centre in pixel coordinates (column, row)
  double r2[kMaxRadii];  // squared radii
};

template <typename T, int BR, int BC>
static __global__ void psf_metrics_kernel(MetricArgs a) {
  const int item = blockIdx.y;
  __shared__ double sh[kPwThreads / 64];
  const cx<T>* f = reinterpret_cast<const cx<T>*>(a.field) + (size_t)item * a.item_stride;
  double acc[4 + kMaxRadii];
#pragma unroll
  for (int k = 0; k < 4 + kMaxRadii; ++k) acc[k] = 0.0;
  const size_t total = a.item_stride;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, a.n, a.pitch, r, c)) continue;
    const double x = (double)f[m].x, y = (double)f[m].y;
    const double I = x * x + y * y;
    acc[0] += I;
    acc[1] += I * (double)c;
    acc[2] += I * (double)r;
    acc[3] = fmax(acc[3], I);
    const double dx = (double)c - a.cx, dy = (double)r - a.cy;
    const double d2 = dx * dx + dy * dy;
#pragma unroll
    for (int k = 0; k < kMaxRadii; ++k)
      if (k < a.nr && d2 <= a.r2[k]) acc[4 + k] += I;
  }
  double* out = a.partial + ((size_t)item * gridDim.x + blockIdx.x) * (4 + a.nr);
  for (int k = 0; k < 4 + a.nr; ++k) {
    double v = acc[k];
    if (k == 3) {  // max-reduce
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
      if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) { double s = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s = fmax(s, sh[w]); out[k] = s; }
    } else {
      const double s = block_sum(v, sh);
      if (threadIdx.x == 0) out[k] = s;
    }
    __syncthreads();
  }
}

static __global__ void psf_metrics_final_kernel(const double* partial, double* out, int nblocks, int nvals) {
  const int item = blockIdx.x, k = threadIdx.x;
  if (k >= nvals) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) {
    const double v = partial[((size_t)item * nblocks + b) * nvals + k];
    s = (k == 3) ? fmax(s, v) : s + v;
  }
  out[(size_t)item * nvals + k] = s;
}

// ---- apertures ------------------------------------------------------------------
// per-item block: [enable, xc, yc, a|w, b|h, theta, obscuration, subpixels]
enum : int { AP_ENABLE = 0, AP_XC, AP_YC, AP_A, AP_B, AP_THETA, AP_OBSC, AP_SUBPIX, AP_STRIDE, AP_SHAPE = 8 };

struct EdgeAcc {
  double area;
  bool touched;
};

// signed area of (triangle O, p, p+d) INTERSECT unit disk -- oracle/aperture_np.py:_edge_term
__device__ __forceinline__ void edge_term(double px, double py, double dx, double dy, EdgeAcc& acc) {
  const double cr = __dsub_rn(__dmul_rn(px, dy), __dmul_rn(py, dx));
  const double pp = __dadd_rn(__dmul_rn(px, px), __dmul_rn(py, py));
  const double pd = __dadd_rn(__dmul_rn(px, dx), __dmul_rn(py, dy));
  const double dd = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
  const double cc = __dsub_rn(pp, 1.0);
  const double disc = __dsub_rn(__dmul_rn(pd, pd), __dmul_rn(dd, cc));
  const bool has = disc > 0.0;
  const double sq = sqrt(has ? disc : 0.0);
  const double qq = -(__dadd_rn(pd, (pd >= 0.0) ? sq : -sq));
  const double ta = qq / dd;
  const double tb = (qq != 0.0) ? cc / qq : ta;
  const double t1 = fmin(ta, tb), t2 = fmax(ta, tb);
  const double t1c = fmin(fmax(t1, 0.0), 1.0), t2c = fmin(fmax(t2, 0.0), 1.0);
  const bool part = has && (t1c < t2c);
  double val;
  if (part) {
    const double a_in = atan2(__dmul_rn(t1c, cr), __dadd_rn(pp, __dmul_rn(t1c, pd)));
    const double a_out =
        atan2(__dmul_rn(__dsub_rn(1.0, t2c), cr),
              __dadd_rn(__dadd_rn(pp, __dmul_rn(__dadd_rn(1.0, t2c), pd)), __dmul_rn(t2c, dd)));
    val = __dmul_rn(0.5, __dadd_rn(__dadd_rn(a_in, __dmul_rn(__dsub_rn(t2c, t1c), cr)), a_out));
  } else {
    val = __dmul_rn(0.5, atan2(cr, __dadd_rn(pp, pd)));
  }
  acc.area = val;
  acc.touched = part;
}

__device__ __forceinline__ double ellipse_pixel(int kx, int ky, double xc, double yc, double a,
                                                double b, double ct, double st, double full_disk) {
  const double x0 = __dsub_rn((double)kx - 0.5, xc), x1 = __dsub_rn((double)kx + 0.5, xc);
  const double y0 = __dsub_rn((double)ky - 0.5, yc), y1 = __dsub_rn((double)ky + 0.5, yc);
  auto ux = [&](double x, double y) { return __dadd_rn(__dmul_rn(x, ct), __dmul_rn(y, st)) / a; };
  auto uy = [&](double x, double y) { return __dsub_rn(__dmul_rn(y, ct), __dmul_rn(x, st)) / b; };
  const double c0x = ux(x0, y0), c0y = uy(x0, y0);
  const double c1x = ux(x1, y0), c1y = uy(x1, y0);
  const double c2x = ux(x1, y1), c2y = uy(x1, y1);
  const double c3x = ux(x0, y1), c3y = uy(x0, y1);
  auto inside = [](double x, double y) { return __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)) <= 1.0; };
  const bool i0 = inside(c0x, c0y), i1 = inside(c1x, c1y), i2 = inside(c2x, c2y), i3 = inside(c3x, c3y);
  if (i0 && i1 && i2 && i3) return 1.0;
  EdgeAcc e0, e1, e2, e3;
  edge_term(c0x, c0y, __dsub_rn(c1x, c0x), __dsub_rn(c1y, c0y), e0);
  edge_term(c1x, c1y, __dsub_rn(c2x, c1x), __dsub_rn(c2y, c1y), e1);
  edge_term(c2x, c2y, __dsub_rn(c3x, c2x), __dsub_rn(c3y, c2y), e2);
  edge_term(c3x, c3y, __dsub_rn(c0x, c3x), __dsub_rn(c0y, c3y), e3);
  const bool touched = i0 || i1 || i2 || i3 || e0.touched || e1.touched || e2.touched || e3.touched;
  if (!touched) {
    const bool holds = (x0 <= 0.0) && (x1 >= 0.0) && (y0 <= 0.0) && (y1 >= 0.0);
    return holds ? full_disk : 0.0;
  }
  double area = __dmul_rn(__dadd_rn(__dadd_rn(e0.area, e1.area), __dadd_rn(e2.area, e3.area)),
                          __dmul_rn(a, b));
  return fmin(fmax(area, 0.0), 1.0);
}

// #sub-samples (1-D) of pixel k with |x| < half; positions by repeated addition
__device__ __forceinline__ int subpixel_count_1d(int k, double centre, double half, int subpixels) {
  const double step = 1.0 / (double)subpixels;
  double x = __dsub_rn(__dsub_rn((double)k - 0.5, centre), __dmul_rn(0.5, step));
  int cnt = 0;
  for (int s = 0; s < subpixels; ++s) {
    x = __dadd_rn(x, step);
    cnt += (fabs(x) < half) ? 1 : 0;
  }
  return cnt;
}

__device__ __forceinline__ double rect_pixel(int kx, int ky, double xc, double yc, double hw,
                                             double hh, double ct, double st, int subpixels) {
  const double step = 1.0 / (double)subpixels;
  int cnt = 0;
  double x = __dsub_rn(__dsub_rn((double)kx - 0.5, xc), __dmul_rn(0.5, step));
  for (int i = 0; i < subpixels; ++i) {
    x = __dadd_rn(x, step);
    double y = __dsub_rn(__dsub_rn((double)ky - 0.5, yc), __dmul_rn(0.5, step));
    for (int j = 0; j < subpixels; ++j) {
      y = __dadd_rn(y, step);
      const double xt = __dadd_rn(__dmul_rn(y, st), __dmul_rn(x, ct));
      const double yt = __dsub_rn(__dmul_rn(y, ct), __dmul_rn(x, st));
      cnt += (fabs(xt) < hw && fabs(yt) < hh) ? 1 : 0;
    }
  }
  return (double)cnt / (double)(subpixels * subpixels);
}

struct ApertureBox {
  int ixmin, ixmax, iymin, iymax;  // photutils bounding box, max exclusive
};

__device__ __forceinline__ ApertureBox make_box(double xc, double yc, double xe, double ye) {
  ApertureBox b;
  b.ixmin = (int)floor(__dadd_rn(__dsub_rn(xc, xe), 0.5));
  b.ixmax = (int)ceil(__dadd_rn(__dadd_rn(xc, xe), 0.5));
  b.iymin = (int)floor(__dadd_rn(__dsub_rn(yc, ye), 0.5));
  b.iymax = (int)ceil(__dadd_rn(__dadd_rn(yc, ye), 0.5));
  return b;
}

// SHAPE 0: exact ellipse, 1: sub-pixel rectangle.  ``mask_out`` non-null: render instead of
// apply -- row-major mask of item 0 (weights_out == 0, for the aperture object's to_image) or,
// with weights_out != 0, the multiplicative weight (mask, or 1 - mask for an obscuration) of
// EVERY item in the field's own layout, for a PWK_MASK operator riding on an FFT pass.
// Per-item aperture geometry and the per-pixel weight, shared by every kernel that evaluates a mask.
template <int SHAPE>
struct ApertureEval {
  double xc, yc, a, b, theta, ct, st, hw, hh, full_disk;
  bool obsc;
  int subpix;
  ApertureBox box;

  // p = [enable, xc, yc, a|w, b|h, ...]; p2 = [theta, obscuration, subpixels, shape]
  __device__ __forceinline__ void init(const double* p, const double* p2) {
    xc = p[AP_XC]; yc = p[AP_YC]; a = p[AP_A]; b = p[AP_B]; theta = p2[0];
    obsc = p2[1] != 0.0;
    subpix = (int)p2[2];
    ct = cos(theta); st = sin(theta);
    double xe, ye;
    hw = hh = full_disk = 0.0;
    if (SHAPE == 0) {
      xe = sqrt(__dadd_rn(__dmul_rn(__dmul_rn(a, ct), __dmul_rn(a, ct)),
                          __dmul_rn(__dmul_rn(b, st), __dmul_rn(b, st))));
      ye = sqrt(__dadd_rn(__dmul_rn(__dmul_rn(a, st), __dmul_rn(a, st)),
                          __dmul_rn(__dmul_rn(b, ct), __dmul_rn(b, ct))));
      full_disk = fmin(__dmul_rn(__dmul_rn(3.141592653589793, a), b), 1.0);
    } else {
      hw = a / 2.0;
      hh = b / 2.0;
      xe = fmax(fabs(__dsub_rn(__dmul_rn(hw, ct), __dmul_rn(hh, st))),
                fabs(__dadd_rn(__dmul_rn(hw, ct), __dmul_rn(hh, st))));
      ye = fmax(fabs(__dadd_rn(__dmul_rn(hw, st), __dmul_rn(hh, ct))),
                fabs(__dsub_rn(__dmul_rn(hw, st), __dmul_rn(hh, ct))));
    }
    box = make_box(xc, yc, xe, ye);
  }
  // the photutils mask value of pixel (r, c)
  __device__ __forceinline__ double mask(int r, int c) const {
    if (!(c >= box.ixmin && c < box.ixmax && r >= box.iymin && r < box.iymax)) return 0.0;
    if (SHAPE == 0) return ellipse_pixel(c, r, xc, yc, a, b, ct, st, full_disk);
    if (theta == 0.0) {
      const int cxn = subpixel_count_1d(c, xc, hw, subpix);
      const int cyn = subpixel_count_1d(r, yc, hh, subpix);
      return (double)(cxn * cyn) / (double)(subpix * subpix);
    }
    return rect_pixel(c, r, xc, yc, hw, hh, ct, st, subpix);
  }
  // the multiplicative weight: mask, or 1 - mask for an obscuration (wfo.py:273-276)
  __device__ __forceinline__ double weight(double m) const { return obsc ? __dsub_rn(1.0, m) : m; }
};

// SHAPE 0: exact ellipse, 1: sub-pixel rectangle.  ``mask_out`` non-null: render instead of
// apply -- row-major mask of item 0 (weights_out == 0, for the aperture object's to_image) or,
// with weights_out != 0, the multiplicative weight (mask, or 1 - mask for an obscuration) of
// EVERY item in the field's own layout, for a PWK_MASK operator riding on an FFT pass.
template <typename T, int BR, int BC, int SHAPE>
__global__ void aperture_kernel(cx<T>* field, const double* params, const double* params2,
                                int param_stride, int n, unsigned pitch, unsigned item_stride,
                                double* mask_out, int weights_out) {
  const int item = blockIdx.y;
  const double* p = params + (size_t)item * param_stride;
  // second half of the record [theta, obscuration, subpixels, shape]: contiguous for the
  // 8-double blocks of paos_aperture, in the next block set for a pass-program operator
  const double* p2 = params2 ? params2 + (size_t)item * param_stride : p + AP_THETA;
  if (p[AP_ENABLE] == 0.0) return;
  if (weights_out == 1 && SHAPE != (int)p2[3]) return;  // 2: paos_pupil_aperture, shape chosen by the host
  ApertureEval<SHAPE> ap;
  ap.init(p, p2);
  cx<T>* f = field ? field + (size_t)item * item_stride : nullptr;
  const size_t total = item_stride;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const double mask = ap.mask(r, c);
    const double w = ap.weight(mask);
    if (mask_out) {
      if (weights_out) mask_out[(size_t)item * item_stride + m] = w;
      else if (item == 0) mask_out[(size_t)r * n + c] = mask;
      continue;
    }
    // w == 1: untouched (no traffic).  w == 0: the pixel becomes +0 without being read -- most
    // of the grid lies outside a pupil, so this halves the kernel's HBM traffic.  (The
    // reference multiplies, so a non-finite value out there would turn into NaN instead.)
    if (w == 0.0) {
      f[m] = {(T)0, (T)0};
    } else if (w != 1.0) {
      cx<T> v = f[m];
      v.x = (T)__dmul_rn((double)v.x, w);
      v.y = (T)__dmul_rn((double)v.y, w);
      f[m] = v;
    }
  }
}

// ---- the first surface in one go: ones -> aperture -> [make_stop] ------------------------------
// wfo.py:118 (u = 1), :273-276 (u *= w), :200-201 (u /= sqrt(sum |u|^2)) on a field that is still
// the constant `value`.  Stage 1 sums |value w|^2 from the weights alone (no HBM traffic), stage 2
// (norm2_final_kernel) orders the partial sums, stage 3 writes value * w [* 1/sqrt(norm)] once:
// 16 B/px instead of fill 16 + aperture ~8 + norm 16 + scale 32.
template <typename T, int BR, int BC, int SHAPE>
__global__ void start_power_kernel(const double* params, int n, unsigned pitch, unsigned item_stride,
                                   double vre, double vim, double* partial, const double* stop) {
  const int item = blockIdx.y;
  if (stop[item] == 0.0) return;
  __shared__ double sh[kPwThreads / 64];
  const double* p = params + (size_t)item * AP_STRIDE;
  ApertureEval<SHAPE> ap;
  ap.init(p, p + AP_THETA);
  const bool on = p[AP_ENABLE] != 0.0;
  const double v0re = (double)(T)vre, v0im = (double)(T)vim;  // what fill stores
  const size_t total = item_stride;
  double acc = 0.0;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const double w = on ? ap.weight(ap.mask(r, c)) : 1.0;
    // the value aperture_kernel leaves in the field, read back the way norm2_partial_kernel does
    const double x = w == 1.0 ? v0re : (w == 0.0 ? 0.0 : (double)(T)__dmul_rn(v0re, w));
    const double y = w == 1.0 ? v0im : (w == 0.0 ? 0.0 : (double)(T)__dmul_rn(v0im, w));
    acc += __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
  }
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[(size_t)item * gridDim.x + blockIdx.x] = s;
}

template <typename T, int BR, int BC, int SHAPE>
__global__ void start_write_kernel(cx<T>* field, const double* params, int n, unsigned pitch,
                                   unsigned item_stride, double vre, double vim, const double* norm2,
                                   const double* stop, const double* rows, const double* grp_off,
                                   const double* grp_len, const double* grp_members, const double* cols = nullptr) {
  // Items with the same aperture record, stop flag and row window (a wavelength sweep at the entrance pupil, a
  // Monte-Carlo batch) start from the same field: the first of them evaluates the weights once per pixel and writes
  // every member of its group (grp_members[grp_off[item] ...], grp_len[item] of them; 0 = somebody else's member).
  const int item = blockIdx.y;
  const int glen = (int)grp_len[item];
  if (glen == 0) return;
  const double* members = grp_members + (int)grp_off[item];
  const double* p = params + (size_t)item * AP_STRIDE;
  ApertureEval<SHAPE> ap;
  ap.init(p, p + AP_THETA);
  const bool on = p[AP_ENABLE] != 0.0;
  const bool scaled = stop[item] != 0.0;
  const double s = scaled ? 1.0 / sqrt(norm2[item]) : 1.0;
  const double v0re = (double)(T)vre, v0im = (double)(T)vim;
  // ``rows`` ([batch][2], optional): only the rows [lo, hi) are written; the others are left as they are and stand
  // for zeros (paos_start_rows: the caller promises that nobody reads them before a pass program consumes them)
  size_t total = item_stride, first = 0;
  if (rows) {
    first = (size_t)((int)rows[2 * item] / BR) * pitch;
    total = (size_t)(((int)rows[2 * item + 1] + BR - 1) / BR) * pitch;
    if (total > item_stride) total = item_stride;
  }
  // (round 5) ``cols``: only the columns [lo, hi) -- whole blocks -- of those rows are written: the walk covers the
  // in-window part of every block row, `span` elements starting `in_lo` elements into it
  unsigned in_lo = 0, span = pitch;
  if (cols) {
    in_lo = (unsigned)((int)cols[2 * item] / BC) * (BR * BC);
    unsigned h = (unsigned)(((int)cols[2 * item + 1] + BC - 1) / BC) * (BR * BC);
    if (h > (unsigned)n * BR) h = (unsigned)n * BR;
    span = h > in_lo ? h - in_lo : 0;
  }
  const size_t nbr = (total - first) / pitch;  // block rows to write
  const size_t work = cols ? nbr * span : total - first;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < work; j += (size_t)gridDim.x * blockDim.x) {
    const size_t m = cols ? first + (j / span) * pitch + in_lo + (j % span) : first + j;
    int r, c;
    double x = 0.0, y = 0.0;
    if (layout_unmap<BR, BC>(m, n, pitch, r, c)) {
      const double w = on ? ap.weight(ap.mask(r, c)) : 1.0;
      // the same roundings as fill -> aperture_kernel -> stop_scale_kernel, via the field's type
      x = w == 1.0 ? v0re : (w == 0.0 ? 0.0 : (double)(T)__dmul_rn(v0re, w));
      y = w == 1.0 ? v0im : (w == 0.0 ? 0.0 : (double)(T)__dmul_rn(v0im, w));
      if (scaled) { x = __dmul_rn(x, s); y = __dmul_rn(y, s); }
    }
    for (int g = 0; g < glen; ++g) field[(size_t)(int)members[g] * item_stride + m] = {(T)x, (T)y};
  }
}

// everything outside rows [lo, hi) x columns [lo, hi) of every item := 0 (round 5: what makes a box that merely stands for
// zeros around it -- paos_start_box -- a field anybody may read; bounds rounded outward to whole blocks)
template <typename T, int BR, int BC>
__global__ void zero_outside_box_kernel(cx<T>* field, int n, unsigned pitch, unsigned item_stride, const double* rows,
                                        const double* cols) {
  const int item = blockIdx.y;
  const int rlo = ((int)rows[2 * item] / BR) * BR, rhi = (((int)rows[2 * item + 1] + BR - 1) / BR) * BR;
  const int clo = cols ? ((int)cols[2 * item] / BC) * BC : 0, chi = cols ? (((int)cols[2 * item + 1] + BC - 1) / BC) * BC : n;
  cx<T>* f = field + (size_t)item * item_stride;
  for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < item_stride; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    if (r < rlo || r >= rhi || c < clo || c >= chi) f[m] = {(T)0, (T)0};
  }
}

// ---- aperture line records for the frugal pass (frugal_pass.h: MaskLine) -------------------
// One wave per line (axis 0: a row, positions = columns; axis 1: a column, positions = rows).
// The wave walks the line 64 pixels at a time, evaluates the SAME per-pixel functions as
// aperture_kernel, and records where the weight leaves w_out / reaches w_in.  theta must be 0
// (the only case run() produces, run.py:114-121); the host checks that the partial runs fit.
//   ellipse:   w(row, col) = ellipse_pixel(...) [1 - ... for an obscuration], lm = 1
//   rectangle: mask = (cx(col)/S)(cy(row)/S) is separable, so along a row the record holds
//              cx(col)/S and lm = cy(row)/S (and vice versa); obscurations of rectangles are not
//              separable as 1 - mask and stay on the stand-alone kernel.
constexpr int kMaskWaves = 4;  // waves (= lines) per workgroup of mask_lines_kernel

// One rendering: the aperture's two consecutive parameter block sets (`params`, [batch][param_stride] each), the pass
// axis, where the records go, which items only read an earlier item's records (`shared`), and [line0, line_end): the
// lines whose records anybody will read (the pass that carries the aperture skips the tiles of dead lines: three
// quarters of a grid that covers every line were workgroups that only found out they had nothing to do).
struct MaskJob {
  const double* params;
  const double* shared;
  MaskLine* lines;
  double* vals;
  int axis, line0, line_end, shapes;  // shapes: bit s set = some item's aperture has shape s
};
constexpr int kMaskJobs = 8;
// Round 4: the renderings a pass program needs go out in ONE launch (blockIdx.z = job): five relay apertures of a SYN20
// step as five launches of a latency-bound kernel were 1.3 ms of a 62 ms step.
struct MaskJobs {
  MaskJob job[kMaskJobs];
  int windows;       // 1: lines whose two boundary runs fit two 32-pixel windows take the one-evaluation path (PAOS_MASK_SCAN=1: 0)
  int pairs;         // (round 5) 2: a wave renders FOUR lines at once when all fit two 8-pixel windows, else / 1: TWO through 16-pixel windows (PAOS_MASK_PAIRS=1), 0: one line per wave (=0)
  int rect_blocks;   // 1 (round 5): a wave renders 64 lines of a rectangle's records at once -- one column profile for all of them (PAOS_MASK_RECT_BLOCKS=0: 0)
  int batch_stride;  // doubles between the two parameter block sets of a job (= batch * param_stride)
  int param_stride, n;
  int* overflow;
};

// one line by the 64 lanes of a wave (the whole wave calls it: ballots, a wave-private LDS cache)
template <int SHAPE>
__device__ void mask_line_render(const MaskJobs& jobs, const MaskJob& jb, int item, int line, int lane) {
  const int n = jobs.n, axis = jb.axis, param_stride = jobs.param_stride;
  int* const overflow = jobs.overflow;
  if (line >= jb.line_end || line >= n) return;
  const double* p = jb.params + (size_t)item * param_stride;
  const double* p2 = jb.params + jobs.batch_stride + (size_t)item * param_stride;
  MaskLine* out = jb.lines + (size_t)item * n + line;
  double* vout = jb.vals + ((size_t)item * n + line) * (2 * kMaskW);
  if (p[AP_ENABLE] == 0.0 || SHAPE != (int)p2[3]) return;
  const double xc = p[AP_XC], yc = p[AP_YC], a = p[AP_A], b = p[AP_B];
  const bool obsc = p2[1] != 0.0;
  const int subpix = (int)p2[2];
  const double w_in = obsc ? 0.0 : 1.0, w_out = obsc ? 1.0 : 0.0;
  double xe, ye, hw = 0.0, hh = 0.0, full_disk = 0.0, lm = 1.0;
  if (SHAPE == 0) {
    xe = sqrt(__dmul_rn(a, a));  // cos 0 = 1, sin 0 = 0: the theta = 0 form of aperture_kernel's extents
    ye = sqrt(__dmul_rn(b, b));
    full_disk = fmin(__dmul_rn(__dmul_rn(3.141592653589793, a), b), 1.0);
  } else {
    hw = a / 2.0; hh = b / 2.0;
    xe = fabs(hw); ye = fabs(hh);
  }
  const ApertureBox box = make_box(xc, yc, xe, ye);
  if (SHAPE == 1) {
    const int c_other = axis == 0 ? subpixel_count_1d(line, yc, hh, subpix) : subpixel_count_1d(line, xc, hw, subpix);
    const bool in_box = axis == 0 ? (line >= box.iymin && line < box.iymax) : (line >= box.ixmin && line < box.ixmax);
    lm = in_box ? (double)c_other / (double)subpix : 0.0;
  }
  int p0 = n, p1 = -1, p2i = -1, p3 = 0;  // first non-out, first in, last in + 1, last non-out + 1
  // Only the part of the line inside the bounding box can differ from w_out: lines that miss
  // the box are done, the others are scanned over the box's extent along the line only.
  const bool line_in_box = axis == 0 ? (line >= box.iymin && line < box.iymax) : (line >= box.ixmin && line < box.ixmax);
  const int scan_lo = line_in_box ? max(0, (axis == 0 ? box.ixmin : box.iymin)) & ~63 : 0;
  const int scan_hi = line_in_box ? min(n, axis == 0 ? box.ixmax : box.iymax) : 0;
  // Round 4: the two boundary windows of the chord at once.  The partially covered pixels of a line lie where the
  // ellipse crosses the strip of the line: between the chord at the strip's edge nearer the centre and the chord at its
  // far edge, on either side.  When both of these runs fit 32 pixels (with a margin of 3; every line but the few near
  // the tips of the ellipse), lanes 0-31 evaluate the exact overlap of the left window and lanes 32-63 of the right one
  // -- ONE evaluation of ellipse_pixel per line instead of one per 64-pixel chunk that touches the boundary (2-4) --
  // and the rest of the line is classified by the same two rectangle tests the chunks use (see below: exactly
  // consistent with the per-pixel rule), applied to the span between the windows (inside) and to the two spans
  // outside them (outside).  Whatever fails -- a window too wide, a test that does not hold -- takes the scan below.
  // Records are identical to the scan's bit for bit (tests/test_gpu_r4.py, PAOS_MASK_SCAN=1 forces the scan).
  if (SHAPE == 0 && line_in_box && jobs.windows != 0) {
    const double sa = axis == 0 ? a : b, sc = axis == 0 ? b : a;          // semi-axes along / across the line
    const double ca = axis == 0 ? xc : yc, cc = axis == 0 ? yc : xc;      // centre along / across
    const double lo_c = __dsub_rn((double)line - 0.5, cc), hi_c = __dsub_rn((double)line + 0.5, cc);
    const double near_c = (lo_c > 0.0 ? lo_c : (hi_c < 0.0 ? -hi_c : 0.0)) / sc;   // strip edge nearer the centre
    const double far_c = fmax(fabs(lo_c), fabs(hi_c)) / sc;
    const double half_long = near_c < 1.0 ? sa * sqrt(1.0 - near_c * near_c) : 0.0;
    const double half_short = far_c < 1.0 ? sa * sqrt(1.0 - far_c * far_c) : 0.0;
    const int box_lo = max(0, axis == 0 ? box.ixmin : box.iymin), box_hi = min(n, axis == 0 ? box.ixmax : box.iymax);
    const int wl0 = max(box_lo, (int)floor(ca - half_long) - 3), wl1 = (int)ceil(ca - half_short) + 3;   // [wl0, wl1]
    const int wr1 = min(box_hi - 1, (int)ceil(ca + half_long) + 3), wr0 = wr1 - 31;                       // [wr0, wr1]
    const int wr_need = (int)floor(ca + half_short) - 3;
    auto sum2 = [&](double al, double ac) { const double u = al / sa, v = ac / sc; return axis == 0 ? __dadd_rn(__dmul_rn(u, u), __dmul_rn(v, v)) : __dadd_rn(__dmul_rn(v, v), __dmul_rn(u, u)); };
    bool ok = wl1 - wl0 < 32 && wr_need >= wr0 && wl0 + 32 <= wr0 && half_long > 0.0;
    if (ok) {
      // between the windows: the four outer corners of the span inside the ellipse => every pixel of it is (monotone)
      const double ia = __dsub_rn((double)(wl0 + 32) - 0.5, ca), ib = __dsub_rn((double)(wr0 - 1) + 0.5, ca);
      ok = sum2(ia, lo_c) <= 1.0 && sum2(ib, lo_c) <= 1.0 && sum2(ia, hi_c) <= 1.0 && sum2(ib, hi_c) <= 1.0;
      // left of the left window and right of the right one: the nearest point of each span outside by a margin
      const double nc = lo_c > 0.0 ? lo_c : (hi_c < 0.0 ? hi_c : 0.0);
      if (ok && wl0 > box_lo) {
        const double hi_a = __dsub_rn((double)(wl0 - 1) + 0.5, ca);
        ok = hi_a < 0.0 && sum2(hi_a, nc) > 1.0 + 1.0e-9;
      }
      if (ok && wr1 + 1 < box_hi) {
        const double lo_a = __dsub_rn((double)(wr1 + 1) - 0.5, ca);
        ok = lo_a > 0.0 && sum2(lo_a, nc) > 1.0 + 1.0e-9;
      }
    }
    if (ok) {  // wave-uniform
      const int pos = lane < 32 ? wl0 + lane : wr0 + (lane - 32);
      const int c = axis == 0 ? pos : line, r = axis == 0 ? line : pos;
      double mask = 0.0;
      if (c >= box.ixmin && c < box.ixmax && r >= box.iymin && r < box.iymax) mask = ellipse_pixel(c, r, xc, yc, a, b, 1.0, 0.0, full_disk);
      const double w = obsc ? __dsub_rn(1.0, mask) : mask;
      const unsigned long long not_out = __ballot(w != w_out), is_in = __ballot(w == w_in);
      const unsigned no_l = (unsigned)not_out, no_r = (unsigned)(not_out >> 32), in_l = (unsigned)is_in, in_r = (unsigned)(is_in >> 32);
      const int i0 = wl0 + 32, i1 = wr0;  // the span between the windows: w_in throughout (i0 <= i1; empty when equal)
      // the scan's four extents over: left window | span | right window
      p0 = no_l ? wl0 + (__ffs((int)no_l) - 1) : (i0 < i1 ? i0 : (no_r ? wr0 + (__ffs((int)no_r) - 1) : n));
      p3 = no_r ? wr0 + 32 - __clz((int)no_r) : (i0 < i1 ? i1 : (no_l ? wl0 + 32 - __clz((int)no_l) : 0));
      p1 = in_l ? wl0 + (__ffs((int)in_l) - 1) : (i0 < i1 ? i0 : (in_r ? wr0 + (__ffs((int)in_r) - 1) : -1));
      p2i = in_r ? wr0 + 32 - __clz((int)in_r) : (i0 < i1 ? i1 : (in_l ? wl0 + 32 - __clz((int)in_l) : -1));
      if (p3 <= p0) { p0 = p1 = p2i = p3 = 0; }
      else if (p1 < 0) { p1 = p2i = p3; }
      if (p1 - p0 > kMaskW || p3 - p2i > kMaskW) {
        if (lane == 0) atomicAdd(overflow, 1);
        p0 = p1 = p2i = p3 = 0;
      }
      if (pos >= p0 && pos < p1) vout[pos - p0] = w;
      if (pos >= p2i && pos < p3) vout[kMaskW + pos - p2i] = w;
      if (lane == 0) *out = {p0, p1, p2i, p3, lm, 0.0};
      return;
    }
  }
  // pass 1: extents.  Chunks of 64 pixels that hold partially covered pixels keep their weights in
  // a small per-wave LDS cache, so that pass 2 does not evaluate the exact overlap a second time.
  constexpr int kCache = 6;                           // chunks per line (two edge zones, a few chunks each)
  __shared__ double cache_w[kMaskWaves][kCache][64];
  __shared__ int cache_base[kMaskWaves][kCache];
  const int wave = threadIdx.x >> 6;
  int ncache = 0;
  for (int base = scan_lo; base < scan_hi; base += 64) {
    const int pos = base + lane;
    const int c = axis == 0 ? pos : line, r = axis == 0 ? line : pos;
    double mask = 0.0;
    // Whole chunks inside or outside the ellipse (wave-uniform tests on the chunk's rectangle, theta = 0):
    //  * inside: ellipse_pixel returns exactly 1 when the four corners of a pixel pass x'^2 + y'^2 <= 1; along a
    //    row edge that sum is a monotone function of |x'| also in floating point (every operation rounds
    //    monotonically), so when the chunk's four outer corners pass, every pixel corner in between does;
    //  * outside: the point of the rectangle nearest the centre lies outside by a margin no rounding bridges,
    //    so no pixel touches the ellipse and none contains its centre: ellipse_pixel returns 0.
    // A line through the middle of a 1024-pixel pupil evaluates 2-4 chunks instead of 16.
    int chunk = 0;  // 0: per pixel, 1: all inside, 2: all outside
    if (SHAPE == 0) {
      const double lo_a = __dsub_rn((double)base - 0.5, axis == 0 ? xc : yc), hi_a = __dsub_rn((double)(base + 63) + 0.5, axis == 0 ? xc : yc);
      const double lo_c = __dsub_rn((double)line - 0.5, axis == 0 ? yc : xc), hi_c = __dsub_rn((double)line + 0.5, axis == 0 ? yc : xc);
      const double sa = axis == 0 ? a : b, sc = axis == 0 ? b : a;  // semi-axes along / across the line
      auto sum2 = [&](double al, double ac) { const double u = al / sa, v = ac / sc; return axis == 0 ? __dadd_rn(__dmul_rn(u, u), __dmul_rn(v, v)) : __dadd_rn(__dmul_rn(v, v), __dmul_rn(u, u)); };
      const bool whole = base >= (axis == 0 ? box.ixmin : box.iymin) && base + 64 <= (axis == 0 ? box.ixmax : box.iymax);
      if (whole && sum2(lo_a, lo_c) <= 1.0 && sum2(hi_a, lo_c) <= 1.0 && sum2(lo_a, hi_c) <= 1.0 && sum2(hi_a, hi_c) <= 1.0) chunk = 1;
      else {
        const double na = lo_a > 0.0 ? lo_a : (hi_a < 0.0 ? hi_a : 0.0), nc = lo_c > 0.0 ? lo_c : (hi_c < 0.0 ? hi_c : 0.0);
        if (sum2(na, nc) > 1.0 + 1.0e-9) chunk = 2;
      }
    }
    if (chunk == 1) mask = 1.0;
    else if (chunk == 0 && c >= box.ixmin && c < box.ixmax && r >= box.iymin && r < box.iymax) {
      if (SHAPE == 0) mask = ellipse_pixel(c, r, xc, yc, a, b, 1.0, 0.0, full_disk);
      else mask = (double)(axis == 0 ? subpixel_count_1d(c, xc, hw, subpix) : subpixel_count_1d(r, yc, hh, subpix)) / (double)subpix;
    }
    const double w = (SHAPE == 0 && obsc) ? __dsub_rn(1.0, mask) : mask;
    const double wi = SHAPE == 0 ? w_in : 1.0, wo = SHAPE == 0 ? w_out : 0.0;
    const unsigned long long not_out = __ballot(w != wo), is_in = __ballot(w == wi);
    if (not_out) {
      p0 = min(p0, base + (int)__ffsll((long long)not_out) - 1);
      p3 = max(p3, base + 64 - (int)__clzll((long long)not_out));
    }
    if (is_in) {
      if (p1 < 0) p1 = base + (int)__ffsll((long long)is_in) - 1;
      p2i = base + 64 - (int)__clzll((long long)is_in);
    }
    if ((not_out & ~is_in) != 0 && ncache < kCache) {  // wave-uniform: some pixel is neither out nor in
      cache_w[wave][ncache][lane] = w;
      if (lane == 0) cache_base[wave][ncache] = base;
      ++ncache;
    }
  }
  // the cache is written and read by this wave only: order the LDS traffic, no workgroup barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (p3 <= p0) { p0 = p1 = p2i = p3 = 0; }           // the line never leaves w_out
  else if (p1 < 0) { p1 = p2i = p3; }                  // no interior: one partial run [p0, p3)
  if (p1 - p0 > kMaskW || p3 - p2i > kMaskW) {
    if (lane == 0) atomicAdd(overflow, 1);
    p0 = p1 = p2i = p3 = 0;
  }
  // pass 2: partial values, from the cache where the chunk was kept
  for (int side = 0; side < 2; ++side) {
    const int lo = side == 0 ? p0 : p2i, hi = side == 0 ? p1 : p3;
    for (int pos = lo + lane; pos < hi; pos += 64) {
      int slot = -1;
      for (int k = 0; k < ncache; ++k)
        if (cache_base[wave][k] == (pos & ~63)) slot = k;
      double w;
      if (slot >= 0) {
        w = cache_w[wave][slot][pos & 63];
      } else {
        const int c = axis == 0 ? pos : line, r = axis == 0 ? line : pos;
        double mask = 0.0;
        if (c >= box.ixmin && c < box.ixmax && r >= box.iymin && r < box.iymax) {
          if (SHAPE == 0) mask = ellipse_pixel(c, r, xc, yc, a, b, 1.0, 0.0, full_disk);
          else mask = (double)(axis == 0 ? subpixel_count_1d(c, xc, hw, subpix) : subpixel_count_1d(r, yc, hh, subpix)) / (double)subpix;
        }
        w = (SHAPE == 0 && obsc) ? __dsub_rn(1.0, mask) : mask;
      }
      vout[side * kMaskW + pos - lo] = w;
    }
  }
  if (lane == 0) *out = {p0, p1, p2i, p3, lm, 0.0};
}

// Round 5: TWO -- then FOUR -- lines per wave.  The window path above spends one exact-overlap evaluation per lane on 2 x 32
// pixels of which two to four are partially covered; everywhere but near the tips of the ellipse the two boundary runs of a
// line fit two 16-pixel windows (margins of 3) and, away from the last few per cent of the lines at either tip, two 8-pixel
// windows (margins of 1: what makes a window exact are the tests below, not its margins).  LPW lines per wave: 64 / LPW lanes
// render one line, the lower half of them its left window, the upper half its right one -- a half or a quarter of the waves,
// the same per-pixel functions on the same pixels, the same span tests: records bit for bit those of the 32-pixel windows and
// of the scan (tests/test_gpu_r4.py, PAOS_MASK_PAIRS=0 / 1, PAOS_MASK_SCAN=1).  Returns false (for the whole wave) when one
// of the lines needs more: the caller then renders them with the next wider windows.
template <int LPW>
__device__ inline bool mask_multi_render(const MaskJobs& jobs, const MaskJob& jb, int item, int first, int lane) {
  constexpr int G = 64 / LPW, W = G / 2, M = W >= 16 ? 3 : 1;  // lanes per line, pixels per window, margin
  constexpr unsigned kWin = (1u << W) - 1u;
  const int n = jobs.n, axis = jb.axis, param_stride = jobs.param_stride;
  const int grp = lane / G, lg = lane % G;
  const int line = first + grp;
  const bool valid = line < jb.line_end && line < n;
  const double* p = jb.params + (size_t)item * param_stride;
  const double* p2 = jb.params + jobs.batch_stride + (size_t)item * param_stride;
  if (p[AP_ENABLE] == 0.0 || 0 != (int)p2[3]) return true;  // (not this kernel's shape: nothing to render, as the line path decides)
  const double xc = p[AP_XC], yc = p[AP_YC], a = p[AP_A], b = p[AP_B];
  const bool obsc = p2[1] != 0.0;
  const double w_in = obsc ? 0.0 : 1.0, w_out = obsc ? 1.0 : 0.0;
  const double xe = sqrt(__dmul_rn(a, a)), ye = sqrt(__dmul_rn(b, b));
  const double full_disk = fmin(__dmul_rn(__dmul_rn(3.141592653589793, a), b), 1.0);
  const ApertureBox box = make_box(xc, yc, xe, ye);
  const bool in_box = valid && (axis == 0 ? (line >= box.iymin && line < box.iymax) : (line >= box.ixmin && line < box.ixmax));
  bool ok = false;
  int wl0 = 0, wr0 = 0;
  if (in_box) {
    const double sa = axis == 0 ? a : b, sc = axis == 0 ? b : a;          // semi-axes along / across the line
    const double ca = axis == 0 ? xc : yc, cc = axis == 0 ? yc : xc;      // centre along / across
    const double lo_c = __dsub_rn((double)line - 0.5, cc), hi_c = __dsub_rn((double)line + 0.5, cc);
    const double near_c = (lo_c > 0.0 ? lo_c : (hi_c < 0.0 ? -hi_c : 0.0)) / sc;
    const double far_c = fmax(fabs(lo_c), fabs(hi_c)) / sc;
    const double half_long = near_c < 1.0 ? sa * sqrt(1.0 - near_c * near_c) : 0.0;
    const double half_short = far_c < 1.0 ? sa * sqrt(1.0 - far_c * far_c) : 0.0;
    const int box_lo = max(0, axis == 0 ? box.ixmin : box.iymin), box_hi = min(n, axis == 0 ? box.ixmax : box.iymax);
    wl0 = max(box_lo, (int)floor(ca - half_long) - M);
    const int wl1 = (int)ceil(ca - half_short) + M;
    const int wr1 = min(box_hi - 1, (int)ceil(ca + half_long) + M);
    wr0 = wr1 - (W - 1);
    const int wr_need = (int)floor(ca + half_short) - M;
    auto sum2 = [&](double al, double ac) { const double u = al / sa, v = ac / sc; return axis == 0 ? __dadd_rn(__dmul_rn(u, u), __dmul_rn(v, v)) : __dadd_rn(__dmul_rn(v, v), __dmul_rn(u, u)); };
    ok = wl1 - wl0 < W && wr_need >= wr0 && wl0 + W <= wr0 && half_long > 0.0;
    if (ok) {
      const double ia = __dsub_rn((double)(wl0 + W) - 0.5, ca), ib = __dsub_rn((double)(wr0 - 1) + 0.5, ca);
      ok = sum2(ia, lo_c) <= 1.0 && sum2(ib, lo_c) <= 1.0 && sum2(ia, hi_c) <= 1.0 && sum2(ib, hi_c) <= 1.0;
      const double nc = lo_c > 0.0 ? lo_c : (hi_c < 0.0 ? hi_c : 0.0);
      if (ok && wl0 > box_lo) {
        const double hi_a = __dsub_rn((double)(wl0 - 1) + 0.5, ca);
        ok = hi_a < 0.0 && sum2(hi_a, nc) > 1.0 + 1.0e-9;
      }
      if (ok && wr1 + 1 < box_hi) {
        const double lo_a = __dsub_rn((double)(wr1 + 1) - 0.5, ca);
        ok = lo_a > 0.0 && sum2(lo_a, nc) > 1.0 + 1.0e-9;
      }
    }
  }
  // a line outside the loop's range or outside the bounding box needs no window at all
  if (!__all(!valid || !in_box || ok)) return false;  // wave-uniform
  const int pos = lg < W ? wl0 + lg : wr0 + (lg - W);
  double w = w_out;
  if (in_box) {
    const int c = axis == 0 ? pos : line, r = axis == 0 ? line : pos;
    double mask = 0.0;
    if (c >= box.ixmin && c < box.ixmax && r >= box.iymin && r < box.iymax) mask = ellipse_pixel(c, r, xc, yc, a, b, 1.0, 0.0, full_disk);
    w = obsc ? __dsub_rn(1.0, mask) : mask;
  }
  const unsigned long long not_out64 = __ballot(w != w_out), is_in64 = __ballot(w == w_in);
  if (!valid) return true;
  MaskLine* out = jb.lines + (size_t)item * n + line;
  double* vout = jb.vals + ((size_t)item * n + line) * (2 * kMaskW);
  if (!in_box) {  // the line never leaves w_out (what the scan finds for it)
    if (lg == 0) *out = {0, 0, 0, 0, 1.0, 0.0};
    return true;
  }
  const unsigned not_out = (unsigned)(not_out64 >> (G * grp)), is_in = (unsigned)(is_in64 >> (G * grp));
  const unsigned no_l = not_out & kWin, no_r = (not_out >> W) & kWin, in_l = is_in & kWin, in_r = (is_in >> W) & kWin;
  const int i0 = wl0 + W, i1 = wr0;  // the span between the windows: w_in throughout
  int p0 = no_l ? wl0 + (__ffs((int)no_l) - 1) : (i0 < i1 ? i0 : (no_r ? wr0 + (__ffs((int)no_r) - 1) : n));
  int p3 = no_r ? wr0 + 32 - __clz((int)no_r) : (i0 < i1 ? i1 : (no_l ? wl0 + 32 - __clz((int)no_l) : 0));
  int p1 = in_l ? wl0 + (__ffs((int)in_l) - 1) : (i0 < i1 ? i0 : (in_r ? wr0 + (__ffs((int)in_r) - 1) : -1));
  int p2i = in_r ? wr0 + 32 - __clz((int)in_r) : (i0 < i1 ? i1 : (in_l ? wl0 + 32 - __clz((int)in_l) : -1));
  if (p3 <= p0) { p0 = p1 = p2i = p3 = 0; }
  else if (p1 < 0) { p1 = p2i = p3; }
  if (p1 - p0 > kMaskW || p3 - p2i > kMaskW) {
    if (lg == 0) atomicAdd(jobs.overflow, 1);
    p0 = p1 = p2i = p3 = 0;
  }
  if (pos >= p0 && pos < p1) vout[pos - p0] = w;
  if (pos >= p2i && pos < p3) vout[kMaskW + pos - p2i] = w;
  if (lg == 0) *out = {p0, p1, p2i, p3, 1.0, 0.0};
  return true;
}

// Round 5: the records of a RECTANGLE, 64 lines per wave.  The mask is separable -- along a line the record holds the counts of
// the line's axis over S, the same for every line inside the box, and `lm` the count of the other axis over S -- so the column
// profile the one-line path scans for every line (32 sub-sample tests per pixel) is scanned once per wave and copied; lane l
// owns line first + l and forms its `lm`.  The same per-pixel function on the same pixels: records bit for bit those of
// mask_line_render<1> (tests/test_gpu_r5.py, PAOS_MASK_RECT_BLOCKS=0).  `prof`: 2 kMaskW doubles of LDS of this wave.
__device__ inline void mask_rect_block_render(const MaskJobs& jobs, const MaskJob& jb, int item, int first, int lane, double* prof) {
  const int n = jobs.n, axis = jb.axis, param_stride = jobs.param_stride;
  if (first >= jb.line_end || first >= n) return;
  const double* p = jb.params + (size_t)item * param_stride;
  const double* p2 = jb.params + jobs.batch_stride + (size_t)item * param_stride;
  if (p[AP_ENABLE] == 0.0 || 1 != (int)p2[3]) return;
  const double xc = p[AP_XC], yc = p[AP_YC], a = p[AP_A], b = p[AP_B];
  const int subpix = (int)p2[2];
  const double hw = a / 2.0, hh = b / 2.0;
  const ApertureBox box = make_box(xc, yc, fabs(hw), fabs(hh));
  const int line = first + lane;
  const bool valid = line < jb.line_end && line < n;
  const bool in_box = valid && (axis == 0 ? (line >= box.iymin && line < box.iymax) : (line >= box.ixmin && line < box.ixmax));
  const int c_other = axis == 0 ? subpixel_count_1d(line, yc, hh, subpix) : subpixel_count_1d(line, xc, hw, subpix);
  const double lm = in_box ? (double)c_other / (double)subpix : 0.0;
  const unsigned long long in_lines = __ballot(in_box);
  int p0 = n, p1 = -1, p2i = -1, p3 = 0;  // first non-out, first in, last in + 1, last non-out + 1 -- of every line inside the box
  auto weight = [&](int pos) {  // mask_line_render<1>'s pixel rule for a line inside the box
    const bool inside = axis == 0 ? (pos >= box.ixmin && pos < box.ixmax) : (pos >= box.iymin && pos < box.iymax);
    return inside ? (double)(axis == 0 ? subpixel_count_1d(pos, xc, hw, subpix) : subpixel_count_1d(pos, yc, hh, subpix)) / (double)subpix : 0.0;
  };
  if (in_lines) {
    const int scan_lo = max(0, (axis == 0 ? box.ixmin : box.iymin)) & ~63, scan_hi = min(n, axis == 0 ? box.ixmax : box.iymax);
    for (int base = scan_lo; base < scan_hi; base += 64) {
      const double w = weight(base + lane);
      const unsigned long long not_out = __ballot(w != 0.0), is_in = __ballot(w == 1.0);
      if (not_out) {
        p0 = min(p0, base + (int)__ffsll((long long)not_out) - 1);
        p3 = max(p3, base + 64 - (int)__clzll((long long)not_out));
      }
      if (is_in) {
        if (p1 < 0) p1 = base + (int)__ffsll((long long)is_in) - 1;
        p2i = base + 64 - (int)__clzll((long long)is_in);
      }
    }
  }
  if (p3 <= p0) { p0 = p1 = p2i = p3 = 0; }           // the lines never leave w_out
  else if (p1 < 0) { p1 = p2i = p3; }                  // no interior: one partial run [p0, p3)
  if (p1 - p0 > kMaskW || p3 - p2i > kMaskW) {
    if (lane == 0) atomicAdd(jobs.overflow, __popcll(in_lines));
    p0 = p1 = p2i = p3 = 0;
  }
  for (int side = 0; side < 2; ++side) {
    const int lo = side == 0 ? p0 : p2i, hi = side == 0 ? p1 : p3;
    for (int pos = lo + lane; pos < hi; pos += 64) prof[side * kMaskW + pos - lo] = weight(pos);
  }
  // the profile is written and read by this wave only: order the LDS traffic, no workgroup barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (valid) {
    MaskLine* out = jb.lines + (size_t)item * n + line;
    if (in_box) *out = {p0, p1, p2i, p3, lm, 0.0}; else *out = {0, 0, 0, 0, lm, 0.0};
  }
  const int len0 = p1 - p0, len1 = p3 - p2i;
  for (unsigned long long todo = in_lines; todo; todo &= todo - 1) {
    const int k = (int)__ffsll((long long)todo) - 1;
    double* vout = jb.vals + ((size_t)item * n + (first + k)) * (2 * kMaskW);
    for (int j = lane; j < len0; j += 64) vout[j] = prof[j];
    for (int j = lane; j < len1; j += 64) vout[kMaskW + j] = prof[kMaskW + j];
  }
}

template <int SHAPE>
__global__ void __launch_bounds__(kMaskWaves * 64) mask_lines_kernel(MaskJobs jobs) {
  const MaskJob& jb = jobs.job[blockIdx.z];
  if (!(jb.shapes & (1 << SHAPE))) return;
  const int item = blockIdx.y;
  if (jb.shared[item] != 0.0) return;  // reads the records of an earlier, identical item
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (SHAPE == 1 && jobs.rect_blocks != 0) {  // 64 lines per wave (the grid covers a 64th of the waves: launch_mask_jobs)
    __shared__ double prof[kMaskWaves][2 * kMaskW];
    mask_rect_block_render(jobs, jb, item, jb.line0 + 64 * wave, lane, prof[threadIdx.x >> 6]);
    return;
  }
  if (!(SHAPE == 0 && jobs.pairs != 0)) {  // one line per wave
    mask_line_render<SHAPE>(jobs, jb, item, jb.line0 + wave, lane);
    return;
  }
  const int per = jobs.pairs >= 2 ? 4 : 2;  // lines per wave (the grid covers a half / a quarter as many waves: launch_mask_jobs)
  const int first = jb.line0 + per * wave;
  if (first >= jb.line_end || first >= jobs.n) return;
  if (per == 4 && mask_multi_render<4>(jobs, jb, item, first, lane)) return;
  for (int f = first; f < first + per; f += 2) {
    if (mask_multi_render<2>(jobs, jb, item, f, lane)) continue;
    mask_line_render<0>(jobs, jb, item, f, lane);
    mask_line_render<0>(jobs, jb, item, f + 1, lane);
  }
}

// ---- Zernike phase ----------------------------------------------------------------
// shared table (doubles): [0]=nmax, then for am in 0..nmax, k in 0..KMAXP-1 the Jacobi
// recurrence constants A, B, C of P_k^{(am,0)}:  P_k = (A x + B) P_{k-1} - C P_{k-2}.
// per-item block: [enable, dx, dy, radius, origin_is_y, cos_off, sin_off, inv_wl,
//                  then coefC[am][k], coefS[am][k]] with the (-1)^k norm Z product folded in.
enum : int { ZP_ENABLE = 0, ZP_DX, ZP_DY, ZP_RADIUS, ZP_ORIGIN_Y, ZP_COS_OFF, ZP_SIN_OFF, ZP_INV_WL, ZP_HEAD };

// NMAXC > 0: the loops over the azimuthal order and the radial index are unrolled for orders up to NMAXC (the host
// picks this build when nmax <= NMAXC): the recurrence constants and coefficients of a whole order are then fetched
// in one go instead of five dependent scalar loads per term (round 3: the kernel waited four times as long as it
// computed, profiles/r03_sq_counters.txt).  NMAXC = 0: any order, rolled loops.
template <typename T, int BR, int BC, int NMAXC = 0>
__global__ void zernike_kernel(cx<T>* field, const double* table, const double* params,
                               int param_stride, int n, unsigned pitch, unsigned item_stride,
                               int nmax, int kdim, double* wfe_out, const double* pupil,
                               unsigned m_first, unsigned m_end, const double* grp_off, const double* grp_len,
                               const double* grp_members, const double* grp_twins = nullptr) {
  // Items whose records differ in the wavelength only (a wavelength sweep through one lens file: same sampling,
  // same coefficients) have the same wfe map: the first of them evaluates the polynomials once per pixel and applies
  // the phase to every member of its group (grp_members[grp_off[item] ...], grp_len[item] of them; 0 = this item is
  // somebody else's member, or switched off).  The arithmetic per item is unchanged.
  const int item = blockIdx.y;
  const int glen = (int)grp_len[item];
  if (glen == 0) return;
  const double* members = grp_members + (int)grp_off[item];
  // (round 5) grp_twins[item] != 0: the caller vouches that the fields of this group's members are copies of the leader's
  // (the surface right behind the start of a sweep, paos_zernike_like): one load per pixel instead of one per member
  const bool twins = grp_twins && grp_twins[item] != 0.0;
  const double* p = params + (size_t)item * param_stride;
  const double dx = p[ZP_DX], dy = p[ZP_DY], radius = p[ZP_RADIUS];
  const bool origin_y = p[ZP_ORIGIN_Y] != 0.0;
  const double co = p[ZP_COS_OFF], so = p[ZP_SIN_OFF];
  const double* coef_c = p + ZP_HEAD;
  const double* coef_s = coef_c + (size_t)(nmax + 1) * kdim;
  // [m_first, m_end): the block rows that meet the unit disk of some item (the host bounds them); outside, the
  // kernel changes nothing
  const size_t total = m_end < item_stride ? m_end : item_stride;
  size_t m = (size_t)m_first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    if (!layout_unmap<BR, BC>(m, n, pitch, r, c)) continue;
    const double x = (double)(c - n / 2) * dx, y = (double)(r - n / 2) * dy;
    const double rr = sqrt(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)));
    const double rho = rr / radius;
    // outside the unit disk, or outside the pupil the polynomials were orthonormalised on
    const bool masked = rho > 1.0 || (pupil && pupil[(size_t)item * item_stride + m] == 0.0);
    double wfe = 0.0;
    if (!masked) {
      double c1 = 1.0, s1 = 0.0;
      if (rr > 0.0) {
        c1 = (origin_y ? y : x) / rr;
        s1 = (origin_y ? x : y) / rr;
      }
      const double cr = c1 * co - s1 * so, sr = s1 * co + c1 * so;
      const double xj = 1.0 - 2.0 * rho * rho;
      double rho_pow = 1.0, cm = 1.0, sm = 0.0;
      auto order = [&](int am, int k_end) __attribute__((always_inline)) {
        double pkm1 = 0.0, pk = 1.0;
        for (int k = 0; k <= k_end; ++k) {
          if (k > 0) {
            const double* abc = table + ((size_t)am * kdim + k) * 3;
            const double pn = (abc[0] * xj + abc[1]) * pk - abc[2] * pkm1;
            pkm1 = pk;
            pk = pn;
          }
          const size_t ci = (size_t)am * kdim + k;
          wfe += (rho_pow * pk) * (coef_c[ci] * cm + coef_s[ci] * sm);
        }
        rho_pow *= rho;
        const double cn = cm * cr - sm * sr;
        sm = sm * cr + cm * sr;
        cm = cn;
      };
      if constexpr (NMAXC > 0) {  // same operations in the same order, loops unrolled
#pragma unroll
        for (int am = 0; am <= NMAXC; ++am) {
          if (am > nmax) break;
          const int kmax = (nmax - am) / 2;
          double pkm1 = 0.0, pk = 1.0;
#pragma unroll
          for (int k = 0; k <= (NMAXC - am) / 2; ++k) {
            if (k > kmax) break;
            if (k > 0) {
              const double* abc = table + ((size_t)am * kdim + k) * 3;
              const double pn = (abc[0] * xj + abc[1]) * pk - abc[2] * pkm1;
              pkm1 = pk;
              pk = pn;
            }
            const size_t ci = (size_t)am * kdim + k;
            wfe += (rho_pow * pk) * (coef_c[ci] * cm + coef_s[ci] * sm);
          }
          rho_pow *= rho;
          const double cn = cm * cr - sm * sr;
          sm = sm * cr + cm * sr;
          cm = cn;
        }
      } else {
        for (int am = 0; am <= nmax; ++am) order(am, (nmax - am) / 2);
      }
      const double turns = __dmul_rn(6.283185307179586, wfe);
      // (round 5) eight members at a time, their loads issued before the first store: the fields of different items never
      // alias, which the compiler cannot know -- one load -> sincos -> store chain per member was a 32-deep latency chain
      // per pixel (0.37 ms per step of a 32-wavelength sweep); the arithmetic per member is unchanged
      constexpr int kZernikeGroup = 8;  // (4: 0.37 -> 0.29 ms per step, 8: see profiles/r05_ab_variants_bench.txt)
      cx<double> lead = {0.0, 0.0};
      if (twins) { const cx<T>* fl = field + (size_t)item * item_stride + m; lead = {(double)fl->x, (double)fl->y}; }
      int g = 0;
      for (; g + kZernikeGroup <= glen; g += kZernikeGroup) {
        cx<T>* f4[kZernikeGroup];
        cx<double> v4[kZernikeGroup];
        double iw[kZernikeGroup];
#pragma unroll
        for (int q = 0; q < kZernikeGroup; ++q) {
          const int it = (int)members[g + q];
          f4[q] = field + (size_t)it * item_stride + m;
          iw[q] = params[(size_t)it * param_stride + ZP_INV_WL];
        }
        if (twins) {
#pragma unroll
          for (int q = 0; q < kZernikeGroup; ++q) v4[q] = lead;
        } else {
#pragma unroll
          for (int q = 0; q < kZernikeGroup; ++q) v4[q] = {(double)f4[q]->x, (double)f4[q]->y};
        }
#pragma unroll
        for (int q = 0; q < kZernikeGroup; ++q) {
          double sn, cs;
          sincos_fast(__dmul_rn(turns, iw[q]), &sn, &cs);
          *f4[q] = {(T)__dsub_rn(__dmul_rn(v4[q].x, cs), __dmul_rn(v4[q].y, sn)),
                    (T)__dadd_rn(__dmul_rn(v4[q].x, sn), __dmul_rn(v4[q].y, cs))};
        }
      }
      for (; g < glen; ++g) {
        const int it = (int)members[g];
        const double arg = __dmul_rn(turns, params[(size_t)it * param_stride + ZP_INV_WL]);
        double sn, cs;
        sincos_fast(arg, &sn, &cs);
        cx<T>* f = field + (size_t)it * item_stride;
        const cx<double> v = twins ? lead : cx<double>{(double)f[m].x, (double)f[m].y};
        f[m] = {(T)__dsub_rn(__dmul_rn(v.x, cs), __dmul_rn(v.y, sn)),
                (T)__dadd_rn(__dmul_rn(v.x, sn), __dmul_rn(v.y, cs))};
      }
    }
    if (wfe_out && item == 0) wfe_out[(size_t)r * n + c] = masked ? __longlong_as_double(0x7ff8000000000000LL) : wfe;
  }
}

// ---- PolyOrthoNorm: Gram sums of the Zernike polynomials over the pupil -----------------------
// zernike.py:293-318 (cov) needs mean(Z_i Z_j) over the unmasked pixels for the first K
// polynomials.  Per 128-pixel chunk, 128 threads evaluate the K values of their pixel into LDS
// (one Jacobi recurrence + one rotation per pixel, like zernike_kernel), then the 256 threads of
// the workgroup each own up to kGramAcc (i, j) pairs and add the chunk's products.  Fixed chunk ->
// workgroup mapping and a second, ordered stage over workgroups: bit-reproducible sums.
constexpr int kGramMaxK = 64, kGramPix = 128, kGramThreads = 256, kGramRow = kGramPix + 1;
constexpr int kGramAcc = (kGramMaxK * (kGramMaxK + 1) / 2 + kGramThreads - 1) / kGramThreads;

// slots[(am * kdim + k) * 2 + {0: cos, 1: sin}] = polynomial index j or -1; fac[j] = its
// normalisation with the (-1)^k of the Jacobi form folded in (doubles: they ride the arena).
template <int BR, int BC>
__global__ void __launch_bounds__(kGramThreads)
    zernike_gram_kernel(const double* table, const double* params, int param_stride, int n, unsigned pitch,
                        unsigned item_stride, int nmax, int kdim, int K, const double* slots,
                        const double* fac, const double* pupil, double* partial) {
  extern __shared__ double zbuf[];  // [K][kGramRow]
  const int item = blockIdx.y, tid = threadIdx.x;
  const double* p = params + (size_t)item * param_stride;
  const int npairs = K * (K + 1) / 2;
  double* out = partial + ((size_t)item * gridDim.x + blockIdx.x) * (npairs + 1);
  if (p[ZP_ENABLE] == 0.0) {
    for (int q = tid; q <= npairs; q += kGramThreads) out[q] = 0.0;
    return;
  }
  const double dx = p[ZP_DX], dy = p[ZP_DY], radius = p[ZP_RADIUS];
  const bool origin_y = p[ZP_ORIGIN_Y] != 0.0;
  const double co = p[ZP_COS_OFF], so = p[ZP_SIN_OFF];
  int pi[kGramAcc], pj[kGramAcc];
  double acc[kGramAcc];
#pragma unroll
  for (int a = 0; a < kGramAcc; ++a) {
    acc[a] = 0.0;
    // pair q = (i, j), i <= j, enumerated row by row: q = i K - i (i - 1) / 2 + (j - i)
    int q = tid + a * kGramThreads, i = 0;
    if (q < npairs) {
      while (q >= K - i) { q -= K - i; ++i; }
      pi[a] = i; pj[a] = i + q;
    } else {
      pi[a] = pj[a] = -1;
    }
  }
  double count = 0.0;
  const size_t total = item_stride;
  for (size_t chunk = blockIdx.x; chunk * kGramPix < total; chunk += gridDim.x) {
    int valid = 0;
    if (tid < kGramPix) {
      const size_t m = chunk * kGramPix + tid;
      int r = 0, c = 0;
      double rr = 0.0, rho = 2.0, x = 0.0, y = 0.0;
      if (m < total && layout_unmap<BR, BC>(m, n, pitch, r, c)) {
        x = (double)(c - n / 2) * dx;
        y = (double)(r - n / 2) * dy;
        rr = sqrt(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)));
        rho = rr / radius;
        valid = rho <= 1.0 && (!pupil || pupil[(size_t)item * item_stride + m] != 0.0);
      }
      if (valid) {
        double c1 = 1.0, s1 = 0.0;
        if (rr > 0.0) {
          c1 = (origin_y ? y : x) / rr;
          s1 = (origin_y ? x : y) / rr;
        }
        const double cr = c1 * co - s1 * so, sr = s1 * co + c1 * so;
        const double xj = 1.0 - 2.0 * rho * rho;
        double rho_pow = 1.0, cm = 1.0, sm = 0.0;
        for (int am = 0; am <= nmax; ++am) {
          const int kmax = (nmax - am) / 2;
          double pkm1 = 0.0, pk = 1.0;
          for (int k = 0; k <= kmax; ++k) {
            if (k > 0) {
              const double* abc = table + ((size_t)am * kdim + k) * 3;
              const double pn = (abc[0] * xj + abc[1]) * pk - abc[2] * pkm1;
              pkm1 = pk;
              pk = pn;
            }
            const size_t ci = (size_t)am * kdim + k;
            const int jc = (int)slots[2 * ci], js = (int)slots[2 * ci + 1];
            const double base = rho_pow * pk;
            if (jc >= 0) zbuf[jc * kGramRow + tid] = fac[jc] * base * cm;
            if (js >= 0) zbuf[js * kGramRow + tid] = fac[js] * base * sm;
          }
          rho_pow *= rho;
          const double cn = cm * cr - sm * sr;
          sm = sm * cr + cm * sr;
          cm = cn;
        }
        count += 1.0;
      } else {
        for (int j = 0; j < K; ++j) zbuf[j * kGramRow + tid] = 0.0;
      }
    }
    if (!__syncthreads_or(valid)) continue;  // (also publishes zbuf)
#pragma unroll
    for (int a = 0; a < kGramAcc; ++a) {
      if (pi[a] < 0) continue;
      const double* zi = zbuf + pi[a] * kGramRow;
      const double* zj = zbuf + pj[a] * kGramRow;
      double sum = 0.0;
      for (int px = 0; px < kGramPix; ++px) sum += zi[px] * zj[px];
      acc[a] += sum;
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < kGramAcc; ++a)
    if (pi[a] >= 0) out[tid + a * kGramThreads] = acc[a];
  // pixel count of this workgroup: ordered sum over the 128 counting threads
  __syncthreads();
  if (tid < kGramPix) zbuf[tid] = count;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int k = 0; k < kGramPix; ++k) t += zbuf[k];
    out[npairs] = t;
  }
}

// out[item][q] = sum over workgroups, in order
static __global__ void zernike_gram_final_kernel(const double* partial, double* out, int nblocks, int nvals) {
  const int item = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nvals) return;
  double t = 0.0;
  for (int b = 0; b < nblocks; ++b) t += partial[((size_t)item * nblocks + b) * nvals + q];
  out[(size_t)item * nvals + q] = t;
}

// row-major host weights (staging) -> one item of the pupil weight map, in the field's layout
template <int BR, int BC>
__global__ void import_weights_kernel(const double* src, double* dst, int n, unsigned pitch, unsigned item_stride) {
  const size_t total = item_stride;
  size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; m < total; m += (size_t)gridDim.x * blockDim.x) {
    int r, c;
    dst[m] = layout_unmap<BR, BC>(m, n, pitch, r, c) ? src[(size_t)r * n + c] : 0.0;
  }
}

}  // namespace paos
