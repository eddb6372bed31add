// membench: what HBM rate can an in-place read-modify-write of a 2 GiB c128 batch reach on
// MI355X, as a function of access shape and occupancy?  The yardstick for the FFT passes
// (each pass reads and writes every field element exactly once, in place).
//   hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o build/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// one 16-byte element per thread
__global__ void rmw_simple(v2d* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { v2d v = p[i]; v = v * 1.0000001; p[i] = v; }
}

// U elements per thread, each wave-instruction one contiguous KiB; the workgroup owns a
// contiguous chunk of U * blockDim.x elements; all loads first, then all stores (like a tile)
template <int U, bool NT, bool OUT>
__global__ void rmw_chunk(v2d* p, v2d* q, size_t n, int lds_bytes) {
  extern __shared__ unsigned char smem[];
  if (lds_bytes < 0) smem[threadIdx.x] = 0;  // keep the allocation
  const size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x;
  v2d v[U];
#pragma unroll
  for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(&p[base + (size_t)k * blockDim.x]) : p[base + (size_t)k * blockDim.x];
#pragma unroll
  for (int k = 0; k < U; ++k) v[k] = v[k] * 1.0000001;
  v2d* o = OUT ? q : p;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    if (NT) __builtin_nontemporal_store(v[k], &o[base + (size_t)k * blockDim.x]);
    else o[base + (size_t)k * blockDim.x] = v[k];
  }
}

// the FFT row-tile shape: 2 rows of 4096 in the 4x2-blocked layout = 64-byte halves of 128-byte
// lines (stride 128 B), sibling workgroup (w + 8) takes the other halves
__global__ void rmw_halfblocks(v2d* p, size_t n_tiles) {
  extern __shared__ unsigned char smem[];
  int tile = blockIdx.x;
  const int grp = tile / 16, in = tile % 16;
  tile = (grp * 8 + in % 8) * 2 + in / 8;           // XCD-sibling order
  const int tid = threadIdx.x;
  const int bc = tid % 2, br = (tid / 2) % 2, q = tid / 4;   // 128 column pairs per sweep
  // block row = tile / 2, rows (tile % 2) * 2 + br inside the block; element offset inside block: row * 2 + col
  const size_t blockrow = (size_t)(tile / 2) * (4096 / 2 + 3) * 8;  // pitch: 2048 blocks + 3 pad, 8 elements each
  v2d v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int blk = q + k * 128;
    v[k] = p[blockrow + (size_t)blk * 8 + ((tile % 2) * 2 + br) * 2 + bc];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = v[k] * 1.0000001;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int blk = q + k * 128;
    p[blockrow + (size_t)blk * 8 + ((tile % 2) * 2 + br) * 2 + bc] = v[k];
  }
}

// MODE 0: read+write, 1: read only (a reduction keeps the loads alive), 2: write only
// SEQ: one workgroup takes rows (0,1) and then rows (2,3) of its block row instead of siblings
template <int MODE, bool SEQ>
__global__ void rmw_halfblocks2(v2d* p, double* sink) {
  extern __shared__ unsigned char smem[];
  int tile = blockIdx.x;
  if (!SEQ) { const int grp = tile / 16, in = tile % 16; tile = (grp * 8 + in % 8) * 2 + in / 8; }
  const int tid = threadIdx.x;
  const int bc = tid % 2, br = (tid / 2) % 2, q = tid / 4;
  double acc = 0;
  for (int half = 0; half < (SEQ ? 2 : 1); ++half) {
    const int brow = SEQ ? tile : tile / 2, sub = SEQ ? half : tile % 2;
    const size_t blockrow = (size_t)brow * (4096 / 2 + 3) * 8;
    v2d v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int blk = q + k * 128;
      if (MODE != 2) v[k] = p[blockrow + (size_t)blk * 8 + (sub * 2 + br) * 2 + bc];
      else v[k] = v2d{(double)k, (double)tid};
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = v[k] * 1.0000001;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int blk = q + k * 128;
      if (MODE != 1) p[blockrow + (size_t)blk * 8 + (sub * 2 + br) * 2 + bc] = v[k];
      else acc += v[k].x + v[k].y;
    }
  }
  if (MODE == 1 && acc == 1.2345e-300) sink[0] = acc;
}

// 2x2 blocks (64 B).  AXIS 0: a row tile (2 rows) is one contiguous block row.  AXIS 1: a column
// tile (2 columns) is one 64-byte block per block row, stride = block-row pitch; tiles c and
// c + 1 share 128-byte lines and run as XCD siblings.
template <int AXIS, int PAD>
__global__ void rmw_blk2x2(v2d* p) {
  extern __shared__ unsigned char smem[];
  const int item = blockIdx.y + blockIdx.x / 2048;
  int tile = blockIdx.x % 2048;
  const int tid = threadIdx.x;
  const int bc = tid % 2, br = (tid / 2) % 2, q = tid / 4;
  const size_t pitch = (size_t)(4096 / 2 + PAD) * 4;  // elements per block row: 2048 blocks + pad, 4 each
  v2d* f = p + (size_t)item * 2048 * pitch;
  size_t idx[16];
  if (AXIS == 1) { const int grp = tile / 16, in = tile % 16; tile = (grp * 8 + in % 8) * 2 + in / 8; }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int b = q + k * 128;  // block index along the line
    idx[k] = AXIS == 0 ? (size_t)tile * pitch + (size_t)b * 4 + br * 2 + bc
                       : (size_t)b * pitch + (size_t)tile * 4 + br * 2 + bc;
  }
  v2d v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = f[idx[k]];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = v[k] * 1.0000001;
#pragma unroll
  for (int k = 0; k < 16; ++k) f[idx[k]] = v[k];
}

// 4x2 blocks (128 B), column tile = one block column: whole lines, stride = block-row pitch
__global__ void rmw_cols4x2(v2d* p) {
  extern __shared__ unsigned char smem[];
  const int item = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int bc = tid % 2, br = (tid / 2) % 4, q = tid / 8;   // 64 blocks per sweep
  const size_t pitch = (size_t)(4096 / 2 + 3) * 8;
  v2d* f = p + (size_t)item * 1024 * pitch;
  v2d v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = f[(size_t)(q + k * 64) * pitch + (size_t)tile * 8 + br * 2 + bc];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = v[k] * 1.0000001;
#pragma unroll
  for (int k = 0; k < 16; ++k) f[(size_t)(q + k * 64) * pitch + (size_t)tile * 8 + br * 2 + bc] = v[k];
}

template <typename F>
float timeit(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const size_t n = (size_t)8 * 4096 * (4096 + 256);  // elements (16 B); room for every padded pitch used below (<= 2048 + 64 blocks of 4, 1024 + ... of 8)
  const size_t n_use = (size_t)8 * 4096 * 4096;
  v2d *p, *q;
  CK(hipMalloc(&p, n * 16)); CK(hipMalloc(&q, n * 16));
  CK(hipMemset(p, 0, n * 16)); CK(hipMemset(q, 0, n * 16));
  const double bytes = 2.0 * n_use * 16;
  auto report = [&](const char* name, float ms) { printf("%-58s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6); fflush(stdout); };

  report("simple: 16 B per thread, 256 thr", timeit([&] { hipLaunchKernelGGL(rmw_simple, dim3(n_use / 256), dim3(256), 0, 0, p, n_use); }, reps));
  report("chunk U=4, 256 thr, in place", timeit([&] { hipLaunchKernelGGL((rmw_chunk<4, false, false>), dim3(n_use / (256 * 4)), dim3(256), 0, 0, p, q, n_use, 0); }, reps));
  report("chunk U=4, 256 thr, out of place", timeit([&] { hipLaunchKernelGGL((rmw_chunk<4, false, true>), dim3(n_use / (256 * 4)), dim3(256), 0, 0, p, q, n_use, 0); }, reps));
  report("chunk U=16, 512 thr (tile-sized, 128 KiB), in place", timeit([&] { hipLaunchKernelGGL((rmw_chunk<16, false, false>), dim3(n_use / (512 * 16)), dim3(512), 0, 0, p, q, n_use, 0); }, reps));
  {
    auto k = rmw_chunk<16, false, false>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    report("  same, 70 KiB LDS (2 WG/CU)", timeit([&] { hipLaunchKernelGGL(k, dim3(n_use / (512 * 16)), dim3(512), 70 * 1024, 0, p, q, n_use, 0); }, reps));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    report("  same, 140 KiB LDS (1 WG/CU)", timeit([&] { hipLaunchKernelGGL(k, dim3(n_use / (512 * 16)), dim3(512), 140 * 1024, 0, p, q, n_use, 0); }, reps));
  }
  {
    auto k = rmw_chunk<16, true, false>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    report("  nontemporal, 70 KiB LDS (2 WG/CU)", timeit([&] { hipLaunchKernelGGL(k, dim3(n_use / (512 * 16)), dim3(512), 70 * 1024, 0, p, q, n_use, 0); }, reps));
  }
  {
    auto k = rmw_chunk<16, false, true>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    report("  out of place, 70 KiB LDS (2 WG/CU)", timeit([&] { hipLaunchKernelGGL(k, dim3(n_use / (512 * 16)), dim3(512), 70 * 1024, 0, p, q, n_use, 0); }, reps));
  }
  {
    CK(hipFuncSetAttribute((const void*)rmw_halfblocks, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    // per item 2048 tiles; run 8 items back to back as one launch over contiguous block rows
    report("FFT row-tile shape (64-B halves, sibling WGs), 2 WG/CU", timeit([&] { hipLaunchKernelGGL(rmw_halfblocks, dim3(8 * 2048), dim3(512), 70 * 1024, 0, p, (size_t)8 * 2048); }, reps));
  }
  double* sink; CK(hipMalloc(&sink, 8));
  auto run = [&](const char* name, auto kern, dim3 grid) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    report(name, timeit([&] { hipLaunchKernelGGL(kern, grid, dim3(512), 70 * 1024, 0, p, sink); }, reps));
  };
  run("  half blocks, siblings, read only (x2 for r+w rate)", rmw_halfblocks2<1, false>, dim3(8 * 2048));
  run("  half blocks, siblings, write only (x2 for r+w rate)", rmw_halfblocks2<2, false>, dim3(8 * 2048));
  run("  half blocks, one WG takes both halves in turn", rmw_halfblocks2<0, true>, dim3(8 * 1024));
  auto run1 = [&](const char* name, auto kern, dim3 grid) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    report(name, timeit([&] { hipLaunchKernelGGL(kern, grid, dim3(512), 70 * 1024, 0, p); }, reps));
  };
  run1("2x2 blocks: row tile (contiguous block row), pad 6", rmw_blk2x2<0, 6>, dim3(2048, 8));
  run1("2x2 blocks: row tile (contiguous block row), pad 0", rmw_blk2x2<0, 0>, dim3(2048, 8));
  run1("2x2 blocks: row tile (contiguous block row), pad 64 (4 KiB)", rmw_blk2x2<0, 64>, dim3(2048, 8));
  run1("2x2 blocks: row tile, pad 0, 1-D grid", rmw_blk2x2<0, 0>, dim3(2048 * 8, 1));
  run1("2x2 blocks: column tile (64-B blocks, stride pitch, siblings), pad 6", rmw_blk2x2<1, 6>, dim3(2048, 8));
  run1("2x2 blocks: column tile, pad 0", rmw_blk2x2<1, 0>, dim3(2048, 8));
  run1("4x2 blocks: column tile (128-B lines, stride pitch)", rmw_cols4x2, dim3(2048, 8));
  {
    auto k = rmw_chunk<16, false, false>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    report("again at the end: chunk U=16, 512 thr, 70 KiB LDS (2 WG/CU)", timeit([&] { hipLaunchKernelGGL(k, dim3(n_use / (512 * 16)), dim3(512), 70 * 1024, 0, p, q, n_use, 0); }, reps));
  }
  return 0;
}