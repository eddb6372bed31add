// paos_hip.hip -- C ABI (include/paos_hip.h) over the gfx950 kernels.
//
// Host side of the library: context / buffer management, the per-call parameter
// arena (pinned ring -> device), launch geometry per grid size, error mapping.
// No PyTorch, no hipFFT, no CPU fallback: every operator is a kernel launch.
// Built either as one translation unit (PAOS_PART undefined) or as six in parallel:
// -DPAOS_PART=0 (everything but the frugal pass-kernel families) and -DPAOS_PART=1..5 (one family
// each), linked together -- see Makefile / __graft_entry__.build().
#ifndef PAOS_PART
#define PAOS_PART -1
#endif
#include "../../include/paos_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <type_traits>
#include <vector>

#include "frugal_pass.h"
#if PAOS_PART <= 0
#include "pointwise.h"
#endif

using namespace paos;

// ---- build-time layout choice -------------------------------------------------------
// A field is stored as blocks of 4 rows x 2 columns (128 B of complex128, 64 B of complex64),
// blocks row-major, so whole or half cache lines are the unit of work for the row pass (2 or 4
// rows per workgroup) and for the column pass (2 columns per workgroup) alike (DESIGN.md
// section 2).  PAD_BLOCKS extra blocks per block row de-tune the power-of-two stride of
// the column pass (measured: 3.5 -> 4.8 TB/s at 4096^2, profiles/r01_fftbench_v2_pitchpad.txt).
#ifndef PAOS_BR
#define PAOS_BR 4
#endif
#ifndef PAOS_PAD_BLOCKS
#define PAOS_PAD_BLOCKS 3
#endif
static constexpr int BR = PAOS_BR;  // complex128 and the base case of complex64
// complex64 at N >= 2048 (where the frugal kernels serve it): blocks of 8 rows x 2 columns = 128 B like a complex128
// block, so that column tiles own whole lines (round 3; profiles/r02_fftbench_c64_8row_blocks_experiment.txt).  Below
// 2048 the generic kernels keep 4 x 2 (a column tile needs N / 16 >= block height threads per line).
#ifndef PAOS_F32_BR
#define PAOS_F32_BR 8
#endif
template <typename T, int N>
constexpr int block_rows() { return (sizeof(T) == 4 && N >= 2048) ? PAOS_F32_BR : PAOS_BR; }
// one float launch for either block height: FBR is the compile-time block height inside the statement
#define F32_BR_SWITCH(c, ...)                                                            \
  do {                                                                                   \
    if ((c)->br == PAOS_F32_BR) { constexpr int FBR = PAOS_F32_BR; __VA_ARGS__; }        \
    else { constexpr int FBR = PAOS_BR; __VA_ARGS__; }                                   \
  } while (0)
[[maybe_unused]] static constexpr int kNormSlots = PAOS_NORM_SLOTS;  // outstanding paos_norm2_enqueue results
template <typename T>
struct Lay {
  // 2 columns for both types: a block is 128 B of complex128 or 64 B of complex64.  (4 columns
  // of complex64 would fill the line but make the column tile 4 lines = 1024 threads, which
  // spills; measured 126 wavefronts/s vs the 2-column shape below.)
  static constexpr int BC = 2;
};

namespace {

thread_local std::string g_err;

// Parameter arena: kArenaSlabs slabs of `cap` doubles each (pinned host mirror + device), filled one after the other.
// Round 5: a slab is reused only after the work that read it has run -- told by an EVENT recorded one slab switch after it
// was left (by then every launch that reads it has been enqueued: a call's pushes and its launches never span more than
// two slabs) -- instead of by a hipStreamSynchronize at every wrap of one ring: at 512 wavefronts per step the ring
// wrapped every second step, and each wrap made the host wait for the GPU to drain (1024^2: 33 ms per step where the GPU
// needs 25 and the host 18).  With four slabs the host may run two to three steps ahead and no further.
constexpr int kArenaSlabs = 4;
struct Arena {
  double* host = nullptr;  // pinned, kArenaSlabs * cap doubles
  double* dev = nullptr;
  size_t cap = 0, head = 0;  // doubles per slab / fill of the current slab
  int cur = 0;               // slab being filled
  int left = -1;             // the slab left at the last switch: its fence is recorded at the NEXT switch
  hipEvent_t fence[kArenaSlabs] = {};
  bool fenced[kArenaSlabs] = {};  // fence[k] has been recorded since slab k was last filled
  bool used[kArenaSlabs] = {};
};

}  // namespace

struct paos_ctx {
  int device = 0, n = 0, batch = 0, precision = 0;
  int br = PAOS_BR;  // block height of this context's layout (block_rows<T, N>())
  unsigned pitch = 0, item_stride = 0;
  hipStream_t stream = nullptr;
  double* psf = nullptr;  // batch x item_stride intensities kept on the device, blocked like the field (paos_psf_keep)
  double* map_dev = nullptr;      // one n x n phase map kept on the device (paos_phase_map_items) and the key it was uploaded under
  unsigned long long map_key = 0;
  // (round 5) the power sums of the last start, kept with everything they depend on (shape, constant, aperture records,
  // stop flags): the entrance pupil of a wavelength sweep or a Monte-Carlo study is the same step after step, and the sums
  // are a pure function of those -- the next start with the same key copies them instead of evaluating the exact pixel
  // overlaps again (start_power_kernel + norm2_final_kernel: 0.08 ms of a 14.6 ms SYN20 step).  PAOS_START_POWER_MEMO=0: never.
  std::vector<double> start_key;
  double* start_norm2 = nullptr;  // [batch]
  cx<double>* psd_scratch = nullptr;  // one item in the field's layout: the spectrum of a PSD screen (paos_psd_screen)
  int* psd_bad = nullptr;
  double* psf_partial = nullptr;  // per-workgroup sums of a pass that stores the PSF (paos_run_program: final_intensity)
  int psf_nparts = 0;
  double* pow_partial = nullptr;  // per-workgroup sums of |u|^2 of a pass that stores the FIELD (final_intensity = 2)
  int pow_nparts = 0;
  // [batch] factors every frugal pass multiplies into the scale of its middle slot (FrugalArgs::dyn_scale): ones, except
  // between paos_stop_defer_last_power and the pass (or settle_scale) that applies the stop's 1 / sqrt(power)
  cx<double>* ptab = nullptr;  // [ptab_slots][batch][n] phase factors by position: a table per operator slot of the launches of the
  int ptab_slots = 0;          // program about to run (FrugalSlot::table; stage_groups)
  double* dyn_scale = nullptr;
  bool dyn_pending = false;
  // c->norm2 holds sum |u|^2 of the field exactly as it is stored: set by a pass program whose last pass summed it on the
  // way out (paos_run_program, final_intensity = 2), cleared by every entry point that reads or rewrites the field or
  // reduces into c->norm2 (SETTLE_SCALE / DROP_SCALE sit at the top of all of them).  paos_stop_scale_last_power and
  // paos_stop_defer_last_power trust c->norm2 only while it is set and run paos_make_stop otherwise.
  bool norm2_of_field = false;
  // What the PSF buffer (and psf_partial) is known to hold after a pass stored it: for item i the lines along
  // psf_zero_axis outside [psf_zero_lo[i], psf_zero_hi[i]) are zero (their per-workgroup sums too).  The next
  // PSF-storing pass with the same live lines need not write those zeros again; -1 = nothing known.
  int psf_zero_axis = -1;
  std::vector<double> psf_zero_lo, psf_zero_hi;
  void* bounce[2] = {nullptr, nullptr};  // pinned host buffers for device -> pageable host copies
  hipEvent_t bounce_ev[2] = {nullptr, nullptr};
  void* field = nullptr;
  void* tw = nullptr;
  void* staging = nullptr;  // n*n*16 bytes, row-major
  cx<double>* tables = nullptr;  // kMaxTables x batch x 2n separable phase factors
  double* mask = nullptr;        // batch x item_stride aperture weights (allocated on first use)
  double* metric_partial = nullptr;  // psf metrics scratch
  double* metric_out = nullptr;
  double* metric_host = nullptr;     // pinned
  // Line records of apertures riding on frugal passes: a few rendered sets are kept, keyed by everything the
  // renderer reads (the aperture's two parameter block sets, the pass axis, which items share records), so that a
  // chain whose relays repeat one aperture -- and the next wavefront batch through the same optics (a Monte-Carlo
  // study, a benchmark step) -- find their records instead of rendering them again (round 3: 6 renderings of
  // 0.3 ms per SYN20 step -> 0 in steady state).
  struct MaskSet {
    MaskLine* lines = nullptr;  // batch x n
    double* vals = nullptr;     // batch x n x 2 kMaskW partial weights
    std::vector<double> key;    // empty: holds nothing valid
    unsigned long long used = 0;
    int line_lo = 0, line_hi = 0;  // the lines whose records were rendered (the others hold whatever was there before)
  };
  static constexpr int kMaskSets = 8;  // (SYN20: five relay apertures whose pixel radii differ in the last digits + the field stop)
  MaskSet mask_sets[kMaskSets];
  unsigned long long mask_clock = 0;
  unsigned long long mask_hits = 0, mask_rendered = 0;  // paos_record_set_stats
  // experiment (tools/two_streams.py): extra dynamic LDS per pass workgroup of THIS context (PAOS_LDS_PAD when the
  // context is created): 6 KiB make two of its workgroups too big for one CU but leave room for one of another
  // context's -- two contexts then share every CU one workgroup each (measured: -14 %, profiles/r04_membench6_mixed_kinds.txt)
  size_t lds_pad = 0;
  int* mask_overflow = nullptr;    // device counter: partial runs that did not fit (must stay 0)
  double* partial = nullptr;
  double* norm2 = nullptr;
  double* norm2_host = nullptr;  // pinned, kNormSlots x batch
  bool prune = true;  // skip tiles / loads of lines an aperture has zeroed (paos_ctx_set_pruning)
  int norm_slot = 0;
  bool norm_busy[64] = {};       // ticket handed out and not fetched yet (kNormSlots entries)
  int nparts = 0;
  Arena arena;
  std::string err;
  // optional per-kernel-class timing with HIP events on the context's stream
  int prof_kind = -1;
  std::vector<hipEvent_t> prof_events;  // start/stop pairs
  std::vector<int> prof_tags;           // per pair: 1 = the launch skipped dead tiles / loads (pruned)
  int prof_next_tag = 0;
  double prof_next_bytes = 0.0;         // bytes the planner has the next timed launch load + store (its algorithmic bytes)
  std::vector<double> prof_bytes;       // per pair, like prof_tags
  double prof_next_lines = 0.0;         // 1-D line transforms the next timed launch runs (live lines x transforms that are on, over the batch)
  std::vector<double> prof_lines;       // per pair, like prof_tags
  size_t prof_used = 0;
};

namespace {

int fail(paos_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  g_err = msg;
  return code;
}

#define HIPCHK(c, call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail((c), PAOS_EHIP,                                                           \
                  std::string(#call) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" +   \
                      std::to_string(__LINE__) + ")");                                      \
  } while (0)

size_t elem_bytes(const paos_ctx* c) { return c->precision == PAOS_F64 ? 16 : 8; }

// Make room for `total` doubles of pushes that must ALL stay live until the work enqueued with
// them has run (a pass program: its block table plus one record set per pass).  A ring wrap in the
// middle of such a sequence would overwrite parameters that later launches still read, so the
// wrap (one stream synchronisation) or a growth of the arena happens here, before the first push.
// next slab: record the fence of the slab left one switch ago, wait for the readers of the slab about to be refilled
int arena_switch(paos_ctx* c) {
  Arena& a = c->arena;
  if (a.left >= 0) {
    HIPCHK(c, hipEventRecord(a.fence[a.left], c->stream));
    a.fenced[a.left] = true;
  }
  a.left = a.cur;
  const int next = (a.cur + 1) % kArenaSlabs;
  if (a.used[next]) {
    if (a.fenced[next]) HIPCHK(c, hipEventSynchronize(a.fence[next]));  // (recorded two switches ago: normally long done)
    else HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  a.fenced[next] = false;
  a.used[next] = true;
  a.cur = next;
  a.head = 0;
  return PAOS_OK;
}

int arena_reserve(paos_ctx* c, size_t total) {
  Arena& a = c->arena;
  total += 64;  // rounding of the individual pushes
  if (total > a.cap) {  // (grow every slab: one synchronisation, once)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    size_t cap = a.cap;
    while (cap < total) cap *= 2;
    double *h = nullptr, *d = nullptr;
    HIPCHK(c, hipHostMalloc(&h, kArenaSlabs * cap * sizeof(double)));
    if (hipMalloc(&d, kArenaSlabs * cap * sizeof(double)) != hipSuccess) {
      (void)hipHostFree(h);
      return fail(c, PAOS_EHIP, "hipMalloc(arena growth)");
    }
    (void)hipHostFree(a.host);
    (void)hipFree(a.dev);
    a.host = h; a.dev = d; a.cap = cap; a.head = 0; a.cur = 0; a.left = -1;
    for (int k = 0; k < kArenaSlabs; ++k) a.fenced[k] = a.used[k] = false;
    a.used[0] = true;
    return PAOS_OK;
  }
  if (a.head + total > a.cap) return arena_switch(c);
  return PAOS_OK;
}

// copy `count` doubles into the arena; returns the device pointer through *dev
int arena_push(paos_ctx* c, const double* src, size_t count, const double** dev) {
  Arena& a = c->arena;
  if (count > a.cap) return fail(c, PAOS_EINVAL, "parameter block larger than the arena");
  if (a.head + count > a.cap) {
    int rc = arena_switch(c);
    if (rc) return rc;
  }
  double* h = a.host + (size_t)a.cur * a.cap + a.head;
  double* d = a.dev + (size_t)a.cur * a.cap + a.head;
  std::memcpy(h, src, count * sizeof(double));
  HIPCHK(c, hipMemcpyAsync(d, h, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  *dev = d;
  a.head += (count + 15) & ~size_t(15);
  return PAOS_OK;
}

// Device -> caller's (pageable) host buffer.  hipMemcpy into pageable memory pins the target pages
// on the fly: 65-85 ms for a 4 MiB array every time the allocator hands out fresh pages.  Instead
// arrays of up to 4 MiB (grids up to 512^2) cross PCIe into a pinned buffer and are copied out by the CPU (measured:
// run() at 512^2 3.5-4.7 ms every time instead of 4 / 85 ms alternating); larger ones keep the
// runtime's path, which is faster per byte (4096^2 PSFs: 26 vs 20 wavefronts/s).  Synchronises.
constexpr size_t kBounceBytes = size_t(4) << 20;
int copy_to_host(paos_ctx* c, void* host, const void* dev, size_t bytes) {
  if (bytes > kBounceBytes) {  // large arrays: the runtime's own pageable path moves them faster
    HIPCHK(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PAOS_OK;
  }
  for (int i = 0; i < 2; ++i)
    if (!c->bounce[i]) {
      HIPCHK(c, hipHostMalloc(&c->bounce[i], kBounceBytes));
      HIPCHK(c, hipEventCreateWithFlags(&c->bounce_ev[i], hipEventDisableTiming));
    }
  const size_t chunks = (bytes + kBounceBytes - 1) / kBounceBytes;
  auto len = [&](size_t k) { return k + 1 < chunks ? kBounceBytes : bytes - k * kBounceBytes; };
  for (size_t k = 0; k <= chunks; ++k) {
    if (k < chunks) {
      HIPCHK(c, hipMemcpyAsync(c->bounce[k & 1], (const char*)dev + k * kBounceBytes, len(k), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipEventRecord(c->bounce_ev[k & 1], c->stream));
    }
    if (k > 0) {
      HIPCHK(c, hipEventSynchronize(c->bounce_ev[(k - 1) & 1]));
      std::memcpy((char*)host + (k - 1) * kBounceBytes, c->bounce[(k - 1) & 1], len(k - 1));
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PAOS_OK;
}

#if PAOS_PART <= 0
int pw_blocks(const paos_ctx* c) {
  const size_t total = (size_t)c->item_stride;
  size_t b = (total + kPwThreads - 1) / kPwThreads;
  return (int)(b < 2048 ? b : 2048);
}
#endif

// ---- FFT launch geometry per grid size ----------------------------------------------
// Row pass: LINES = BR rows per tile; column pass: LINES = BC columns per tile.
// E elements per thread; TILES tiles per workgroup keep small grids at >= 128 threads.
template <typename T, int N>
struct FftCfg {
  static constexpr int BC = Lay<T>::BC;
  static constexpr int BR = block_rows<T, N>();
  // 16 points per thread.  (32 points of complex64 fill the same 64 data VGPRs, but the
  // radix-32 butterflies spill: 256 VGPRs + 60-70 AGPRs measured, so E stays 16 for both types.)
  static constexpr int E = 16;
  // rows of a block row handled by one row tile: all four, except at N = 4096 where four
  // lines of 4096 points do not fit the register file of a spill-free workgroup
  static constexpr int ROW_LINES = (N >= 2048) ? PAOS_BR / 2 : PAOS_BR;  // 256-thread tiles at 2048 measured +29 %
  // the frugal kernels' row tile: half a block row -- 2 rows of complex128, 4 rows of complex64 at N >= 2048 (1024
  // threads at <= 64 VGPRs at 4096: two workgroups per CU either way)
  static constexpr int FR_ROW_LINES = (N >= 2048) ? BR / 2 : BR;
  static constexpr int COL_LINES = BC;
  static constexpr int ROW_THREADS = ROW_LINES * N / E, COL_THREADS = COL_LINES * N / E;
  static constexpr int ROW_TILES = (ROW_THREADS >= 128) ? 1 : 128 / ROW_THREADS;
  static constexpr int COL_TILES = (COL_THREADS >= 128) ? 1 : 128 / COL_THREADS;
  // Exchange whole complex numbers through LDS (two barriers per exchange) whenever the
  // tile fits; otherwise real and imaginary parts take turns in half the space.  Register
  // pressure (~230 VGPRs) already limits these kernels to 8 waves per CU, so using up to
  // 144 KiB of the 160 KiB LDS costs no occupancy (measured +3..7 %, profiles/r01_fftbench_v4).
  static constexpr bool ROW_SPLIT = (size_t)ROW_LINES * ROW_TILES * line_lds_bytes<T, N, false>() > 144 * 1024;
  static constexpr bool COL_SPLIT = (size_t)COL_LINES * COL_TILES * line_lds_bytes<T, N, false>() > 144 * 1024;
  static constexpr int MINW = 1;
};

// The launch timer (paos_profile_begin): every timed launch site brackets its launch with this pair, so that event
// pair i and tag i always belong to the same launch -- the tag is stored where (and only where) the closing event is.
bool timed_launch_begin(paos_ctx* c, int kind) {
  const bool timed = (c->prof_kind == kind || c->prof_kind == PAOS_KERNEL_PASS_ANY) && (c->prof_used + 2 <= c->prof_events.size());
  // (a failure to record the opening event switches the timing of this launch off; the launch itself goes ahead)
  return timed && hipEventRecord(c->prof_events[c->prof_used], c->stream) == hipSuccess;
}
hipError_t timed_launch_end(paos_ctx* c, int tag) {
  const hipError_t e = hipEventRecord(c->prof_events[c->prof_used + 1], c->stream);
  if (e != hipSuccess) return e;
  c->prof_tags.resize(c->prof_used / 2, 0);  // pair i <-> tag i, whatever happened before
  c->prof_tags.push_back(tag);
  c->prof_bytes.resize(c->prof_tags.size() - 1, 0.0);
  c->prof_bytes.push_back(c->prof_next_bytes);
  c->prof_lines.resize(c->prof_tags.size() - 1, 0.0);
  c->prof_lines.push_back(c->prof_next_lines);
  c->prof_used += 2;
  return hipSuccess;
}

// The power tickets are handed out round the ring, but a caller may keep a ticket for long (a result it reads at the
// end): the next free slot is looked for instead of declaring the ring full at the first busy one.
int next_norm_slot(paos_ctx* c) {
  for (int k = 0; k < kNormSlots; ++k) {
    const int slot = (c->norm_slot + k) % kNormSlots;
    if (!c->norm_busy[slot]) { c->norm_slot = slot; return slot; }
  }
  return c->norm_slot;  // every slot is outstanding: the caller's check of norm_busy[] reports it
}

int opt_in_lds(paos_ctx* c, const void* kern, size_t lds) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> configured;
  std::lock_guard<std::mutex> lock(mu);
  const std::pair<int, const void*> key(c->device, kern);
  if (configured.count(key)) return PAOS_OK;
  HIPCHK(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  configured.insert(key);
  return PAOS_OK;
}

template <typename Kern>
int launch_pass(paos_ctx* c, Kern kern, dim3 grid, dim3 block, size_t lds, const PassArgs& a, int kind) {
  // kernels that need more than the default 64 KiB of dynamic LDS opt in once PER DEVICE (the
  // attribute belongs to the device's copy of the function; one process may drive several GPUs)
  if (lds > 48 * 1024) {
    int rc = opt_in_lds(c, (const void*)kern, lds);
    if (rc) return rc;
  }
  const bool timed = timed_launch_begin(c, kind);
  hipLaunchKernelGGL(kern, grid, block, lds, c->stream, a);
  HIPCHK(c, hipGetLastError());
  if (timed) HIPCHK(c, timed_launch_end(c, 0));
  return PAOS_OK;
}

template <typename T, int N, int AXIS, int FEAT>
int pass_launch(paos_ctx* c, const PassArgs& a) {
  using C = FftCfg<T, N>;
  constexpr int BC = C::BC;
  constexpr int LINES = AXIS == 0 ? C::ROW_LINES : C::COL_LINES;
  constexpr int TILES = AXIS == 0 ? C::ROW_TILES : C::COL_TILES;
  constexpr bool SPLIT = AXIS == 0 ? C::ROW_SPLIT : C::COL_SPLIT;
  const dim3 grid(N / LINES / TILES, a.batch), block(TILES * LINES * N / C::E);  // (a.batch: c->batch, or 1 for the PSD scratch item)
  const size_t lds = (size_t)TILES * LINES * line_lds_bytes<T, N, SPLIT>();
  return launch_pass(c, fused_pass_kernel<T, N, C::E, LINES, TILES, AXIS, C::BR, BC, SPLIT, C::MINW, 1, 0, FEAT>,
                     grid, block, lds, a, AXIS == 0 ? PAOS_KERNEL_PASS_ROWS : PAOS_KERNEL_PASS_COLS);
}

// feat != 0: the build with the optional operators (phase tables, aperture weight maps)
template <typename T, int N>
int pass_n(paos_ctx* c, int axis, const PassArgs& a, int feat) {
  if (feat) return axis == 0 ? pass_launch<T, N, 0, 3>(c, a) : pass_launch<T, N, 1, 3>(c, a);
  return axis == 0 ? pass_launch<T, N, 0, 0>(c, a) : pass_launch<T, N, 1, 0>(c, a);
}

template <typename T>
int pass_t(paos_ctx* c, int axis, const PassArgs& a, int feat) {
  switch (c->n) {
    case 64: return pass_n<T, 64>(c, axis, a, feat);
    case 128: return pass_n<T, 128>(c, axis, a, feat);
    case 256: return pass_n<T, 256>(c, axis, a, feat);
    case 512: return pass_n<T, 512>(c, axis, a, feat);
    case 1024: return pass_n<T, 1024>(c, axis, a, feat);
    case 2048: return pass_n<T, 2048>(c, axis, a, feat);
    case 4096: return pass_n<T, 4096>(c, axis, a, feat);
  }
  return fail(c, PAOS_EUNSUPPORTED, "grid size must be a power of two in 64..4096");
}

int check_ops(paos_ctx* c, const paos_pw_op* ops, int count, int n_blocks) {
  if (count < 0 || count > PAOS_MAX_PW) return fail(c, PAOS_EINVAL, "too many pointwise operators in a pass");
  for (int o = 0; o < count; ++o) {
    if (ops[o].kind < PAOS_PW_SIGN || ops[o].kind > PAOS_PW_MASK) return fail(c, PAOS_EINVAL, "unknown pointwise operator");
    if (ops[o].block < 0 || ops[o].block >= n_blocks) return fail(c, PAOS_EINVAL, "operator block index out of range");
    if (ops[o].kind == PAOS_PW_MASK && ops[o].block + 1 >= n_blocks) return fail(c, PAOS_EINVAL, "an aperture operator needs two parameter blocks");
  }
  return PAOS_OK;
}

// Separable phase tables with an exact rounding correction (fft_kernels.h: table_phase) replace
// the per-pixel sincos by ~25 fp64 instructions + one L2-resident 16-byte load.  Measured on
// MI355X (round 1) the extra loads cost more than the arithmetic they save at 2 waves/SIMD
// (90 vs 97 wavefronts/s), so the path is opt-in: PAOS_PHASE_TABLES=1.
bool use_tables() {
  static const bool on = [] { const char* e = getenv("PAOS_PHASE_TABLES"); return e && e[0] == '1'; }();
  return on;
}

// ---- frugal path (N >= 1024, complex128): compile-time pass shapes, <= 128 VGPRs ---------------
// PAOS_NO_FRUGAL=1 keeps every pass on the generic kernel (A/B tests).
bool use_frugal() {
  static const bool on = [] { const char* e = getenv("PAOS_NO_FRUGAL"); return !(e && e[0] == '1'); }();
  return on;
}

// Express pass p as   load | sign*scale*K phases | FFT | sign*scale*K phases | [FFT] | store.
bool lower_frugal(const paos_ctx* c, const paos_pass& p, const double* blocks /*host*/, std::vector<FrugalItem>& items,
                  int& kpre, int& kmid, int& nfft, int& mask_block, int& mask_slot, std::vector<double>& mask_shared,
                  std::vector<int>& mask_rep) {
  if (p.axis != 0 && p.axis != 1) return false;
  if (p.fft1 < 0 || p.n_post != 0) return false;
  const paos_pw_op* lists[2] = {p.pre, p.mid};
  const int counts[2] = {p.n_pre, p.n_mid};
  int k[2] = {0, 0};
  mask_block = -1; mask_slot = -1;
  for (int l = 0; l < 2; ++l)
    for (int o = 0; o < counts[l]; ++o) {
      const int kind = lists[l][o].kind;
      if (kind == PAOS_PW_QPHASE_CENTRED || kind == PAOS_PW_QPHASE_NATURAL) {
        ++k[l];
        for (int it = 0; it < c->batch; ++it) {  // the kernels fold the sign into the coefficient: it must be +-1
          const double* q = blocks + ((size_t)lists[l][o].block * c->batch + it) * FP_STRIDE;
          if (q[FP_ENABLE] != 0.0 && std::fabs(q[FP_SGN]) != 1.0) return false;
        }
      }
      else if (kind == PAOS_PW_MASK) {
        if (mask_block >= 0) return false;  // one aperture per pass (one set of line records)
        mask_block = lists[l][o].block; mask_slot = l;
      } else if (kind != PAOS_PW_SIGN && kind != PAOS_PW_SCALE) return false;
    }
  if (k[0] > kFrugalMaxPre || k[1] > kFrugalMaxMid) return false;
  if (mask_block >= 0) {  // can this aperture be held as per-line records along the pass axis?
    for (int it = 0; it < c->batch; ++it) {
      const double* q = blocks + ((size_t)mask_block * c->batch + it) * FP_STRIDE;
      const double* q2 = blocks + ((size_t)(mask_block + 1) * c->batch + it) * FP_STRIDE;
      if (q[0] == 0.0) continue;
      const double a = q[3], b = q[4], theta = q2[0], obsc = q2[1], subpix = q2[2], shape = q2[3];
      if (theta != 0.0 || !(a > 0.0) || !(b > 0.0)) return false;
      if (shape == PAOS_SHAPE_ELLIPSE) {
        // longest partial run near the tips of the ellipse: ~ 2 a sqrt(3 / b) along rows
        const double along = p.axis == 0 ? a : b, across = p.axis == 0 ? b : a;
        if (!(across >= 2.0) || 2.0 * along * std::sqrt(3.0 / across) + 8.0 > kMaskW) return false;
      } else {
        const int sp = (int)subpix;
        if (obsc != 0.0 || sp <= 0 || (sp & (sp - 1)) != 0) return false;
      }
    }
  }
  kpre = k[0]; kmid = k[1]; nfft = p.fft2 >= 0 ? 2 : 1;
  items.assign(c->batch, FrugalItem{});
  mask_shared.assign(c->batch, 0.0);
  mask_rep.assign(c->batch, -1);
  auto blk = [&](int b, int it) { return blocks + ((size_t)b * c->batch + it) * FP_STRIDE; };
  for (int it = 0; it < c->batch; ++it) {
    FrugalItem& fi = items[it];
    const double* c1 = blk(p.fft1, it);
    fi.fft1_on = c1[FC_ENABLE] != 0.0; fi.fft1_inv = c1[FC_INVERSE] != 0.0;
    if (p.fft2 >= 0) { const double* c2 = blk(p.fft2, it); fi.fft2_on = c2[FC_ENABLE] != 0.0; fi.fft2_inv = c2[FC_INVERSE] != 0.0; }
    bool active = fi.fft1_on != 0.0 || fi.fft2_on != 0.0;
    FrugalSlot* slots[2] = {&fi.pre, &fi.mid};
    FrugalPhase* phases[2] = {fi.pre_ph, fi.mid_ph};
    int sign_bits[2] = {0, 0};
    for (int l = 0; l < 2; ++l) {
      slots[l]->sign_on = 0.0; slots[l]->scale = 1.0;
      sign_bits[l] = 0;
      slots[l]->mask_on = 0.0; slots[l]->w_in = 1.0; slots[l]->w_out = 0.0;
      slots[l]->lines = nullptr; slots[l]->vals = nullptr;
      int j = 0;
      for (int o = 0; o < counts[l]; ++o) {
        const paos_pw_op& op = lists[l][o];
        const double* q = blk(op.block, it);
        const bool on = q[FP_ENABLE] != 0.0;
        active = active || on;
        if (op.kind == PAOS_PW_MASK) {
          const double* q2 = blk(op.block + 1, it);
          slots[l]->mask_on = on ? 1.0 : 0.0;
          const bool obsc = q2[1] != 0.0 && q2[3] == PAOS_SHAPE_ELLIPSE;
          slots[l]->w_in = obsc ? 0.0 : 1.0; slots[l]->w_out = obsc ? 1.0 : 0.0;
          // items with the same aperture on the same sampling (a Monte-Carlo batch: one wavelength, many
          // wavefront-error draws) share one set of line records: only the first of them is rendered
          int rep = it;
          for (int j = 0; j < it; ++j)
            if (!std::memcmp(blk(op.block, j), q, FP_STRIDE * sizeof(double)) &&
                !std::memcmp(blk(op.block + 1, j), q2, FP_STRIDE * sizeof(double))) { rep = j; break; }
          mask_shared[it] = rep != it ? 1.0 : 0.0;
          mask_rep[it] = rep;  // the record set is chosen later (assign_mask_set): pointers are filled in there
        } else if (op.kind == PAOS_PW_SIGN) {
          // bit 0: (-1)^position along the line, bit 1: (-1)^line; the checkerboard flips both
          if (on) {
            const bool x_only = op.flags & PAOS_PWF_X_ONLY, y_only = op.flags & PAOS_PWF_Y_ONLY;
            const int along = p.axis == 0 ? (y_only ? 0 : 1) : (x_only ? 0 : 1);
            const int across = p.axis == 0 ? (x_only ? 0 : 1) : (y_only ? 0 : 1);
            sign_bits[l] ^= along | (across << 1);
          }
        }
        else if (op.kind == PAOS_PW_SCALE) { if (on) slots[l]->scale *= q[FP_COEF]; }
        else {
          FrugalPhase& ph = phases[l][j++];
          ph.natural = op.kind == PAOS_PW_QPHASE_NATURAL ? 1.0 : 0.0;
          // the sign rides on the coefficient: exp(i sgn m2 fl(coef s)) = exp(i m2 fl((sgn coef) s)), sgn = +-1
          if (on) { ph.sx = q[FP_SX]; ph.sy = q[FP_SY]; ph.coef = q[FP_COEF] * q[FP_SGN]; ph.sgn = 1.0; ph.m2 = (op.flags & PAOS_PWF_MUL2PI) ? 6.283185307179586 : 1.0; }
          else { ph.sx = ph.sy = 0.0; ph.coef = 0.0; ph.sgn = 1.0; ph.m2 = 1.0; }  // exp(i 0) = 1 exactly
        }
      }
      // frugal_slot: 1 = (-1)^(line + position), 2 = (-1)^position, 3 = (-1)^line
      slots[l]->sign_on = sign_bits[l] == 3 ? 1.0 : (sign_bits[l] == 1 ? 2.0 : (sign_bits[l] == 2 ? 3.0 : 0.0));
    }
    fi.active = active ? 1.0 : 0.0;
    fi.line_lo = 0.0; fi.line_hi = (double)c->n; fi.line_fill = 0.0; fi.pos_lo = 0.0; fi.pos_hi = (double)c->n;
    fi.spos_lo = 0.0; fi.spos_hi = (double)c->n;
  }
  // The KPRE = 0 shapes take the slot in front of the first transform to be empty (frugal_slot: PLAIN).  The
  // rare pass that has a sign, a scale or an aperture there but no phase runs on the KPRE = 1 shape; its
  // phase record is all zeros: exp(i 0) = 1 exactly.
  if (kpre == 0)
    for (const FrugalItem& fi : items)
      if (fi.active != 0.0 && (fi.pre.sign_on != 0.0 || fi.pre.scale != 1.0 || fi.pre.mask_on != 0.0)) { kpre = 1; break; }
  return true;
}

// ---- pruning of dead lines (frugal_pass.h: FrugalItem::line_lo ...) ------------------------------
// One pass of a program, lowered for the frugal kernels before anything is launched, so that the
// planner below can look ahead.
struct LoweredPass {
  bool ok = false;
  std::vector<FrugalItem> items;
  int kpre = 0, kmid = 0, nfft = 1, mask_block = -1, mask_slot = -1;
  std::vector<double> mask_shared;  // [batch] 1: the item reads the line records of an earlier, identical item
  std::vector<int> mask_rep;        // [batch] item whose records this item reads (-1: none)
  int mask_set = -1;                // which of the context's record sets this pass reads
  int mask_shapes = 3;              // bit s: some item's aperture has shape s (0 ellipse, 1 rectangle) and renders its own records
  int mask_lo = 0, mask_hi = 0;     // lines whose records this pass reads (run_passes_impl, behind the pruning plan)
  bool mask_render = false;         // ... and whether it has to be rendered first
};

// Lines (rows for a row pass, columns for a column pass) outside the returned range get weight
// exactly 0 from the aperture of this pass: photutils' bounding box, ixmin = floor(c - e + 0.5),
// ixmax = ceil(c + e + 0.5) (pointwise.h: make_box; theta = 0 here), widened by one pixel and then
// rounded outward to whole block rows so that a tile is either wholly dead or processed.
bool mask_live_range(const paos_ctx* c, const double* q, const double* q2, int axis, int* lo, int* hi) {
  if (q[0] == 0.0 || q2[0] != 0.0 || q2[1] != 0.0) return false;  // off, tilted, or an obscuration (outside weight 1)
  const double centre = axis == 0 ? q[2] : q[1];
  double ext = axis == 0 ? q[4] : q[3];
  if (q2[3] != PAOS_SHAPE_ELLIPSE) ext = ext / 2.0;
  if (!std::isfinite(centre) || !std::isfinite(ext) || !(ext > 0.0)) return false;
  const double a = std::floor(centre - ext + 0.5) - 1.0, b = std::ceil(centre + ext + 0.5) + 1.0;
  const int n = c->n;
  int l = a < 0.0 ? 0 : (a > n ? n : (int)a), h = b < 0.0 ? 0 : (b > n ? n : (int)b);
  const int br = c->br;
  l = (l / br) * br;
  h = ((h + br - 1) / br) * br;
  if (h > n) h = n;
  if (l >= h) { l = 0; h = br; }  // aperture off the grid along this axis: keep one block row live
  *lo = l; *hi = h;
  return true;
}

// Fill in the pruning fields of a whole program.  Per item the planner carries, forwards, the BOX outside which the
// field is known to be zero -- rows [r.lo, r.hi) x columns [c.lo, c.hi); physically (zeros in memory: what a stand-alone
// aperture leaves, passed in through entry_rows) or virtually (tiles that were skipped hold stale data that STANDS for
// zeros) -- and, backwards, the box of each pass's output that the next pass reads at all:
//   forwards   a pass keeps dead lines dead; its transforms spread the live positions over the whole line; an aperture
//              riding on it clips both ranges to its bounding box (positions: unless a transform follows it);
//   backwards  a pass processes the lines that are alive AND wanted, loads the live positions of those lines and stores
//              the positions the next pass reads; what it reads is what the pass in front of it has to deliver.
// Every load therefore falls inside what the previous pass stored (or is known to be zero and not loaded), and nothing
// else is ever looked at: tiles nobody processes keep whatever they held.  The last pass an item takes part in delivers
// the whole grid: it stores every position of its lines and writes zeros to the dead ones (line_fill).
// Round 4: the box (both axes at once, and the backward half) is what lets the separable pass programs (passes.py:
// SeparableCompiler) run an aperture-to-aperture stretch on the live rows and the wanted columns only; for the
// operator-by-operator programs it yields the ranges of round 2's one-axis planner and round 3's "stores nobody reads".
struct LineRange { int lo, hi; };
void plan_pruning(const paos_ctx* c, const paos_pass* passes, int n_passes, const double* blocks,
                  std::vector<LoweredPass>& low, const double* entry_rows, bool entry_stale, const double* entry_cols = nullptr) {
  const int n = c->n, br = c->br;
  // never empty, always inside `a`: an aperture off the live range keeps one block row of `a` (which it then zeroes)
  auto meet = [br](LineRange a, LineRange b) {
    LineRange r{a.lo > b.lo ? a.lo : b.lo, a.hi < b.hi ? a.hi : b.hi};
    if (r.lo >= r.hi) { r.lo = a.lo; r.hi = a.lo + br < a.hi ? a.lo + br : a.hi; }
    return r;
  };
  std::vector<int> act;
  std::vector<LineRange> lines, loads;
  for (int it = 0; it < c->batch; ++it) {
    act.clear();
    for (int q = 0; q < n_passes; ++q)
      if (low[q].items[it].active != 0.0) act.push_back(q);
    if (act.empty()) continue;
    LineRange box[2] = {{0, n}, {0, n}};  // [0]: rows, [1]: columns
    bool clean = false;                   // rows outside box[0] are zeros in memory and nothing has touched them
    LineRange rows0{0, n};
    if (entry_rows) {
      int l = (int)entry_rows[2 * it], h = (int)entry_rows[2 * it + 1];
      l = l < 0 ? 0 : (l / br) * br;
      h = h > n ? n : ((h + br - 1) / br) * br;
      if (h > n) h = n;
      if (l < h && (l > 0 || h < n)) { box[0] = {l, h}; clean = !entry_stale; }
      rows0 = box[0];
    }
    if (entry_cols && entry_stale) {
      // (round 5: paos_start_box) the columns outside stand for zeros too, inside the live rows: the first pass loads the box
      // only, and every later pass reads what its predecessor stored -- nobody ever looks at them
      int l = (int)entry_cols[2 * it], h = (int)entry_cols[2 * it + 1];
      l = l < 0 ? 0 : (l / br) * br;
      h = h > n ? n : ((h + br - 1) / br) * br;
      if (h > n) h = n;
      if (l < h && (l > 0 || h < n)) box[1] = {l, h};
    }
    // forwards
    lines.assign(act.size(), LineRange{0, n});
    loads.assign(act.size(), LineRange{0, n});
    for (size_t k = 0; k < act.size(); ++k) {
      const int q = act[k], ax = passes[q].axis;
      const FrugalItem& fi = low[q].items[it];
      LineRange& L = box[ax];      // along the lines of this pass (rows for a row pass)
      LineRange& P = box[1 - ax];  // along the positions of a line
      LineRange ml{0, n}, mp{0, n};
      bool masked = false;
      if (low[q].mask_block >= 0) {
        const double* mq = blocks + ((size_t)low[q].mask_block * c->batch + it) * FP_STRIDE;
        const double* mq2 = blocks + ((size_t)(low[q].mask_block + 1) * c->batch + it) * FP_STRIDE;
        masked = mask_live_range(c, mq, mq2, ax, &ml.lo, &ml.hi) && mask_live_range(c, mq, mq2, 1 - ax, &mp.lo, &mp.hi);
      }
      if (masked) L = meet(L, ml);
      lines[k] = L;
      LineRange pos = P;
      if (masked && low[q].mask_slot == 0) pos = meet(pos, mp);  // in front of the first transform: no need to load what it zeroes
      loads[k] = pos;
      if (fi.fft1_on != 0.0) pos = {0, n};
      if (masked && low[q].mask_slot == 1) pos = meet(pos, mp);
      if (low[q].nfft >= 2 && fi.fft2_on != 0.0) pos = {0, n};
      P = pos;
    }
    // backwards
    LineRange want[2] = {{0, n}, {0, n}};
    for (size_t k = act.size(); k-- > 0;) {
      const int q = act[k], ax = passes[q].axis;
      FrugalItem& fi = low[q].items[it];
      const LineRange proc = meet(lines[k], want[ax]);
      fi.line_lo = proc.lo; fi.line_hi = proc.hi;
      fi.pos_lo = loads[k].lo; fi.pos_hi = loads[k].hi;
      fi.spos_lo = want[1 - ax].lo; fi.spos_hi = want[1 - ax].hi;
      want[ax] = proc;
      want[1 - ax] = loads[k];
    }
    // zeros nobody has written: the last pass writes them
    for (size_t k = 0; k < act.size(); ++k) {
      const FrugalItem& fi = low[act[k]].items[it];
      if (passes[act[k]].axis != 0 || (int)fi.line_lo != rows0.lo || (int)fi.line_hi != rows0.hi || fi.spos_lo > 0.0 ||
          fi.spos_hi < (double)n)
        clean = false;
    }
    FrugalItem& last = low[act.back()].items[it];
    if ((last.line_lo > 0.0 || last.line_hi < (double)n) && !clean) last.line_fill = 1.0;
  }
}

#ifndef PAOS_LONG_ONE_LINE
#define PAOS_LONG_ONE_LINE 1
#endif
#ifndef PAOS_SINGLE_ONE_LINE
#define PAOS_SINGLE_ONE_LINE 1   // 0 (A/B builds): single table passes keep the two-line workgroups whatever they load and store
#endif
template <typename T, int N, int AXIS, int KPRE, int KMID, int NFFT, int STORE = 0, int TAB = 0, int LONG = 0, int ONE = 0>
int frugal_launch(paos_ctx* c, const FrugalArgs& args) {
  using C = FftCfg<T, N>;
  // Round 5: the launches that run two or three passes of a chain (LONG builds) and store the field are bound by the latency
  // chain of a workgroup -- exchanges, barriers, table reads -- not by bytes (they move a sixteenth of the grid): at 4096^2
  // complex128 they run on ONE-line workgroups of 256 threads, four per CU with one wave each per SIMD instead of two of
  // 512 threads (-6 ... -8 % rows, -2 ... -3 % columns, bit-identical: profiles/r05_fftbench_fused_variants.txt).  The 16- /
  // 32-byte pieces such tiles take out of every 128-byte block, which rule them out for byte-bound passes, cost nothing
  // here.  (The PSF- / power-summing builds keep the two-line tiles: their partial sums are laid out per two-line tile.)
  // ONE = 1: a single table pass that loads AND stores at most half of its positions (the two passes of the first stretch since
  // the start box) is as latency-bound as the fused launches and takes the same shape (launch_lowered decides per launch).
  constexpr bool kOneLine = PAOS_LONG_ONE_LINE != 0 && (LONG != 0 || ONE != 0) && STORE == 0 && sizeof(T) == 8 && N == 4096;
  // ... and at 2048^2, where a line is 128 threads and the workgroup already two lines of them: four workgroups per CU instead
  // of three (frugal_pass.h: OCC)
  // (1024^2: measured too -- four rows per workgroup keep four twiddles per thread in registers and the shapes spill 30-100 B:
  // fused two-pass launches 0.866 -> 0.878 ms, three-pass 1.22 -> 1.31: stays on three workgroups per CU.  profiles/r05_ab_variants_bench.txt)
  constexpr int kOcc = (PAOS_LONG_ONE_LINE != 0 && (LONG != 0 || ONE != 0) && STORE == 0 && sizeof(T) == 8 && N == 2048) ? 1 : 0;
  constexpr int LINES = kOneLine ? 1 : (AXIS == 0 ? C::FR_ROW_LINES : C::COL_LINES);
  constexpr int TILES = AXIS == 0 ? C::ROW_TILES : C::COL_TILES;
  // several workgroups share the 160 KiB of LDS: c128 exchanges re and im in turn; a c64 line
  // fits whole (the same 35 KiB) and so needs half the barriers -- except in the 4-line row tiles
  constexpr bool SPLIT = sizeof(T) == 8 || LINES > 2;
  FrugalArgs a = args;
  unsigned groups = N / LINES / TILES;
  a.wg0 = 0;
  // TileMap renumbers the tiles that share 128-byte lines inside aligned groups of workgroups (siblings 8 apart: one XCD):
  // 16 for half-block row tiles and whole-block column tiles, 32 for the quarter-block row tiles of the one-line builds
  constexpr unsigned kAlign = (kOneLine && AXIS == 0) ? 32 : 16;
  static_assert((N / LINES / TILES) % kAlign == 0, "TileMap renumbers tiles inside aligned groups of workgroups");
  if (a.live_hi > a.live_lo) {  // launch the workgroups of live lines only, in whole groups
    const unsigned per = LINES * TILES;
    a.wg0 = (a.live_lo / per) / kAlign * kAlign;
    unsigned end = ((a.live_hi + per - 1) / per + kAlign - 1) / kAlign * kAlign;
    if (end > groups) end = groups;
    groups = end - a.wg0;
  }
  const dim3 grid(groups, c->batch), block(TILES * LINES * N / C::E);
  constexpr size_t kMaxPad = 8192;
  const size_t lds = frugal_lds_bytes<T, N, LINES, TILES, SPLIT, KPRE, KMID, C::E, STORE, kOcc>() + (c->lds_pad < kMaxPad ? c->lds_pad : kMaxPad);
  auto kern = frugal_pass_kernel<T, N, C::E, LINES, TILES, AXIS, C::BR, C::BC, SPLIT, KPRE, KMID, NFFT, STORE, TAB, LONG, kOcc>;
  {
    int rc = opt_in_lds(c, (const void*)kern, frugal_lds_bytes<T, N, LINES, TILES, SPLIT, KPRE, KMID, C::E, STORE, kOcc>() + kMaxPad);
    if (rc) return rc;
  }
  const int kind = AXIS == 0 ? PAOS_KERNEL_PASS_ROWS : PAOS_KERNEL_PASS_COLS;
  const bool timed = timed_launch_begin(c, kind);
  hipLaunchKernelGGL(kern, grid, block, lds, c->stream, PAOS_FRUGAL_PASS(a));
  HIPCHK(c, hipGetLastError());
  if (timed) HIPCHK(c, timed_launch_end(c, c->prof_next_tag));
  return PAOS_OK;
}

// workgroups per batch item of a frugal pass along `axis` (the partial sums of a PSF-storing pass)
template <typename T, int N>
int frugal_groups(int axis) {
  using C = FftCfg<T, N>;
  return axis == 0 ? N / C::ROW_LINES / C::ROW_TILES : N / C::COL_LINES / C::COL_TILES;
}

template <typename T, int N, int AXIS, int KPRE, int KMID>
int frugal_nfft(paos_ctx* c, const FrugalArgs& a, int nfft) {
  // (the digit-swapped two-transform variant NFFT = 3 of frugal_pass.h is built by tools/fftbench.hip only:
  // measured in round 2 with parity unchanged and no gain, profiles/r02_fftbench_digit_swapped_experiment.txt)
  if (a.tab && a.fuse) {  // ... and the launch runs the next pass -- or the next two -- of the program as well (LONG builds)
    if constexpr (KPRE == 1 && KMID == 1) {
      if (nfft < 2) return fail(c, PAOS_EINVAL, "a fused chain starts with a two-transform pass");
#define PAOS_LONG_CASE(L)                                                           \
  case L:                                                                           \
    if (a.psf) return frugal_launch<T, N, AXIS, 1, 1, 2, 1, 1, L>(c, a);            \
    if (a.pow_partial) return frugal_launch<T, N, AXIS, 1, 1, 2, 2, 1, L>(c, a);    \
    return frugal_launch<T, N, AXIS, 1, 1, 2, 0, 1, L>(c, a);
      switch (a.fuse) {
        PAOS_LONG_CASE(1)
        PAOS_LONG_CASE(2)
        PAOS_LONG_CASE(3)
        PAOS_LONG_CASE(4)
      }
#undef PAOS_LONG_CASE
      return fail(c, PAOS_EINVAL, "a launch runs at most three passes");
    } else {
      return fail(c, PAOS_EINVAL, "no fused build of this pass shape");
    }
  }
  if (a.tab) {  // the slots read their factors from tables: one build for any number of phases per slot
    if constexpr (KPRE <= 1 && KMID <= 1 && KPRE + KMID > 0) {
      if (a.psf) return nfft >= 2 ? frugal_launch<T, N, AXIS, KPRE, KMID, 2, 1, 1>(c, a) : frugal_launch<T, N, AXIS, KPRE, KMID, 1, 1, 1>(c, a);
      if (a.pow_partial) return nfft >= 2 ? frugal_launch<T, N, AXIS, KPRE, KMID, 2, 2, 1>(c, a) : frugal_launch<T, N, AXIS, KPRE, KMID, 1, 2, 1>(c, a);
      if constexpr (sizeof(T) == 8 && (N == 4096 || N == 2048) && PAOS_LONG_ONE_LINE != 0) {
        if (a.one_line && PAOS_SINGLE_ONE_LINE != 0) return nfft >= 2 ? frugal_launch<T, N, AXIS, KPRE, KMID, 2, 0, 1, 0, 1>(c, a) : frugal_launch<T, N, AXIS, KPRE, KMID, 1, 0, 1, 0, 1>(c, a);
      }
      return nfft >= 2 ? frugal_launch<T, N, AXIS, KPRE, KMID, 2, 0, 1>(c, a) : frugal_launch<T, N, AXIS, KPRE, KMID, 1, 0, 1>(c, a);
    } else {
      return fail(c, PAOS_EINVAL, "no table build of this pass shape");
    }
  }
  if constexpr (KPRE <= 1 && KMID <= 1) {  // the shapes a chain can end on: also built with the PSF store (KPRE = 1: round 4,
    // the last column pass of a separable program usually has the column half of a phase in front of its first transform)
    if (a.psf) return nfft >= 2 ? frugal_launch<T, N, AXIS, KPRE, KMID, 2, 1>(c, a) : frugal_launch<T, N, AXIS, KPRE, KMID, 1, 1>(c, a);
  } else {
    if (a.psf) return fail(c, PAOS_EUNSUPPORTED, "no PSF-storing build of this pass shape");
  }
  if constexpr (KPRE <= 1) {  // ... and the shapes a program that ends on a saved surface ends with: field + its power
    if (a.pow_partial) return nfft >= 2 ? frugal_launch<T, N, AXIS, KPRE, KMID, 2, 2>(c, a) : frugal_launch<T, N, AXIS, KPRE, KMID, 1, 2>(c, a);
  } else {
    if (a.pow_partial) return fail(c, PAOS_EUNSUPPORTED, "no power-summing build of this pass shape");
  }
  return nfft >= 2 ? frugal_launch<T, N, AXIS, KPRE, KMID, 2>(c, a) : frugal_launch<T, N, AXIS, KPRE, KMID, 1>(c, a);
}
template <typename T, int N, int AXIS, int KPRE>
int frugal_kmid(paos_ctx* c, const FrugalArgs& a, int kmid, int nfft) {
  switch (kmid) {
    case 0: return frugal_nfft<T, N, AXIS, KPRE, 0>(c, a, nfft);
    case 1: return frugal_nfft<T, N, AXIS, KPRE, 1>(c, a, nfft);
    case 2: return frugal_nfft<T, N, AXIS, KPRE, 2>(c, a, nfft);
    default: return frugal_nfft<T, N, AXIS, KPRE, 3>(c, a, nfft);
  }
}
template <typename T, int N, int AXIS>
int frugal_kpre(paos_ctx* c, const FrugalArgs& a, int kpre, int kmid, int nfft) {
  switch (kpre) {
    case 0: return frugal_kmid<T, N, AXIS, 0>(c, a, kmid, nfft);
    case 1: return frugal_kmid<T, N, AXIS, 1>(c, a, kmid, nfft);
    default: return frugal_kmid<T, N, AXIS, 2>(c, a, kmid, nfft);
  }
}
template <typename T, int N>
int frugal_axis(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft) {
  return axis == 0 ? frugal_kpre<T, N, 0>(c, a, kpre, kmid, nfft) : frugal_kpre<T, N, 1>(c, a, kpre, kmid, nfft);
}

}  // namespace

// The 48 pass-kernel shapes of one (type, N) family behind one ordinary function, so that the
// library can be compiled as six translation units in parallel (PAOS_PART, see the top of the
// file): -1 = everything here, 0 = all but these families, 1..5 = one family each.
#define PAOS_HIDDEN __attribute__((visibility("hidden")))
PAOS_HIDDEN int paos_frugal_d1024(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft);
PAOS_HIDDEN int paos_frugal_d2048(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft);
PAOS_HIDDEN int paos_frugal_d4096(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft);
PAOS_HIDDEN int paos_frugal_f2048(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft);
PAOS_HIDDEN int paos_frugal_f4096(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft);
#if PAOS_PART < 0 || PAOS_PART == 1
int paos_frugal_d1024(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft) {
  return frugal_axis<double, 1024>(c, a, axis, kpre, kmid, nfft);
}
#endif
#if PAOS_PART < 0 || PAOS_PART == 2
int paos_frugal_d2048(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft) {
  return frugal_axis<double, 2048>(c, a, axis, kpre, kmid, nfft);
}
#endif
#if PAOS_PART < 0 || PAOS_PART == 3
int paos_frugal_d4096(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft) {
  return frugal_axis<double, 4096>(c, a, axis, kpre, kmid, nfft);
}
#endif
#if PAOS_PART < 0 || PAOS_PART == 4
int paos_frugal_f2048(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft) {
  return frugal_axis<float, 2048>(c, a, axis, kpre, kmid, nfft);
}
#endif
#if PAOS_PART < 0 || PAOS_PART == 5
int paos_frugal_f4096(paos_ctx* c, const FrugalArgs& a, int axis, int kpre, int kmid, int nfft) {
  return frugal_axis<float, 4096>(c, a, axis, kpre, kmid, nfft);
}
#endif

#if PAOS_PART <= 0  // ---- everything below belongs to the main translation unit ----------------
namespace {

// ---- a stop whose scaling is left to the next pass (paos_stop_defer_last_power) ---------------------------------------
__global__ void dyn_scale_set_kernel(double* dyn, const double* norm2, const double* enable, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) dyn[i] = (!enable || enable[i] != 0.0) ? 1.0 / sqrt(norm2[i]) : 1.0;  // stop_scale_kernel's own expression
}
__global__ void dyn_scale_reset_kernel(double* dyn, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) dyn[i] = 1.0;
}
template <typename T>
__global__ void scale_by_kernel(cx<T>* field, const double* dyn, unsigned item_stride) {
  const int item = blockIdx.y;
  const double s = dyn[item];
  if (s == 1.0) return;
  cx<T>* f = field + (size_t)item * item_stride;
  for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < item_stride; m += (size_t)gridDim.x * blockDim.x)
    f[m] = {(T)__dmul_rn((double)f[m].x, s), (T)__dmul_rn((double)f[m].y, s)};
}
// Whatever is about to read or rewrite the field other than a pass program that can take the factor along: the stop's
// scaling is applied now, by the sweep make_stop would have run (same factor, same products: bit-identical).
int settle_scale(paos_ctx* c, bool field_is_overwritten = false) {
  if (!c->dyn_pending) return PAOS_OK;
  if (!field_is_overwritten) {
    const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
    if (c->precision == PAOS_F64)
      hipLaunchKernelGGL(scale_by_kernel<double>, grid, block, 0, c->stream, (cx<double>*)c->field, c->dyn_scale, c->item_stride);
    else
      hipLaunchKernelGGL(scale_by_kernel<float>, grid, block, 0, c->stream, (cx<float>*)c->field, c->dyn_scale, c->item_stride);
  }
  hipLaunchKernelGGL(dyn_scale_reset_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream, c->dyn_scale, c->batch);
  HIPCHK(c, hipGetLastError());
  c->dyn_pending = false;
  return PAOS_OK;
}
#define SETTLE_SCALE(c) do { if (c) { (c)->norm2_of_field = false; if ((c)->dyn_pending) { int rc_ = settle_scale(c); if (rc_) return rc_; } } } while (0)
#define DROP_SCALE(c) do { if (c) { (c)->norm2_of_field = false; if ((c)->dyn_pending) { int rc_ = settle_scale(c, true); if (rc_) return rc_; } } } while (0)


bool frugal_sizes(const paos_ctx* c) {
  // instantiated for complex128 at N >= 1024 and complex64 at N >= 2048
  return use_frugal() && c->n >= (c->precision == PAOS_F64 ? 1024 : 2048);
}

int ensure_mask_store(paos_ctx* c) {
  if (!c->mask_overflow) {
    HIPCHK(c, hipMalloc(&c->mask_overflow, sizeof(int)));
    HIPCHK(c, hipMemsetAsync(c->mask_overflow, 0, sizeof(int), c->stream));
  }
  return PAOS_OK;
}

// Pick the record set pass `lp` reads -- one that already holds its aperture, or the least recently used one, to
// be rendered -- and point the items' slots at it.  Called for the passes of a program in order, before anything is
// launched: launches follow in the same order on one stream, so a set re-used further down the program is
// overwritten only after the pass that read it.
// How ellipse records are rendered (read per call: tests switch it).  PAOS_MASK_SCAN=1: the chunk scan of rounds 2-3 (0);
// PAOS_MASK_PAIRS=0: one line per wave through two 32-pixel windows (1), =1: two lines per wave through 16-pixel windows
// where they fit (2); default: four lines per wave through 8-pixel windows where they fit (3).
int mask_render_mode() {
  const char* e = getenv("PAOS_MASK_SCAN");
  if (e && e[0] == '1') return 0;
  const char* pe = getenv("PAOS_MASK_PAIRS");
  return (pe && pe[0] == '0') ? 1 : ((pe && pe[0] == '1') ? 2 : 3);
}
// ... and rectangle records: 64 lines per wave from one column profile (round 5); PAOS_MASK_RECT_BLOCKS=0: one line per wave
bool mask_rect_blocks() {
  const char* e = getenv("PAOS_MASK_RECT_BLOCKS");
  return !(e && e[0] == '0');
}

int assign_mask_set(paos_ctx* c, const paos_pass& p, LoweredPass& lp, const double* blocks) {
  std::vector<double> key;
  key.reserve((size_t)2 * c->batch * FP_STRIDE + c->batch + 3);
  key.push_back((double)p.axis);
  key.push_back((double)(mask_render_mode() + (mask_rect_blocks() ? 0 : 8)));  // (the renderers give the same records bit for bit -- and a test that says so must render twice)
  const double* ap = blocks + (size_t)lp.mask_block * c->batch * FP_STRIDE;
  key.insert(key.end(), ap, ap + (size_t)2 * c->batch * FP_STRIDE);  // the two consecutive block sets
  key.insert(key.end(), lp.mask_shared.begin(), lp.mask_shared.end());
  int hit = -1, victim = 0;
  for (int k = 0; k < paos_ctx::kMaskSets; ++k) {
    const paos_ctx::MaskSet& ms = c->mask_sets[k];
    if (!ms.key.empty() && ms.key.size() == key.size() && !std::memcmp(ms.key.data(), key.data(), key.size() * sizeof(double))) hit = k;
    if (ms.used < c->mask_sets[victim].used) victim = k;
  }
  const int k = hit >= 0 ? hit : victim;
  paos_ctx::MaskSet& ms = c->mask_sets[k];
  if (!ms.lines) {
    // (every set of the context at the first use of any: a sweep walks through all of them within two or three programs, and an
    // allocation of this size in the middle of a later program stalls the stream -- a five-step bench run behind ONE warm-up step
    // measured 7 % low for it)
    for (paos_ctx::MaskSet& m2 : c->mask_sets) {
      if (m2.lines) continue;
      HIPCHK(c, hipMalloc(&m2.lines, (size_t)c->batch * c->n * sizeof(MaskLine)));
      HIPCHK(c, hipMalloc(&m2.vals, (size_t)c->batch * c->n * 2 * kMaskW * sizeof(double)));
    }
  }
  ms.used = ++c->mask_clock;
  lp.mask_set = k;
  lp.mask_render = hit < 0;
  lp.mask_shapes = 0;  // which of the two renderers have anything to do (the other one's launch would only exit)
  for (int it = 0; it < c->batch; ++it) {
    const double* a1 = ap + (size_t)it * FP_STRIDE;
    const double* a2 = a1 + (size_t)c->batch * FP_STRIDE;
    if (a1[0] != 0.0 && (it >= (int)lp.mask_shared.size() || lp.mask_shared[it] == 0.0)) lp.mask_shapes |= (int)a2[3] == 0 ? 1 : 2;
  }
  if (hit < 0) { ms.key = std::move(key); ++c->mask_rendered; } else ++c->mask_hits;
  for (int it = 0; it < c->batch; ++it) {
    if (lp.mask_rep[it] < 0) continue;
    FrugalSlot& sl = lp.mask_slot == 0 ? lp.items[it].pre : lp.items[it].mid;
    sl.lines = ms.lines + (size_t)lp.mask_rep[it] * c->n;
    sl.vals = ms.vals + (size_t)lp.mask_rep[it] * c->n * 2 * kMaskW;
  }
  return PAOS_OK;
}

void forget_mask_sets(paos_ctx* c) {  // after a failed program: what the sets hold is no longer known
  for (auto& ms : c->mask_sets) ms.key.clear();
}

MaskJob mask_job(paos_ctx* c, const paos_pass& p, const LoweredPass& lp, const double* ap, const double* dshared) {
  const paos_ctx::MaskSet& ms = c->mask_sets[lp.mask_set];
  const int line0 = lp.mask_lo, line_end = lp.mask_hi > lp.mask_lo ? lp.mask_hi : c->n;
  return MaskJob{ap, dshared, ms.lines, ms.vals, p.axis, line0, line_end, lp.mask_shapes};
}
// one launch per shape that occurs for `count` renderings (blockIdx.z), the grid sized for the widest line window
int launch_mask_jobs(paos_ctx* c, MaskJobs& jobs, int count) {
  jobs.batch_stride = c->batch * (int)FP_STRIDE; jobs.param_stride = (int)FP_STRIDE; jobs.n = c->n; jobs.overflow = c->mask_overflow;
  {
    const int mode = mask_render_mode();
    jobs.windows = mode != 0 ? 1 : 0;
    jobs.pairs = mode >= 2 ? mode - 1 : 0;
    jobs.rect_blocks = mask_rect_blocks() ? 1 : 0;
  }
  int widest = 0, shapes = 0;
  for (int j = 0; j < count; ++j) {
    widest = std::max(widest, jobs.job[j].line_end - jobs.job[j].line0);
    shapes |= jobs.job[j].shapes;
  }
  const dim3 block(256);
  // (ellipses: a wave renders four / two lines, the grid covers a quarter / half as many waves)
  const int per_wg = 4 * (jobs.pairs >= 2 ? 4 : (jobs.pairs ? 2 : 1));
  if (shapes & 1) hipLaunchKernelGGL(mask_lines_kernel<0>, dim3((widest + per_wg - 1) / per_wg, c->batch, count), block, 0, c->stream, jobs);
  const int rect_per_wg = 4 * (jobs.rect_blocks ? 64 : 1);
  if (shapes & 2) hipLaunchKernelGGL(mask_lines_kernel<1>, dim3((widest + rect_per_wg - 1) / rect_per_wg, c->batch, count), block, 0, c->stream, jobs);
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

// Do all phases of the pass vary along its lines only (the row / column factors of the separable programs)?  Then their
// factors come from tables by position (frugal_pass.h: FrugalSlot::table).  complex128 only (the complex64 slots use the
// hardware sin / cos).  PAOS_LINE_TABLES=0: every slot evaluates.
bool phases_along_lines(const paos_ctx* c, const paos_pass& p, const LoweredPass& lp, bool or_none = false) {
  static const bool want = [] { const char* e = getenv("PAOS_LINE_TABLES"); return !(e && e[0] == '0'); }();
  if (!want || (lp.kpre + lp.kmid == 0 && !or_none)) return false;
  const int counts[2] = {lp.kpre, lp.kmid};
  for (int l = 0; l < 2; ++l)
    for (const FrugalItem& fi : lp.items) {
      if (fi.active == 0.0) continue;
      const FrugalPhase* ph = l == 0 ? fi.pre_ph : fi.mid_ph;
      for (int j = 0; j < counts[l]; ++j)
        if ((p.axis == 0 ? ph[j].sy : ph[j].sx) != 0.0) return false;
    }
  return true;
}

// Can the launch of pass `p` go on with the next pass `p2` of the program (frugal_pass.h: LONG builds)?  Same axis, both on
// table slots with phases in every slot, a two-transform pass in front, and every item takes both or neither, on the same lines.  PAOS_FUSE_PAIRS=0: never.
bool can_fuse_pair(const paos_ctx* c, const paos_pass& p, const LoweredPass& lp, const paos_pass& p2, const LoweredPass& lp2) {
  const char* e = getenv("PAOS_FUSE_PAIRS");
  if (e && e[0] == '0') return false;
  if (!lp.ok || !lp2.ok || p.axis != p2.axis || lp.nfft != 2) return false;
  // (an aperture may ride on either pass -- the second one's slots read their line records themselves -- but the two must not
  // share a record set that the second would have to re-render between them: sets are assigned per pass, in program order)
  if (lp.mask_block >= 0 && lp2.mask_block >= 0 && lp.mask_set == lp2.mask_set) return false;
  // (a slot without phases takes part with a table of ones: (v 1) f is v f bit for bit)
  if (!phases_along_lines(c, p, lp, true) || !phases_along_lines(c, p2, lp2, true)) return false;
  for (int it = 0; it < c->batch; ++it) {
    const FrugalItem &f1 = lp.items[it], &f2 = lp2.items[it];
    if ((f1.active != 0.0) != (f2.active != 0.0)) return false;
    if (f1.active != 0.0 && (f1.line_lo != f2.line_lo || f1.line_hi != f2.line_hi || f1.line_fill != 0.0)) return false;
  }
  return true;
}

// launch a pass that lower_frugal accepted -- and, with `next`, the pass behind it in the same launch (can_fuse_pair)
// One launch of a pass program: a pass and the one or two behind it that ride along (can_fuse_pair).
struct FusedGroup {
  LoweredPass* lp[3] = {nullptr, nullptr, nullptr};
  int q = 0, count = 1, axis = 0;
  bool tables = false;   // its slots read their phase factors from tables
  size_t item_base = 0;  // index of its first item record in the staged array
};

// Stage the launches `groups`: decide which of them run on table slots (all phases of all their passes along the lines;
// a fused group always does), point their slots at tables of the context's store, copy the item records of all of them to
// the device in ONE transfer and build all tables with ONE launch -- in front of the program's first launch, so that
// nothing but the pass kernels themselves stands between two passes (round 4: a copy and a table launch in front of
// every pass cost ~50 us of an 1.3 ms launch).  *ditems: the device array; groups[g].item_base indexes it.
int stage_groups(paos_ctx* c, std::vector<FusedGroup>& groups, const FrugalItem** ditems) {
  std::vector<FrugalItem> blob;
  std::vector<PhaseSlotDesc> descs;
  int slots = 0;
  for (FusedGroup& g : groups) {
    paos_pass axis_only{};
    axis_only.axis = g.axis;
    // (complex64: a slot evaluates its factors with the hardware sin / cos for less than the table's loads cost -- measured: dense
    // launches 2.03 -> 2.41 ms with tables -- so only the fused groups, whose one build reads tables, use them)
    g.tables = g.count > 1 || (c->precision == PAOS_F64 && phases_along_lines(c, axis_only, *g.lp[0]));
    if (g.tables && g.count == 1) {
      // (a single pass over most of the grid is bound by HBM, and the table's reads are traffic too: dense launches
      // 3.75 ms evaluated, 3.85-4.0 with tables -- tables where the pass works on at most half of the lines)
      double lines = 0.0, active = 0.0;
      for (const FrugalItem& fi : g.lp[0]->items)
        if (fi.active != 0.0) { lines += fi.line_hi - fi.line_lo; active += 1.0; }
      if (active > 0.0 && lines > 0.5 * active * c->n) g.tables = false;
    }
    if (g.tables) slots += 2 * g.count;
  }
  if (slots > c->ptab_slots) {
    if (c->ptab) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->ptab); c->ptab = nullptr; c->ptab_slots = 0; }
    HIPCHK(c, hipMalloc(&c->ptab, (size_t)slots * c->batch * c->n * sizeof(cx<double>)));
    c->ptab_slots = slots;
  }
  int slot = 0;
  for (FusedGroup& g : groups) {
    g.item_base = blob.size();
    for (int k = 0; k < g.count; ++k) {
      LoweredPass& l = *g.lp[k];
      const int counts[2] = {l.kpre, l.kmid};
      for (int s = 0; s < 2; ++s) {
        // (in a fused group a slot without phases rides with a table of ones)
        const bool has = g.tables && (counts[s] > 0 || g.count > 1);
        for (int it = 0; it < c->batch; ++it)
          (s == 0 ? l.items[it].pre : l.items[it].mid).table = has ? c->ptab + ((size_t)slot * c->batch + it) * c->n : nullptr;
        if (has) descs.push_back(PhaseSlotDesc{(int)blob.size(), s, counts[s], g.axis});
        if (g.tables) ++slot;
      }
      blob.insert(blob.end(), l.items.begin(), l.items.end());
    }
  }
  static_assert(sizeof(FrugalItem) % sizeof(double) == 0 && sizeof(PhaseSlotDesc) == 2 * sizeof(double), "records of doubles");
  const size_t item_doubles = blob.size() * sizeof(FrugalItem) / sizeof(double);
  std::vector<double> flat(item_doubles + 2 * descs.size());
  std::memcpy(flat.data(), blob.data(), item_doubles * sizeof(double));
  if (!descs.empty()) std::memcpy(flat.data() + item_doubles, descs.data(), descs.size() * sizeof(PhaseSlotDesc));
  const double* dflat = nullptr;
  int rc = arena_push(c, flat.data(), flat.size(), &dflat);
  if (rc) return rc;
  *ditems = reinterpret_cast<const FrugalItem*>(dflat);
  if (!descs.empty()) {
    const PhaseTableArgs ta{*ditems, reinterpret_cast<const cx<double>*>(c->tw), reinterpret_cast<const PhaseSlotDesc*>(dflat + item_doubles), c->n,
                            c->precision == PAOS_F64 ? 0 : 1};
    hipLaunchKernelGGL(phase_table_kernel<0>, dim3(c->n / 256, c->batch, (unsigned)descs.size()), dim3(256), 0, c->stream, ta);
    HIPCHK(c, hipGetLastError());
  }
  return PAOS_OK;
}

int launch_lowered(paos_ctx* c, const paos_pass& p, LoweredPass& lp, const double* dblocks, bool store_psf = false,
                   bool sum_power = false, LoweredPass* next = nullptr, LoweredPass* next2 = nullptr,
                   const FrugalItem* staged = nullptr, bool staged_tables = false) {
  LoweredPass* const last = next2 ? next2 : next;  // the pass whose stores leave the launch (nullptr: this one)
  // PAOS_DUMP_PASSES=1: one line per pass launch on stderr (shape and what item 0's two slots carry)
  static const bool dump = [] { const char* e = getenv("PAOS_DUMP_PASSES"); return e && e[0] == '1'; }();
  if (dump)
    for (const LoweredPass* l : {(const LoweredPass*)&lp, (const LoweredPass*)next, (const LoweredPass*)next2}) {
      if (!l || l->items.empty()) continue;
      const FrugalItem& f = l->items[0];  // ("+pass": rides in the launch of the pass above it)
      std::fprintf(stderr, "%spass axis %d kpre %d kmid %d nfft %d | pre: sign %g scale %g mask %g | fft1 on %g inv %g | mid: sign %g scale %g mask %g | "
                   "fft2 on %g inv %g | lines [%g, %g) fill %g loads [%g, %g) stores [%g, %g)\n", l == &lp ? "" : "+", p.axis, l->kpre, l->kmid, l->nfft,
                   f.pre.sign_on, f.pre.scale, f.pre.mask_on, f.fft1_on, f.fft1_inv, f.mid.sign_on, f.mid.scale, f.mid.mask_on, f.fft2_on, f.fft2_inv,
                   f.line_lo, f.line_hi, f.line_fill, f.pos_lo, f.pos_hi, f.spos_lo, f.spos_hi);
    }
  for (LoweredPass* l : {&lp, next, next2}) {  // render the records along the pass axis, right before the pass (every pass of a chain)
    if (!l || l->mask_block < 0 || !l->mask_render) continue;
    const double* ap = dblocks + (size_t)l->mask_block * c->batch * FP_STRIDE;
    const double* dshared = nullptr;
    int rcs = arena_push(c, l->mask_shared.data(), l->mask_shared.size(), &dshared);
    if (rcs) return rcs;
    MaskJobs jobs{};
    jobs.job[0] = mask_job(c, p, *l, ap, dshared);  // (a pair runs along one axis)
    if ((rcs = launch_mask_jobs(c, jobs, 1))) return rcs;
    l->mask_render = false;
  }
  // The item records (with the pointers to their slots' phase tables) and the tables themselves: staged for the whole
  // program in front of its first launch (stage_groups: one copy, one table launch), or here for this launch alone.
  bool tables = staged_tables;
  const FrugalItem* ditems_f = staged;
  if (!ditems_f) {
    FusedGroup g;
    g.lp[0] = &lp; g.lp[1] = next; g.lp[2] = next2; g.count = next2 ? 3 : (next ? 2 : 1);
    g.axis = p.axis;
    std::vector<FusedGroup> one{g};
    int rcg = stage_groups(c, one, &ditems_f);
    if (rcg) return rcg;
    tables = one[0].tables;
    ditems_f += one[0].item_base;
  }
  const double* ditems = reinterpret_cast<const double*>(ditems_f);
  FrugalArgs a{c->field, c->tw, reinterpret_cast<const FrugalItem*>(ditems), c->pitch, c->item_stride, nullptr, nullptr, nullptr};
  if (store_psf) { a.psf = c->psf; a.psf_partial = c->psf_partial; }
  if (sum_power) a.pow_partial = c->pow_partial;
  a.dyn_scale = c->dyn_scale;
  {  // the lines some item still works on: the grid need not cover the others when their tiles have nothing to write
    // PAOS_COMPACT_GRID=0 launches the full grid (dead workgroups exit in their prologue)
    static const bool want = [] { const char* e = getenv("PAOS_COMPACT_GRID"); return !(e && e[0] == '0'); }();
    double lo = (double)c->n, hi = 0.0;
    bool fill = false;
    // (a fused pair: the lines are the same for both passes, what is stored and filled is the second pass's business)
    const std::vector<FrugalItem>& out_items = last ? last->items : lp.items;
    for (const FrugalItem& fi : out_items) {
      if (fi.active == 0.0) continue;
      lo = fi.line_lo < lo ? fi.line_lo : lo;
      hi = fi.line_hi > hi ? fi.line_hi : hi;
      if (fi.line_fill != 0.0 && store_psf) {  // dead tiles write PSF zeros inside [spos_lo, spos_hi) only (run_passes_impl)
        lo = fi.spos_lo < lo ? fi.spos_lo : lo;
        hi = fi.spos_hi > hi ? fi.spos_hi : hi;
      } else {
        fill = fill || fi.line_fill != 0.0;
      }
    }
    a.live_lo = a.live_hi = a.wg0 = 0;
    if (want && !fill && hi > lo && (lo > 0.0 || hi < (double)c->n)) {
      a.live_lo = (unsigned)lo;
      a.live_hi = (unsigned)hi;
    }
  }
  const int nfft = lp.nfft;
  // for the launch timer: what does this launch skip?  bit 0: whole tiles of dead lines, bit 1: loads of dead
  // positions, bit 2: stores nobody reads, bit 3: it stores the PSF instead of the field
  c->prof_next_tag = store_psf ? 8 : 0;
  c->prof_next_bytes = 0.0;
  c->prof_next_lines = 0.0;
  if (next) c->prof_next_tag |= next2 ? 32 : 16;  // bit 4: the launch ran two passes of the program, bit 5: three
  for (int it = 0; it < c->batch; ++it) {
    const FrugalItem& fi = lp.items[it];
    const FrugalItem& fo = last ? last->items[it] : fi;  // the pass whose stores leave the launch
    if (fi.active == 0.0) continue;
    if (fi.line_lo > 0.0 || fi.line_hi < (double)c->n) c->prof_next_tag |= 1;
    if (fi.pos_lo > 0.0 || fi.pos_hi < (double)c->n) c->prof_next_tag |= 2;
    if (!store_psf && (fo.spos_lo > 0.0 || fo.spos_hi < (double)c->n)) c->prof_next_tag |= 4;
    // what the plan has this launch move: its live lines' loaded and stored positions (the PSF store: doubles, every position)
    c->prof_next_bytes += (fi.line_hi - fi.line_lo) * ((fi.pos_hi - fi.pos_lo) * (double)elem_bytes(c) +
                                                        (store_psf ? (double)c->n * 8.0 : (fo.spos_hi - fo.spos_lo) * (double)elem_bytes(c)));
    // ... and compute: every live line goes through the transforms that are switched on for this item, in every pass of the launch
    for (const LoweredPass* l : {(const LoweredPass*)&lp, (const LoweredPass*)next, (const LoweredPass*)next2}) {
      if (!l) continue;
      const FrugalItem& fk = l->items[it];
      c->prof_next_lines += (fi.line_hi - fi.line_lo) * ((fk.fft1_on != 0.0 ? 1.0 : 0.0) + (l->nfft >= 2 && fk.fft2_on != 0.0 ? 1.0 : 0.0));
    }
  }
  a.tab = tables ? 1 : 0;  // (the TAB builds take "has phases" for the number of phases: their slots read one factor)
  {  // a single table pass whose every item loads and stores at most half of its positions: one-line workgroups (frugal_launch)
    bool light = tables && !next && !store_psf && !sum_power;
    for (int it = 0; it < c->batch && light; ++it) {
      const FrugalItem& fi = lp.items[it];
      if (fi.active == 0.0) continue;
      light = 2.0 * (fi.pos_hi - fi.pos_lo) <= (double)c->n && 2.0 * (fi.spos_hi - fi.spos_lo) <= (double)c->n;
    }
    a.one_line = light ? 1 : 0;
  }
  a.fuse = next2 ? 2 + next2->nfft : (next ? next->nfft : 0);  // LONG: the transforms of the passes that ride along
  // (a fused pair runs on the one build whose four slots all read tables)
  const int kpre = next ? 1 : (tables && lp.kpre > 1 ? 1 : lp.kpre), kmid = next ? 1 : (tables && lp.kmid > 1 ? 1 : lp.kmid);
  if (c->precision == PAOS_F64) {
    switch (c->n) {
      case 1024: return paos_frugal_d1024(c, a, p.axis, kpre, kmid, nfft);
      case 2048: return paos_frugal_d2048(c, a, p.axis, kpre, kmid, nfft);
      default: return paos_frugal_d4096(c, a, p.axis, kpre, kmid, nfft);
    }
  }
  switch (c->n) {
    case 2048: return paos_frugal_f2048(c, a, p.axis, kpre, kmid, nfft);
    default: return paos_frugal_f4096(c, a, p.axis, kpre, kmid, nfft);
  }
}

int launch_one_pass(paos_ctx* c, const paos_pass& p, const double* dblocks, int n_blocks,
                    const int* table_of_op /* [3][PAOS_MAX_PW], -1 = none */) {
  int rc;
  if ((rc = check_ops(c, p.pre, p.n_pre, n_blocks))) return rc;
  if ((rc = check_ops(c, p.mid, p.n_mid, n_blocks))) return rc;
  if ((rc = check_ops(c, p.post, p.n_post, n_blocks))) return rc;
  if (p.fft1 >= n_blocks || p.fft2 >= n_blocks) return fail(c, PAOS_EINVAL, "transform control block out of range");
  {  // at most one aperture operator per pass: its weight map is rendered right before it
    const paos_pw_op* lists3[3] = {p.pre, p.mid, p.post};
    const int counts3[3] = {p.n_pre, p.n_mid, p.n_post};
    int nmask = 0;
    for (int l = 0; l < 3; ++l)
      for (int o = 0; o < counts3[l]; ++o)
        if (lists3[l][o].kind == PAOS_PW_MASK) {
          if (++nmask > 1) return fail(c, PAOS_EINVAL, "one aperture operator per pass");
          if (!c->mask) HIPCHK(c, hipMalloc(&c->mask, (size_t)c->batch * c->item_stride * sizeof(double)));
          const double* ap = dblocks + (size_t)lists3[l][o].block * c->batch * FP_STRIDE;
          const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
          const double* ap2 = ap + (size_t)c->batch * FP_STRIDE;  // the next block set
          if (c->precision == PAOS_F64) {
            hipLaunchKernelGGL((aperture_kernel<double, BR, Lay<double>::BC, 0>), grid, block, 0, c->stream, (cx<double>*)nullptr, ap, ap2, FP_STRIDE, c->n, c->pitch, c->item_stride, c->mask, 1);
            hipLaunchKernelGGL((aperture_kernel<double, BR, Lay<double>::BC, 1>), grid, block, 0, c->stream, (cx<double>*)nullptr, ap, ap2, FP_STRIDE, c->n, c->pitch, c->item_stride, c->mask, 1);
          } else {
            F32_BR_SWITCH(c, hipLaunchKernelGGL((aperture_kernel<float, FBR, Lay<float>::BC, 0>), grid, block, 0, c->stream, (cx<float>*)nullptr, ap, ap2, FP_STRIDE, c->n, c->pitch, c->item_stride, c->mask, 1));
            F32_BR_SWITCH(c, hipLaunchKernelGGL((aperture_kernel<float, FBR, Lay<float>::BC, 1>), grid, block, 0, c->stream, (cx<float>*)nullptr, ap, ap2, FP_STRIDE, c->n, c->pitch, c->item_stride, c->mask, 1));
          }
          HIPCHK(c, hipGetLastError());
        }
  }
  PassArgs a{};
  a.field = c->field; a.tw = c->tw; a.blocks = dblocks; a.tables = c->tables; a.mask = c->mask; a.batch = c->batch;
  a.fft1 = p.fft1; a.fft2 = p.fft2; a.n_pre = p.n_pre; a.n_mid = p.n_mid; a.n_post = p.n_post;
  static_assert(sizeof(PwOp) == sizeof(paos_pw_op), "ABI op layout");
  std::memcpy(a.pre, p.pre, sizeof(a.pre));
  std::memcpy(a.mid, p.mid, sizeof(a.mid));
  std::memcpy(a.post, p.post, sizeof(a.post));
  PwOp* lists[3] = {a.pre, a.mid, a.post};
  const int counts[3] = {a.n_pre, a.n_mid, a.n_post};
  int feat = 0;
  for (int l = 0; l < 3; ++l)
    for (int o = 0; o < PAOS_MAX_PW; ++o) {
      lists[l][o].flags &= (1 << kTableShift) - 1;
      const int t = table_of_op[l * PAOS_MAX_PW + o];
      if (t >= 0) { lists[l][o].flags |= (t + 1) << kTableShift; feat = 3; }
      if (o < counts[l] && lists[l][o].kind == PWK_MASK) feat = 3;
    }
  a.pitch = c->pitch; a.item_stride = c->item_stride;
  if (p.axis == 0 || p.axis == 1)
    return c->precision == PAOS_F64 ? pass_t<double>(c, p.axis, a, feat) : pass_t<float>(c, p.axis, a, feat);
  if (p.axis != -1) return fail(c, PAOS_EINVAL, "pass axis must be 0, 1 or -1");
  if (p.fft1 >= 0 || p.fft2 >= 0 || p.n_mid || p.n_post)
    return fail(c, PAOS_EINVAL, "a transform-free pass carries its operators in the pre list");
  const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((pointwise_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream, a, c->n);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((pointwise_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream, a, c->n));
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

// PAOS_NO_PRUNE=1 processes every tile (A/B tests of the dead-line pruning).
bool use_pruning() {
  static const bool on = [] { const char* e = getenv("PAOS_NO_PRUNE"); return !(e && e[0] == '1'); }();
  return on;
}

int zero_outside_rows(paos_ctx* c, const double* live_rows);
int zero_outside_box(paos_ctx* c, const double* live_rows, const double* live_cols);
int psf_power_ticket(paos_ctx* c, const double* partial, int nparts, int* ticket, const double* source = nullptr);
int psf_keep_power_impl(paos_ctx* c, int* ticket);

// entry_rows / entry_stale: see paos_program_opts.  final_ticket != nullptr: the caller wants |u|^2 and its sum of
// the field the program ends with, not the field.
int run_passes_impl(paos_ctx* c, const paos_pass* passes, int n_passes, const double* blocks, int n_blocks,
                    const double* entry_rows, bool entry_stale, int* final_ticket, int final_mode, const double* entry_cols);

// final_mode (with final_ticket): 1 = the PSF instead of the field, 2 = the field as usual plus the ticket of its power
int run_passes(paos_ctx* c, const paos_pass* passes, int n_passes, const double* blocks, int n_blocks,
               const double* entry_rows = nullptr, bool entry_stale = false, int* final_ticket = nullptr, int final_mode = 1,
               const double* entry_cols = nullptr) {
  const int rc = run_passes_impl(c, passes, n_passes, blocks, n_blocks, entry_rows, entry_stale, final_ticket, final_mode, entry_cols);
  if (rc != PAOS_OK && c) forget_mask_sets(c);  // a program that stopped half way: which records were rendered is moot
  return rc;
}

int run_passes_impl(paos_ctx* c, const paos_pass* passes, int n_passes, const double* blocks, int n_blocks,
                    const double* entry_rows, bool entry_stale, int* final_ticket, int final_mode, const double* entry_cols) {
  if (!c || !passes || !blocks || n_passes < 0 || n_blocks < 1) return fail(c, PAOS_EINVAL, "bad pass program");
  c->norm2_of_field = false;  // (set again at the end when the last pass sums the power of what it stores)
  // A program that ends on the PSF gives the field up for it: the free power-ticket slot it will need is checked
  // BEFORE anything is launched (found full afterwards, the context held neither field nor ticket).
  if (final_ticket && c->norm_busy[next_norm_slot(c)])
    return fail(c, PAOS_EINVAL, "64 power reductions outstanding: fetch earlier tickets (paos_norm2_fetch) first");
  // The device sincos has no huge-argument path: bound every enabled phase operator here.
  for (int i = 0; i < n_passes; ++i) {
    const paos_pw_op* lists[3] = {passes[i].pre, passes[i].mid, passes[i].post};
    const int counts[3] = {passes[i].n_pre, passes[i].n_mid, passes[i].n_post};
    for (int l = 0; l < 3; ++l)
      for (int o = 0; o < counts[l] && o < PAOS_MAX_PW; ++o) {
        const paos_pw_op& op = lists[l][o];
        if (op.kind != PAOS_PW_QPHASE_CENTRED && op.kind != PAOS_PW_QPHASE_NATURAL) continue;
        if (op.block < 0 || op.block >= n_blocks) return fail(c, PAOS_EINVAL, "operator block index out of range");
        for (int it = 0; it < c->batch; ++it) {
          const double* p = blocks + ((size_t)op.block * c->batch + it) * FP_STRIDE;
          if (p[FP_ENABLE] == 0.0) continue;
          const double hx = 0.5 * c->n * p[FP_SX], hy = 0.5 * c->n * p[FP_SY];
          double arg = std::fabs(p[FP_COEF]) * (hx * hx + hy * hy);
          if (op.flags & PAOS_PWF_MUL2PI) arg *= 6.283185307179586;
          if (!(arg < kMaxPhaseArg))
            return fail(c, PAOS_EUNSUPPORTED, "quadratic phase exceeds 1e12 rad at the grid corner (or is not finite)");
        }
      }
  }
  const double* dblocks = nullptr;
  // everything this program pushes stays live until its last pass has run: the block table and,
  // per pass, one FrugalItem record per batch item (each push is rounded up to 16 doubles)
  // plus, for a pass that carries an aperture, the [batch] "shares its line records" vector (launch_lowered)
  // ... and two table descriptors (stage_groups)
  const size_t per_pass = (((size_t)c->batch * sizeof(FrugalItem) / sizeof(double) + 15) & ~size_t(15)) +
                          (((size_t)c->batch + 15) & ~size_t(15)) + 16;
  int rc = arena_reserve(c, (size_t)n_blocks * c->batch * FP_STRIDE + 16 + (size_t)n_passes * per_pass);
  if (rc) return rc;
  rc = arena_push(c, blocks, (size_t)n_blocks * c->batch * FP_STRIDE, &dblocks);
  if (rc) return rc;
  // Lower every pass for the frugal kernels first; when the whole program runs on them, plan the
  // pruning of dead lines across it (look-ahead), then launch.
  std::vector<LoweredPass> low(n_passes);
  bool all_frugal = !use_tables() && frugal_sizes(c) && n_passes > 0;
  if (!use_tables() && frugal_sizes(c)) {
    if ((rc = ensure_mask_store(c))) return rc;
    for (int q = 0; q < n_passes; ++q) {
      LoweredPass& lp = low[q];
      lp.ok = lower_frugal(c, passes[q], blocks, lp.items, lp.kpre, lp.kmid, lp.nfft, lp.mask_block, lp.mask_slot, lp.mask_shared,
                           lp.mask_rep);
      if (lp.ok && lp.mask_block >= 0 && (rc = assign_mask_set(c, passes[q], lp, blocks))) return rc;
      all_frugal = all_frugal && lp.ok;
    }
  }
  if (c->dyn_pending) {
    // a stop's scaling is waiting: the first pass takes it along in its middle slot when it is a frugal pass every item
    // takes part in; otherwise the field gets it now
    bool ride = n_passes > 0 && low[0].ok;
    if (ride)
      for (const FrugalItem& fi : low[0].items) ride = ride && fi.active != 0.0;
    if (!ride && (rc = settle_scale(c))) return rc;
  }
  const bool pruned = all_frugal && use_pruning() && c->prune;
  if (pruned) plan_pruning(c, passes, n_passes, blocks, low, entry_rows, entry_stale, entry_cols);
  // Which lines' aperture records does each pass read?  Those of its live tiles only: a workgroup whose lines are dead
  // for its item (frugal_pass_kernel: outside [line_lo, line_hi), bounds that are multiples of the tile height)
  // leaves before it looks at a record.  Only these lines are rendered (a quarter of them behind a clear aperture at
  // zoom 4), and a set found in the context's store is good only if it was rendered that far.
  {
    for (int q = 0; q < n_passes; ++q) {
      LoweredPass& lp = low[q];
      if (!lp.ok || lp.mask_block < 0) continue;
      double lo = (double)c->n, hi = 0.0;
      for (const FrugalItem& fi : lp.items) {
        if (fi.active == 0.0) continue;
        lo = fi.line_lo < lo ? fi.line_lo : lo;
        hi = fi.line_hi > hi ? fi.line_hi : hi;
      }
      const bool window = hi > lo;
      lp.mask_lo = window ? (int)lo : 0;
      lp.mask_hi = window ? (int)hi : c->n;
      paos_ctx::MaskSet& ms = c->mask_sets[lp.mask_set];
      if (!lp.mask_render && !(ms.line_lo <= lp.mask_lo && ms.line_hi >= lp.mask_hi)) {
        lp.mask_render = true;  // found, but rendered for a narrower window than this pass reads
        --c->mask_hits; ++c->mask_rendered;
      }
      if (lp.mask_render) { ms.line_lo = lp.mask_lo; ms.line_hi = lp.mask_hi; }
    }
  }
  if (entry_stale && entry_rows) {
    // Rows that merely stand for zeros must become zeros wherever the program will not consume them: everywhere when
    // the planner is off, and for an item no pass of the program touches.
    bool need = !pruned;
    for (int it = 0; it < c->batch && !need; ++it) {
      bool active = false;
      for (int q = 0; q < n_passes; ++q) active = active || low[q].items[it].active != 0.0;
      need = !active;
    }
    if (need) {
      // (all items at once: the ones the planner handles lose nothing but a little time, and the planner was told
      // "stale", which is also right for zeros)
      if ((rc = entry_cols ? zero_outside_box(c, entry_rows, entry_cols) : zero_outside_rows(c, entry_rows))) return rc;
    }
  }
  // The PSF instead of the field: the last pass stores |u|^2 and its per-workgroup sums (frugal_pass.h: STORE) when
  // it runs on the frugal kernels in a shape built for it and every item takes part; otherwise the program runs as
  // usual and the intensity sweep follows.
  bool fused_store = false, fused_power = false;
  int power_groups = 0;
  if (final_ticket && final_mode == 2) {
    // The field is kept AND its power is wanted (a saved surface, run.py:218-223 callers): the last pass sums |u|^2 of
    // its tiles while it stores them (FrugalArgs::pow_partial) -- when it runs on the frugal kernels and every item
    // takes part; otherwise the ordinary reduction follows the program.
    fused_power = n_passes > 0 && low[n_passes - 1].ok && low[n_passes - 1].kpre <= 1;  // (the shapes built with STORE = 2)
    if (fused_power)
      for (const FrugalItem& fi : low[n_passes - 1].items) fused_power = fused_power && fi.active != 0.0;
    if (fused_power) {
      const int lines = passes[n_passes - 1].axis == 0 ? (c->n >= 2048 ? c->br / 2 : c->br) : 2;  // FftCfg: FR_ROW_LINES / COL_LINES
      power_groups = c->n / lines;
      if (c->pow_nparts < c->n / 2) {  // (sized for the finest tiling of either axis)
        if (c->pow_partial) (void)hipFree(c->pow_partial);
        c->pow_partial = nullptr; c->pow_nparts = 0;
        HIPCHK(c, hipMalloc(&c->pow_partial, (size_t)c->batch * (c->n / 2) * sizeof(double)));
        c->pow_nparts = c->n / 2;
      }
      // dead tiles and workgroups that are not launched at all contribute nothing: zeros
      HIPCHK(c, hipMemsetAsync(c->pow_partial, 0, (size_t)c->batch * power_groups * sizeof(double), c->stream));
    }
  }
  if (final_ticket && final_mode != 2) {
    if (!c->psf) HIPCHK(c, hipMalloc(&c->psf, (size_t)c->batch * c->item_stride * sizeof(double)));
    fused_store = n_passes > 0 && low[n_passes - 1].ok && low[n_passes - 1].kpre <= 1 && low[n_passes - 1].kmid <= 1;
    if (fused_store)
      for (const FrugalItem& fi : low[n_passes - 1].items) fused_store = fused_store && fi.active != 0.0;
    if (fused_store) {
      const int lines = passes[n_passes - 1].axis == 0 ? (c->n >= 2048 ? c->br / 2 : c->br) : 2;  // FftCfg: FR_ROW_LINES / COL_LINES
      const int groups = c->n / lines;
      if (c->psf_nparts < groups) {
        if (c->psf_partial) (void)hipFree(c->psf_partial);
        c->psf_partial = nullptr; c->psf_nparts = 0;
        c->psf_zero_axis = -1;
        HIPCHK(c, hipMalloc(&c->psf_partial, (size_t)c->batch * groups * sizeof(double)));
        c->psf_nparts = groups;
      }
      // dead tiles of the storing pass write PSF zeros (line_fill) -- unless the buffer already holds zeros there:
      // the previous storing pass had the same live lines (the next batch of a sweep, the next Monte-Carlo batch)
      static const bool reuse = [] { const char* e = getenv("PAOS_PSF_ZERO_REUSE"); return !(e && e[0] == '0'); }();
      const int axis = passes[n_passes - 1].axis;
      std::vector<FrugalItem>& last = low[n_passes - 1].items;
      // Round 4: per item.  What the buffer may still hold of the previous storing pass lies inside that pass's live
      // lines [psf_zero_lo, psf_zero_hi); only dead tiles of THIS pass that meet them write zeros (the range travels in
      // the item's -- here otherwise unused -- store-position fields).  A walked sweep changes the sampling at the
      // image plane from batch to batch, so the live lines are never the same twice, but they move by a few lines.
      const bool known = reuse && c->psf_zero_axis == axis && (int)c->psf_zero_lo.size() == c->batch;
      c->psf_zero_lo.resize(c->batch); c->psf_zero_hi.resize(c->batch);
      for (int it = 0; it < c->batch; ++it) {
        const double old_lo = known ? c->psf_zero_lo[it] : 0.0, old_hi = known ? c->psf_zero_hi[it] : (double)c->n;
        const bool covered = old_lo >= last[it].line_lo && old_hi <= last[it].line_hi;  // every old line is rewritten
        last[it].line_fill = covered ? 0.0 : 1.0;
        last[it].spos_lo = covered ? 0.0 : old_lo;
        last[it].spos_hi = covered ? (double)c->n : old_hi;
        c->psf_zero_lo[it] = last[it].line_lo; c->psf_zero_hi[it] = last[it].line_hi;
      }
      c->psf_zero_axis = -1;  // set again below once the pass is on the stream
    }
  }
  // The aperture line records this program has to render (not found in the context's kept sets): all of them in ONE
  // launch per shape, in front of the first pass -- when every rendering goes to a set no EARLIER pass of this program
  // reads (always, unless a program carries more distinct apertures than there are sets; then each is rendered in place,
  // right before its pass).  PAOS_BATCHED_RECORDS=0: in place, one launch per aperture, as before.
  {
    static const bool want = [] { const char* e = getenv("PAOS_BATCHED_RECORDS"); return !(e && e[0] == '0'); }();
    std::vector<int> todo;
    bool safe = want && all_frugal;
    for (int q = 0; q < n_passes && safe; ++q) {
      if (low[q].mask_block < 0) continue;
      if (low[q].mask_render) {
        for (int r = 0; r < q; ++r) safe = safe && !(low[r].mask_block >= 0 && low[r].mask_set == low[q].mask_set);
        todo.push_back(q);
      }
    }
    if (safe && todo.size() > 1) {
      for (size_t at = 0; at < todo.size(); at += kMaskJobs) {
        MaskJobs jobs{};
        const int count = (int)std::min<size_t>(kMaskJobs, todo.size() - at);
        for (int j = 0; j < count; ++j) {
          LoweredPass& lp = low[todo[at + j]];
          const double* dshared = nullptr;
          if ((rc = arena_push(c, lp.mask_shared.data(), lp.mask_shared.size(), &dshared))) return rc;
          jobs.job[j] = mask_job(c, passes[todo[at + j]], lp, dblocks + (size_t)lp.mask_block * c->batch * FP_STRIDE, dshared);
          lp.mask_render = false;
        }
        if ((rc = launch_mask_jobs(c, jobs, count))) return rc;
      }
    }
  }
  // The launches of the program: a pass, or two / three consecutive passes of one row / column chain (can_fuse_pair).
  // When every pass runs on the frugal kernels their item records and phase tables are staged here, once, for all of
  // them (stage_groups); otherwise each launch stages its own.  PAOS_STAGE_PROGRAM=0: each launch stages its own.
  std::vector<FusedGroup> groups;
  std::vector<int> group_of(n_passes, -1);
  const FrugalItem* staged_items = nullptr;
  {
    for (int q = 0; q < n_passes;) {
      FusedGroup g;
      g.q = q; g.axis = passes[q].axis; g.lp[0] = &low[q];
      if (low[q].ok && all_frugal && pruned) {
        const bool pair = q + 1 < n_passes && can_fuse_pair(c, passes[q], low[q], passes[q + 1], low[q + 1]);
        // (and a third: the five-transform chains -- ptp, stw, ptp -- that end a SYN20-like prescription)
        const bool triple = pair && q + 2 < n_passes && can_fuse_pair(c, passes[q + 1], low[q + 1], passes[q + 2], low[q + 2]) &&
                            !(low[q].mask_block >= 0 && low[q + 2].mask_block >= 0 && low[q].mask_set == low[q + 2].mask_set);
        g.count = triple ? 3 : (pair ? 2 : 1);
        for (int k = 1; k < g.count; ++k) g.lp[k] = &low[q + k];
      }
      group_of[q] = (int)groups.size();
      groups.push_back(g);
      q += g.count;
    }
    const char* e = getenv("PAOS_STAGE_PROGRAM");
    // (a table per operator slot of every pass: 2 MiB each at 4096^2 x 32; a program that would need more than 1 GiB of
    // them -- hundreds of passes -- stages launch by launch, on six tables)
    const size_t table_bytes = (size_t)2 * n_passes * c->batch * c->n * sizeof(cx<double>);
    if (all_frugal && !(e && e[0] == '0') && table_bytes <= (size_t(1) << 30))
      if ((rc = stage_groups(c, groups, &staged_items))) return rc;
  }
  // Walk the program in chunks whose phase operators fit the table store: fill the tables of
  // a chunk with one small launch, then run its passes.
  int i = 0;
  while (i < n_passes) {
    TableArgs ta{};
    ta.blocks = dblocks; ta.tables = c->tables; ta.batch = c->batch; ta.n = c->n; ta.count = 0;
    std::vector<int> assign;  // per pass of the chunk: [3][PAOS_MAX_PW]
    int j = i;
    for (; j < n_passes; ++j) {
      const paos_pass& p = passes[j];
      const paos_pw_op* lists[3] = {p.pre, p.mid, p.post};
      const int counts[3] = {p.n_pre, p.n_mid, p.n_post};
      int need = 0;
      for (int l = 0; l < 3; ++l)
        for (int o = 0; o < counts[l] && o < PAOS_MAX_PW; ++o)
          need += (lists[l][o].kind == PAOS_PW_QPHASE_CENTRED || lists[l][o].kind == PAOS_PW_QPHASE_NATURAL);
      if (use_tables() && ta.count + need > kMaxTables) {
        if (j == i) return fail(c, PAOS_EINVAL, "one pass needs more phase tables than the store holds");
        break;
      }
      const size_t base = assign.size();
      assign.resize(base + 3 * PAOS_MAX_PW, -1);
      if (!use_tables()) continue;
      for (int l = 0; l < 3; ++l)
        for (int o = 0; o < counts[l] && o < PAOS_MAX_PW; ++o) {
          const paos_pw_op& op = lists[l][o];
          if (op.kind != PAOS_PW_QPHASE_CENTRED && op.kind != PAOS_PW_QPHASE_NATURAL) continue;
          if (op.block < 0 || op.block >= n_blocks) return fail(c, PAOS_EINVAL, "operator block index out of range");
          ta.jobs[ta.count] = {op.block, op.kind, op.flags & PAOS_PWF_MUL2PI};
          assign[base + l * PAOS_MAX_PW + o] = ta.count++;
        }
    }
    if (ta.count > 0) {
      if (!c->tables) {  // opt-in feature: allocate on first use
        HIPCHK(c, hipMalloc(&c->tables, (size_t)kMaxTables * c->batch * 2 * c->n * sizeof(cx<double>)));
        ta.tables = c->tables;
      }
      const dim3 grid((2 * c->n + 255) / 256, c->batch, ta.count);
      hipLaunchKernelGGL(phase_table_kernel, grid, dim3(256), 0, c->stream, ta);
      HIPCHK(c, hipGetLastError());
    }
    for (int q = i; q < j; ++q) {
      if (low[q].ok) {
        // (a group of up to three passes of one row / column chain runs in ONE launch: frugal_pass.h, LONG builds)
        const FusedGroup& g = groups[group_of[q]];
        const int last = q + g.count - 1;
        if (last >= j) return fail(c, PAOS_EINVAL, "a fused group crosses a table chunk");
        if ((rc = launch_lowered(c, passes[q], low[q], dblocks, fused_store && last == n_passes - 1, fused_power && last == n_passes - 1,
                                 g.lp[1], g.lp[2], staged_items ? staged_items + g.item_base : nullptr, g.tables))) return rc;
        q = last;
        if (c->dyn_pending) {  // (first launch of the program) the stop's factor has gone into the field: ones again for the next
          hipLaunchKernelGGL(dyn_scale_reset_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream, c->dyn_scale, c->batch);
          HIPCHK(c, hipGetLastError());
          c->dyn_pending = false;
        }
        continue;
      }
      if ((rc = launch_one_pass(c, passes[q], dblocks, n_blocks, &assign[(size_t)(q - i) * 3 * PAOS_MAX_PW]))) return rc;
    }
    i = j;
  }
  if (final_ticket && final_mode == 2) {
    // (the fallback: a generic-kernel pass, or an item that sat the last pass out)
    const int rcp = fused_power ? psf_power_ticket(c, c->pow_partial, power_groups, final_ticket) : paos_norm2_enqueue(c, final_ticket);
    if (rcp == PAOS_OK) c->norm2_of_field = true;  // c->norm2 = the power of every item's field as stored
    return rcp;
  }
  if (final_ticket) {
    if (fused_store) {
      c->psf_zero_axis = passes[n_passes - 1].axis;  // every launch of the program went through
      const int lines = passes[n_passes - 1].axis == 0 ? (c->n >= 2048 ? c->br / 2 : c->br) : 2;
      return psf_power_ticket(c, c->psf_partial, c->n / lines, final_ticket);
    }
    return psf_keep_power_impl(c, final_ticket);
  }
  return PAOS_OK;
}

// ---- the reference's primitives as pass programs --------------------------------------
struct Program {
  std::vector<paos_pass> passes;
  std::vector<double> blocks;
  int batch;
  explicit Program(int b) : batch(b) {}
  // block set with the enable flags of `src` (stride FP_STRIDE) and the given payload
  int add_block(const double* src, double v1, double v2, double v3, double v4, bool copy) {
    const int id = (int)(blocks.size() / ((size_t)batch * FP_STRIDE));
    for (int i = 0; i < batch; ++i) {
      const double* s = src + (size_t)i * FP_STRIDE;
      if (copy) blocks.insert(blocks.end(), s, s + FP_STRIDE);
      else blocks.insert(blocks.end(), {s[FP_ENABLE], v1, v2, v3, v4});
    }
    return id;
  }
  paos_pass& add_pass(int axis, int fft1, int fft2) {
    paos_pass p{};
    p.axis = axis; p.fft1 = fft1; p.fft2 = fft2;
    passes.push_back(p);
    return passes.back();
  }
};
void push(paos_pw_op* list, int& count, int kind, int block, int flags = 0) { list[count++] = {kind, flags, block}; }

enum FftOp { OP_PTP, OP_STW, OP_WTS };

int fft_op(paos_ctx* c, FftOp op, const double* params, int inverse) {
  if (!c || !params) return fail(c, PAOS_EINVAL, "null argument");
  Program g(c->batch);
  const double inv_n = 1.0 / c->n;
  const int par = g.add_block(params, 0, 0, 0, 0, true);
  const int scl = g.add_block(params, 0, 0, inv_n, 0, false);
  if (op == OP_PTP) {
    const int fwd = g.add_block(params, 0, 0, 0, 0, false), inv = g.add_block(params, 1, 0, 0, 0, false);
    g.add_pass(0, fwd, -1);
    { paos_pass& p = g.add_pass(1, fwd, inv); push(p.mid, p.n_mid, PAOS_PW_QPHASE_NATURAL, par); push(p.mid, p.n_mid, PAOS_PW_SCALE, scl); }
    { paos_pass& p = g.add_pass(0, inv, -1); push(p.mid, p.n_mid, PAOS_PW_SCALE, scl); }
  } else {
    const int ctl = g.add_block(params, inverse ? 1.0 : 0.0, 0, 0, 0, false);
    paos_pass& r = g.add_pass(0, ctl, -1);
    push(r.pre, r.n_pre, PAOS_PW_SIGN, par);
    if (op == OP_WTS) push(r.pre, r.n_pre, PAOS_PW_QPHASE_CENTRED, par);
    paos_pass& q = g.add_pass(1, ctl, -1);
    push(q.mid, q.n_mid, PAOS_PW_SIGN, par);
    if (op == OP_STW) push(q.mid, q.n_mid, PAOS_PW_QPHASE_CENTRED, par);
    push(q.mid, q.n_mid, PAOS_PW_SCALE, scl);
  }
  return run_passes(c, g.passes.data(), (int)g.passes.size(), g.blocks.data(),
                    (int)(g.blocks.size() / ((size_t)c->batch * FP_STRIDE)));
}

template <typename T>
std::vector<std::complex<T>> twiddles(int n) {
  std::vector<std::complex<T>> tw(n);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int m = 0; m < n; ++m) {
    // exact octant symmetry keeps the table correctly rounded and conj-symmetric
    const long double a = two_pi * (long double)m / (long double)n;
    tw[m] = std::complex<T>((T)cosl(a), (T)-sinl(a));
  }
  return tw;
}

#define DISPATCH_T(c, expr_d, expr_f) ((c)->precision == PAOS_F64 ? (expr_d) : (expr_f))

}  // namespace

static int check_mask_overflow(paos_ctx* c);

extern "C" {

const char* paos_last_error(const paos_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

const char* paos_build_info(void) {
  static const std::string info = std::string("libpaoship gfx950 layout=") + std::to_string(BR) + "(c64 at N>=2048: " + std::to_string(PAOS_F32_BR) + ")x(" +
                                  std::to_string(Lay<double>::BC) + "|" + std::to_string(Lay<float>::BC) +
                                  ") pad_blocks=" + std::to_string(PAOS_PAD_BLOCKS);
  return info.c_str();
}

void* paos_stream(paos_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int paos_ctx_create(int device, int n, int batch, int precision, paos_ctx** out) {
  if (!out) return fail(nullptr, PAOS_EINVAL, "out is null");
  *out = nullptr;
  if (n < 64 || n > 4096 || (n & (n - 1))) return fail(nullptr, PAOS_EUNSUPPORTED, "grid size must be a power of two in 64..4096");
  if (batch < 1) return fail(nullptr, PAOS_EINVAL, "batch must be >= 1");
  if (precision != PAOS_F64 && precision != PAOS_F32) return fail(nullptr, PAOS_EINVAL, "precision must be PAOS_F64 or PAOS_F32");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, PAOS_EHIP, "no HIP device available: libpaoship has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(nullptr, PAOS_EINVAL, "device index out of range");
  paos_ctx* c = new paos_ctx();
  c->device = device; c->n = n; c->batch = batch; c->precision = precision;
  const int bc = precision == PAOS_F64 ? Lay<double>::BC : Lay<float>::BC;
  c->br = (precision == PAOS_F32 && n >= 2048) ? PAOS_F32_BR : PAOS_BR;  // block_rows<T, N>()
  c->pitch = (unsigned)n * c->br + (unsigned)PAOS_PAD_BLOCKS * c->br * bc;
  c->item_stride = c->pitch * (unsigned)(n / c->br);
  const size_t eb = elem_bytes(c);
  auto bail = [&](hipError_t e, const char* what) {
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    paos_ctx_destroy(c);
    return fail(nullptr, PAOS_EHIP, msg);
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  if (const char* pad = getenv("PAOS_LDS_PAD")) c->lds_pad = (size_t)std::max(0, std::atoi(pad));
  if ((e = hipMalloc(&c->dyn_scale, (size_t)batch * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(dyn_scale)");
  {
    std::vector<double> ones((size_t)batch, 1.0);
    if ((e = hipMemcpy(c->dyn_scale, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(dyn_scale)");
  }
  if ((e = hipMalloc(&c->field, (size_t)c->item_stride * batch * eb)) != hipSuccess) return bail(e, "hipMalloc(field)");
  if ((e = hipMemsetAsync(c->field, 0, (size_t)c->item_stride * batch * eb, c->stream)) != hipSuccess) return bail(e, "hipMemset(field)");
  if ((e = hipMalloc(&c->tw, (size_t)n * eb)) != hipSuccess) return bail(e, "hipMalloc(tw)");
  if ((e = hipMalloc(&c->staging, (size_t)n * n * 16)) != hipSuccess) return bail(e, "hipMalloc(staging)");
  c->nparts = 1024;
  if ((e = hipMalloc(&c->partial, (size_t)batch * c->nparts * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(partial)");
  if ((e = hipMalloc(&c->norm2, (size_t)batch * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(norm2)");
  if ((e = hipHostMalloc(&c->norm2_host, (size_t)kNormSlots * batch * sizeof(double))) != hipSuccess) return bail(e, "hipHostMalloc(norm2)");
  c->arena.cap = (size_t)1 << 18;  // four slabs of 2 MiB of doubles; a large batch starts with room for one of its programs per slab
  if (c->arena.cap < (size_t)batch * 2048) c->arena.cap = (size_t)batch * 2048;
  if ((e = hipHostMalloc(&c->arena.host, kArenaSlabs * c->arena.cap * sizeof(double))) != hipSuccess) return bail(e, "hipHostMalloc(arena)");
  if ((e = hipMalloc(&c->arena.dev, kArenaSlabs * c->arena.cap * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(arena)");
  for (int k = 0; k < kArenaSlabs; ++k)
    if ((e = hipEventCreateWithFlags(&c->arena.fence[k], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate(arena fence)");
  c->arena.used[0] = true;
  if (precision == PAOS_F64) {
    auto tw = twiddles<double>(n);
    e = hipMemcpy(c->tw, tw.data(), (size_t)n * eb, hipMemcpyHostToDevice);
  } else {
    auto tw = twiddles<float>(n);
    e = hipMemcpy(c->tw, tw.data(), (size_t)n * eb, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) return bail(e, "hipMemcpy(tw)");
  *out = c;
  return PAOS_OK;
}

int paos_profile_begin(paos_ctx* c, int kernel_kind, int max_launches) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || max_launches < 0) return fail(c, PAOS_EINVAL, "bad profile request");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  while (c->prof_events.size() < (size_t)2 * max_launches) {
    hipEvent_t e;
    HIPCHK(c, hipEventCreate(&e));
    c->prof_events.push_back(e);
  }
  c->prof_kind = kernel_kind;
  c->prof_used = 0;
  c->prof_tags.clear();
  c->prof_bytes.clear();
  c->prof_lines.clear();
  return PAOS_OK;
}

static int profile_end(paos_ctx* c, int* launches, double* total_ms, int* pruned_launches, double* pruned_ms) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !launches || !total_ms) return fail(c, PAOS_EINVAL, "null argument");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double sum = 0.0, psum = 0.0;
  int pcount = 0;
  for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->prof_events[i], c->prof_events[i + 1]));
    sum += ms;
    if (i / 2 < c->prof_tags.size() && c->prof_tags[i / 2]) { psum += ms; ++pcount; }
  }
  *launches = (int)(c->prof_used / 2);
  *total_ms = sum;
  if (pruned_launches) *pruned_launches = pcount;
  if (pruned_ms) *pruned_ms = psum;
  c->prof_kind = -1;
  c->prof_used = 0;
  return PAOS_OK;
}

int paos_profile_end_launches(paos_ctx* c, int capacity, double* ms_out, int* tag_out, int* count) {
  if (c) (void)hipSetDevice(c->device);
  if (!c || !ms_out || !tag_out || !count || capacity < 0) return fail(c, PAOS_EINVAL, "null argument");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int n = (int)(c->prof_used / 2);
  if (n > capacity) return fail(c, PAOS_EINVAL, "more launches were timed than the caller's arrays hold");
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
    ms_out[i] = ms;
    tag_out[i] = (size_t)i < c->prof_tags.size() ? c->prof_tags[i] : 0;
  }
  *count = n;
  c->prof_kind = -1;
  c->prof_used = 0;
  return PAOS_OK;
}

// bytes the pruning plan had each timed launch so far load + store (call BEFORE paos_profile_end_launches, which resets)
int paos_profile_planned_bytes(paos_ctx* c, int capacity, double* bytes_out, int* count) {
  if (!c || !bytes_out || !count || capacity < 0) return fail(c, PAOS_EINVAL, "null argument");
  const int n = (int)(c->prof_used / 2);
  if (n > capacity) return fail(c, PAOS_EINVAL, "more launches were timed than the caller's array holds");
  for (int i = 0; i < n; ++i) bytes_out[i] = (size_t)i < c->prof_bytes.size() ? c->prof_bytes[i] : 0.0;
  *count = n;
  return PAOS_OK;
}

// 1-D line transforms each timed launch so far ran (call BEFORE paos_profile_end_launches, which resets)
int paos_profile_line_transforms(paos_ctx* c, int capacity, double* lines_out, int* count) {
  if (!c || !lines_out || !count || capacity < 0) return fail(c, PAOS_EINVAL, "bad profile request");
  const int n = (int)(c->prof_used / 2);
  if (n > capacity) return fail(c, PAOS_EINVAL, "profile buffer too small");
  for (int i = 0; i < n; ++i) lines_out[i] = (size_t)i < c->prof_lines.size() ? c->prof_lines[i] : 0.0;
  *count = n;
  return PAOS_OK;
}

int paos_profile_end(paos_ctx* c, int* launches, double* total_ms) {
  return profile_end(c, launches, total_ms, nullptr, nullptr);
}

int paos_profile_end_split(paos_ctx* c, int* launches, double* total_ms, int* pruned_launches, double* pruned_ms) {
  if (!pruned_launches || !pruned_ms) return fail(c, PAOS_EINVAL, "null argument");
  return profile_end(c, launches, total_ms, pruned_launches, pruned_ms);
}

int paos_ctx_destroy(paos_ctx* c) {
  if (!c) return PAOS_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
  if (c->field) (void)hipFree(c->field);
  if (c->tw) (void)hipFree(c->tw);
  if (c->staging) (void)hipFree(c->staging);
  if (c->tables) (void)hipFree(c->tables);
  if (c->mask) (void)hipFree(c->mask);
  if (c->metric_partial) (void)hipFree(c->metric_partial);
  if (c->metric_out) (void)hipFree(c->metric_out);
  if (c->metric_host) (void)hipHostFree(c->metric_host);
  for (auto& ms : c->mask_sets) {
    if (ms.lines) (void)hipFree(ms.lines);
    if (ms.vals) (void)hipFree(ms.vals);
  }
  if (c->mask_overflow) (void)hipFree(c->mask_overflow);
  if (c->partial) (void)hipFree(c->partial);
  if (c->norm2) (void)hipFree(c->norm2);
  if (c->norm2_host) (void)hipHostFree(c->norm2_host);
  if (c->psf) (void)hipFree(c->psf);
  if (c->map_dev) (void)hipFree(c->map_dev);
  if (c->psd_scratch) (void)hipFree(c->psd_scratch);
  if (c->start_norm2) (void)hipFree(c->start_norm2);
  if (c->psd_bad) (void)hipFree(c->psd_bad);
  if (c->pow_partial) (void)hipFree(c->pow_partial);
  if (c->dyn_scale) (void)hipFree(c->dyn_scale);
  if (c->ptab) (void)hipFree(c->ptab);
  if (c->psf_partial) (void)hipFree(c->psf_partial);
  for (int i = 0; i < 2; ++i) {
    if (c->bounce[i]) (void)hipHostFree(c->bounce[i]);
    if (c->bounce_ev[i]) (void)hipEventDestroy(c->bounce_ev[i]);
  }
  if (c->arena.host) (void)hipHostFree(c->arena.host);
  if (c->arena.dev) (void)hipFree(c->arena.dev);
  for (int k = 0; k < kArenaSlabs; ++k)
    if (c->arena.fence[k]) (void)hipEventDestroy(c->arena.fence[k]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return PAOS_OK;
}

// after a synchronisation: did an aperture's partial run overflow its line records?
static int check_mask_overflow(paos_ctx* c) {
  if (!c->mask_overflow) return PAOS_OK;
  int n = 0;
  HIPCHK(c, hipMemcpy(&n, c->mask_overflow, sizeof(int), hipMemcpyDeviceToHost));
  if (n != 0) {
    (void)hipMemset(c->mask_overflow, 0, sizeof(int));
    for (auto& ms : c->mask_sets) ms.key.clear();
    return fail(c, PAOS_EUNSUPPORTED, "aperture line records overflowed (partial run longer than kMaskW): results are invalid");
  }
  return PAOS_OK;
}

int paos_sync(paos_ctx* c) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c) return fail(c, PAOS_EINVAL, "null context");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_mask_overflow(c);
}

int paos_fill(paos_ctx* c, double re, double im) {
  DROP_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c) return fail(c, PAOS_EINVAL, "null context");
  const size_t total = (size_t)c->item_stride * c->batch;
  // padding blocks are filled too; they are never read by any operator
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL(fill_kernel<double>, dim3(2048), dim3(kPwThreads), 0, c->stream,
                       (cx<double>*)c->field, total, re, im);
  else
    hipLaunchKernelGGL(fill_kernel<float>, dim3(2048), dim3(kPwThreads), 0, c->stream,
                       (cx<float>*)c->field, total, (float)re, (float)im);
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

static int check_rows(paos_ctx* c, const double* rows) {
  for (int i = 0; i < c->batch; ++i)
    if (!(rows[2 * i] >= 0.0) || !(rows[2 * i + 1] <= (double)c->n) || !(rows[2 * i] <= rows[2 * i + 1]))
      return fail(c, PAOS_EINVAL, "row range must satisfy 0 <= lo <= hi <= n");
  return PAOS_OK;
}

// Column windows as every consumer uses them: rounded outward to whole multiples of the block height c->br -- the
// granularity at which plan_pruning lets a pass load positions -- so that what paos_start_box writes, what
// paos_norm2_enqueue_box sums, what paos_zero_outside_box keeps and what the first pass of a program loads are ONE window
// (round 5: written to whole blocks of two columns only, the first pass read up to two stale columns at either edge).
static std::vector<double> rounded_cols(const paos_ctx* c, const double* cols) {
  std::vector<double> out((size_t)2 * c->batch);
  for (int i = 0; i < c->batch; ++i) {
    int l = (int)cols[2 * i], h = (int)cols[2 * i + 1];
    l = l < 0 ? 0 : (l / c->br) * c->br;
    h = ((h + c->br - 1) / c->br) * c->br;
    if (h > c->n) h = c->n;
    out[2 * i] = l; out[2 * i + 1] = h;
  }
  return out;
}

static int start_impl(paos_ctx* c, double re, double im, int shape, const double* aperture, const double* stop,
                      const double* write_rows, const double* write_cols = nullptr) {
  if (c) (void)hipSetDevice(c->device);
  if (!c || !aperture) return fail(c, PAOS_EINVAL, "null argument");
  if (shape != PAOS_SHAPE_ELLIPSE && shape != PAOS_SHAPE_RECT) return fail(c, PAOS_EINVAL, "unknown aperture shape");
  std::vector<double> flags(c->batch, 0.0);
  bool any_stop = false;
  if (stop)
    for (int i = 0; i < c->batch; ++i) { flags[i] = stop[i] != 0.0 ? 1.0 : 0.0; any_stop |= flags[i] != 0.0; }
  // items with identical aperture records (a wavelength sweep at the entrance pupil, a Monte-Carlo batch)
  // have identical power sums: stage 1 runs for the first of them, stage 2 sums ITS partials for all
  std::vector<double> power_of(c->batch), compute(c->batch);
  for (int i = 0; i < c->batch; ++i) {
    int rep = i;
    for (int j = 0; j < i; ++j)
      if (flags[j] != 0.0 && !std::memcmp(aperture + (size_t)j * AP_STRIDE, aperture + (size_t)i * AP_STRIDE, AP_STRIDE * sizeof(double))) { rep = j; break; }
    power_of[i] = (double)rep;
    compute[i] = (flags[i] != 0.0 && rep == i) ? 1.0 : 0.0;
  }
  const double *dp = nullptr, *ds = nullptr, *dcompute = nullptr, *dpower_of = nullptr;
  int rc;
  if ((rc = arena_push(c, aperture, (size_t)c->batch * AP_STRIDE, &dp))) return rc;
  if ((rc = arena_push(c, flags.data(), flags.size(), &ds))) return rc;
  if ((rc = arena_push(c, compute.data(), compute.size(), &dcompute))) return rc;
  if ((rc = arena_push(c, power_of.data(), power_of.size(), &dpower_of))) return rc;
  const double* drows = nullptr;
  if (write_rows) {
    if ((rc = check_rows(c, write_rows))) return rc;
    if ((rc = arena_push(c, write_rows, (size_t)2 * c->batch, &drows))) return rc;
  }
  const double* dcols = nullptr;
  if (write_cols) {
    if (!write_rows) return fail(c, PAOS_EINVAL, "a column window needs a row window");
    if ((rc = check_rows(c, write_cols))) return rc;
    const std::vector<double> wc = rounded_cols(c, write_cols);
    if ((rc = arena_push(c, wc.data(), wc.size(), &dcols))) return rc;
  }
  // groups of items that start from the same field (start_write_kernel)
  static const bool share_start = [] { const char* e = getenv("PAOS_SHARE_START"); return !(e && e[0] == '0'); }();
  std::vector<double> goff(c->batch, 0.0), glen(c->batch, 0.0), members;
  {
    std::vector<int> lead(c->batch);
    for (int i = 0; i < c->batch; ++i) {
      lead[i] = i;
      if (!share_start) continue;
      for (int j = 0; j < i; ++j)
        if (lead[j] == j && flags[j] == flags[i] &&
            !std::memcmp(aperture + (size_t)j * AP_STRIDE, aperture + (size_t)i * AP_STRIDE, AP_STRIDE * sizeof(double)) &&
            (!write_rows || !std::memcmp(write_rows + 2 * j, write_rows + 2 * i, 2 * sizeof(double))) &&
            (!write_cols || !std::memcmp(write_cols + 2 * j, write_cols + 2 * i, 2 * sizeof(double)))) { lead[i] = j; break; }
    }
    for (int i = 0; i < c->batch; ++i) {
      if (lead[i] != i) continue;
      goff[i] = (double)members.size();
      for (int j = i; j < c->batch; ++j)
        if (lead[j] == i) members.push_back((double)j);
      glen[i] = (double)members.size() - goff[i];
    }
  }
  const double *dgoff = nullptr, *dglen = nullptr, *dmembers = nullptr;
  if ((rc = arena_push(c, goff.data(), goff.size(), &dgoff))) return rc;
  if ((rc = arena_push(c, glen.data(), glen.size(), &dglen))) return rc;
  if ((rc = arena_push(c, members.data(), members.size(), &dmembers))) return rc;
  const dim3 block(kPwThreads);
  // the power sums: found (same shape, constant, aperture records and stop flags as the last start) or evaluated and kept
  bool power_found = false;
  if (any_stop) {
    static const bool memo = [] { const char* e = getenv("PAOS_START_POWER_MEMO"); return !(e && e[0] == '0'); }();
    std::vector<double> key;
    key.reserve(3 + (size_t)c->batch * (AP_STRIDE + 1));
    key.push_back((double)shape); key.push_back(re); key.push_back(im);
    key.insert(key.end(), aperture, aperture + (size_t)c->batch * AP_STRIDE);
    key.insert(key.end(), flags.begin(), flags.end());
    power_found = memo && c->start_norm2 && key.size() == c->start_key.size() &&
                  !std::memcmp(key.data(), c->start_key.data(), key.size() * sizeof(double));
    if (power_found) {
      HIPCHK(c, hipMemcpyAsync(c->norm2, c->start_norm2, (size_t)c->batch * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else {
      c->start_key.clear();  // (set again behind the launches below)
      if (memo) c->start_key = std::move(key);
    }
  }
#define START_LAUNCH(T, BRV, S)                                                                         \
  do {                                                                                                  \
    if (any_stop && !power_found) {                                                                     \
      hipLaunchKernelGGL((start_power_kernel<T, BRV, Lay<T>::BC, S>), dim3(c->nparts, c->batch), block, 0, \
                         c->stream, dp, c->n, c->pitch, c->item_stride, re, im, c->partial, dcompute);  \
      hipLaunchKernelGGL(norm2_final_kernel, dim3(c->batch), block, 0, c->stream, c->partial, c->norm2,   \
                         c->nparts, ds, 1, dpower_of);                                                  \
    }                                                                                                   \
    hipLaunchKernelGGL((start_write_kernel<T, BRV, Lay<T>::BC, S>), dim3(pw_blocks(c), c->batch), block, 0, \
                       c->stream, (cx<T>*)c->field, dp, c->n, c->pitch, c->item_stride, re, im,          \
                       (const double*)c->norm2, ds, drows, dgoff, dglen, dmembers, dcols);              \
  } while (0)
  if (c->precision == PAOS_F64) {
    if (shape == PAOS_SHAPE_ELLIPSE) START_LAUNCH(double, BR, 0); else START_LAUNCH(double, BR, 1);
  } else {
    if (shape == PAOS_SHAPE_ELLIPSE) F32_BR_SWITCH(c, START_LAUNCH(float, FBR, 0)); else F32_BR_SWITCH(c, START_LAUNCH(float, FBR, 1));
  }
#undef START_LAUNCH
  HIPCHK(c, hipGetLastError());
  if (any_stop && !power_found && !c->start_key.empty()) {  // keep the sums just evaluated (c->norm2 is rewritten by every reduction)
    if (!c->start_norm2) HIPCHK(c, hipMalloc(&c->start_norm2, (size_t)c->batch * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(c->start_norm2, c->norm2, (size_t)c->batch * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  return PAOS_OK;
}

int paos_start(paos_ctx* c, double re, double im, int shape, const double* aperture, const double* stop) {
  DROP_SCALE(c);
  return start_impl(c, re, im, shape, aperture, stop, nullptr);
}

int paos_start_rows(paos_ctx* c, double re, double im, int shape, const double* aperture, const double* stop,
                    const double* write_rows) {
  DROP_SCALE(c);
  return start_impl(c, re, im, shape, aperture, stop, write_rows);
}

int paos_start_box(paos_ctx* c, double re, double im, int shape, const double* aperture, const double* stop,
                   const double* write_rows, const double* write_cols) {
  DROP_SCALE(c);
  return start_impl(c, re, im, shape, aperture, stop, write_rows, write_cols);
}

int paos_zero_outside_box(paos_ctx* c, const double* live_rows, const double* live_cols) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c || !live_rows) return fail(c, PAOS_EINVAL, "null argument");
  int rc = check_rows(c, live_rows);
  if (rc) return rc;
  if (live_cols && (rc = check_rows(c, live_cols))) return rc;
  return zero_outside_box(c, live_rows, live_cols);
}

int paos_zero_outside_rows(paos_ctx* c, const double* live_rows) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c || !live_rows) return fail(c, PAOS_EINVAL, "null argument");
  int rc = check_rows(c, live_rows);
  if (rc) return rc;
  return zero_outside_rows(c, live_rows);
}

int paos_import(paos_ctx* c, int item, const void* host) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !host || item < 0 || item >= c->batch) return fail(c, PAOS_EINVAL, "bad item or null buffer");
  const size_t bytes = (size_t)c->n * c->n * 16;
  HIPCHK(c, hipMemcpyAsync(c->staging, host, bytes, hipMemcpyHostToDevice, c->stream));
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((import_kernel<double, BR, Lay<double>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (cx<double>*)c->field + (size_t)item * c->item_stride, (const cx<double>*)c->staging,
                       c->n, c->pitch);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((import_kernel<float, FBR, Lay<float>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (cx<float>*)c->field + (size_t)item * c->item_stride, (const cx<double>*)c->staging,
                       c->n, c->pitch));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the host buffer is only borrowed
  return PAOS_OK;
}

static int export_impl(paos_ctx* c, int item, int what, void* host_out, bool pinned) {
  if (!c || !host_out || item < 0 || item >= c->batch || what < 0 || what > 3)
    return fail(c, PAOS_EINVAL, "bad item/what or null buffer");
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((export_kernel<double, BR, Lay<double>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (const cx<double>*)c->field + (size_t)item * c->item_stride, (double*)c->staging,
                       c->n, c->pitch, what);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((export_kernel<float, FBR, Lay<float>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (const cx<float>*)c->field + (size_t)item * c->item_stride, (double*)c->staging,
                       c->n, c->pitch, what));
  HIPCHK(c, hipGetLastError());
  const size_t bytes = (size_t)c->n * c->n * (what == PAOS_WHAT_FIELD ? 16 : 8);
  if (pinned) {  // one DMA into page-locked memory
    HIPCHK(c, hipMemcpyAsync(host_out, c->staging, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  } else {
    int rc = copy_to_host(c, host_out, c->staging, bytes);
    if (rc) return rc;
  }
  return check_mask_overflow(c);
}

int paos_export(paos_ctx* c, int item, int what, void* host_out) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  return export_impl(c, item, what, host_out, false);
}

int paos_export_pinned(paos_ctx* c, int item, int what, void* pinned_out) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  return export_impl(c, item, what, pinned_out, true);
}

int paos_psf_keep(paos_ctx* c) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c) return fail(c, PAOS_EINVAL, "null context");
  if (!c->psf) HIPCHK(c, hipMalloc(&c->psf, (size_t)c->batch * c->item_stride * sizeof(double)));
  c->psf_zero_axis = -1;  // the whole buffer is rewritten
  const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((intensity_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream,
                       (const cx<double>*)c->field, c->psf, c->n, c->pitch, c->item_stride);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((intensity_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream,
                       (const cx<float>*)c->field, c->psf, c->n, c->pitch, c->item_stride));
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

int paos_psf_fetch(paos_ctx* c, int item, double* host_out) {
  if (c) (void)hipSetDevice(c->device);
  if (!c || !host_out || item < 0 || item >= c->batch) return fail(c, PAOS_EINVAL, "bad item or null buffer");
  if (!c->psf) return fail(c, PAOS_EINVAL, "no PSF kept (paos_psf_keep)");
  // the PSF buffer is blocked like the field: row-major through the staging buffer
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((psf_unblock_kernel<BR, Lay<double>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (const double*)c->psf + (size_t)item * c->item_stride, (double*)c->staging, c->n, c->pitch);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((psf_unblock_kernel<FBR, Lay<float>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (const double*)c->psf + (size_t)item * c->item_stride, (double*)c->staging, c->n, c->pitch));
  HIPCHK(c, hipGetLastError());
  return copy_to_host(c, host_out, c->staging, (size_t)c->n * c->n * sizeof(double));
}

int paos_host_alloc(unsigned long long bytes, void** out) {
  if (!out || bytes == 0) return fail(nullptr, PAOS_EINVAL, "null argument");
  *out = nullptr;
  const hipError_t e = hipHostMalloc(out, (size_t)bytes);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(nullptr, PAOS_EHIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  }
  return PAOS_OK;
}

int paos_host_free(void* p) {
  if (!p) return PAOS_OK;
  const hipError_t e = hipHostFree(p);
  if (e != hipSuccess) return fail(nullptr, PAOS_EHIP, std::string("hipHostFree: ") + hipGetErrorString(e));
  return PAOS_OK;
}

static int aperture_launch(paos_ctx* c, int shape, const double* dp, int nitems, double* mask_out) {
  const dim3 grid(pw_blocks(c), nitems), block(kPwThreads);
#define AP_LAUNCH(T, BRV, S)                                                                     \
  hipLaunchKernelGGL((aperture_kernel<T, BRV, Lay<T>::BC, S>), grid, block, 0, c->stream,                \
                     mask_out ? (cx<T>*)nullptr : (cx<T>*)c->field, dp, (const double*)nullptr,    \
                     AP_STRIDE, c->n, c->pitch, c->item_stride, mask_out, 0)
  if (c->precision == PAOS_F64) {
    if (shape == PAOS_SHAPE_ELLIPSE) AP_LAUNCH(double, BR, 0); else AP_LAUNCH(double, BR, 1);
  } else {
    if (shape == PAOS_SHAPE_ELLIPSE) F32_BR_SWITCH(c, AP_LAUNCH(float, FBR, 0)); else F32_BR_SWITCH(c, AP_LAUNCH(float, FBR, 1));
  }
#undef AP_LAUNCH
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

int paos_aperture(paos_ctx* c, int shape, const double* params) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !params) return fail(c, PAOS_EINVAL, "null argument");
  if (shape != PAOS_SHAPE_ELLIPSE && shape != PAOS_SHAPE_RECT) return fail(c, PAOS_EINVAL, "unknown aperture shape");
  const double* dp = nullptr;
  int rc = arena_push(c, params, (size_t)c->batch * AP_STRIDE, &dp);
  if (rc) return rc;
  return aperture_launch(c, shape, dp, c->batch, nullptr);
}

int paos_aperture_render(paos_ctx* c, int shape, const double* params1, double* host_mask) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !params1 || !host_mask) return fail(c, PAOS_EINVAL, "null argument");
  if (shape != PAOS_SHAPE_ELLIPSE && shape != PAOS_SHAPE_RECT) return fail(c, PAOS_EINVAL, "unknown aperture shape");
  const double* dp = nullptr;
  int rc = arena_push(c, params1, AP_STRIDE, &dp);
  if (rc) return rc;
  rc = aperture_launch(c, shape, dp, 1, (double*)c->staging);
  if (rc) return rc;
  return copy_to_host(c, host_mask, c->staging, (size_t)c->n * c->n * 8);
}

static int norm2_launch(paos_ctx* c, const double* den) {
  const dim3 grid(c->nparts, c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((norm2_partial_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream,
                       (const cx<double>*)c->field, c->partial, c->n, c->pitch, c->item_stride, den, 1);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((norm2_partial_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream,
                       (const cx<float>*)c->field, c->partial, c->n, c->pitch, c->item_stride, den, 1));
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(norm2_final_kernel, dim3(c->batch), block, 0, c->stream, c->partial, c->norm2,
                     c->nparts, den, 1);
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

int paos_make_stop(paos_ctx* c, const double* enable) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c) return fail(c, PAOS_EINVAL, "null context");
  const double* den = nullptr;
  if (enable) {
    int rc = arena_push(c, enable, (size_t)c->batch, &den);
    if (rc) return rc;
  }
  int rc = norm2_launch(c, den);
  if (rc) return rc;
  const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL(stop_scale_kernel<double>, grid, block, 0, c->stream, (cx<double>*)c->field,
                       c->norm2, c->item_stride, den, 1);
  else
    hipLaunchKernelGGL(stop_scale_kernel<float>, grid, block, 0, c->stream, (cx<float>*)c->field,
                       c->norm2, c->item_stride, den, 1);
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

int paos_stop_scale_last_power(paos_ctx* c, const double* enable) {
  if (!c) return fail(c, PAOS_EINVAL, "null context");
  // something has touched the field or c->norm2 since the program summed the power (or no program did): reduce now
  if (!c->norm2_of_field) return paos_make_stop(c, enable);
  SETTLE_SCALE(c);  // (clears norm2_of_field: behind the stop c->norm2 no longer is the field's power)
  (void)hipSetDevice(c->device);
  const double* den = nullptr;
  if (enable) {
    int rc = arena_push(c, enable, (size_t)c->batch, &den);
    if (rc) return rc;
  }
  const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL(stop_scale_kernel<double>, grid, block, 0, c->stream, (cx<double>*)c->field, c->norm2, c->item_stride, den, 1);
  else
    hipLaunchKernelGGL(stop_scale_kernel<float>, grid, block, 0, c->stream, (cx<float>*)c->field, c->norm2, c->item_stride, den, 1);
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

int paos_stop_defer_last_power(paos_ctx* c, const double* enable) {
  if (!c) return fail(c, PAOS_EINVAL, "null context");
  // (a second stop with nothing in between lands here too: the first one consumed the flag, so this one reduces the
  // field as it then is -- make_stop settles the first one's factor first)
  if (!c->norm2_of_field) return paos_make_stop(c, enable);
  (void)hipSetDevice(c->device);
  SETTLE_SCALE(c);  // (clears norm2_of_field)
  const double* den = nullptr;
  if (enable) {
    int rc = arena_push(c, enable, (size_t)c->batch, &den);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(dyn_scale_set_kernel, dim3((c->batch + 255) / 256), dim3(256), 0, c->stream, c->dyn_scale, c->norm2, den, c->batch);
  HIPCHK(c, hipGetLastError());
  c->dyn_pending = true;
  return PAOS_OK;
}

int paos_psf_metrics(paos_ctx* c, int nr, const double* radii_px, double cx_px, double cy_px, double* host_out) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !host_out || nr < 0 || nr > kMaxRadii || (nr > 0 && !radii_px)) return fail(c, PAOS_EINVAL, "bad metrics request");
  const int nvals = 4 + nr, nblocks = 512;
  if (!c->metric_partial) {
    HIPCHK(c, hipMalloc(&c->metric_partial, (size_t)c->batch * nblocks * (4 + kMaxRadii) * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->metric_out, (size_t)c->batch * (4 + kMaxRadii) * sizeof(double)));
    HIPCHK(c, hipHostMalloc(&c->metric_host, (size_t)c->batch * (4 + kMaxRadii) * sizeof(double)));
  }
  MetricArgs a{};
  a.field = c->field; a.partial = c->metric_partial; a.n = c->n; a.nr = nr; a.pitch = c->pitch;
  a.item_stride = c->item_stride; a.cx = cx_px; a.cy = cy_px;
  for (int k = 0; k < nr; ++k) a.r2[k] = radii_px[k] * radii_px[k];
  const dim3 grid(nblocks, c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64) hipLaunchKernelGGL((psf_metrics_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream, a);
  else F32_BR_SWITCH(c, hipLaunchKernelGGL((psf_metrics_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream, a));
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(psf_metrics_final_kernel, dim3(c->batch), dim3(64), 0, c->stream, c->metric_partial, c->metric_out, nblocks, nvals);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->metric_host, c->metric_out, (size_t)c->batch * nvals * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(host_out, c->metric_host, (size_t)c->batch * nvals * sizeof(double));
  return check_mask_overflow(c);
}

int paos_norm2_enqueue(paos_ctx* c, int* ticket) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !ticket) return fail(c, PAOS_EINVAL, "null argument");
  const int slot = next_norm_slot(c);
  if (c->norm_busy[slot])  // the ring is full: the oldest ticket has not been fetched
    return fail(c, PAOS_EINVAL, "64 power reductions outstanding: fetch earlier tickets (paos_norm2_fetch) first");
  int rc = norm2_launch(c, nullptr);
  if (rc) return rc;
  c->norm_busy[slot] = true;
  c->norm_slot = (slot + 1) % kNormSlots;
  HIPCHK(c, hipMemcpyAsync(c->norm2_host + (size_t)slot * c->batch, c->norm2, (size_t)c->batch * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  *ticket = slot;
  return PAOS_OK;
}

}  // extern "C"

namespace {

// partial sums -> norm2 -> a ticket of the power ring
int psf_power_ticket(paos_ctx* c, const double* partial, int nparts, int* ticket, const double* source) {
  const int slot = next_norm_slot(c);
  if (c->norm_busy[slot])
    return fail(c, PAOS_EINVAL, "64 power reductions outstanding: fetch earlier tickets (paos_norm2_fetch) first");
  hipLaunchKernelGGL(norm2_final_kernel, dim3(c->batch), dim3(kPwThreads), 0, c->stream, partial, c->norm2, nparts,
                     (const double*)nullptr, 1, source);
  HIPCHK(c, hipGetLastError());
  c->norm_busy[slot] = true;
  c->norm_slot = (slot + 1) % kNormSlots;
  HIPCHK(c, hipMemcpyAsync(c->norm2_host + (size_t)slot * c->batch, c->norm2, (size_t)c->batch * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  *ticket = slot;
  return PAOS_OK;
}

int psf_keep_power_impl(paos_ctx* c, int* ticket) {
  if (c->norm_busy[next_norm_slot(c)])
    return fail(c, PAOS_EINVAL, "64 power reductions outstanding: fetch earlier tickets (paos_norm2_fetch) first");
  if (!c->psf) HIPCHK(c, hipMalloc(&c->psf, (size_t)c->batch * c->item_stride * sizeof(double)));
  c->psf_zero_axis = -1;  // the whole buffer is rewritten
  const dim3 grid(c->nparts, c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((intensity_power_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream,
                       (const cx<double>*)c->field, c->psf, c->partial, c->n, c->pitch, c->item_stride);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((intensity_power_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream,
                       (const cx<float>*)c->field, c->psf, c->partial, c->n, c->pitch, c->item_stride));
  HIPCHK(c, hipGetLastError());
  return psf_power_ticket(c, c->partial, c->nparts, ticket);
}

// rows outside [lo, hi) of every item := 0 (whole block rows, which are contiguous in memory)
int zero_outside_rows(paos_ctx* c, const double* live_rows) {
  const size_t eb = elem_bytes(c);
  for (int i = 0; i < c->batch; ++i) {
    const int br = c->br;
    int lo = ((int)live_rows[2 * i] / br) * br, hi = (((int)live_rows[2 * i + 1] + br - 1) / br) * br;
    if (hi > c->n) hi = c->n;
    if (lo >= hi) { lo = 0; hi = 0; }
    char* base = (char*)c->field + (size_t)i * c->item_stride * eb;
    const size_t row_bytes = (size_t)c->pitch / br * eb;  // one row's share of a block row
    if (lo > 0) HIPCHK(c, hipMemsetAsync(base, 0, (size_t)lo * row_bytes, c->stream));
    if (hi < c->n) HIPCHK(c, hipMemsetAsync(base + (size_t)hi * row_bytes, 0, (size_t)(c->n - hi) * row_bytes, c->stream));
  }
  return PAOS_OK;
}

// ... and outside the columns [lo, hi) of the rows in between (round 5: a sweep over the field; this is the rare path --
// something wants to read a field that was started inside its aperture's box only)
int zero_outside_box(paos_ctx* c, const double* live_rows, const double* live_cols) {
  if (!live_cols) return zero_outside_rows(c, live_rows);
  const double *drows = nullptr, *dcols = nullptr;
  int rc;
  if ((rc = arena_push(c, live_rows, (size_t)2 * c->batch, &drows))) return rc;
  const std::vector<double> lc = rounded_cols(c, live_cols);
  if ((rc = arena_push(c, lc.data(), lc.size(), &dcols))) return rc;
  const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((zero_outside_box_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream, (cx<double>*)c->field, c->n,
                       c->pitch, c->item_stride, drows, dcols);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((zero_outside_box_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream,
                                        (cx<float>*)c->field, c->n, c->pitch, c->item_stride, drows, dcols));
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

}  // namespace

extern "C" {

int paos_psf_keep_power(paos_ctx* c, int* ticket) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c || !ticket) return fail(c, PAOS_EINVAL, "null argument");
  return psf_keep_power_impl(c, ticket);
}

int paos_run_program(paos_ctx* c, const paos_pass* passes, int n_passes, const double* blocks, int n_blocks,
                     const paos_program_opts* opts) {
  if (c) (void)hipSetDevice(c->device);
  if (!opts) return run_passes(c, passes, n_passes, blocks, n_blocks);
  if (c && opts->live_rows) {
    int rc = check_rows(c, opts->live_rows);
    if (rc) return rc;
  }
  if (opts->final_intensity && !opts->power_ticket) return fail(c, PAOS_EINVAL, "final_intensity needs a place for the power ticket");
  if (opts->final_intensity < 0 || opts->final_intensity > 2) return fail(c, PAOS_EINVAL, "final_intensity: 0, 1 (PSF instead of the field) or 2 (field + its power)");
  const bool stale = opts->live_rows && opts->rows_stale != 0;
  if (c && opts->live_cols && stale) {
    int rc = check_rows(c, opts->live_cols);
    if (rc) return rc;
  }
  return run_passes(c, passes, n_passes, blocks, n_blocks, opts->live_rows, stale,
                    opts->final_intensity ? opts->power_ticket : nullptr, opts->final_intensity == 2 ? 2 : 1,
                    stale ? opts->live_cols : nullptr);
}

static int norm2_enqueue_rows_impl(paos_ctx* c, const double* live_rows, const double* same_as, int* ticket,
                                   const double* live_cols = nullptr) {
  if (c) (void)hipSetDevice(c->device);
  if (!c || !ticket || !live_rows) return fail(c, PAOS_EINVAL, "null argument");
  int rc = check_rows(c, live_rows);
  if (rc) return rc;
  const int slot = next_norm_slot(c);
  if (c->norm_busy[slot])
    return fail(c, PAOS_EINVAL, "64 power reductions outstanding: fetch earlier tickets (paos_norm2_fetch) first");
  const double *drows = nullptr, *dlead = nullptr, *dsame = nullptr, *dcols = nullptr;
  if ((rc = arena_push(c, live_rows, (size_t)2 * c->batch, &drows))) return rc;
  if (live_cols) {
    const std::vector<double> lc = rounded_cols(c, live_cols);
    if ((rc = arena_push(c, lc.data(), lc.size(), &dcols))) return rc;
  }
  if (same_as) {  // items whose fields the caller knows to be copies of another item's: summed once
    std::vector<double> lead(c->batch);
    for (int i = 0; i < c->batch; ++i) {
      const int j = (int)same_as[i];
      if (!(same_as[i] >= 0.0) || j >= c->batch || (double)j != same_as[i] || (int)same_as[j] != j)
        return fail(c, PAOS_EINVAL, "same_as must name an item that stands for itself");
      if (live_rows[2 * i] != live_rows[2 * j] || live_rows[2 * i + 1] != live_rows[2 * j + 1] ||
          (live_cols && (live_cols[2 * i] != live_cols[2 * j] || live_cols[2 * i + 1] != live_cols[2 * j + 1])))
        return fail(c, PAOS_EINVAL, "items that share a sum must share their row (and column) window");
      lead[i] = j == i ? 1.0 : 0.0;
    }
    if ((rc = arena_push(c, lead.data(), lead.size(), &dlead))) return rc;
    if ((rc = arena_push(c, same_as, (size_t)c->batch, &dsame))) return rc;
  }
  const dim3 grid(c->nparts, c->batch), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((norm2_partial_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream,
                       (const cx<double>*)c->field, c->partial, c->n, c->pitch, c->item_stride, dlead, 1, drows, dcols);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((norm2_partial_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream,
                       (const cx<float>*)c->field, c->partial, c->n, c->pitch, c->item_stride, dlead, 1, drows, dcols));
  HIPCHK(c, hipGetLastError());
  return psf_power_ticket(c, c->partial, c->nparts, ticket, dsame);
}

int paos_norm2_enqueue_box(paos_ctx* c, const double* live_rows, const double* live_cols, const double* same_as, int* ticket) {
  SETTLE_SCALE(c);
  if (c && live_cols) {
    int rc = check_rows(c, live_cols);
    if (rc) return rc;
  }
  return norm2_enqueue_rows_impl(c, live_rows, same_as, ticket, live_cols);
}

int paos_norm2_enqueue_rows(paos_ctx* c, const double* live_rows, int* ticket) {
  SETTLE_SCALE(c);
  return norm2_enqueue_rows_impl(c, live_rows, nullptr, ticket);
}

int paos_norm2_enqueue_rows_like(paos_ctx* c, const double* live_rows, const double* same_as, int* ticket) {
  SETTLE_SCALE(c);
  if (!same_as) return fail(c, PAOS_EINVAL, "null argument");
  return norm2_enqueue_rows_impl(c, live_rows, same_as, ticket);
}

int paos_norm2_fetch(paos_ctx* c, int ticket, double* host_out) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !host_out || ticket < 0 || ticket >= kNormSlots) return fail(c, PAOS_EINVAL, "bad ticket");
  if (!c->norm_busy[ticket]) return fail(c, PAOS_EINVAL, "ticket is not outstanding (already fetched, or never issued)");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->norm_busy[ticket] = false;
  std::memcpy(host_out, c->norm2_host + (size_t)ticket * c->batch, (size_t)c->batch * sizeof(double));
  return check_mask_overflow(c);
}

int paos_norm2_release(paos_ctx* c, int ticket) {
  if (!c || ticket < 0 || ticket >= kNormSlots) return fail(c, PAOS_EINVAL, "bad ticket");
  c->norm_busy[ticket] = false;  // the caller does not want the value; the slot may be handed out again
  return PAOS_OK;
}

int paos_norm2(paos_ctx* c, double* host_out) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !host_out) return fail(c, PAOS_EINVAL, "null argument");
  int rc = norm2_launch(c, nullptr);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->norm2_host, c->norm2, (size_t)c->batch * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(host_out, c->norm2_host, (size_t)c->batch * sizeof(double));
  return PAOS_OK;
}

int paos_phase(paos_ctx* c, const double* params, int mul2pi) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  if (!c || !params) return fail(c, PAOS_EINVAL, "null argument");
  Program g(c->batch);
  const int par = g.add_block(params, 0, 0, 0, 0, true);
  paos_pass& p = g.add_pass(-1, -1, -1);
  push(p.pre, p.n_pre, PAOS_PW_QPHASE_CENTRED, par, mul2pi ? PAOS_PWF_MUL2PI : 0);
  return run_passes(c, g.passes.data(), 1, g.blocks.data(), 1);
}

int paos_phase_map(paos_ctx* c, int item, const double* host_wfe, double wl) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c || !host_wfe || item < 0 || item >= c->batch) return fail(c, PAOS_EINVAL, "bad item or null buffer");
  if (!(wl > 0.0) || !std::isfinite(wl)) return fail(c, PAOS_EINVAL, "wavelength must be positive and finite");
  const size_t count = (size_t)c->n * c->n;
  for (size_t i = 0; i < count; ++i)  // the device sincos handles any finite argument; reject the rest here
    if (!std::isfinite(host_wfe[i])) return fail(c, PAOS_EINVAL, "the phase map holds a non-finite value (fill masked pixels with 0)");
  HIPCHK(c, hipMemcpyAsync(c->staging, host_wfe, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((phase_map_kernel<double, BR, Lay<double>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (cx<double>*)c->field + (size_t)item * c->item_stride, (const double*)c->staging, c->n, c->pitch, wl);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((phase_map_kernel<float, FBR, Lay<float>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                       (cx<float>*)c->field + (size_t)item * c->item_stride, (const double*)c->staging, c->n, c->pitch, wl));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the host buffer is only borrowed
  return PAOS_OK;
}

int paos_phase_map_items(paos_ctx* c, const double* host_wfe, unsigned long long key, int n_items, const double* items,
                         const double* wl) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c || !items || !wl || n_items < 1 || n_items > c->batch) return fail(c, PAOS_EINVAL, "bad item list or null buffer");
  if (!host_wfe && (key == 0 || key != c->map_key || !c->map_dev)) return fail(c, PAOS_EINVAL, "no host map, and no map kept on the device under this key");
  for (int k = 0; k < n_items; ++k) {
    if (!(items[k] >= 0.0) || items[k] >= (double)c->batch || items[k] != (double)(int)items[k]) return fail(c, PAOS_EINVAL, "bad item index");
    if (!(wl[k] > 0.0) || !std::isfinite(wl[k])) return fail(c, PAOS_EINVAL, "wavelength must be positive and finite");
  }
  const size_t count = (size_t)c->n * c->n;
  if (!c->map_dev) HIPCHK(c, hipMalloc(&c->map_dev, count * sizeof(double)));
  if (key == 0 || key != c->map_key) {  // (the same key again: the caller vouches that the map is the one uploaded under it)
    for (size_t i = 0; i < count; ++i)  // the device sincos handles any finite argument; reject the rest here
      if (!std::isfinite(host_wfe[i])) return fail(c, PAOS_EINVAL, "the phase map holds a non-finite value (fill masked pixels with 0)");
    c->map_key = 0;
    HIPCHK(c, hipMemcpyAsync(c->map_dev, host_wfe, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the host buffer is only borrowed
    c->map_key = key;
  }
  const double *ditems = nullptr, *dwl = nullptr;
  int rc;
  if ((rc = arena_push(c, items, (size_t)n_items, &ditems))) return rc;
  if ((rc = arena_push(c, wl, (size_t)n_items, &dwl))) return rc;
  const dim3 grid(pw_blocks(c), n_items), block(kPwThreads);
  if (c->precision == PAOS_F64)
    hipLaunchKernelGGL((phase_map_items_kernel<double, BR, Lay<double>::BC>), grid, block, 0, c->stream, (cx<double>*)c->field,
                       c->item_stride, (const double*)c->map_dev, c->n, c->pitch, ditems, dwl);
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((phase_map_items_kernel<float, FBR, Lay<float>::BC>), grid, block, 0, c->stream,
                                        (cx<float>*)c->field, c->item_stride, (const double*)c->map_dev, c->n, c->pitch, ditems, dwl));
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

int paos_psd_screen(paos_ctx* c, const double* host_noise, const double* host_rough, const double* params, unsigned long long key,
                    double* host_out) {
  if (c) (void)hipSetDevice(c->device);
  if (!c || !host_noise || !params || key == 0) return fail(c, PAOS_EINVAL, "null argument (or key 0)");
  if (c->precision != PAOS_F64) return fail(c, PAOS_EUNSUPPORTED, "PSD screens are built on complex128 contexts (build the map on the host for fp32 mode)");
  PsdParams p{params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8], params[9], params[10], params[11]};
  for (int k = 0; k < 12; ++k)
    if (std::isnan(params[k])) return fail(c, PAOS_EINVAL, "a PSD parameter is NaN");
  const size_t count = (size_t)c->n * c->n;
  if (!c->map_dev) HIPCHK(c, hipMalloc(&c->map_dev, count * sizeof(double)));
  if (!c->psd_scratch) HIPCHK(c, hipMalloc(&c->psd_scratch, (size_t)c->item_stride * sizeof(cx<double>)));
  if (!c->psd_bad) HIPCHK(c, hipMalloc(&c->psd_bad, sizeof(int)));
  c->map_key = 0;
  double* noise = (double*)c->staging;  // n x n x 16 bytes: the white noise, then the roughness draw
  double* rough = (host_rough && p.SR != 0.0) ? noise + count : nullptr;
  HIPCHK(c, hipMemcpyAsync(noise, host_noise, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (rough) HIPCHK(c, hipMemcpyAsync(rough, host_rough, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->psd_bad, 0, sizeof(int), c->stream));
  // control blocks of the four passes (one item): forward, inverse, scale 1 / n
  std::vector<double> hb((size_t)3 * FP_STRIDE, 0.0);
  hb[0 * FP_STRIDE + FP_ENABLE] = 1.0;
  hb[1 * FP_STRIDE + FP_ENABLE] = 1.0; hb[1 * FP_STRIDE + 1] = 1.0;
  hb[2 * FP_STRIDE + FP_ENABLE] = 1.0; hb[2 * FP_STRIDE + 3] = 1.0 / c->n;
  const double* dblocks = nullptr;
  int rc;
  if ((rc = arena_push(c, hb.data(), hb.size(), &dblocks))) return rc;
  const dim3 grid(pw_blocks(c)), block(kPwThreads);
  hipLaunchKernelGGL((psd_load_kernel<BR, Lay<double>::BC>), grid, block, 0, c->stream, c->psd_scratch, (const double*)noise, c->n, c->pitch);
  PassArgs a{};
  a.field = c->psd_scratch; a.tw = c->tw; a.blocks = dblocks; a.tables = nullptr; a.mask = nullptr; a.batch = 1;
  a.fft1 = 0; a.fft2 = -1; a.pitch = c->pitch; a.item_stride = c->item_stride;
  for (int axis = 0; axis < 2; ++axis)  // spectrum = fft2(noise)
    if ((rc = pass_t<double>(c, axis, a, 0))) return rc;
  hipLaunchKernelGGL((psd_filter_kernel<BR, Lay<double>::BC>), grid, block, 0, c->stream, c->psd_scratch, c->n, c->pitch, p);
  a.fft1 = 1; a.n_mid = 1; a.mid[0] = {PAOS_PW_SCALE, 0, 2};
  for (int axis = 0; axis < 2; ++axis)  // ifft2: each axis carries its 1 / n
    if ((rc = pass_t<double>(c, axis, a, 0))) return rc;
  hipLaunchKernelGGL((psd_finish_kernel<BR, Lay<double>::BC>), grid, block, 0, c->stream, c->map_dev, (const cx<double>*)c->psd_scratch,
                     (const double*)rough, c->n, c->pitch, p, c->psd_bad);
  HIPCHK(c, hipGetLastError());
  int bad = 0;
  HIPCHK(c, hipMemcpyAsync(&bad, c->psd_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // (the host buffers were only borrowed)
  if (bad) return fail(c, PAOS_EINVAL, "the PSD screen holds a non-finite value");
  if (host_out) {
    rc = copy_to_host(c, host_out, c->map_dev, count * sizeof(double));
    if (rc) return rc;
  }
  c->map_key = key;
  return PAOS_OK;
}

int paos_run_passes(paos_ctx* c, const paos_pass* passes, int n_passes, const double* blocks, int n_blocks) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  return run_passes(c, passes, n_passes, blocks, n_blocks);
}

int paos_record_set_stats(paos_ctx* c, unsigned long long* found, unsigned long long* rendered) {
  if (!c || !found || !rendered) return fail(c, PAOS_EINVAL, "bad record-set request");
  *found = c->mask_hits;
  *rendered = c->mask_rendered;
  return PAOS_OK;
}

int paos_copy_yardstick(paos_ctx* c, int reps, double* ms_per_launch, double* bytes_per_launch) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c || !ms_per_launch || !bytes_per_launch || reps < 1) return fail(c, PAOS_EINVAL, "bad yardstick request");
  const size_t total = (size_t)c->item_stride * c->batch;
  const int blocks = (int)((total + kPwThreads - 1) / kPwThreads < 65536 * 4 ? (total + kPwThreads - 1) / kPwThreads : 65536 * 4);
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  auto launch = [&] {
    if (c->precision == PAOS_F64)
      hipLaunchKernelGGL(rmw_copy_kernel<double>, dim3(blocks), dim3(kPwThreads), 0, c->stream, (cx<double>*)c->field, total);
    else
      hipLaunchKernelGGL(rmw_copy_kernel<float>, dim3(blocks), dim3(kPwThreads), 0, c->stream, (cx<float>*)c->field, total);
  };
  launch();  // warm
  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int r = 0; r < reps; ++r) launch();
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  *bytes_per_launch = 2.0 * (double)total * (double)elem_bytes(c);  // pitch padding included: it is moved too
  return PAOS_OK;
}

int paos_ctx_set_pruning(paos_ctx* c, int on) {
  if (!c) return fail(c, PAOS_EINVAL, "null context");
  c->prune = on != 0;
  return PAOS_OK;
}

int paos_run_passes_live(paos_ctx* c, const paos_pass* passes, int n_passes, const double* blocks, int n_blocks,
                         const double* live_rows) {
  if (c) (void)hipSetDevice(c->device);
  if (c && live_rows)
    for (int i = 0; i < c->batch; ++i)
      if (!(live_rows[2 * i] >= 0.0) || !(live_rows[2 * i + 1] <= (double)c->n) || !(live_rows[2 * i] <= live_rows[2 * i + 1]))
        return fail(c, PAOS_EINVAL, "live row range must satisfy 0 <= lo <= hi <= n");
  return run_passes(c, passes, n_passes, blocks, n_blocks, live_rows);
}

int paos_ptp(paos_ctx* c, const double* params) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  return fft_op(c, OP_PTP, params, 0);
}
int paos_stw(paos_ctx* c, const double* params, int inverse) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  return fft_op(c, OP_STW, params, inverse);
}
int paos_wts(paos_ctx* c, const double* params, int inverse) {
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  return fft_op(c, OP_WTS, params, inverse);
}

static int zernike_apply(paos_ctx* c, int nmax, int kdim, const double* table, const double* params,
                         int param_stride, double* host_wfe, bool use_pupil, const double* same_as = nullptr) {
  if (!c || !table || !params) return fail(c, PAOS_EINVAL, "null argument");
  if (nmax < 0 || kdim < nmax / 2 + 1 || param_stride < ZP_HEAD + 2 * (nmax + 1) * kdim)
    return fail(c, PAOS_EINVAL, "inconsistent Zernike table dimensions");
  if (use_pupil && !c->mask) return fail(c, PAOS_EINVAL, "no pupil defined (paos_pupil_aperture / paos_pupil_upload)");
  const double *dt = nullptr, *dp = nullptr;
  int rc = arena_push(c, table, (size_t)(nmax + 1) * kdim * 3, &dt);
  if (rc) return rc;
  rc = arena_push(c, params, (size_t)c->batch * param_stride, &dp);
  if (rc) return rc;
  // groups of items with one wfe map: records equal in everything but the wavelength (and no per-item pupil)
  static const bool share_wfe = [] { const char* e = getenv("PAOS_SHARE_WFE"); return !(e && e[0] == '0'); }();
  std::vector<double> goff(c->batch, 0.0), glen(c->batch, 0.0), members;
  {
    std::vector<int> lead(c->batch, -1);
    auto same_map = [&](int a, int b) {
      const double *qa = params + (size_t)a * param_stride, *qb = params + (size_t)b * param_stride;
      for (int k = 0; k < param_stride; ++k)
        if (k != ZP_INV_WL && std::memcmp(qa + k, qb + k, sizeof(double))) return false;
      return true;
    };
    for (int i = 0; i < c->batch; ++i) {
      if (params[(size_t)i * param_stride + ZP_ENABLE] == 0.0) continue;
      lead[i] = i;
      if (share_wfe && !use_pupil)
        for (int j = 0; j < i; ++j)
          if (lead[j] == j && same_map(j, i)) { lead[i] = j; break; }
    }
    for (int i = 0; i < c->batch; ++i) {
      if (lead[i] != i) continue;
      goff[i] = (double)members.size();
      for (int j = i; j < c->batch; ++j)
        if (lead[j] == i) members.push_back((double)j);
      glen[i] = (double)members.size() - goff[i];
    }
    if (members.empty()) members.push_back(0.0);
  }
  const double *dgoff = nullptr, *dglen = nullptr, *dmembers = nullptr, *dtwins = nullptr;
  if ((rc = arena_push(c, goff.data(), goff.size(), &dgoff))) return rc;
  if ((rc = arena_push(c, glen.data(), glen.size(), &dglen))) return rc;
  if ((rc = arena_push(c, members.data(), members.size(), &dmembers))) return rc;
  if (same_as) {  // a group whose members all hold copies of one field reads the leader's (paos_zernike_like)
    std::vector<double> twins(c->batch, 0.0);
    bool any = false;
    for (int i = 0; i < c->batch; ++i) {
      if (!(same_as[i] >= 0.0) || same_as[i] >= (double)c->batch || same_as[i] != (double)(int)same_as[i])
        return fail(c, PAOS_EINVAL, "same_as must hold item indices");
      if (glen[i] < 2.0) continue;
      bool all = true;
      for (int g = 0; g < (int)glen[i]; ++g) all = all && same_as[(int)members[(size_t)goff[i] + g]] == same_as[i];
      twins[i] = all ? 1.0 : 0.0;
      any = any || all;
    }
    if (any && (rc = arena_push(c, twins.data(), twins.size(), &dtwins))) return rc;
  }
  const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
  double* wfe = host_wfe ? (double*)c->staging : nullptr;
  const double* pupil = use_pupil ? c->mask : nullptr;
  // The kernel changes pixels with rho <= 1 only: |row - n/2| dy <= radius.  Bound the rows of the union of the
  // items' disks (two pixels of margin, whole block rows) and walk only that part of memory -- unless the caller
  // wants the wfe map, which is written (NaN) outside the disk too.
  unsigned m_first = 0, m_end = c->item_stride;
  if (!host_wfe) {
    double half = -1.0;
    bool known = true;
    for (int i = 0; i < c->batch; ++i) {
      const double* q = params + (size_t)i * param_stride;
      if (q[ZP_ENABLE] == 0.0) continue;
      const double h = q[ZP_RADIUS] / q[ZP_DY];
      if (!(h >= 0.0) || !std::isfinite(h)) { known = false; break; }
      if (h > half) half = h;
    }
    if (known && half >= 0.0 && half < (double)c->n) {
      int lo = (int)std::floor((double)(c->n / 2) - half) - 2, hi = (int)std::ceil((double)(c->n / 2) + half) + 3;
      lo = lo < 0 ? 0 : (lo / c->br) * c->br;
      hi = hi > c->n ? c->n : ((hi + c->br - 1) / c->br) * c->br;
      if (hi > c->n) hi = c->n;
      m_first = (unsigned)(lo / c->br) * c->pitch;
      m_end = (unsigned)(hi / c->br) * c->pitch;
    }
  }
  // orders up to 8 (45 polynomials) run on the unrolled build
#define ZK_LAUNCH(T, BRV, NC)                                                                                          \
  hipLaunchKernelGGL((zernike_kernel<T, BRV, Lay<T>::BC, NC>), grid, block, 0, c->stream, (cx<T>*)c->field, dt, dp,   \
                     param_stride, c->n, c->pitch, c->item_stride, nmax, kdim, wfe, pupil, m_first, m_end, dgoff, dglen, \
                     dmembers, dtwins)
  if (c->precision == PAOS_F64) {
    if (nmax <= 8) ZK_LAUNCH(double, BR, 8); else ZK_LAUNCH(double, BR, 0);
  } else {
    if (nmax <= 8) F32_BR_SWITCH(c, ZK_LAUNCH(float, FBR, 8)); else F32_BR_SWITCH(c, ZK_LAUNCH(float, FBR, 0));
  }
#undef ZK_LAUNCH
  HIPCHK(c, hipGetLastError());
  if (host_wfe) return copy_to_host(c, host_wfe, c->staging, (size_t)c->n * c->n * 8);
  return PAOS_OK;
}

int paos_zernike(paos_ctx* c, int nmax, int kdim, const double* table, const double* params,
                 int param_stride, double* host_wfe) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);  // one process may drive several GPUs
  return zernike_apply(c, nmax, kdim, table, params, param_stride, host_wfe, false);
}

int paos_zernike_like(paos_ctx* c, int nmax, int kdim, const double* table, const double* params,
                      int param_stride, const double* same_as, double* host_wfe) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  return zernike_apply(c, nmax, kdim, table, params, param_stride, host_wfe, false, same_as);
}

int paos_zernike_pupil(paos_ctx* c, int nmax, int kdim, const double* table, const double* params,
                       int param_stride, double* host_wfe) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  return zernike_apply(c, nmax, kdim, table, params, param_stride, host_wfe, true);
}

static int ensure_pupil(paos_ctx* c) {
  if (!c->mask) HIPCHK(c, hipMalloc(&c->mask, (size_t)c->batch * c->item_stride * sizeof(double)));
  return PAOS_OK;
}

int paos_pupil_aperture(paos_ctx* c, int shape, const double* params) {
  if (c) (void)hipSetDevice(c->device);
  if (!c || !params) return fail(c, PAOS_EINVAL, "null argument");
  if (shape != PAOS_SHAPE_ELLIPSE && shape != PAOS_SHAPE_RECT) return fail(c, PAOS_EINVAL, "unknown aperture shape");
  int rc = ensure_pupil(c);
  if (rc) return rc;
  const double* dp = nullptr;
  rc = arena_push(c, params, (size_t)c->batch * AP_STRIDE, &dp);
  if (rc) return rc;
  const dim3 grid(pw_blocks(c), c->batch), block(kPwThreads);
  // the mask values of the aperture OBJECT, whatever its obscuration flag says (run.py:136-141)
  // (the element type is irrelevant without a field; the block height is the context's)
  if (shape == PAOS_SHAPE_ELLIPSE)
    F32_BR_SWITCH(c, hipLaunchKernelGGL((aperture_kernel<double, FBR, Lay<double>::BC, 0>), grid, block, 0, c->stream, (cx<double>*)nullptr,
                       dp, (const double*)nullptr, AP_STRIDE, c->n, c->pitch, c->item_stride, c->mask, 2));
  else
    F32_BR_SWITCH(c, hipLaunchKernelGGL((aperture_kernel<double, FBR, Lay<double>::BC, 1>), grid, block, 0, c->stream, (cx<double>*)nullptr,
                       dp, (const double*)nullptr, AP_STRIDE, c->n, c->pitch, c->item_stride, c->mask, 2));
  HIPCHK(c, hipGetLastError());
  return PAOS_OK;
}

int paos_pupil_upload(paos_ctx* c, int item, const double* host_weights) {
  if (c) (void)hipSetDevice(c->device);
  if (!c || !host_weights) return fail(c, PAOS_EINVAL, "null argument");
  if (item < 0 || item >= c->batch) return fail(c, PAOS_EINVAL, "item out of range");
  int rc = ensure_pupil(c);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->staging, host_weights, (size_t)c->n * c->n * 8, hipMemcpyHostToDevice, c->stream));
  F32_BR_SWITCH(c, hipLaunchKernelGGL((import_weights_kernel<FBR, Lay<double>::BC>), dim3(pw_blocks(c)), dim3(kPwThreads), 0, c->stream,
                     (const double*)c->staging, c->mask + (size_t)item * c->item_stride, c->n, c->pitch, c->item_stride));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // host_weights is borrowed
  return PAOS_OK;
}

int paos_zernike_gram(paos_ctx* c, int nmax, int kdim, const double* table, const double* params,
                      int param_stride, int K, const double* poly, int use_pupil, double* host_out) {
  SETTLE_SCALE(c);
  if (c) (void)hipSetDevice(c->device);
  if (!c || !table || !params || !poly || !host_out) return fail(c, PAOS_EINVAL, "null argument");
  if (nmax < 0 || kdim < nmax / 2 + 1 || param_stride < ZP_HEAD)
    return fail(c, PAOS_EINVAL, "inconsistent Zernike table dimensions");
  if (K < 1 || K > kGramMaxK) return fail(c, PAOS_EUNSUPPORTED, "1 <= K <= 64 polynomials");
  if (use_pupil && !c->mask) return fail(c, PAOS_EINVAL, "no pupil defined (paos_pupil_aperture / paos_pupil_upload)");
  // poly[j] = {|m|, k, is_sin, factor}  ->  slot table + factors
  const size_t ncoef = (size_t)(nmax + 1) * kdim;
  std::vector<double> slots(2 * ncoef, -1.0), fac(K);
  for (int j = 0; j < K; ++j) {
    const int am = (int)poly[4 * j], k = (int)poly[4 * j + 1], is_sin = poly[4 * j + 2] != 0.0;
    if (am < 0 || am > nmax || k < 0 || k >= kdim || am + 2 * k > nmax)
      return fail(c, PAOS_EINVAL, "polynomial outside the recurrence table");
    double& slot = slots[2 * ((size_t)am * kdim + k) + (is_sin ? 1 : 0)];
    if (slot >= 0.0) return fail(c, PAOS_EINVAL, "polynomial listed twice");
    slot = (double)j;
    fac[j] = poly[4 * j + 3];
  }
  const double *dt = nullptr, *dp = nullptr, *dslots = nullptr, *dfac = nullptr;
  int rc;
  if ((rc = arena_push(c, table, ncoef * 3, &dt))) return rc;
  if ((rc = arena_push(c, params, (size_t)c->batch * param_stride, &dp))) return rc;
  if ((rc = arena_push(c, slots.data(), slots.size(), &dslots))) return rc;
  if ((rc = arena_push(c, fac.data(), fac.size(), &dfac))) return rc;
  const int nvals = K * (K + 1) / 2 + 1;
  const size_t chunks = ((size_t)c->item_stride + kGramPix - 1) / kGramPix;
  const int nblocks = (int)(chunks < 512 ? chunks : 512);
  double *partial = nullptr, *sums = nullptr;
  HIPCHK(c, hipMalloc(&partial, (size_t)c->batch * nblocks * nvals * sizeof(double)));
  if (hipMalloc(&sums, (size_t)c->batch * nvals * sizeof(double)) != hipSuccess) {
    (void)hipFree(partial);
    return fail(c, PAOS_EHIP, "hipMalloc(gram sums)");
  }
  const size_t lds = (size_t)K * kGramRow * sizeof(double);
  auto kern = c->br == PAOS_F32_BR ? zernike_gram_kernel<PAOS_F32_BR, Lay<double>::BC> : zernike_gram_kernel<BR, Lay<double>::BC>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(kern, dim3(nblocks, c->batch), dim3(kGramThreads), lds, c->stream, dt, dp, param_stride, c->n,
                       c->pitch, c->item_stride, nmax, kdim, K, dslots, dfac,
                       use_pupil ? (const double*)c->mask : (const double*)nullptr, partial);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(zernike_gram_final_kernel, dim3((nvals + 255) / 256, c->batch), dim3(256), 0, c->stream,
                       (const double*)partial, sums, nblocks, nvals);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(host_out, sums, (size_t)c->batch * nvals * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(partial);
  (void)hipFree(sums);
  if (e != hipSuccess) return fail(c, PAOS_EHIP, std::string("zernike gram: ") + hipGetErrorString(e));
  return PAOS_OK;
}

}  // extern "C"

#endif  // PAOS_PART <= 0
