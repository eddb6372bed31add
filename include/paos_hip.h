/* paos_hip.h -- C ABI of libpaoship.so, the MI355X (gfx950) wavefront-propagation core.
 *
 * The reference (arielmission-space/PAOS v1.2.12) is pure Python and has no FFI;
 * the boundary this library replaces is the field arithmetic inside the methods of
 * paos.classes.wfo.WFO (paos/classes/wfo.py) as driven by paos.core.run.run
 * (paos/core/run.py:30-228).  Each entry point below names the reference lines it
 * stands in for.  The scalar "pilot Gaussian beam" bookkeeping of those methods
 * stays on the host (paos_amd/planner.py); only N x N field work crosses this ABI.
 *
 * Conventions
 *   - every function returns 0 on success, a PAOS_E* code otherwise;
 *     paos_last_error(ctx) gives the message (ctx may be NULL for create errors);
 *   - a context owns `batch` fields of n x n complex numbers (double or float) in
 *     HBM, one HIP stream, and all scratch; calls are enqueued on that stream and
 *     return without waiting unless they hand data to the host;
 *   - per-item parameter blocks are plain `double` arrays, `batch` blocks long,
 *     borrowed for the duration of the call; block[0] is an enable flag (0 = leave
 *     that batch item untouched) so that one launch serves wavelengths / Monte-Carlo
 *     draws whose planners disagree on whether a step runs;
 *   - host field buffers are row-major [y][x] complex128, like WFO._wfo;
 *   - contexts are not thread-safe; different contexts are independent.
 */
#ifndef PAOS_HIP_H
#define PAOS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct paos_ctx paos_ctx;

enum { PAOS_OK = 0, PAOS_EINVAL = 1, PAOS_EHIP = 2, PAOS_EUNSUPPORTED = 3 };
enum { PAOS_F64 = 0, PAOS_F32 = 1 };
enum { PAOS_SHAPE_ELLIPSE = 0, PAOS_SHAPE_RECT = 1 };
enum { PAOS_KERNEL_PASS_ROWS = 0, PAOS_KERNEL_PASS_COLS = 1, PAOS_KERNEL_PASS_ANY = 2 };
/* pointwise operators that ride on an FFT pass (paos_run_passes) */
enum { PAOS_PW_SIGN = 1, PAOS_PW_QPHASE_CENTRED = 2, PAOS_PW_QPHASE_NATURAL = 3, PAOS_PW_SCALE = 4,
       PAOS_PW_MASK = 5 };
enum { PAOS_PWF_MUL2PI = 1,
       /* PAOS_PW_SIGN only (round 4, the separable pass programs): (-1)^column, resp. (-1)^row, instead of the
          checkerboard (-1)^(row + column) -- the two factors the checkerboard of wfo.py:491-545 splits into */
       PAOS_PWF_X_ONLY = 2, PAOS_PWF_Y_ONLY = 4 };
enum { PAOS_MAX_PW = 6 };
enum { PAOS_NORM_SLOTS = 64 };  /* outstanding paos_norm2_enqueue tickets */
enum { PAOS_WHAT_FIELD = 0, PAOS_WHAT_AMPLITUDE = 1, PAOS_WHAT_PHASE = 2, PAOS_WHAT_INTENSITY = 3 };

/* parameter-block layouts (doubles per batch item) */
enum { PAOS_PHASE_STRIDE = 5 };    /* enable, sx, sy, coef, sgn                         */
enum { PAOS_APERTURE_STRIDE = 8 }; /* enable, xc, yc, a|w, b|h, theta, obscuration, subpixels */
enum { PAOS_ZERNIKE_HEAD = 8 };    /* enable, dx, dy, radius, origin_is_y, cos_off, sin_off, 1/wl;
                                      then coefC[(nmax+1)*kdim], coefS[(nmax+1)*kdim]           */

/* One HBM pass over every field of the batch:
 *   load -> pre operators -> [1-D FFTs along `axis`] -> mid operators -> [1-D FFTs] -> post -> store.
 * `block` / `fft1` / `fft2` index parameter block sets of PAOS_PHASE_STRIDE doubles per batch
 * item: operators read [enable, sx, sy, coef, sgn] (SIGN uses enable only, SCALE multiplies by
 * coef); a transform control block reads [enable, inverse].  MASK (WFO.aperture, wfo.py:236-276,
 * riding on a pass) reads two consecutive block sets: `block` = [enable, xc, yc, a|w, b|h] and
 * `block`+1 = [theta, obscuration, subpixels, shape, 0]; at most one MASK per pass.  axis = -1: no transform, the operators of `pre` are applied in a stand-alone pass. */
typedef struct { int kind, flags, block; } paos_pw_op;
typedef struct {
  int axis;                    /* 0 = along rows, 1 = along columns, -1 = no transform */
  int fft1, fft2;              /* control block index, or -1 for "no transform here"  */
  int n_pre, n_mid, n_post;
  paos_pw_op pre[PAOS_MAX_PW], mid[PAOS_MAX_PW], post[PAOS_MAX_PW];
} paos_pass;

/* Options of paos_run_program (round 3); a NULL pointer or all zeros = paos_run_passes.
 *   live_rows      [batch][2] or NULL: on entry the rows of item i outside [lo, hi) are zero (see
 *                  paos_run_passes_live) ...
 *   rows_stale     != 0: ... or rather, they hold old data that STANDS for zeros (what paos_start_rows leaves):
 *                  no pass reads them; the program consumes them (a pass along columns rewrites the whole field) or,
 *                  where it cannot, writes the zeros itself before it ends.
 *   final_intensity  != 0: the caller wants |u|^2 (the PSF, plot.py:125-130) and its sum of the field the program
 *                  ends with, not the field: the last pass stores |u|^2 into the context's PSF buffer (as
 *                  paos_psf_keep_power would afterwards) and *power_ticket receives a ticket for paos_norm2_fetch.
 *                  The field content is UNDEFINED after the call.  (Saves writing the last field and reading it
 *                  back: 32 B/px.)
 *                  == 2 (round 4): the field is stored as usual AND *power_ticket receives the ticket of its
 *                  sum |u|^2 -- the power a caller reports next to a saved surface (push_results, run.py:12-27,
 *                  218-223) -- summed by the last pass while it stores its tiles instead of by a separate sweep
 *                  that reads the field back (16 B/px). */
typedef struct {
  const double* live_rows;
  int rows_stale;
  int final_intensity;
  int* power_ticket;
  /* round 5: [batch][2] or NULL -- with rows_stale, columns outside [lo, hi) of the live rows hold old data that stands
   * for zeros as well (what paos_start_box leaves: the first field is written inside the aperture's box only). */
  const double* live_cols;
} paos_program_opts;

/* ---- lifetime -------------------------------------------------------------------- */
/* WFO.__init__ (wfo.py:99-120): allocates `batch` n x n fields (n = 2^k, 64..4096).
 * The field content is undefined until paos_fill / paos_import. */
int paos_ctx_create(int device, int n, int batch, int precision, paos_ctx** out);
int paos_ctx_destroy(paos_ctx* ctx);
const char* paos_last_error(const paos_ctx* ctx);
int paos_sync(paos_ctx* ctx);
/* describe the build: gfx arch, layout block, pitch padding (for logs) */
const char* paos_build_info(void);
/* round 5: the first 32 hex digits of sha256 over the library's sources as they were when it was built (Makefile: HASHED,
 * in that order) -- __graft_entry__.build() compares it with the tree and rebuilds on a mismatch, so a prebuilt library that
 * travelled with a tree it was not built from is noticed */
const char* paos_source_hash(void);
/* the HIP stream handle (hipStream_t) of the context, for event timing */
void* paos_stream(paos_ctx* ctx);

/* Measurement aid (no reference counterpart): time every launch of one kernel class with
 * HIP events on the context's stream between begin and end; end synchronises and returns
 * the launch count and the summed durations.  Used by bench.py for the roofline figure. */
int paos_profile_begin(paos_ctx* ctx, int kernel_kind, int max_launches);
int paos_profile_end(paos_ctx* ctx, int* launches, double* total_ms);
/* the same, also telling apart the launches that skipped dead tiles or loads (pruned passes move fewer
 * bytes, so a bandwidth figure must be taken over the others) */
int paos_profile_end_split(paos_ctx* ctx, int* launches, double* total_ms, int* pruned_launches, double* pruned_ms);
/* the same launch by launch, in launch order: ms_out[i], tag_out[i] (bit 0: the launch skipped whole tiles of dead
 * lines, bit 1: loads of dead positions, bit 2: stores nobody reads, bit 3: it stored the PSF instead of the
 * field, bits 4 / 5 (round 4): the launch ran two / three consecutive passes of the program; 0 = a full pass); *count
 * launches, at most `capacity` */
int paos_profile_end_launches(paos_ctx* ctx, int capacity, double* ms_out, int* tag_out, int* count);
/* round 4: the bytes the pruning plan had each launch timed so far load + store (live lines x (loaded + stored positions)
 * x element size, summed over the batch items): the launch's algorithmic bytes.  Call before paos_profile_end_*. */
int paos_profile_planned_bytes(paos_ctx* ctx, int capacity, double* bytes_out, int* count);
/* round 5: the 1-D line transforms each launch timed so far ran (live lines x the transforms switched on for the item,
 * over every pass the launch carries and every batch item): x 5 N log2 N = the launch's nominal flops, what bench.py
 * prices the fused launches with -- they move a sixteenth of the grid and are bound by fp64 issue, not by bytes.
 * Call before paos_profile_end_*. */
int paos_profile_line_transforms(paos_ctx* ctx, int capacity, double* lines_out, int* count);

/* ---- field I/O ---------------------------------------------------------------------- */
/* u[:] = re + i im for every batch item -- np.ones(..., complex128), wfo.py:118 */
int paos_fill(paos_ctx* ctx, double re, double im);
/* The first surface in one go: u = re + i im (wfo.py:118), u *= aperture weight (wfo.py:236-276),
 * and, for items with stop[i] != 0, u /= sqrt(sum |u|^2) (wfo.py:195-201) -- what paos_fill,
 * paos_aperture and paos_make_stop do one after the other, with the same roundings, but the power
 * is summed from the weights alone and the field is written once (16 B/px instead of ~72).
 * aperture: [batch][PAOS_APERTURE_STRIDE] (enable = 0: no aperture on that item); stop may be NULL. */
int paos_start(paos_ctx* ctx, double re, double im, int shape, const double* aperture, const double* stop);
/* paos_start that writes only the rows [write_rows[2 i], write_rows[2 i + 1]) of item i (rounded outward to
 * whole blocks of rows; they must contain every row the aperture leaves non-zero).  The other rows keep whatever
 * they held and merely STAND for the zeros wfo.py:273-276 would have put there: until a pass program has
 * consumed them (paos_run_program with rows_stale) only paos_zernike, paos_norm2_enqueue_rows and
 * paos_zero_outside_rows may touch the field.  Saves three quarters of the first field write at zoom 4. */
int paos_start_rows(paos_ctx* ctx, double re, double im, int shape, const double* aperture, const double* stop,
                    const double* write_rows);
/* make such rows real zeros (whole blocks of rows outside [lo, hi) of every item are cleared) */
int paos_zero_outside_rows(paos_ctx* ctx, const double* live_rows);
/* Round 5: paos_start_rows that also leaves the COLUMNS outside [write_cols[2 i], write_cols[2 i + 1]) (rounded outward to
 * whole blocks) of the written rows alone -- the field is stored inside the aperture's bounding box only, a sixteenth of the
 * grid at zoom 4 instead of a quarter (wfo.py:118 -> :236-276 -> :195-201).  Until a pass program has consumed the box
 * (paos_run_program with rows_stale and live_cols) only paos_zernike, paos_norm2_enqueue_box and paos_zero_outside_box may
 * touch the field. */
int paos_start_box(paos_ctx* ctx, double re, double im, int shape, const double* aperture, const double* stop,
                   const double* write_rows, const double* write_cols);
/* make everything outside the box rows x cols of every item real zeros (live_cols may be NULL: whole rows) */
int paos_zero_outside_box(paos_ctx* ctx, const double* live_rows, const double* live_cols);
/* host row-major complex128 -> batch item (WFO._wfo assignment in notebooks/tests) */
int paos_import(paos_ctx* ctx, int item, const void* host_c128);
/* batch item -> host.  what = FIELD: complex128 copy (wfo.py:162-164); AMPLITUDE: |u|
 * (wfo.py:166-168); PHASE: angle(u) (wfo.py:170-172); INTENSITY: |u|^2 = the PSF
 * definition of paos/core/plot.py:125-130.  Synchronises. */
int paos_export(paos_ctx* ctx, int item, int what, void* host_out);

/* PSF = |u|^2 (plot.py:125-130) of EVERY batch item written to a device buffer and kept there
 * (doubles, laid out like the field; paos_psf_fetch hands out row-major arrays): the final intensity write of a propagation whose results are consumed on
 * the GPU or fetched later.  paos_psf_fetch copies one item's PSF to the host (synchronises). */
int paos_psf_keep(paos_ctx* ctx);
/* paos_psf_keep and paos_norm2_enqueue of the same field in one sweep (the saved last surface of a chain:
 * |u|^2 is written and summed while the field is read once); *ticket as paos_norm2_enqueue, the sum is
 * bit-identical to it. */
int paos_psf_keep_power(paos_ctx* ctx, int* ticket);
int paos_psf_fetch(paos_ctx* ctx, int item, double* host_out);
/* Page-locked host memory for results (no reference counterpart).  paos_export into pageable
 * memory is bounded by first-touch page faults and on-the-fly pinning (~3 GB/s); into a buffer from
 * paos_host_alloc it is one DMA.  paos_export_pinned requires such a buffer. */
int paos_host_alloc(unsigned long long bytes, void** out);
int paos_host_free(void* p);
int paos_export_pinned(paos_ctx* ctx, int item, int what, void* pinned_out);

/* ---- operators ------------------------------------------------------------------------ */
/* WFO.aperture (wfo.py:203-278): multiply by the exact ellipse mask
 * (EllipticalAperture.to_mask("exact")) or the 32x32 sub-pixel rectangle mask
 * (RectangularAperture.to_mask("subpixel", subpixels=32)); obscuration uses 1 - mask. */
int paos_aperture(paos_ctx* ctx, int shape, const double* params);
/* the mask alone, row-major doubles, for the aperture object's
 * .to_mask(...).to_image(shape) used at run.py:136-141, plot.py:164-184 */
int paos_aperture_render(paos_ctx* ctx, int shape, const double* params1, double* host_mask);
/* WFO.make_stop (wfo.py:195-201): u /= sqrt(sum |u|^2), per enabled item.
 * enable may be NULL (= all). */
int paos_make_stop(paos_ctx* ctx, const double* enable);
/* make_stop (wfo.py:195-201) for a field whose power the context has JUST reduced: the pass program that stored the
 * field ended with final_intensity = 2 (paos_run_program) and nothing has run on the context since.  Only the scaling
 * sweep u *= 1/sqrt(power) is launched -- the reduction that would read the field back has already been done by the
 * pass that stored it.  Round 5: the context checks the precondition itself -- a flag set by such a program and cleared by
 * every entry point that reads or rewrites the field or reduces a power (including this one and the next: behind a stop
 * the reduced power no longer is the field's); when it is clear, both functions run paos_make_stop (same result, one
 * reduction more). */
int paos_stop_scale_last_power(paos_ctx* ctx, const double* enable);
/* The same stop, with even the scaling sweep left out: 1 / sqrt(power) is kept per item on the device and the NEXT pass
 * program's first pass multiplies it into its middle slot (one more factor in a multiplication it performs anyway), so
 * the stop costs neither a read nor a write of the field.  Anything else that touches the field first -- a download, a
 * reduction, an aperture, another stop, a pass program whose first pass cannot take it (generic kernel, an item that
 * sits it out) -- applies it then, by the sweep paos_stop_scale_last_power would have run: same factor, same products,
 * bit-identical.  Same precondition as above. */
int paos_stop_defer_last_power(paos_ctx* ctx, const double* enable);
/* sum |u|^2 per item to the host (np.sum(np.abs(u)**2), wfo.py:200).  Synchronises. */
int paos_norm2(paos_ctx* ctx, double* host_out);
/* The same without stalling the host: enqueue the reduction and its copy to pinned memory,
 * get a ticket, fetch later (the fetch synchronises).  At most PAOS_NORM_SLOTS tickets may be
 * outstanding: one more paos_norm2_enqueue fails with PAOS_EINVAL until a ticket is fetched, and
 * a ticket can be fetched once. */
int paos_norm2_enqueue(paos_ctx* ctx, int* ticket);
/* paos_norm2_enqueue that reads only the rows [lo, hi) of each item: the caller knows the others to be zero
 * (or to stand for zeros, paos_start_rows).  Bit-identical to the full sum of the zero-filled field. */
int paos_norm2_enqueue_rows(paos_ctx* ctx, const double* live_rows, int* ticket);
/* ... and that sums only once what the caller knows to be copies: same_as[i] (a double holding an index) names the
 * item whose field equals item i's (itself for the first of a group; items of a group share their row window) --
 * the wavelengths of a sweep right behind paos_start_rows, whose aperture records agree (wfo.py:195-201 at the
 * entrance pupil does not depend on the wavelength).  Every item gets its leader's sum. */
int paos_norm2_enqueue_rows_like(paos_ctx* ctx, const double* live_rows, const double* same_as, int* ticket);
/* round 5: the same sum over the box rows x cols (live_cols: [batch][2]; same_as may be NULL) -- elements outside are zero
 * or stand for zeros and are not read; bit-identical to the full sum of the zero-filled field. */
int paos_norm2_enqueue_box(paos_ctx* ctx, const double* live_rows, const double* live_cols, const double* same_as, int* ticket);
int paos_norm2_fetch(paos_ctx* ctx, int ticket, double* host_out);
/* give a ticket back without reading it (no synchronisation) */
int paos_norm2_release(paos_ctx* ctx, int ticket);
/* PSF metrics on the GPU for Monte-Carlo studies (the encircled-energy workflow of
 * docs/source/user/montecarlo/index.rst:26-66; PSF = |u|^2, plot.py:125-130).  Per item:
 * [sum I, sum I*col, sum I*row, max I, then nr values: sum of I over pixels whose centre lies
 * within radii_px[k] of (cx_px, cy_px)], nr <= 16.  host_out: [batch][4 + nr].  Synchronises. */
int paos_psf_metrics(paos_ctx* ctx, int nr, const double* radii_px, double cx_px, double cy_px,
                     double* host_out);
/* quadratic phase u *= exp(i sgn [2 pi] coef ((x sx)^2 + (y sy)^2)), x, y centred pixel
 * indices: the field part of WFO.lens (wfo.py:359-366) with mul2pi = 1, sgn = -1,
 * coef = 0.5 lens_phase / wl. */
int paos_phase(paos_ctx* ctx, const double* params, int mul2pi);
/* Tabulated phase screen, the field part of WFO.grid_sag (wfo.py:869-871) and WFO.psd
 * (wfo.py:945-949): u[item] *= exp(2 pi i wfe / wl) with wfe a host map in metres (row-major
 * n x n doubles, finite: masked pixels filled with 0 as the reference does).  The resampling of a
 * sag map and the random draw of a PSD screen stay on the host (paos_amd/wfo.py).  Synchronises. */
int paos_phase_map(paos_ctx* ctx, int item, const double* host_wfe, double wl);
/* Round 5: the same for `n_items` items that share ONE map (a measured surface map is the same for every wavelength of a
 * sweep and every draw of a Monte-Carlo study): items[k] (indices, as doubles) get u *= exp(2 pi i wfe / wl[k]).  The map
 * crosses PCIe once and stays on the device; `key` != 0 names its content: a later call with the same key skips the
 * validation and the upload (the caller vouches that the host buffer still holds what was uploaded under that key;
 * 0 = always upload).  `host_wfe` may be NULL when a map is kept under `key` (paos_psd_screen).  Does not synchronise when
 * the key matches. */
int paos_phase_map_items(paos_ctx* ctx, const double* host_wfe, unsigned long long key, int n_items, const double* items,
                         const double* wl);
/* Round 5: the random screen of WFO.psd built ON THE DEVICE (wfo.py:908-943 calling psd.py:100-160): the host draws the
 * white noise -- `host_noise`, then `host_rough` (n x n doubles each; NumPy's generator is the reference's, so a seeded run
 * stays comparable) -- and the library runs  fft2(noise) * sqrt(A / (B + (rho / fknee)^C) / (2 pi rho) * cell) * gain,
 * zero outside [fmin, fmax]  ->  ifft2  ->  (Re + SR * rough) * 2 * unit  on its own passes, in the reference's order of
 * operations.  params[12] = {1 / (n dx), 1 / (n dy), A, B, C, fknee, fmin, fmax, cell, gain = sqrt(n0 n1), SR, unit}.
 * The map stays on the device under `key` (!= 0): paos_phase_map_items(ctx, NULL, key, ...) applies it; `host_out`
 * (n x n doubles or NULL) receives a copy (the `wfe` entry the reference returns).  complex128 contexts only.
 * Synchronises.  Host restatement, and the bit-exact reference for the fixtures: paos_amd/phase_maps.py psd_map. */
int paos_psd_screen(paos_ctx* ctx, const double* host_noise, const double* host_rough, const double* params,
                    unsigned long long key, double* host_out);
/* WFO.ptp (wfo.py:462-472): ifft2(exp(-i coef (fx^2+fy^2)) fft2(u)), ortho norms, shifts
 * cancelled; sx, sy = 1/(n dx), 1/(n dy) (np.fft.fftfreq spacing), coef = pi wl dz. */
int paos_ptp(paos_ctx* ctx, const double* params);
/* WFO.stw (wfo.py:491-509): fftshift(exp(+i coef f^2) FFT(ifftshift u)); inverse != 0
 * selects ifft2 (dz < 0).  sx, sy, coef as for ptp. */
int paos_stw(paos_ctx* ctx, const double* params, int inverse);
/* WFO.wts (wfo.py:528-545): fftshift(FFT(ifftshift(exp(i coef (x^2+y^2)) u))); sx, sy =
 * dx, dy; coef = pi / (dz wl). */
int paos_wts(paos_ctx* ctx, const double* params, int inverse);
/* A whole stretch of the propagation loop (run.py:193-207 over consecutive surfaces) as a
 * program of passes: lens phases (wfo.py:359-366), the checkerboard signs that replace
 * fftshift/ifftshift, the quadratic phases and ortho scalings of ptp / stw / wts
 * (wfo.py:462-545) ride on the FFT passes, and the last pass of one propagator merges with the
 * first pass of the next because a 2-D FFT may run rows-then-columns or columns-then-rows.
 * `blocks` = n_blocks sets of [batch][PAOS_PHASE_STRIDE] doubles.  paos_ptp / paos_stw /
 * paos_wts / paos_phase above are one-operator programs of the same machinery. */
int paos_run_passes(paos_ctx* ctx, const paos_pass* passes, int n_passes, const double* blocks,
                    int n_blocks);
/* The same with a promise about the field on entry: for item i, the rows outside
 * [live_rows[2 i], live_rows[2 i + 1]) are exactly zero in memory (what WFO.aperture, wfo.py:273-276,
 * leaves outside the bounding box of a clear aperture).  Transforms of zero rows are zero rows, so the
 * first passes skip them -- the results are the same, only the traffic is not.  (Inside a program the
 * library tracks by itself which rows / columns the apertures riding on its passes have zeroed.) */
/* Measurement aid: an in-place copy of the whole batch (every element read and written back unchanged,
 * 16 B per lane, unit stride), `reps` launches timed with HIP events on the context's stream: the
 * yardstick bench.py prints next to the pass kernels' rate.  The field is left as it was. */
int paos_copy_yardstick(paos_ctx* ctx, int reps, double* ms_per_launch, double* bytes_per_launch);
/* Measurement aid: how often, since the context was created, a pass that carries an aperture (WFO.aperture,
 * wfo.py:246-276, riding on a pass as line records) found its records in one of the context's kept sets
 * (*found) and how often it had to render them (*rendered).  bench.py reports both per step of its walked sweep. */
int paos_record_set_stats(paos_ctx* ctx, unsigned long long* found, unsigned long long* rendered);
/* Measurement / test aid: on = 0 makes every pass process every tile (no dead-line pruning). */
int paos_ctx_set_pruning(paos_ctx* ctx, int on);
int paos_run_passes_live(paos_ctx* ctx, const paos_pass* passes, int n_passes, const double* blocks,
                         int n_blocks, const double* live_rows);
/* paos_run_passes with options (paos_program_opts above): entry rows that are zero or stand for zeros, and the
 * PSF + power of the final field instead of the field itself (run.py:222-224 -> plot.py:125-130 in one sweep). */
int paos_run_program(paos_ctx* ctx, const paos_pass* passes, int n_passes, const double* blocks, int n_blocks,
                     const paos_program_opts* opts);
/* WFO.zernikes (wfo.py:620-652) with Zernike polynomials (zernike.py:85-109,245-247):
 * u *= exp(2 pi i wfe / wl) inside rho <= 1.  `table` holds the Jacobi recurrence
 * constants [(nmax+1)][kdim][3]; `params` the per-item blocks (PAOS_ZERNIKE_HEAD +
 * 2 (nmax+1) kdim doubles).  If host_wfe != NULL the wfe map of item 0 is returned
 * (row-major doubles, NaN where rho > 1 -- the masked array of wfo.py:654). */
int paos_zernike(paos_ctx* ctx, int nmax, int kdim, const double* table, const double* params,
                 int param_stride, double* host_wfe);
/* Round 5: paos_zernike for a caller who knows that some items hold COPIES of one field -- the surface right behind the
 * start of a wavelength sweep or of a Monte-Carlo batch (wfo.py:118 fills every wavefront with the same constant under the
 * same aperture): same_as[i] = index of an item whose field equals item i's.  Items that share their wfe map (records equal
 * but for the wavelength) AND their field are served by one load per pixel; the values written are those of paos_zernike
 * bit for bit.  NULL = paos_zernike. */
int paos_zernike_like(paos_ctx* ctx, int nmax, int kdim, const double* table, const double* params,
                      int param_stride, const double* same_as, double* host_wfe);


/* ---- PolyOrthoNorm / Zorthonorm ------------------------------------------------ */
/* The pupil the polynomials are orthonormalised on -- run.py:133-141: the pixels where the exact
 * mask of this surface's aperture object is non-zero (whatever its obscuration flag).  One weight
 * map per batch item stays in HBM until the next call.  params as for paos_aperture. */
int paos_pupil_aperture(paos_ctx* ctx, int shape, const double* params);
/* the same from a host array (row-major doubles, 0 = masked): the `mask` argument of
 * WFO.zernikes, wfo.py:583,623-627.  Synchronises. */
int paos_pupil_upload(paos_ctx* ctx, int item, const double* host_weights);
/* Zernike.cov (zernike.py:293-318) without its final division: per item, the sums over the
 * unmasked pixels (rho <= 1, and pupil weight != 0 when use_pupil) of Z_i Z_j for the first K
 * polynomials, i <= j enumerated row by row, followed by the pixel count:
 * host_out[batch][K (K + 1) / 2 + 1].  poly[K][4] = {|m|, k = (n - |m|) / 2, is_sin, factor}
 * with factor = (-1)^k norm; table / params as for paos_zernike (the coefficient planes of
 * params are not read).  K <= 64.  Synchronises. */
int paos_zernike_gram(paos_ctx* ctx, int nmax, int kdim, const double* table, const double* params,
                      int param_stride, int K, const double* poly, int use_pupil, double* host_out);
/* paos_zernike restricted to the pupil: pixels outside it keep their value and read NaN in
 * host_wfe (the mask of PolyOrthoNorm's polynomials, zernike.py:396-400). */
int paos_zernike_pupil(paos_ctx* ctx, int nmax, int kdim, const double* table, const double* params,
                       int param_stride, double* host_wfe);

#ifdef __cplusplus
}
#endif
#endif /* PAOS_HIP_H */
