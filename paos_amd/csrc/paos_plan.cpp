// paos_plan.cpp -- include/paos_plan.h: the pilot-beam scalars of the propagation loop for a batch.
//
// A transcription of the scalar statements of paos/classes/wfo.py (lines cited per function) in the
// reference's operation order.  Compile with -ffp-contract=off.  `sq(x)` is the libm pow(x, 2.0) that
// the reference's `x**2` on NumPy / Python scalars evaluates to -- NOT x * x: glibc's pow is not
// correctly rounded and the two differ in the last bit for about one argument in a thousand
// (measured); it is called through a volatile pointer so that the compiler cannot rewrite it.
#include "../../include/paos_plan.h"

#include <cmath>

namespace {

double (*volatile libm_pow)(double, double) = std::pow;
inline double sq(double x) { return libm_pow(x, 2.0); }

constexpr double kPi = 3.141592653589793;  // np.pi
constexpr double kRayleighFactor = 2.0;    // wfo.py:111

struct Beam {
  double wl, z, w0, zw0, zr, dx, dy, C, fratio, prop;
};
static_assert(sizeof(Beam) == PAOS_BEAM_STRIDE * sizeof(double), "beam record layout");

// wfo.py:280-302: 'I' within rayleigh_factor * zr of the waist
inline bool inside(const Beam& b, double z) { return std::fabs(z - b.zw0) < kRayleighFactor * b.zr; }

inline double wz_of(const Beam& b) { return b.w0 * std::sqrt(1.0 + sq((b.z - b.zw0) / b.zr)); }

inline void off(double* blk) { blk[0] = blk[1] = blk[2] = blk[3] = blk[4] = 0.0; }
inline void set(double* blk, double sx, double sy, double coef, double sgn) {
  blk[0] = 1.0; blk[1] = sx; blk[2] = sy; blk[3] = coef; blk[4] = sgn;
}

// wfo.py:386-416 (run.py:195 passes (Mt, Ms) as (My, Mx))
int magnification(Beam& b, double My, double Mx) {
  if (!(Mx > 0.0) || !(My > 0.0)) return PAOS_PLAN_NEGATIVE_MAGNIFICATION;
  b.dx *= Mx;
  b.dy *= My;
  if (std::fabs(Mx - 1.0) < 1.0e-8) return PAOS_PLAN_OK;
  double gap = b.z - b.zw0;
  double wz = b.w0 * std::sqrt(1.0 + sq((b.z - b.zw0) / b.zr));
  gap *= sq(Mx);
  wz *= Mx;
  b.w0 *= Mx;
  b.zr *= sq(Mx);
  b.zw0 = b.z - gap;
  b.fratio = std::fabs(gap) / (2 * wz);
  return PAOS_PLAN_OK;
}

// wfo.py:434-443
void change_medium(Beam& b, double n1n2) {
  double gap = b.z - b.zw0;
  gap /= n1n2;
  b.zr /= n1n2;
  b.wl *= n1n2;
  b.zw0 = b.z - gap;
  b.fratio /= n1n2;
}

// wfo.py:318-366: the block of exp(2 pi i * (-(x^2+y^2) * (0.5 lens_phase / wl)))
void lens(Beam& b, double fl, double* blk) {
  const double wz = b.w0 * std::sqrt(1.0 + sq((b.z - b.zw0) / b.zr));
  double gap = b.z - b.zw0;
  const bool in0 = inside(b, b.z);
  const double curv_in = gap / (sq(gap) + sq(b.zr));
  const double curv_out = curv_in - 1.0 / fl;
  b.w0 = wz / std::sqrt(1.0 + sq(kPi * sq(wz) * curv_out / b.wl));
  b.zw0 = -curv_out / (sq(curv_out) + sq(b.wl / (kPi * sq(wz)))) + b.z;
  b.zr = kPi * sq(b.w0) / b.wl;
  const bool in1 = inside(b, b.z);
  const double ref_in = (in0 || b.C == 0.0) ? 0.0 : 1 / gap;
  gap = b.z - b.zw0;
  const double ref_out = in1 ? 0.0 : 1 / gap;
  b.C = ref_out;
  double power;
  if (in0 && in1) power = 1.0 / fl;
  else if (in0 && !in1) power = 1 / fl + ref_out;
  else if (!in0 && in1) power = 1.0 / fl - ref_in;
  else power = 1.0 / fl - ref_in + ref_out;
  b.fratio = std::fabs(gap) / (2 * wz);
  set(blk, b.dx, b.dy, 0.5 * power / b.wl, -1.0);
}

// wfo.py:454-472 head
int ptp(Beam& b, int n, double dz, double* blk) {
  if (std::fabs(dz) < 0.001 * b.wl) return PAOS_PLAN_OK;
  if (b.C != 0) return PAOS_PLAN_PTP_NOT_PLANAR;
  const double fsx = 1.0 / (n * b.dx), fsy = 1.0 / (n * b.dy);
  set(blk, fsx, fsy, kPi * b.wl * dz, -1.0);
  b.z = b.z + dz;
  return PAOS_PLAN_OK;
}

// wfo.py:483-509 head
int stw(Beam& b, int n, double dz, double* blk, double* inverse) {
  if (std::fabs(dz) < 0.001 * b.wl) return PAOS_PLAN_OK;
  if (b.C == 0.0) return PAOS_PLAN_STW_PLANAR;
  const double fsx = 1.0 / (n * b.dx), fsy = 1.0 / (n * b.dy);
  set(blk, fsx, fsy, kPi * b.wl * dz, 1.0);
  b.z = b.z + dz;
  b.C = 0.0;
  b.dx = (fsx - 0.0) * b.wl * std::fabs(dz);
  b.dy = (fsy - 0.0) * b.wl * std::fabs(dz);
  *inverse = (dz >= 0) ? 0.0 : 1.0;
  return PAOS_PLAN_OK;
}

// wfo.py:520-545 head
int wts(Beam& b, int n, double dz, double* blk, double* inverse) {
  if (std::fabs(dz) < 0.001 * b.wl) return PAOS_PLAN_OK;
  if (b.C != 0.0) return PAOS_PLAN_WTS_NOT_PLANAR;
  set(blk, b.dx, b.dy, kPi / (dz * b.wl), 1.0);
  b.z = b.z + dz;
  b.C = 1 / (b.z - b.zw0);
  b.dx = b.wl * std::fabs(dz) / (n * b.dx);
  b.dy = b.wl * std::fabs(dz) / (n * b.dy);
  *inverse = (dz >= 0) ? 0.0 : 1.0;
  return PAOS_PLAN_OK;
}

// wfo.py:556-572
int propagate(Beam& b, int n, double dz, double* bstw, double* bptp, double* bwts, double* inv_stw, double* inv_wts) {
  const bool in0 = inside(b, b.z), in1 = inside(b, b.z + dz);
  const double z1 = b.z, z2 = b.z + dz;
  int rc = PAOS_PLAN_OK;
  if (in0 && in1) {
    rc = ptp(b, n, dz, bptp);
  } else if (!in0 && in1) {
    rc = stw(b, n, b.zw0 - z1, bstw, inv_stw);
    if (!rc) rc = ptp(b, n, z2 - b.zw0, bptp);
  } else if (in0 && !in1) {
    rc = ptp(b, n, b.zw0 - z1, bptp);
    if (!rc) rc = wts(b, n, z2 - b.zw0, bwts, inv_wts);
  } else {
    rc = stw(b, n, b.zw0 - z1, bstw, inv_stw);
    if (!rc) rc = wts(b, n, z2 - b.zw0, bwts, inv_wts);
  }
  if (!rc) b.prop = in0 ? (in1 ? 1.0 : 2.0) : (in1 ? 3.0 : 4.0);  // II, IO, OI, OO
  return rc;
}

}  // namespace

extern "C" {

int paos_plan_init(int batch, double beam_diameter, const double* wavelengths, int grid, double zoom, double* beams) {
  if (batch < 0 || !wavelengths || !beams || grid <= 0) return -1;
  Beam* b = reinterpret_cast<Beam*>(beams);
  for (int i = 0; i < batch; ++i) {
    const double wl = wavelengths[i];
    b[i].wl = wl;
    b[i].z = 0.0;
    b[i].w0 = beam_diameter / 2.0;
    b[i].zw0 = 0.0;
    b[i].zr = kPi * sq(b[i].w0) / wl;
    b[i].dx = beam_diameter * zoom / grid;
    b[i].dy = beam_diameter * zoom / grid;
    b[i].C = 0.0;
    b[i].fratio = INFINITY;
    b[i].prop = 0.0;
  }
  return 0;
}

int paos_plan_readout(int batch, const double* beams, double* wz, double* distancetofocus) {
  if (batch < 0 || !beams || !wz || !distancetofocus) return -1;
  const Beam* b = reinterpret_cast<const Beam*>(beams);
  for (int i = 0; i < batch; ++i) {
    wz[i] = wz_of(b[i]);
    distancetofocus[i] = b[i].zw0 - b[i].z;
  }
  return 0;
}

int paos_plan_surface(int batch, int grid, double* beams, const double* Mt, const double* Ms, const double* fl,
                      const double* T, const double* n1n2, double* lens_blk, double* stw_blk, double* ptp_blk,
                      double* wts_blk, double* inv_stw, double* inv_wts, int* status) {
  if (batch < 0 || !beams || !Mt || !Ms || !fl || !T || !n1n2 || !lens_blk || !stw_blk || !ptp_blk || !wts_blk ||
      !inv_stw || !inv_wts || !status)
    return -1;
  Beam* b = reinterpret_cast<Beam*>(beams);
  int bad = 0;
  for (int i = 0; i < batch; ++i) {
    double* bl = lens_blk + 5 * i;
    double* bs = stw_blk + 5 * i;
    double* bp = ptp_blk + 5 * i;
    double* bw = wts_blk + 5 * i;
    off(bl); off(bs); off(bp); off(bw);
    inv_stw[i] = inv_wts[i] = 0.0;
    int rc = PAOS_PLAN_OK;
    if (Mt[i] != 1.0 || Ms[i] != 1.0) rc = magnification(b[i], Mt[i], Ms[i]);
    if (!rc && std::fabs(n1n2[i]) != 1.0) change_medium(b[i], n1n2[i]);
    if (!rc && std::isfinite(fl[i])) lens(b[i], fl[i], bl);
    if (!rc && std::isfinite(T[i]) && std::fabs(T[i]) > 1e-10)
      rc = propagate(b[i], grid, T[i], bs, bp, bw, inv_stw + i, inv_wts + i);
    status[i] = rc;
    bad += rc != 0;
  }
  return bad;
}

}  // extern "C"
