/* paos_plan.h -- the scalar half of the propagation loop for a whole batch, C ABI (part of libpaoship.so).
 *
 * Every WFO method of the reference mixes a few dozen floating-point operations on the pilot Gaussian
 * beam with the N x N array work (paos/classes/wfo.py:318-357 lens, :386-416 Magnification, :434-443
 * ChangeMedium, :454-460 / :483-508 / :520-544 the heads of ptp / stw / wts, :556-572 propagate), and
 * paos.core.run.run gates them per surface (paos/core/run.py:181-207).  Those scalars DECIDE which
 * propagator runs, so they are reproduced here in IEEE double with the reference's operation order --
 * no FMA contraction, `x**2` as the libm pow(x, 2.0) that NumPy / Python scalars call (it differs from
 * x * x in the last bit for ~0.1 % of arguments), sqrt correctly rounded -- and return, for every
 * wavefront of the batch, the parameter blocks the device passes need.  Doing this in C for the whole
 * batch at once instead of per wavefront in Python removes the host bound of grids <= 1024^2.
 *
 * A beam is PAOS_BEAM_STRIDE doubles: wl, z, w0, zw0, zr, dx, dy, C, fratio, propagator code
 * (0 "", 1 II, 2 IO, 3 OI, 4 OO).
 */
#ifndef PAOS_PLAN_H
#define PAOS_PLAN_H

#ifdef __cplusplus
extern "C" {
#endif

enum { PAOS_BEAM_WL = 0, PAOS_BEAM_Z = 1, PAOS_BEAM_W0 = 2, PAOS_BEAM_ZW0 = 3, PAOS_BEAM_ZR = 4, PAOS_BEAM_DX = 5,
       PAOS_BEAM_DY = 6, PAOS_BEAM_C = 7, PAOS_BEAM_FRATIO = 8, PAOS_BEAM_PROP = 9, PAOS_BEAM_STRIDE = 10 };
/* per-item status of paos_plan_surface: the exception the reference raises */
enum { PAOS_PLAN_OK = 0, PAOS_PLAN_NEGATIVE_MAGNIFICATION = 1, PAOS_PLAN_PTP_NOT_PLANAR = 2,
       PAOS_PLAN_STW_PLANAR = 3, PAOS_PLAN_WTS_NOT_PLANAR = 4 };

/* WFO.__init__ (wfo.py:99-120): beams[batch][PAOS_BEAM_STRIDE] from the beam diameter, one wavelength per
 * item, the grid size and the zoom. */
int paos_plan_init(int batch, double beam_diameter, const double* wavelengths, int grid, double zoom, double* beams);
/* wz and distancetofocus of every beam (wfo.py:142-150), the two derived read-outs of push_results. */
int paos_plan_readout(int batch, const double* beams, double* wz, double* distancetofocus);
/* One surface of run.py:181-207 for every item: Magnification if Mt != 1 or Ms != 1, ChangeMedium if
 * |n1n2| != 1, lens if fl is finite, propagate if T is finite and |T| > 1e-10 -- with fl = cout / power
 * (inf when power == 0) and T = cout * thickness prepared by the caller.  Outputs, all [batch][5] blocks
 * [enable, sx, sy, coef, sgn] of include/paos_hip.h with enable = 0 where the step does not run:
 * lens, stw, ptp, wts; inverse flags [batch] for stw and wts (dz < 0); status [batch] (PAOS_PLAN_*: the
 * item's beam is left where the reference would have raised).  Returns the number of items with a
 * non-zero status. */
int paos_plan_surface(int batch, int grid, double* beams, const double* Mt, const double* Ms, const double* fl,
                      const double* T, const double* n1n2, double* lens, double* stw, double* ptp, double* wts,
                      double* inv_stw, double* inv_wts, int* status);

#ifdef __cplusplus
}
#endif
#endif /* PAOS_PLAN_H */
