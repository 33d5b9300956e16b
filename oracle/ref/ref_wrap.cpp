// ref_wrap.cpp -- thin C entry points around the reference's OWN cost functors.
//
// TEST INFRASTRUCTURE ONLY.  The functor bodies are not in this file and not in this
// repository: build_ref.sh cuts them, at build time, out of /root/reference/src/main.cpp
// (struct collinearity :1315-1401, struct collinearityMachineLearnedSimple :2033-2120, the five
// other projection functors of :2133-2562, struct collinearityOrthographic :1412-1490, struct
// omniCollinearityMachineLearnedSimple :2700-2790, the statistics and angle helpers :791-823 and
// :884-1019 (calcMedian, calcStdDev, calcStatistics, incidenceAngle, refractionAngle, radialDistance,
// sort_index, calculateAIC, calculateBIC), struct constrainPoint :1057-1085, and the APSCALE
// define :55) into a temporary directory
// that is on the include path only while this file compiles.  The functors are templates
// over the scalar type, so they are instantiated here with
//   T = double                 -> the reference's residuals, bit for bit;
//   T = std::complex<double>   -> complex-step derivatives d r / d p = Im r(p + i h) / h,
//                                 exact to rounding for these analytic expressions.
// Only standard-library types are used; nothing stands in for Ceres.  The fisheye functors call
// atan2, which the standard library does not define for complex arguments: the overload below
// is its first-order continuation (value of the real parts, imaginary part from the gradient),
// which is all a complex step of size 1e-30 can see.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <vector>

inline std::complex<double> atan2(const std::complex<double> &y, const std::complex<double> &x) {
  const double yr = y.real(), xr = x.real();
  return std::complex<double>(std::atan2(yr, xr), (xr * y.imag() - yr * x.imag()) / (xr * xr + yr * yr));
}

#include "slice_apscale.inc"
#include "slice_constrainPoint.inc"
#include "slice_collinearity.inc"
#include "slice_collinearityMLS.inc"
#include "slice_stereographicMLS.inc"
#include "slice_equidistantMLS.inc"
#include "slice_equisolidMLS.inc"
#include "slice_orthographicMLS.inc"
#include "slice_fisheyeStereoMLS.inc"
#include "slice_orthographic.inc"
#include "slice_omniMLS.inc"
#include "slice_pi.inc"
#include "slice_stats1.inc"
#include "slice_stats2.inc"

typedef std::complex<double> cplx;

// collinearityOrthographic has no correction arguments (src/main.cpp:1414): give it the 8-argument face of the others
struct orthographicNoCorrection : collinearityOrthographic {
  orthographicNoCorrection(double x, double y, double sx, double sy, double xp, double yp, double, double)
      : collinearityOrthographic(x, y, sx, sy, xp, yp) {}
};

template <class F>
static void eval_functor(const double *eop, const double *xyz, const double *iop, const double *ap,
                         const double *obs, double *r) {
  F f(obs[0], obs[1], obs[2], obs[3], iop[0], iop[1], obs[4], obs[5]);
  f(eop, xyz, iop, ap, r);
}

template <class F>
static void jacobian_functor(const double *eop, const double *xyz, const double *iop, const double *ap,
                             const double *obs, double h, double *J) {
  F f(obs[0], obs[1], obs[2], obs[3], iop[0], iop[1], obs[4], obs[5]);
  cplx p[28], r[2];
  for (int col = 0; col < 28; ++col) {
    for (int k = 0; k < 6; ++k) p[k] = eop[k];
    for (int k = 0; k < 3; ++k) p[6 + k] = xyz[k];
    for (int k = 0; k < 3; ++k) p[9 + k] = iop[k];
    for (int k = 0; k < 16; ++k) p[12 + k] = ap[k];
    p[col] += cplx(0.0, h);
    f(p, p + 6, p + 9, p + 12, r);
    J[col] = r[0].imag() / h;
    J[28 + col] = r[1].imag() / h;
  }
}

extern "C" {

// obs = x, y, sx, sy, xcorr, ycorr.  xp_/yp_ constructor arguments are stored but unused by
// the functors (src/main.cpp:1398-1399); iop[0], iop[1] are passed as the call site does
// (src/main.cpp:3847).
void ref_collinearity(const double *eop, const double *xyz, const double *iop, const double *ap,
                      const double *obs, double *r) {
  collinearity f(obs[0], obs[1], obs[2], obs[3], iop[0], iop[1]);
  f(eop, xyz, iop, ap, r);
}

void ref_collinearity_mls(const double *eop, const double *xyz, const double *iop, const double *ap,
                          const double *obs, double *r) {
  collinearityMachineLearnedSimple f(obs[0], obs[1], obs[2], obs[3], iop[0], iop[1], obs[4], obs[5]);
  f(eop, xyz, iop, ap, r);
}

// J is row-major 2x28 (EOP 6 | XYZ 3 | IOP 3 | AP 16), one complex-step evaluation per column.
void ref_collinearity_mls_jacobian(const double *eop, const double *xyz, const double *iop,
                                   const double *ap, const double *obs, double h, double *J) {
  collinearityMachineLearnedSimple f(obs[0], obs[1], obs[2], obs[3], iop[0], iop[1], obs[4], obs[5]);
  cplx p[28], r[2];
  for (int col = 0; col < 28; ++col) {
    for (int k = 0; k < 6; ++k) p[k] = eop[k];
    for (int k = 0; k < 3; ++k) p[6 + k] = xyz[k];
    for (int k = 0; k < 3; ++k) p[9 + k] = iop[k];
    for (int k = 0; k < 16; ++k) p[12 + k] = ap[k];
    p[col] += cplx(0.0, h);
    f(p, p + 6, p + 9, p + 12, r);
    J[col] = r[0].imag() / h;
    J[28 + col] = r[1].imag() / h;
  }
}

// model: 0 collinearity, 1 collinearity stereographic, 2..5 fisheye equidistant, equisolid-angle,
// orthographic, stereographic (the "MachineLearnedSimple" functors), 6 collinearityOrthographic (ignores the
// corrections obs[4], obs[5]), 7 omniCollinearityMachineLearnedSimple.  Returns -1 for another model.
int ref_projection(int model, const double *eop, const double *xyz, const double *iop, const double *ap,
                   const double *obs, double *r) {
  switch (model) {
    case 0: eval_functor<collinearityMachineLearnedSimple>(eop, xyz, iop, ap, obs, r); return 0;
    case 1: eval_functor<collinearityStereographicMachineLearnedSimple>(eop, xyz, iop, ap, obs, r); return 0;
    case 2: eval_functor<fisheyeEquidistantMachineLearnedSimple>(eop, xyz, iop, ap, obs, r); return 0;
    case 3: eval_functor<fisheyeEquisolidAngleMachineLearnedSimple>(eop, xyz, iop, ap, obs, r); return 0;
    case 4: eval_functor<fisheyeOrthographicMachineLearnedSimple>(eop, xyz, iop, ap, obs, r); return 0;
    case 5: eval_functor<fisheyeStereographicMachineLearnedSimple>(eop, xyz, iop, ap, obs, r); return 0;
    case 6: eval_functor<orthographicNoCorrection>(eop, xyz, iop, ap, obs, r); return 0;
    case 7: eval_functor<omniCollinearityMachineLearnedSimple>(eop, xyz, iop, ap, obs, r); return 0;
  }
  return -1;
}

int ref_projection_jacobian(int model, const double *eop, const double *xyz, const double *iop,
                            const double *ap, const double *obs, double h, double *J) {
  switch (model) {
    case 0: jacobian_functor<collinearityMachineLearnedSimple>(eop, xyz, iop, ap, obs, h, J); return 0;
    case 1: jacobian_functor<collinearityStereographicMachineLearnedSimple>(eop, xyz, iop, ap, obs, h, J); return 0;
    case 2: jacobian_functor<fisheyeEquidistantMachineLearnedSimple>(eop, xyz, iop, ap, obs, h, J); return 0;
    case 3: jacobian_functor<fisheyeEquisolidAngleMachineLearnedSimple>(eop, xyz, iop, ap, obs, h, J); return 0;
    case 4: jacobian_functor<fisheyeOrthographicMachineLearnedSimple>(eop, xyz, iop, ap, obs, h, J); return 0;
    case 5: jacobian_functor<fisheyeStereographicMachineLearnedSimple>(eop, xyz, iop, ap, obs, h, J); return 0;
    case 6: jacobian_functor<orthographicNoCorrection>(eop, xyz, iop, ap, obs, h, J); return 0;
    case 7: jacobian_functor<omniCollinearityMachineLearnedSimple>(eop, xyz, iop, ap, obs, h, J); return 0;
  }
  return -1;
}

// out = median, mean, stdev, min, max, RMSE (calcStatistics with six outputs, src/main.cpp:884)
void ref_statistics(const double *v, int n, double *out) {
  std::vector<double> s(v, v + n);
  calcStatistics(s, out[0], out[1], out[2], out[3], out[4], out[5]);
}
double ref_median(const double *v, int n) { return calcMedian(std::vector<double>(v, v + n)); }
double ref_stddev(const double *v, int n) { return calcStdDev(std::vector<double>(v, v + n)); }
void ref_sort_index(const double *v, int n, int *out) {
  const std::vector<int> idx = sort_index(std::vector<double>(v, v + n));
  for (int i = 0; i < n; ++i) out[i] = idx[i];
}
double ref_aic(double n, double mse, double k) { return calculateAIC(n, mse, k); }
double ref_bic(double n, double mse, double k) { return calculateBIC(n, mse, k); }
double ref_incidence_angle(const double *eop, const double *xyz) { return incidenceAngle(eop, xyz); }
double ref_refraction_angle(double x, double y, const double *iop, const double *ap) { return refractionAngle(x, y, iop, ap); }
double ref_radial_distance(double x, double y, const double *iop, const double *ap) { return radialDistance(x, y, iop, ap); }

void ref_constrain_point(const double *p, const double *p0, const double *sigma, double *r) {
  constrainPoint f(p0[0], p0[1], p0[2], sigma[0], sigma[1], sigma[2]);
  f(p, r);
}

}  // extern "C"
