// ref_wrap.cpp -- thin C entry points around the reference's OWN cost functors.
//
// TEST INFRASTRUCTURE ONLY.  The functor bodies are not in this file and not in this
// repository: build_ref.sh cuts them, at build time, out of /root/reference/src/main.cpp
// (struct collinearity :1315-1401, struct collinearityMachineLearnedSimple :2033-2120,
// struct constrainPoint :1057-1085, and the APSCALE define :55) into a temporary directory
// that is on the include path only while this file compiles.  The functors are templates
// over the scalar type, so they are instantiated here with
//   T = double                 -> the reference's residuals, bit for bit;
//   T = std::complex<double>   -> complex-step derivatives d r / d p = Im r(p + i h) / h,
//                                 exact to rounding for these analytic expressions.
// Only standard-library types are used; nothing stands in for Ceres.
#include <cmath>
#include <complex>

#include "slice_apscale.inc"
#include "slice_constrainPoint.inc"
#include "slice_collinearity.inc"
#include "slice_collinearityMLS.inc"

typedef std::complex<double> cplx;

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

void ref_constrain_point(const double *p, const double *p0, const double *sigma, double *r) {
  constrainPoint f(p0[0], p0[1], p0[2], sigma[0], sigma[1], sigma[2]);
  f(p, r);
}

}  // extern "C"
