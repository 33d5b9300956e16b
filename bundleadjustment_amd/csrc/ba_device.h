// ba_device.h -- data layout in HBM and the per-observation device math shared by the kernels.
//
// Layout (all fp64 unless noted; "e" = observation in point-sorted order, "jj" = point in
// internal order, "i" = station in file order, "c" = camera in file order):
//   obs     [e]  8 doubles (64 B): x, y, sx, sy, xcorr, ycorr, {int32 station, int32 file row}, pad
//   sta     [i]  24 doubles (192 B): R(9), Xo Yo Zo, sin k, cos k, column scale(6), {int32 cam}, pad(3)
//   cam     [c]  CamRec: xp yp c, AP(16), column scale of the free parameters, free mask, offset
//   pt      [jj] xyz(3) | prior(3) | 1/sigma(3) | column scale(3) | Vinv(6) | g(3)   (separate arrays)
//   rec     [e]  16 doubles (128 B, one cache line): Js row0(6), Js row1(6), r(2), slot(2)
//   Jp      [e]  6 doubles: row0(3), row1(3)
//   Jc      [e]  2*ncmax doubles: row0(ncmax), row1(ncmax)  (free camera parameters only)
//   Mo      [e]  3 doubles: I - Jp Vinv Jp^T (symmetric 2x2)
//   Y       [e]  24 doubles: Js^T Jp L with Vinv = L L^T, 6 rows padded to 4 (block-sparse path only)
// Jacobians in rec/Jp/Jc carry the robust-loss correction sqrt(rho') and the Jacobi column
// scaling, i.e. they are rows of the matrix the trust-region step is computed from.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define BA_OBS_STRIDE 8
#define BA_STA_STRIDE 24
#define BA_REC_STRIDE 16
#define BA_Y_STRIDE 18  // 6 rows x 3 doubles = 144 B, no padding: a record starts on a 16-byte boundary, its second half 8 bytes past one
#define BA_NCAMPAR 19
#define BA_NCAMPAR_BROWN 10  // xp yp c a1 a2 k1 k2 k3 p1 p2: the camera columns when no empirical term is in use
#define BA_APSCALE 1000.0  // src/main.cpp:55

struct ObsIn {
  double x, y, sx, sy, cx, cy;
  int sta, row;
};

__device__ __forceinline__ ObsIn load_obs(const double *__restrict__ obs, int64_t e) {
  const double2 *p = reinterpret_cast<const double2 *>(obs + e * BA_OBS_STRIDE);
  const double2 a = p[0], b = p[1], c = p[2], d = p[3];
  ObsIn o;
  o.x = a.x; o.y = a.y; o.sx = b.x; o.sy = b.y; o.cx = c.x; o.cy = c.y;
  o.sta = __double2loint(d.x);
  o.row = __double2hiint(d.x);
  return o;
}

struct CamIn {
  double xp, yp, c;
  double ap[16];
};

// One camera in device memory.  `mask` has bit p set when camera parameter p (0..2 = xp yp c,
// 3..18 = AP[0..15]) is estimated; the free parameters occupy columns off .. off+nc-1 of the
// camera part of the reduced system, in ascending p.
struct CamRec {
  double iop[3];
  double ap[16];
  double scale[BA_NCAMPAR];  // Jacobi column scale of free parameter k (compacted index)
  double prior[3];           // prior mean of xp yp c
  double isig[3];            // 1/sigma of the prior
  uint32_t mask;
  int32_t nc, off;
  int32_t in_program;        // bit0: IOP block is variable, bit1: AP block is variable and observed
  double lo[3], hi[3];       // box of xp yp c (src/main.cpp:4252-4256); -DBL_MAX / DBL_MAX = no bound
};

__device__ __forceinline__ CamIn load_cam(const CamRec *__restrict__ cam, int c) {
  CamIn cm;
  const CamRec &r = cam[c];
  cm.xp = r.iop[0]; cm.yp = r.iop[1]; cm.c = r.iop[2];
#pragma unroll
  for (int k = 0; k < 16; ++k) cm.ap[k] = r.ap[k];
  return cm;
}

// Residual of one image observation and, if JAC, its derivatives.
//   src/main.cpp:1322-1388 (collinearity) with the constant correction of :2096-2097.
// R, C, sk, ck come from the station record; dcol[p][row] (p = 0..18: xp yp c a1 a2 k1 k2 k3 p1 p2
// ep1..ep9) are the camera columns before selection of the free ones.
// GEN = false is the collinearity projection x = -c Xc/Zc.  GEN = true takes `model` (uniform over a
// launch) and covers the reference's other functors, whose projections all have the form
// (x, y) = c h(rho, Zc) (Xc, Yc) with rho = sqrt(Xc^2 + Yc^2):
//   1 collinearity stereographic  h = 2/(d - Zc), d = |(Xc,Yc,Zc)|            src/main.cpp:2161-2166
//   2 fisheye equidistant         h = theta/rho, theta = atan2(rho, -Zc)      src/main.cpp:2255-2256
//   3 fisheye equisolid angle     h = 2 sin(theta/2)/rho                      src/main.cpp:2341-2342
//   4 fisheye orthographic        h = sin(theta)/rho                          src/main.cpp:2427-2428
//   5 fisheye stereographic       h = 2 tan(theta/2)/rho                      src/main.cpp:2513-2514
//   6 collinearity orthographic   h = 1                                       src/main.cpp:1441-1442
//   7 omnidirectional             model 2 with (|Xc|, |Yc|) for (Xc, Yc)      src/main.cpp:2737-2738
// (model 0 through this path gives h = -1/Zc).  hr = (dh/drho)/rho and hz = dh/dZc give
// d(x,y)/d(Xc,Yc,Zc) = c [ h I + hr (Xc,Yc)(Xc,Yc)^T | hz (Xc,Yc) ].
// EMP = false leaves out the nine empirical terms AP[7..15] (all zero and none of them estimated): the residual is
// the same number, dc0/dc1 then hold BA_NCAMPAR_BROWN entries only.
template <bool JAC, bool GEN = false, bool EMP = true>
__device__ __forceinline__ void eval_obs(const ObsIn &o, const double *__restrict__ st, const double X, const double Y,
                                         const double Z, const CamIn &cm, double &r0, double &r1, double *Js /*12*/,
                                         double *Jp /*6*/, double *dc0 /*19*/, double *dc1 /*19*/, const int model = 0) {
  const double R0 = st[0], R1 = st[1], R2 = st[2], R3 = st[3], R4 = st[4], R5 = st[5], R6 = st[6], R7 = st[7], R8 = st[8];
  const double dX = X - st[9], dY = Y - st[10], dZ = Z - st[11];
  const double Xc = R0 * dX + R1 * dY + R2 * dZ;  // :1338-1340
  const double Yc = R3 * dX + R4 * dY + R5 * dZ;
  const double Zc = R6 * dX + R7 * dY + R8 * dZ;
  double iz = 0.0, x, y;
  double h = 0.0, hr = 0.0, hz = 0.0;
  if (!GEN) {
    iz = 1.0 / Zc;
    x = -cm.c * Xc * iz;  // :1349-1350
    y = -cm.c * Yc * iz;
  } else {
    const double rho2 = Xc * Xc + Yc * Yc;
    if (model == 0) {
      h = -1.0 / Zc;
      hz = h * h;
    } else if (model == 6) {
      h = 1.0;
    } else if (model == 1) {
      const double d = sqrt(rho2 + Zc * Zc), ie = 1.0 / (d - Zc), id = 1.0 / d;
      h = 2.0 * ie;
      hz = h * id;
      hr = rho2 > 0.0 ? -hz * ie : 0.0;
    } else {
      const double rho = sqrt(rho2), ir = 1.0 / rho, id2 = 1.0 / (rho2 + Zc * Zc);
      const double th = atan2(rho, -Zc);
      double f, fth;  // (x, y) = c f(theta) (Xc, Yc)/rho
      if (model == 2 || model == 7) {
        f = th;
        fth = 1.0;
      } else if (model == 3) {
        double sh, ch;
        sincos(0.5 * th, &sh, &ch);
        f = 2.0 * sh;
        fth = ch;
      } else if (model == 4) {
        double s1, c1;
        sincos(th, &s1, &c1);
        f = s1;
        fth = c1;
      } else {
        double sh, ch;
        sincos(0.5 * th, &sh, &ch);
        f = 2.0 * sh / ch;
        fth = 1.0 / (ch * ch);
      }
      h = f * ir;
      hr = (fth * (-Zc * id2) - h) * ir * ir;  // d theta/d rho = -Zc/d^2
      hz = fth * (rho * id2) * ir;             // d theta/d Zc  =  rho/d^2
    }
    x = cm.c * h * Xc;
    y = cm.c * h * Yc;
    if (model == 7) {
      x = fabs(x);
      y = fabs(y);
    }
  }
  const double xb = (o.x - cm.xp) / BA_APSCALE;  // :1357-1362
  const double yb = (o.y - cm.yp) / BA_APSCALE;
  const double rr = xb * xb + yb * yb;
  const double rr2 = rr * rr, rr3 = rr2 * rr;
  double rr4 = 0.0, rr5 = 0.0, rr6 = 0.0, rr7 = 0.0, rr8 = 0.0, rr9 = 0.0, rr10 = 0.0, rr11 = 0.0, rr12 = 0.0, rr13 = 0.0;
  const double *ap = cm.ap;
  // :1371-1376 (AP[10] multiplies rr^7 and rr^8)
  double G = ap[2] * rr + ap[3] * rr2 + ap[4] * rr3;
  if (EMP) {
    rr4 = rr3 * rr; rr5 = rr4 * rr; rr6 = rr5 * rr; rr7 = rr6 * rr; rr8 = rr7 * rr; rr9 = rr8 * rr; rr10 = rr9 * rr;
    rr11 = rr10 * rr; rr12 = rr11 * rr; rr13 = rr12 * rr;
    G += ap[7] * rr4;  // one term at a time: the sum keeps its left-to-right order
    G += ap[8] * rr5;
    G += ap[9] * rr6;
    G += ap[10] * (rr7 + rr8);
    G += ap[11] * rr9;
    G += ap[12] * rr10;
    G += ap[13] * rr11;
    G += ap[14] * rr12;
    G += ap[15] * rr13;
  }
  const double ddx = xb * G + ap[5] * (rr + 2.0 * xb * xb) + 2.0 * ap[6] * xb * yb + ap[0] * xb + ap[1] * yb;
  const double ddy = yb * G + ap[6] * (rr + 2.0 * yb * yb) + 2.0 * ap[5] * xb * yb;
  const double isx = 1.0 / o.sx, isy = 1.0 / o.sy;
  r0 = (x + cm.xp + ddx - o.cx - o.x) * isx;  // :2096-2106
  r1 = (y + cm.yp + ddy - o.cy - o.y) * isy;
  if (!JAC) return;

  const double sk = st[12], ck = st[13];
  const double woX = R1 * dZ - R2 * dY, woY = R4 * dZ - R5 * dY, woZ = R7 * dZ - R8 * dY;  // d cam / d omega
  const double wpX = -ck * Zc, wpY = sk * Zc, wpZ = ck * Xc - sk * Yc;                      // d cam / d phi
  if (!GEN) {
    // d(x,y)/d(Xc,Yc,Zc)
    const double a = -cm.c * iz, bx = -x * iz, by = -y * iz;
    const double ax = a * isx, bxx = bx * isx, ay = a * isy, byy = by * isy;
    // point block: d cam / d XYZ = R
    Jp[0] = ax * R0 + bxx * R6; Jp[1] = ax * R1 + bxx * R7; Jp[2] = ax * R2 + bxx * R8;
    Jp[3] = ay * R3 + byy * R6; Jp[4] = ay * R4 + byy * R7; Jp[5] = ay * R5 + byy * R8;
    // station block: omega, phi, kappa, Xo, Yo, Zo
    Js[0] = ax * woX + bxx * woZ;  Js[6] = ay * woY + byy * woZ;
    Js[1] = ax * wpX + bxx * wpZ;  Js[7] = ay * wpY + byy * wpZ;
    Js[2] = ax * Yc;               Js[8] = -ay * Xc;                                         // d cam / d kappa = (Yc,-Xc,0)
  } else {
    // model 7: the rows of x and y take the signs of Xc and Yc
    const double cx = (model == 7 && Xc < 0.0) ? -cm.c * isx : cm.c * isx;
    const double cy = (model == 7 && Yc < 0.0) ? -cm.c * isy : cm.c * isy;
    const double hxy = hr * Xc * Yc;
    const double ax = cx * (h + hr * Xc * Xc), bx = cx * hxy, gx = cx * hz * Xc;  // row of x wrt (Xc, Yc, Zc)
    const double ay = cy * hxy, by = cy * (h + hr * Yc * Yc), gy = cy * hz * Yc;  // row of y
    Jp[0] = ax * R0 + bx * R3 + gx * R6; Jp[1] = ax * R1 + bx * R4 + gx * R7; Jp[2] = ax * R2 + bx * R5 + gx * R8;
    Jp[3] = ay * R0 + by * R3 + gy * R6; Jp[4] = ay * R1 + by * R4 + gy * R7; Jp[5] = ay * R2 + by * R5 + gy * R8;
    Js[0] = ax * woX + bx * woY + gx * woZ;  Js[6] = ay * woX + by * woY + gy * woZ;
    Js[1] = ax * wpX + bx * wpY + gx * wpZ;  Js[7] = ay * wpX + by * wpY + gy * wpZ;
    Js[2] = ax * Yc - bx * Xc;               Js[8] = ay * Yc - by * Xc;
  }
  Js[3] = -Jp[0]; Js[4] = -Jp[1]; Js[5] = -Jp[2];
  Js[9] = -Jp[3]; Js[10] = -Jp[4]; Js[11] = -Jp[5];
  // camera block
  double dG = ap[2] + 2.0 * ap[3] * rr + 3.0 * ap[4] * rr2;
  if (EMP) {
    dG += 4.0 * ap[7] * rr3;
    dG += 5.0 * ap[8] * rr4;
    dG += 6.0 * ap[9] * rr5;
    dG += ap[10] * (7.0 * rr6 + 8.0 * rr7);
    dG += 9.0 * ap[11] * rr8;
    dG += 10.0 * ap[12] * rr9;
    dG += 11.0 * ap[13] * rr10;
    dG += 12.0 * ap[14] * rr11;
    dG += 13.0 * ap[15] * rr12;
  }
  const double xy2 = 2.0 * xb * yb;
  const double dxx = G + 2.0 * xb * xb * dG + 6.0 * ap[5] * xb + 2.0 * ap[6] * yb + ap[0];
  const double dxy = xy2 * dG + 2.0 * ap[5] * yb + 2.0 * ap[6] * xb + ap[1];
  const double dyx = xy2 * dG + 2.0 * ap[6] * xb + 2.0 * ap[5] * yb;
  const double dyy = G + 2.0 * yb * yb * dG + 6.0 * ap[6] * yb + 2.0 * ap[5] * xb;
  const double is = 1.0 / BA_APSCALE;
  dc0[0] = (1.0 - dxx * is) * isx;  dc1[0] = (-dyx * is) * isy;        // xp
  dc0[1] = (-dxy * is) * isx;       dc1[1] = (1.0 - dyy * is) * isy;   // yp
  if (!GEN) {
    dc0[2] = -Xc * iz * isx;        dc1[2] = -Yc * iz * isy;           // c
  } else {
    dc0[2] = h * (model == 7 ? fabs(Xc) : Xc) * isx;
    dc1[2] = h * (model == 7 ? fabs(Yc) : Yc) * isy;
  }
  const double xs = xb * isx, ys = yb * isy;
  dc0[3] = xs;                      dc1[3] = 0.0;                      // a1
  dc0[4] = yb * isx;                dc1[4] = 0.0;                      // a2
  dc0[5] = xs * rr;                 dc1[5] = ys * rr;                  // k1
  dc0[6] = xs * rr2;                dc1[6] = ys * rr2;                 // k2
  dc0[7] = xs * rr3;                dc1[7] = ys * rr3;                 // k3
  dc0[8] = (rr + 2.0 * xb * xb) * isx; dc1[8] = xy2 * isy;             // p1
  dc0[9] = xy2 * isx;               dc1[9] = (rr + 2.0 * yb * yb) * isy;  // p2
  if (!EMP) return;
  dc0[10] = xs * rr4;               dc1[10] = ys * rr4;                // ep1
  dc0[11] = xs * rr5;               dc1[11] = ys * rr5;
  dc0[12] = xs * rr6;               dc1[12] = ys * rr6;
  dc0[13] = xs * (rr7 + rr8);       dc1[13] = ys * (rr7 + rr8);        // ep4 = AP[10]
  dc0[14] = xs * rr9;               dc1[14] = ys * rr9;
  dc0[15] = xs * rr10;              dc1[15] = ys * rr10;
  dc0[16] = xs * rr11;              dc1[16] = ys * rr11;
  dc0[17] = xs * rr12;              dc1[17] = ys * rr12;
  dc0[18] = xs * rr13;              dc1[18] = ys * rr13;
}

// [Ceres 1.13 HuberLoss::Evaluate + Corrector, rho'' <= 0 branch] returns rho(s); scale = sqrt(rho')
__device__ __forceinline__ double huber_rho(double a, double s, double &scale) {
  if (a > 0.0 && s > a * a) {
    const double r = sqrt(s);
    const double rho1 = fmax(2.2250738585072014e-308, a / r);
    scale = sqrt(rho1);
    return 2.0 * a * r - a * a;
  }
  scale = 1.0;
  return s;
}

// symmetric 3x3 inverse, a = [xx xy xz yy yz zz]; returns false when not positive definite
__device__ __forceinline__ bool inv_sym3(const double a[6], double o[6]) {
  const double c00 = a[3] * a[5] - a[4] * a[4];
  const double c01 = a[2] * a[4] - a[1] * a[5];
  const double c02 = a[1] * a[4] - a[2] * a[3];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id;
  o[4] = (a[1] * a[2] - a[0] * a[4]) * id;
  o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
  return det > 0.0 && a[0] > 0.0 && (a[0] * a[3] - a[1] * a[1]) > 0.0;
}

__device__ __forceinline__ void sym3_mul(const double a[6], double v0, double v1, double v2, double &o0, double &o1,
                                         double &o2) {
  o0 = a[0] * v0 + a[1] * v1 + a[2] * v2;
  o1 = a[1] * v0 + a[3] * v1 + a[4] * v2;
  o2 = a[2] * v0 + a[4] * v1 + a[5] * v2;
}

// wave (64 lanes) and block sum; result valid in thread 0.  lds must hold blockDim.x/64 doubles.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// Sums sixteen values per lane over the wave with 17 exchanges instead of 96: at each of the steps 32, 16, 8, 4 a lane
// keeps one half of what it holds and takes the partner's share of that half, so the data halves while the number of
// lanes summed doubles; two plain steps finish.  The first two steps are what the lane-swap instructions of gfx950
// do in one go: v_permlane32_swap exchanges the upper half of one register with the lower half of another,
// v_permlane16_swap the odd rows of one with the even rows of the other, so "keep mine, fetch yours" is a swap and an
// add with no select and no trip through the LDS crossbar.  On return lane l holds the total of value
// 8 b5 + 4 b4 + 2 b3 + b2 (b = bits of l); the four lanes of a quad hold the same.
__device__ __forceinline__ double swap_add32(double a, double b) {  // lanes < 32: a[l] + a[l+32]; lanes >= 32: b[l-32] + b[l]
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {  // even rows: a[l] + a[l+16]; odd rows: b[l-16] + b[l]
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double wave_sum16(const double (&v)[16], int lane) {
  double a[8], b[4], c[2], d;
  const bool h3 = lane & 8, h2 = lane & 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = swap_add32(v[k], v[8 + k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = swap_add16(a[k], a[4 + k]);
#pragma unroll
  for (int k = 0; k < 2; ++k) c[k] = (h3 ? b[2 + k] : b[k]) + __shfl_xor(h3 ? b[k] : b[2 + k], 8, 64);
  d = (h2 ? c[1] : c[0]) + __shfl_xor(h2 ? c[0] : c[1], 4, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  return d;
}

// dst[k] = sum over the wave of acc[k], k < N, sixteen at a time
template <int N>
__device__ __forceinline__ void wave_sums_to(const double (&acc)[N], double *dst, int lane) {
  const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
#pragma unroll
  for (int g = 0; g < (N + 15) / 16; ++g) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (16 * g + k < N) ? acc[16 * g + k < N ? 16 * g + k : 0] : 0.0;
    const double t = wave_sum16(v, lane);
    if ((lane & 3) == 0 && 16 * g + slot < N) dst[16 * g + slot] = t;
  }
}

__device__ __forceinline__ double block_sum(double v, double *lds) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int k = 0; k < nw; ++k) t += lds[k];
  return t;
}
__device__ __forceinline__ double block_max(double v, double *lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int k = 0; k < nw; ++k) t = fmax(t, lds[k]);
  return t;
}
