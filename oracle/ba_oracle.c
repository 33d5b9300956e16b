/*
 * ba_oracle.c -- CPU restatement of the reference's Levenberg-Marquardt path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY (see ba_oracle.h).  Written to be read next to the reference:
 * every function cites the lines of /root/reference it follows.  No cleverness: sequential
 * sums in file order, dense matrices, one thread.
 *
 * Parts restated from Ceres Solver 1.13 (absent here; parity unpinned) are marked [CERES]:
 *   trust_region_minimizer.cc, levenberg_marquardt_strategy.cc, loss_function.cc (HuberLoss),
 *   corrector.cc, trust_region_step_evaluator.cc, solver.cc (SetSummaryFinalCost).
 */
#include "ba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define APSCALE 1000.0 /* src/main.cpp:55 */
#define NCAMPAR 19     /* xp,yp,c + 16 AP */

static char g_err[512];
const char *bao_last_error(void) { return g_err; }
#define FAIL(code, ...)                        \
  do {                                         \
    snprintf(g_err, sizeof g_err, __VA_ARGS__); \
    return (code);                             \
  } while (0)

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------ */
/* Threads.  The checker (libba_oracle.so) is built without OpenMP: one thread, the pragmas */
/* below vanish and every sum runs in the written order.  libba_oracle_omp.so is the same   */
/* source with -fopenmp, used only to time the all-cores CPU baseline.                      */
/* ------------------------------------------------------------------------------------ */
#ifdef _OPENMP
#include <omp.h>
#define BAO_TID() omp_get_thread_num()
#define BAO_NT() omp_get_max_threads()
#else
#define BAO_TID() 0
#define BAO_NT() 1
#endif

int bao_num_threads(void) { return BAO_NT(); }

/* Scatter-add target of a parallel loop: thread t adds into its own zeroed copy, acc_end folds the copies into
 * dst in thread order.  With one thread the copy is dst itself. */
typedef struct { double *buf, *dst; size_t n; int T; } acc_t;
static void acc_begin(acc_t *a, double *dst, size_t n) {
  a->T = BAO_NT(); a->dst = dst; a->n = n;
  a->buf = (a->T > 1) ? calloc((size_t)a->T * n + 1, sizeof(double)) : NULL;
}
static double *acc_mine(const acc_t *a) { return a->buf ? a->buf + (size_t)BAO_TID() * a->n : a->dst; }
static void acc_end(acc_t *a) {
  if (!a->buf) return;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)a->n; ++i) {
    double v = a->dst[i];
    for (int t = 0; t < a->T; ++t) v += a->buf[(size_t)t * a->n + i];
    a->dst[i] = v;
  }
  free(a->buf);
  a->buf = NULL;
}

/* ------------------------------------------------------------------------------------ */
/* Cost model                                                                             */
/* ------------------------------------------------------------------------------------ */

/* src/main.cpp:1322-1388 with the constant correction of :2096-2097.  Residual arithmetic
 * keeps the reference's operation order (including the rr*rr*... chains and AP[10] used for
 * both rr^7 and rr^8); the Jacobian is the hand derivative of exactly these expressions. */
void bao_collinearity(const double *EOP, const double *XYZ, const double *IOP, const double *AP,
                      const double *obs, double *residual, double *J) {
  bao_projection(0, EOP, XYZ, IOP, AP, obs, residual, J);
}

/* The same cost model with the projection of one of the reference's other functors (their
 * "MachineLearnedSimple" forms, which reduce to the plain ones when the correction is zero):
 *   0 collinearity :2033-2120 (:1315-1401)      1 collinearityStereographic :2133-2225
 *   2 fisheyeEquidistant :2226-2311              3 fisheyeEquisolidAngle :2312-2397
 *   4 fisheyeOrthographic :2398-2483             5 fisheyeStereographic :2484-2571
 *   6 collinearityOrthographic :1412-1490         7 omniCollinearity :2700-2790
 * They differ in the two projection lines only: (x, y) = c h(rho, Zs) (Xs, Ys), rho = sqrt(Xs^2 + Ys^2);
 * model 7 is model 2 with (|Xs|, |Ys|) in place of (Xs, Ys). */
void bao_projection(int model, const double *EOP, const double *XYZ, const double *IOP, const double *AP,
                    const double *obs, double *residual, double *J) {
  const double x_ = obs[0], y_ = obs[1], xStdDev_ = obs[2], yStdDev_ = obs[3];
  const double xMLP_ = obs[4], yMLP_ = obs[5];

  const double so = sin(EOP[0]), co = cos(EOP[0]);
  const double sp = sin(EOP[1]), cp = cos(EOP[1]);
  const double sk = sin(EOP[2]), ck = cos(EOP[2]);

  /* :1325-1335 */
  const double r11 = cp * ck;
  const double r12 = co * sk + so * sp * ck;
  const double r13 = so * sk - co * sp * ck;
  const double r21 = -cp * sk;
  const double r22 = co * ck - so * sp * sk;
  const double r23 = so * ck + co * sp * sk;
  const double r31 = sp;
  const double r32 = -so * cp;
  const double r33 = co * cp;

  /* :1338-1340 */
  const double dX = XYZ[0] - EOP[3], dY = XYZ[1] - EOP[4], dZ = XYZ[2] - EOP[5];
  const double XTemp = r11 * dX + r12 * dY + r13 * dZ;
  const double YTemp = r21 * dX + r22 * dY + r23 * dZ;
  const double ZTemp = r31 * dX + r32 * dY + r33 * dZ;

  double x, y;
  double h = 0.0, hr = 0.0, hz = 0.0; /* h, (dh/drho)/rho, dh/dZs for the Jacobian of models 1..7 */
  if (model == 0) {
    /* :1349-1350 */
    x = -IOP[2] * XTemp / ZTemp;
    y = -IOP[2] * YTemp / ZTemp;
  } else if (model == 1) {
    /* :2161-2166 */
    const double d = sqrt(XTemp * XTemp + YTemp * YTemp + ZTemp * ZTemp);
    x = (2.0 * IOP[2]) / (d - ZTemp) * XTemp;
    y = (2.0 * IOP[2]) / (d - ZTemp) * YTemp;
    const double rho2 = XTemp * XTemp + YTemp * YTemp, e = d - ZTemp;
    h = 2.0 / e;
    hr = rho2 > 0.0 ? -2.0 / (d * e * e) : 0.0;
    hz = 2.0 / (d * e);
  } else if (model == 6) {
    /* :1441-1442 */
    x = IOP[2] * XTemp;
    y = IOP[2] * YTemp;
    h = 1.0;
  } else {
    const double rho = sqrt(XTemp * XTemp + YTemp * YTemp);
    const double th = atan2(rho, -ZTemp);
    const double d2 = rho * rho + ZTemp * ZTemp;
    const double th_r = -ZTemp / d2, th_z = rho / d2; /* d theta / d rho, d theta / d Zs */
    double f, fth; /* (x, y) = c f(theta) (Xs, Ys) / rho */
    if (model == 2) { /* :2255-2256 */
      x = IOP[2] * XTemp * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp) / sqrt(XTemp * XTemp + YTemp * YTemp);
      y = IOP[2] * YTemp * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp) / sqrt(XTemp * XTemp + YTemp * YTemp);
      f = th; fth = 1.0;
    } else if (model == 3) { /* :2341-2342 */
      x = 2.0 * IOP[2] * XTemp * sin(0.5 * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp)) / sqrt(XTemp * XTemp + YTemp * YTemp);
      y = 2.0 * IOP[2] * YTemp * sin(0.5 * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp)) / sqrt(XTemp * XTemp + YTemp * YTemp);
      f = 2.0 * sin(0.5 * th); fth = cos(0.5 * th);
    } else if (model == 4) { /* :2427-2428 */
      x = IOP[2] * XTemp * sin(atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp)) / sqrt(XTemp * XTemp + YTemp * YTemp);
      y = IOP[2] * YTemp * sin(atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp)) / sqrt(XTemp * XTemp + YTemp * YTemp);
      f = sin(th); fth = cos(th);
    } else if (model == 7) { /* :2737-2738 */
      x = IOP[2] * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp) / sqrt((YTemp / XTemp) * (YTemp / XTemp) + 1.0);
      y = IOP[2] * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp) / sqrt((XTemp / YTemp) * (XTemp / YTemp) + 1.0);
      f = th; fth = 1.0;
    } else { /* 5, :2513-2514 */
      x = 2.0 * IOP[2] * XTemp * tan(0.5 * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp)) / sqrt(XTemp * XTemp + YTemp * YTemp);
      y = 2.0 * IOP[2] * YTemp * tan(0.5 * atan2(sqrt(XTemp * XTemp + YTemp * YTemp), -ZTemp)) / sqrt(XTemp * XTemp + YTemp * YTemp);
      const double ch = cos(0.5 * th);
      f = 2.0 * tan(0.5 * th); fth = 1.0 / (ch * ch);
    }
    h = f / rho;
    hr = (fth * th_r / rho - f / (rho * rho)) / rho;
    hz = fth * th_z / rho;
  }

  /* :1357-1362 */
  const double x_bar = (x_ - IOP[0]) / APSCALE;
  const double y_bar = (y_ - IOP[1]) / APSCALE;
  const double rr = x_bar * x_bar + y_bar * y_bar;

  /* :1371-1372 */
  double delta_x = x_bar * (AP[2] * rr + AP[3] * rr * rr + AP[4] * rr * rr * rr) +
                   AP[5] * (rr + 2.0 * x_bar * x_bar) + 2.0 * AP[6] * x_bar * y_bar +
                   AP[0] * x_bar + AP[1] * y_bar;
  double delta_y = y_bar * (AP[2] * rr + AP[3] * rr * rr + AP[4] * rr * rr * rr) +
                   AP[6] * (rr + 2.0 * y_bar * y_bar) + 2.0 * AP[5] * x_bar * y_bar;

  /* :1375-1376 -- AP[10] multiplies both rr^7 and rr^8, as in the reference */
  const double emp =
      AP[7] * rr * rr * rr * rr + AP[8] * rr * rr * rr * rr * rr +
      AP[9] * rr * rr * rr * rr * rr * rr + AP[10] * rr * rr * rr * rr * rr * rr * rr +
      AP[10] * rr * rr * rr * rr * rr * rr * rr * rr +
      AP[11] * rr * rr * rr * rr * rr * rr * rr * rr * rr +
      AP[12] * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr +
      AP[13] * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr +
      AP[14] * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr +
      AP[15] * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr * rr;
  delta_x += x_bar * emp;
  delta_y += y_bar * emp;

  /* :2096-2097 (== :1378-1379 when the correction is zero) */
  const double x_true = x + IOP[0] + delta_x - xMLP_;
  const double y_true = y + IOP[1] + delta_y - yMLP_;

  /* :1382-1388 */
  residual[0] = (x_true - x_) / xStdDev_;
  residual[1] = (y_true - y_) / yStdDev_;

  if (!J) return;

  /* ---- derivative of the rotation entries wrt omega, phi, kappa ---- */
  const double dRo[9] = {0.0, -so * sk + co * sp * ck, co * sk + so * sp * ck,
                         0.0, -so * ck - co * sp * sk, co * ck - so * sp * sk,
                         0.0, -co * cp,                -so * cp};
  const double dRp[9] = {-sp * ck, so * cp * ck,  -co * cp * ck,
                         sp * sk,  -so * cp * sk, co * cp * sk,
                         cp,       so * sp,       -co * sp};
  const double dRk[9] = {-cp * sk, co * ck - so * sp * sk,  so * ck + co * sp * sk,
                         -cp * ck, -co * sk - so * sp * ck, -so * sk + co * sp * ck,
                         0.0,      0.0,                     0.0};
  const double R[9] = {r11, r12, r13, r21, r22, r23, r31, r32, r33};

  /* d(XTemp,YTemp,ZTemp)/d(param) for the 9 geometric parameters */
  double dcam[9][3];
  for (int k = 0; k < 3; ++k) {
    dcam[0][k] = dRo[3 * k] * dX + dRo[3 * k + 1] * dY + dRo[3 * k + 2] * dZ;
    dcam[1][k] = dRp[3 * k] * dX + dRp[3 * k + 1] * dY + dRp[3 * k + 2] * dZ;
    dcam[2][k] = dRk[3 * k] * dX + dRk[3 * k + 1] * dY + dRk[3 * k + 2] * dZ;
    dcam[3][k] = -R[3 * k];     /* Xo */
    dcam[4][k] = -R[3 * k + 1]; /* Yo */
    dcam[5][k] = -R[3 * k + 2]; /* Zo */
    dcam[6][k] = R[3 * k];      /* X  */
    dcam[7][k] = R[3 * k + 1];  /* Y  */
    dcam[8][k] = R[3 * k + 2];  /* Z  */
  }
  double *J0 = J, *J1 = J + 28;
  const double c = IOP[2];
  for (int q = 0; q < 9; ++q) {
    const double dXc = dcam[q][0], dYc = dcam[q][1], dZc = dcam[q][2];
    double dx, dy;
    if (model == 0) {
      dx = -c * (dXc * ZTemp - XTemp * dZc) / (ZTemp * ZTemp);
      dy = -c * (dYc * ZTemp - YTemp * dZc) / (ZTemp * ZTemp);
    } else {
      /* d(x,y)/d(Xs,Ys,Zs) = c [ h I + hr (Xs,Ys)(Xs,Ys)^T | hz (Xs,Ys) ] */
      const double t = hr * (XTemp * dXc + YTemp * dYc) + hz * dZc;
      dx = c * (h * dXc + XTemp * t);
      dy = c * (h * dYc + YTemp * t);
      if (model == 7) { /* x = c h |Xs|, y = c h |Ys| */
        if (XTemp < 0.0) dx = -dx;
        if (YTemp < 0.0) dy = -dy;
      }
    }
    J0[q] = dx / xStdDev_;
    J1[q] = dy / yStdDev_;
  }

  /* ---- distortion: G(rr) = sum of the radial terms, dG = dG/drr ---- */
  const double rr2 = rr * rr, rr3 = rr2 * rr, rr4 = rr3 * rr, rr5 = rr4 * rr, rr6 = rr5 * rr;
  const double rr7 = rr6 * rr, rr8 = rr7 * rr, rr9 = rr8 * rr, rr10 = rr9 * rr;
  const double rr11 = rr10 * rr, rr12 = rr11 * rr, rr13 = rr12 * rr;
  const double G = AP[2] * rr + AP[3] * rr2 + AP[4] * rr3 + emp;
  const double dG = AP[2] + 2.0 * AP[3] * rr + 3.0 * AP[4] * rr2 + 4.0 * AP[7] * rr3 +
                    5.0 * AP[8] * rr4 + 6.0 * AP[9] * rr5 + AP[10] * (7.0 * rr6 + 8.0 * rr7) +
                    9.0 * AP[11] * rr8 + 10.0 * AP[12] * rr9 + 11.0 * AP[13] * rr10 +
                    12.0 * AP[14] * rr11 + 13.0 * AP[15] * rr12;
  const double ddx_dxb = G + 2.0 * x_bar * x_bar * dG + 6.0 * AP[5] * x_bar + 2.0 * AP[6] * y_bar + AP[0];
  const double ddx_dyb = 2.0 * x_bar * y_bar * dG + 2.0 * AP[5] * y_bar + 2.0 * AP[6] * x_bar + AP[1];
  const double ddy_dxb = 2.0 * x_bar * y_bar * dG + 2.0 * AP[6] * x_bar + 2.0 * AP[5] * y_bar;
  const double ddy_dyb = G + 2.0 * y_bar * y_bar * dG + 6.0 * AP[6] * y_bar + 2.0 * AP[5] * x_bar;

  /* IOP: xp, yp (through x_bar,y_bar = (obs - xp)/APSCALE and the +IOP term), c */
  J0[9] = (1.0 - ddx_dxb / APSCALE) / xStdDev_;
  J0[10] = (-ddx_dyb / APSCALE) / xStdDev_;
  J0[11] = (model == 0 ? -XTemp / ZTemp : h * (model == 7 ? fabs(XTemp) : XTemp)) / xStdDev_;
  J1[9] = (-ddy_dxb / APSCALE) / yStdDev_;
  J1[10] = (1.0 - ddy_dyb / APSCALE) / yStdDev_;
  J1[11] = (model == 0 ? -YTemp / ZTemp : h * (model == 7 ? fabs(YTemp) : YTemp)) / yStdDev_;

  /* AP: a1 a2 k1 k2 k3 p1 p2 ep1..ep9 */
  const double pw[16] = {0, 0, rr, rr2, rr3, 0, 0, rr4, rr5, rr6, rr7 + rr8, rr9, rr10, rr11, rr12, rr13};
  for (int k = 0; k < 16; ++k) {
    J0[12 + k] = x_bar * pw[k] / xStdDev_;
    J1[12 + k] = y_bar * pw[k] / yStdDev_;
  }
  J0[12 + 0] = x_bar / xStdDev_;
  J1[12 + 0] = 0.0;
  J0[12 + 1] = y_bar / xStdDev_;
  J1[12 + 1] = 0.0;
  J0[12 + 5] = (rr + 2.0 * x_bar * x_bar) / xStdDev_;
  J1[12 + 5] = (2.0 * x_bar * y_bar) / yStdDev_;
  J0[12 + 6] = (2.0 * x_bar * y_bar) / xStdDev_;
  J1[12 + 6] = (rr + 2.0 * y_bar * y_bar) / yStdDev_;
}

/* src/main.cpp:1064-1071 */
void bao_constrain_point(const double *p, const double *p0, const double *sigma, double *r) {
  for (int k = 0; k < 3; ++k) {
    r[k] = p[k] - p0[k];
    r[k] /= sigma[k];
  }
}

/* src/main.cpp:3821-3831: std::find returns the first match, std::distance its position */
void bao_index_first_match(const int32_t *ids, int32_t n_ids, const int32_t *keys, int32_t n_keys,
                           int32_t *out) {
  for (int32_t n = 0; n < n_keys; ++n) {
    int32_t i = 0;
    while (i < n_ids && ids[i] != keys[n]) ++i;
    out[n] = i;
  }
}

/* [CERES] HuberLoss::Evaluate (loss_function.cc) */
static void huber(double a, double s, double rho[3]) {
  const double b = a * a;
  if (a > 0.0 && s > b) {
    const double r = sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = fmax(DBL_MIN, a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s;
    rho[1] = 1.0;
    rho[2] = 0.0;
  }
}

/* ------------------------------------------------------------------------------------ */
/* Problem layout                                                                         */
/* ------------------------------------------------------------------------------------ */

typedef struct {
  int N, S, P, C;
  const ba_inputs *in;
  int *sta_used;          /* station referenced by >= 1 observation                */
  int *cam_used;          /* camera referenced by >= 1 observation                 */
  int *cam_nc, *cam_off;  /* free camera parameters per camera, offset into cam cols */
  int *cam_par;           /* [C*19] ids (0..2 IOP, 3..18 AP) of the free parameters */
  int ncam;               /* sum of cam_nc                                          */
  int ncmax;
  int nt;                 /* tangent size: 6S + 3P + ncam                           */
  int nr;                 /* reduced size: 6S + ncam                                */
  int *pt_start, *pt_obs; /* observations grouped by point, file order inside       */
  /* linearisation at the current point (loss-corrected, column-scaled) */
  double *Js, *Jp, *Jc, *res; /* per observation: 2x6, 2x3, 2xncmax, 2               */
  double *rp, *jp;            /* point prior residual / jacobian diag, [3P]          */
  double *rc, *jc;            /* IOP prior residual / jacobian diag over cam cols    */
  double *scale;              /* [nt] Jacobi scaling                                 */
  double *grad;               /* [nt] gradient J^T r from the unscaled Jacobian      */
  double cost, fixed_cost;
  const double *iop_lo, *iop_hi; /* [3C] box of the IOP blocks, NULL = none (src/main.cpp:4252-4256) */
} prob_t;

static int popcount32(unsigned v) {
  int n = 0;
  while (v) {
    n += v & 1u;
    v >>= 1;
  }
  return n;
}

static void prob_free(prob_t *q) {
  free(q->sta_used); free(q->cam_used); free(q->cam_nc); free(q->cam_off); free(q->cam_par);
  free(q->pt_start); free(q->pt_obs); free(q->Js); free(q->Jp); free(q->Jc); free(q->res);
  free(q->rp); free(q->jp); free(q->rc); free(q->jc); free(q->scale); free(q->grad);
  memset(q, 0, sizeof *q);
}

static int prob_init(prob_t *q, const ba_dims *d, const ba_inputs *in, int need_lin) {
  memset(q, 0, sizeof *q);
  q->N = d->n_obs; q->S = d->n_sta; q->P = d->n_pts; q->C = d->n_cam;
  q->in = in;
  const int N = q->N, S = q->S, P = q->P, C = q->C;
  if (N < 0 || S < 0 || P < 0 || C < 0) FAIL(BA_ERR_INVALID, "negative dimension");
  if (in->model < 0 || in->model > 7) FAIL(BA_ERR_UNSUPPORTED, "model %d not implemented", in->model);
  for (int n = 0; n < N; ++n) {
    if (in->obs_pt[n] < 0 || in->obs_pt[n] >= P) FAIL(BA_ERR_INVALID, "obs %d: point index %d out of range", n, in->obs_pt[n]);
    if (in->obs_sta[n] < 0 || in->obs_sta[n] >= S) FAIL(BA_ERR_INVALID, "obs %d: station index %d out of range", n, in->obs_sta[n]);
  }
  for (int i = 0; i < S; ++i)
    if (in->sta_cam[i] < 0 || in->sta_cam[i] >= C) FAIL(BA_ERR_INVALID, "station %d: camera index %d out of range", i, in->sta_cam[i]);
  q->sta_used = calloc(S + 1, sizeof(int));
  q->cam_used = calloc(C + 1, sizeof(int));
  q->cam_nc = calloc(C + 1, sizeof(int));
  q->cam_off = calloc(C + 1, sizeof(int));
  q->cam_par = calloc((size_t)C * NCAMPAR + 1, sizeof(int));
  for (int n = 0; n < N; ++n) {
    q->sta_used[in->obs_sta[n]] = 1;
    q->cam_used[in->sta_cam[in->obs_sta[n]]] = 1;
  }
  int off = 0;
  for (int c = 0; c < C; ++c) {
    int k = 0;
    /* IOP block always has its prior (src/main.cpp:4629), so it is referenced even when no
       observation uses the camera; the AP block of an unused camera is unreferenced and
       dropped from the reduced program [CERES]. */
    for (int b = 0; b < 3; ++b)
      if (in->iop_free_mask[c] & (1u << b)) q->cam_par[c * NCAMPAR + k++] = b;
    if (q->cam_used[c])
      for (int b = 0; b < 16; ++b)
        if (in->ap_free_mask[c] & (1u << b)) q->cam_par[c * NCAMPAR + k++] = 3 + b;
    q->cam_nc[c] = k;
    q->cam_off[c] = off;
    off += k;
    if (k > q->ncmax) q->ncmax = k;
  }
  q->ncam = off;
  q->nt = 6 * S + 3 * P + q->ncam;
  q->nr = 6 * S + q->ncam;
  /* observations by point (stable) */
  q->pt_start = calloc(P + 2, sizeof(int));
  q->pt_obs = calloc(N + 1, sizeof(int));
  for (int n = 0; n < N; ++n) q->pt_start[in->obs_pt[n] + 1]++;
  for (int j = 0; j < P; ++j) q->pt_start[j + 1] += q->pt_start[j];
  int *fill = calloc(P + 1, sizeof(int));
  for (int n = 0; n < N; ++n) {
    int j = in->obs_pt[n];
    q->pt_obs[q->pt_start[j] + fill[j]++] = n;
  }
  free(fill);
  if (need_lin) {
    q->Js = calloc((size_t)N * 12 + 1, sizeof(double));
    q->Jp = calloc((size_t)N * 6 + 1, sizeof(double));
    q->Jc = calloc((size_t)N * 2 * (q->ncmax ? q->ncmax : 1) + 1, sizeof(double));
    q->res = calloc((size_t)N * 2 + 1, sizeof(double));
    q->rp = calloc((size_t)3 * P + 1, sizeof(double));
    q->jp = calloc((size_t)3 * P + 1, sizeof(double));
    q->rc = calloc(q->ncam + 1, sizeof(double));
    q->jc = calloc(q->ncam + 1, sizeof(double));
    q->scale = calloc(q->nt + 1, sizeof(double));
    q->grad = calloc(q->nt + 1, sizeof(double));
    for (int i = 0; i < q->nt; ++i) q->scale[i] = 1.0;
  }
  return 0;
}

/* column index helpers in the tangent vector [stations | points | cameras] */
#define COL_STA(q, i) (6 * (i))
#define COL_PT(q, j) (6 * (q)->S + 3 * (j))
#define COL_CAM(q, c) (6 * (q)->S + 3 * (q)->P + (q)->cam_off[c])

/* Cost only.  [CERES] ResidualBlock::Evaluate: cost = 0.5*rho(s) with a loss, 0.5*|r|^2 without.
 * Residual blocks in the order of src/main.cpp:3845-3848, 4596-4599, 4626-4629.
 * Blocks that depend only on constant parameter blocks go to *fixed [CERES] (program removal). */
static double eval_cost(const prob_t *q, const double *eop, const double *xyz, const double *iop,
                        const double *ap, double *fixed, int apply_loss, double *residuals) {
  const ba_inputs *in = q->in;
  double cost = 0.0, fx = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : cost)
  for (int n = 0; n < q->N; ++n) {
    const int i = in->obs_sta[n], j = in->obs_pt[n], c = in->sta_cam[i];
    const double obs[6] = {in->obs_x[n], in->obs_y[n], in->obs_sx[n], in->obs_sy[n], in->obs_cx[n], in->obs_cy[n]};
    double r[2], rho[3];
    bao_projection(in->model, eop + 6 * i, xyz + 3 * j, iop + 3 * c, ap + 16 * c, obs, r, NULL);
    huber(in->huber_a, r[0] * r[0] + r[1] * r[1], rho);
    cost += 0.5 * rho[0];
    if (residuals) {
      const double sc = apply_loss ? sqrt(rho[1]) : 1.0; /* [CERES] Corrector, rho'' <= 0 branch */
      residuals[2 * n] = sc * r[0];
      residuals[2 * n + 1] = sc * r[1];
    }
  }
  for (int j = 0; j < q->P; ++j) {
    double r[3];
    bao_constrain_point(xyz + 3 * j, in->pt_prior + 3 * j, in->pt_sigma + 3 * j, r);
    cost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (residuals) memcpy(residuals + 2 * (size_t)q->N + 3 * j, r, sizeof r);
  }
  for (int c = 0; c < q->C; ++c) {
    double r[3];
    bao_constrain_point(iop + 3 * c, in->iop_prior + 3 * c, in->iop_sigma + 3 * c, r);
    const double v = 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (in->iop_free_mask[c] & 7u) cost += v; else fx += v;
    if (residuals) memcpy(residuals + 2 * (size_t)q->N + 3 * (size_t)q->P + 3 * c, r, sizeof r);
  }
  if (fixed) *fixed = fx;
  return cost;
}

/* Residuals, loss-corrected Jacobian, gradient.  [CERES] ProgramEvaluator::Evaluate +
 * Corrector::CorrectResiduals/CorrectJacobian; gradient = J^T r from the corrected, unscaled J;
 * columns are then scaled (TrustRegionMinimizer::EvaluateGradientAndJacobian, ScaleColumns). */
static void linearize(prob_t *q, const double *eop, const double *xyz, const double *iop, const double *ap) {
  const ba_inputs *in = q->in;
  const int ncm = q->ncmax ? q->ncmax : 1;
  double cost = 0.0, fx = 0.0;
  memset(q->grad, 0, sizeof(double) * q->nt);
  acc_t ga;
  acc_begin(&ga, q->grad, (size_t)q->nt);
#pragma omp parallel for schedule(static) reduction(+ : cost)
  for (int n = 0; n < q->N; ++n) {
    double *grad = acc_mine(&ga);
    const int i = in->obs_sta[n], j = in->obs_pt[n], c = in->sta_cam[i];
    const double obs[6] = {in->obs_x[n], in->obs_y[n], in->obs_sx[n], in->obs_sy[n], in->obs_cx[n], in->obs_cy[n]};
    double r[2], J[56], rho[3];
    bao_projection(in->model, eop + 6 * i, xyz + 3 * j, iop + 3 * c, ap + 16 * c, obs, r, J);
    huber(in->huber_a, r[0] * r[0] + r[1] * r[1], rho);
    cost += 0.5 * rho[0];
    const double sc = sqrt(rho[1]);
    r[0] *= sc;
    r[1] *= sc;
    double *Js = q->Js + 12 * (size_t)n, *Jp = q->Jp + 6 * (size_t)n, *Jc = q->Jc + 2 * (size_t)ncm * n;
    const double *ss = q->scale + COL_STA(q, i), *sp = q->scale + COL_PT(q, j), *scm = q->scale + COL_CAM(q, c);
    double *gs = grad + COL_STA(q, i), *gp = grad + COL_PT(q, j), *gc = grad + COL_CAM(q, c);
    for (int k = 0; k < 6; ++k) {
      const double a = sc * J[k], b = sc * J[28 + k];
      gs[k] += a * r[0] + b * r[1];
      Js[k] = a * ss[k];
      Js[6 + k] = b * ss[k];
    }
    for (int k = 0; k < 3; ++k) {
      const double a = sc * J[6 + k], b = sc * J[28 + 6 + k];
      gp[k] += a * r[0] + b * r[1];
      Jp[k] = a * sp[k];
      Jp[3 + k] = b * sp[k];
    }
    for (int k = 0; k < q->cam_nc[c]; ++k) {
      const int par = q->cam_par[c * NCAMPAR + k];
      const double a = sc * J[9 + par], b = sc * J[28 + 9 + par];
      gc[k] += a * r[0] + b * r[1];
      Jc[k] = a * scm[k];
      Jc[ncm + k] = b * scm[k];
    }
    q->res[2 * (size_t)n] = r[0];
    q->res[2 * (size_t)n + 1] = r[1];
  }
  acc_end(&ga);
  for (int j = 0; j < q->P; ++j) {
    double r[3];
    bao_constrain_point(xyz + 3 * j, in->pt_prior + 3 * j, in->pt_sigma + 3 * j, r);
    cost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; ++k) {
      const double jac = 1.0 / in->pt_sigma[3 * j + k];
      q->grad[COL_PT(q, j) + k] += jac * r[k];
      q->rp[3 * j + k] = r[k];
      q->jp[3 * j + k] = jac * q->scale[COL_PT(q, j) + k];
    }
  }
  for (int c = 0; c < q->C; ++c) {
    double r[3];
    bao_constrain_point(iop + 3 * c, in->iop_prior + 3 * c, in->iop_sigma + 3 * c, r);
    const double v = 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (!(in->iop_free_mask[c] & 7u)) { fx += v; continue; }
    cost += v;
    for (int k = 0; k < q->cam_nc[c]; ++k) {
      const int par = q->cam_par[c * NCAMPAR + k];
      q->rc[q->cam_off[c] + k] = 0.0;
      q->jc[q->cam_off[c] + k] = 0.0;
      if (par < 3) {
        const double jac = 1.0 / in->iop_sigma[3 * c + par];
        q->grad[COL_CAM(q, c) + k] += jac * r[par];
        q->rc[q->cam_off[c] + k] = r[par];
        q->jc[q->cam_off[c] + k] = jac * q->scale[COL_CAM(q, c) + k];
      }
    }
  }
  q->cost = cost;
  q->fixed_cost = fx;
}

/* [CERES] SparseMatrix::SquaredColumnNorm of the (currently scaled) Jacobian */
static void column_sqnorm(const prob_t *q, double *out) {
  const ba_inputs *in = q->in;
  const int ncm = q->ncmax ? q->ncmax : 1;
  memset(out, 0, sizeof(double) * q->nt);
  acc_t oa;
  acc_begin(&oa, out, (size_t)q->nt);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < q->N; ++n) {
    double *o = acc_mine(&oa);
    const int i = in->obs_sta[n], j = in->obs_pt[n], c = in->sta_cam[i];
    const double *Js = q->Js + 12 * (size_t)n, *Jp = q->Jp + 6 * (size_t)n, *Jc = q->Jc + 2 * (size_t)ncm * n;
    for (int k = 0; k < 6; ++k) o[COL_STA(q, i) + k] += Js[k] * Js[k] + Js[6 + k] * Js[6 + k];
    for (int k = 0; k < 3; ++k) o[COL_PT(q, j) + k] += Jp[k] * Jp[k] + Jp[3 + k] * Jp[3 + k];
    for (int k = 0; k < q->cam_nc[c]; ++k) o[COL_CAM(q, c) + k] += Jc[k] * Jc[k] + Jc[ncm + k] * Jc[ncm + k];
  }
  acc_end(&oa);
  for (int j = 0; j < 3 * q->P; ++j) out[6 * q->S + j] += q->jp[j] * q->jp[j];
  for (int k = 0; k < q->ncam; ++k) out[6 * q->S + 3 * q->P + k] += q->jc[k] * q->jc[k];
}

/* out = J * v over all rows; returns -(Jv).(r + Jv/2)  [CERES] ComputeTrustRegionStep */
static double model_cost_change(const prob_t *q, const double *v) {
  const ba_inputs *in = q->in;
  const int ncm = q->ncmax ? q->ncmax : 1;
  double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
  for (int n = 0; n < q->N; ++n) {
    const int i = in->obs_sta[n], j = in->obs_pt[n], c = in->sta_cam[i];
    const double *Js = q->Js + 12 * (size_t)n, *Jp = q->Jp + 6 * (size_t)n, *Jc = q->Jc + 2 * (size_t)ncm * n;
    const double *vs = v + COL_STA(q, i), *vp = v + COL_PT(q, j), *vc = v + COL_CAM(q, c);
    double m0 = 0.0, m1 = 0.0;
    for (int k = 0; k < 6; ++k) { m0 += Js[k] * vs[k]; m1 += Js[6 + k] * vs[k]; }
    for (int k = 0; k < 3; ++k) { m0 += Jp[k] * vp[k]; m1 += Jp[3 + k] * vp[k]; }
    for (int k = 0; k < q->cam_nc[c]; ++k) { m0 += Jc[k] * vc[k]; m1 += Jc[ncm + k] * vc[k]; }
    acc += m0 * (q->res[2 * (size_t)n] + 0.5 * m0) + m1 * (q->res[2 * (size_t)n + 1] + 0.5 * m1);
  }
  for (int j = 0; j < 3 * q->P; ++j) {
    const double m = q->jp[j] * v[6 * q->S + j];
    acc += m * (q->rp[j] + 0.5 * m);
  }
  for (int k = 0; k < q->ncam; ++k) {
    const double m = q->jc[k] * v[6 * q->S + 3 * q->P + k];
    acc += m * (q->rc[k] + 0.5 * m);
  }
  return -acc;
}

/* ------------------------------------------------------------------------------------ */
/* Dense linear algebra                                                                   */
/* ------------------------------------------------------------------------------------ */

/* in-place lower Cholesky of a row-major n x n SPD matrix; returns 0 or the failing row+1 */
static int chol_factor(double *A, int n) {
  for (int j = 0; j < n; ++j) {
    double *Aj = A + (size_t)j * n;
    double d = Aj[j];
    for (int k = 0; k < j; ++k) d -= Aj[k] * Aj[k];
    if (!(d > 0.0) || !isfinite(d)) return j + 1;
    d = sqrt(d);
    Aj[j] = d;
    for (int i = j + 1; i < n; ++i) {
      double *Ai = A + (size_t)i * n;
      double s = Ai[j];
      for (int k = 0; k < j; ++k) s -= Ai[k] * Aj[k];
      Ai[j] = s / d;
    }
  }
  return 0;
}
static void chol_solve(const double *L, int n, double *b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    const double *Li = L + (size_t)i * n;
    for (int k = 0; k < i; ++k) s -= Li[k] * b[k];
    b[i] = s / Li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}

/* symmetric 3x3 inverse, a = [xx xy xz yy yz zz] */
static int inv3(const double a[6], double o[6]) {
  const double c00 = a[3] * a[5] - a[4] * a[4];
  const double c01 = a[2] * a[4] - a[1] * a[5];
  const double c02 = a[1] * a[4] - a[2] * a[3];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  if (!(det > 0.0) || !isfinite(det)) return 1;
  const double id = 1.0 / det;
  o[0] = c00 * id;
  o[1] = c01 * id;
  o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id;
  o[4] = (a[1] * a[2] - a[0] * a[4]) * id;
  o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
  return 0;
}
static void sym3_mul(const double a[6], const double v[3], double o[3]) {
  o[0] = a[0] * v[0] + a[1] * v[1] + a[2] * v[2];
  o[1] = a[1] * v[0] + a[3] * v[1] + a[4] * v[2];
  o[2] = a[2] * v[0] + a[4] * v[1] + a[5] * v[2];
}

/* ------------------------------------------------------------------------------------ */
/* Linear solvers: all solve (J^T J + D^2) y = J^T r for the scaled J, y in tangent layout */
/* ------------------------------------------------------------------------------------ */

/* rhs of the normal equations from the scaled Jacobian */
static void normal_rhs(const prob_t *q, double *g) {
  const ba_inputs *in = q->in;
  const int ncm = q->ncmax ? q->ncmax : 1;
  memset(g, 0, sizeof(double) * q->nt);
  acc_t ga;
  acc_begin(&ga, g, (size_t)q->nt);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < q->N; ++n) {
    double *o = acc_mine(&ga);
    const int i = in->obs_sta[n], j = in->obs_pt[n], c = in->sta_cam[i];
    const double *Js = q->Js + 12 * (size_t)n, *Jp = q->Jp + 6 * (size_t)n, *Jc = q->Jc + 2 * (size_t)ncm * n;
    const double r0 = q->res[2 * (size_t)n], r1 = q->res[2 * (size_t)n + 1];
    for (int k = 0; k < 6; ++k) o[COL_STA(q, i) + k] += Js[k] * r0 + Js[6 + k] * r1;
    for (int k = 0; k < 3; ++k) o[COL_PT(q, j) + k] += Jp[k] * r0 + Jp[3 + k] * r1;
    for (int k = 0; k < q->cam_nc[c]; ++k) o[COL_CAM(q, c) + k] += Jc[k] * r0 + Jc[ncm + k] * r1;
  }
  acc_end(&ga);
  for (int j = 0; j < 3 * q->P; ++j) g[6 * q->S + j] += q->jp[j] * q->rp[j];
  for (int k = 0; k < q->ncam; ++k) g[6 * q->S + 3 * q->P + k] += q->jc[k] * q->rc[k];
}

/* SPARSE_NORMAL_CHOLESKY stand-in: dense normal matrix over all unknowns */
static int solve_full_dense(const prob_t *q, const double *D2, double *y) {
  const ba_inputs *in = q->in;
  const int nt = q->nt, ncm = q->ncmax ? q->ncmax : 1;
  if ((double)nt * nt * 8.0 > 6e9) FAIL(BA_ERR_INVALID, "full dense mode: %d unknowns is too many", nt);
  double *H = calloc((size_t)nt * nt, sizeof(double));
  if (!H) FAIL(BA_ERR_INVALID, "out of memory");
  int idx[9 + NCAMPAR];
  double v0[9 + NCAMPAR], v1[9 + NCAMPAR];
  for (int n = 0; n < q->N; ++n) {
    const int i = in->obs_sta[n], j = in->obs_pt[n], c = in->sta_cam[i];
    const double *Js = q->Js + 12 * (size_t)n, *Jp = q->Jp + 6 * (size_t)n, *Jc = q->Jc + 2 * (size_t)ncm * n;
    int m = 0;
    for (int k = 0; k < 6; ++k, ++m) { idx[m] = COL_STA(q, i) + k; v0[m] = Js[k]; v1[m] = Js[6 + k]; }
    for (int k = 0; k < 3; ++k, ++m) { idx[m] = COL_PT(q, j) + k; v0[m] = Jp[k]; v1[m] = Jp[3 + k]; }
    for (int k = 0; k < q->cam_nc[c]; ++k, ++m) { idx[m] = COL_CAM(q, c) + k; v0[m] = Jc[k]; v1[m] = Jc[ncm + k]; }
    for (int a = 0; a < m; ++a)
      for (int b = 0; b < m; ++b) H[(size_t)idx[a] * nt + idx[b]] += v0[a] * v0[b] + v1[a] * v1[b];
  }
  for (int j = 0; j < 3 * q->P; ++j) H[(size_t)(6 * q->S + j) * nt + 6 * q->S + j] += q->jp[j] * q->jp[j];
  for (int k = 0; k < q->ncam; ++k) {
    const size_t cidx = 6 * q->S + 3 * q->P + k;
    H[cidx * nt + cidx] += q->jc[k] * q->jc[k];
  }
  for (int i = 0; i < nt; ++i) H[(size_t)i * nt + i] += D2[i];
  /* columns of stations without observations are not part of the reduced program */
  for (int i = 0; i < q->S; ++i)
    if (!q->sta_used[i])
      for (int k = 0; k < 6; ++k) H[(size_t)(6 * i + k) * nt + 6 * i + k] = 1.0;
  normal_rhs(q, y);
  const int bad = chol_factor(H, nt);
  if (bad) { free(H); FAIL(BA_ERR_NUMERIC, "Cholesky failed at column %d of %d", bad - 1, nt); }
  chol_solve(H, nt, y);
  free(H);
  return 0;
}

/* Point elimination shared by the two Schur modes: V_j^-1 and g_pj */
static int eliminate_points(const prob_t *q, const double *D2, const double *g, double *Vinv) {
  int bad = q->P; /* first point whose block cannot be inverted */
#pragma omp parallel for schedule(static) reduction(min : bad)
  for (int j = 0; j < q->P; ++j) {
    double V[6] = {0, 0, 0, 0, 0, 0};
    for (int e = q->pt_start[j]; e < q->pt_start[j + 1]; ++e) {
      const double *Jp = q->Jp + 6 * (size_t)q->pt_obs[e];
      V[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3];
      V[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
      V[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
      V[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
      V[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5];
      V[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
    }
    const double *jp = q->jp + 3 * j, *d2 = D2 + 6 * q->S + 3 * j;
    V[0] += jp[0] * jp[0] + d2[0];
    V[3] += jp[1] * jp[1] + d2[1];
    V[5] += jp[2] * jp[2] + d2[2];
    if (inv3(V, Vinv + 6 * (size_t)j) && j < bad) bad = j;
  }
  (void)g;
  if (bad < q->P) FAIL(BA_ERR_NUMERIC, "point block %d is not positive definite", bad);
  return 0;
}

/* camera-side row of one observation as (index,value) lists in the reduced layout
 * [stations | cameras]; returns the count */
static int cam_side(const prob_t *q, int n, int *idx, double *v0, double *v1) {
  const ba_inputs *in = q->in;
  const int ncm = q->ncmax ? q->ncmax : 1;
  const int i = in->obs_sta[n], c = in->sta_cam[i];
  const double *Js = q->Js + 12 * (size_t)n, *Jc = q->Jc + 2 * (size_t)ncm * n;
  int m = 0;
  for (int k = 0; k < 6; ++k, ++m) { idx[m] = 6 * i + k; v0[m] = Js[k]; v1[m] = Js[6 + k]; }
  for (int k = 0; k < q->cam_nc[c]; ++k, ++m) { idx[m] = 6 * q->S + q->cam_off[c] + k; v0[m] = Jc[k]; v1[m] = Jc[ncm + k]; }
  return m;
}

static void back_substitute(const prob_t *q, const double *Vinv, const double *g, const double *yc, double *y) {
  /* y_p = V^-1 (g_p - W^T y_c), W^T y_c = sum_obs Jp^T (Jc_side y_c) */
#pragma omp parallel for schedule(static)
  for (int j = 0; j < q->P; ++j) {
    int idx[6 + NCAMPAR];
    double v0[6 + NCAMPAR], v1[6 + NCAMPAR];
    double t[3] = {0, 0, 0};
    for (int e = q->pt_start[j]; e < q->pt_start[j + 1]; ++e) {
      const int n = q->pt_obs[e];
      const int m = cam_side(q, n, idx, v0, v1);
      double a0 = 0, a1 = 0;
      for (int k = 0; k < m; ++k) { a0 += v0[k] * yc[idx[k]]; a1 += v1[k] * yc[idx[k]]; }
      const double *Jp = q->Jp + 6 * (size_t)n;
      for (int k = 0; k < 3; ++k) t[k] += Jp[k] * a0 + Jp[3 + k] * a1;
    }
    double rhs[3], out[3];
    for (int k = 0; k < 3; ++k) rhs[k] = g[6 * q->S + 3 * j + k] - t[k];
    sym3_mul(Vinv + 6 * (size_t)j, rhs, out);
    for (int k = 0; k < 3; ++k) y[6 * q->S + 3 * j + k] = out[k];
  }
  for (int i = 0; i < 6 * q->S; ++i) y[i] = yc[i];
  for (int k = 0; k < q->ncam; ++k) y[6 * q->S + 3 * q->P + k] = yc[6 * q->S + k];
}

static int solve_schur_dense(const prob_t *q, const double *D2, double *y) {
  const int nr = q->nr, S6 = 6 * q->S;
  if ((double)nr * nr * 8.0 > 6e9) FAIL(BA_ERR_INVALID, "dense Schur mode: reduced size %d is too large", nr);
  double *g = calloc(q->nt + 1, sizeof(double));
  double *Vinv = calloc((size_t)6 * q->P + 1, sizeof(double));
  double *Sm = calloc((size_t)nr * nr + 1, sizeof(double));
  double *b = calloc(nr + 1, sizeof(double));
  int *loc = malloc(sizeof(int) * (nr + 1));
  int rc = 0;
  normal_rhs(q, g);
  if ((rc = eliminate_points(q, D2, g, Vinv))) goto done;
  for (int i = 0; i < nr; ++i) loc[i] = -1;
  int idx[6 + NCAMPAR];
  double v0[6 + NCAMPAR], v1[6 + NCAMPAR];
  /* U part */
  for (int n = 0; n < q->N; ++n) {
    const int m = cam_side(q, n, idx, v0, v1);
    for (int a = 0; a < m; ++a)
      for (int c2 = 0; c2 < m; ++c2) Sm[(size_t)idx[a] * nr + idx[c2]] += v0[a] * v0[c2] + v1[a] * v1[c2];
  }
  for (int k = 0; k < q->ncam; ++k) Sm[(size_t)(S6 + k) * nr + S6 + k] += q->jc[k] * q->jc[k];
  for (int i = 0; i < S6; ++i) Sm[(size_t)i * nr + i] += D2[i];
  for (int k = 0; k < q->ncam; ++k) Sm[(size_t)(S6 + k) * nr + S6 + k] += D2[S6 + 3 * q->P + k];
  for (int i = 0; i < S6; ++i) b[i] = g[i];
  for (int k = 0; k < q->ncam; ++k) b[S6 + k] = g[S6 + 3 * q->P + k];
  /* - Y V^-1 Y^T and - Y V^-1 g_p, point by point over the merged column list */
  {
    int cap = 64;
    int *cols = malloc(sizeof(int) * cap);
    double *Y = malloc(sizeof(double) * 3 * cap), *Z = malloc(sizeof(double) * 3 * cap);
    for (int j = 0; j < q->P; ++j) {
      int m = 0;
      for (int e = q->pt_start[j]; e < q->pt_start[j + 1]; ++e) {
        const int n = q->pt_obs[e];
        const int mm = cam_side(q, n, idx, v0, v1);
        const double *Jp = q->Jp + 6 * (size_t)n;
        for (int a = 0; a < mm; ++a) {
          int l = loc[idx[a]];
          if (l < 0) {
            if (m == cap) {
              cap *= 2;
              cols = realloc(cols, sizeof(int) * cap);
              Y = realloc(Y, sizeof(double) * 3 * cap);
              Z = realloc(Z, sizeof(double) * 3 * cap);
            }
            l = m++;
            loc[idx[a]] = l;
            cols[l] = idx[a];
            Y[3 * l] = Y[3 * l + 1] = Y[3 * l + 2] = 0.0;
          }
          for (int k = 0; k < 3; ++k) Y[3 * l + k] += v0[a] * Jp[k] + v1[a] * Jp[3 + k];
        }
      }
      const double *Vi = Vinv + 6 * (size_t)j;
      double vg[3];
      sym3_mul(Vi, g + S6 + 3 * j, vg);
      for (int a = 0; a < m; ++a) {
        sym3_mul(Vi, Y + 3 * a, Z + 3 * a);
        b[cols[a]] -= Y[3 * a] * vg[0] + Y[3 * a + 1] * vg[1] + Y[3 * a + 2] * vg[2];
      }
      for (int a = 0; a < m; ++a) {
        double *row = Sm + (size_t)cols[a] * nr;
        for (int c2 = 0; c2 < m; ++c2)
          row[cols[c2]] -= Z[3 * a] * Y[3 * c2] + Z[3 * a + 1] * Y[3 * c2 + 1] + Z[3 * a + 2] * Y[3 * c2 + 2];
      }
      for (int a = 0; a < m; ++a) loc[cols[a]] = -1;
    }
    free(cols); free(Y); free(Z);
  }
  for (int i = 0; i < q->S; ++i)
    if (!q->sta_used[i])
      for (int k = 0; k < 6; ++k) Sm[(size_t)(6 * i + k) * nr + 6 * i + k] = 1.0;
  {
    const int bad = chol_factor(Sm, nr);
    if (bad) { snprintf(g_err, sizeof g_err, "reduced Cholesky failed at column %d of %d", bad - 1, nr); rc = BA_ERR_NUMERIC; goto done; }
  }
  chol_solve(Sm, nr, b);
  back_substitute(q, Vinv, g, b, y);
done:
  free(g); free(Vinv); free(Sm); free(b); free(loc);
  return rc;
}

/* implicit reduced operator: out = S p,  S = Jc^T (I - Jp V^-1 Jp^T) Jc + prior + D^2 */
static void schur_apply(const prob_t *q, const double *D2, const double *Vinv, const double *p, double *out, double *tmp2N) {
  const int S6 = 6 * q->S, nr = q->nr;
  for (int i = 0; i < nr; ++i) out[i] = 0.0;
  acc_t oa;
  acc_begin(&oa, out, (size_t)nr);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < q->P; ++j) {
    double *o = acc_mine(&oa);
    int idx[6 + NCAMPAR];
    double v0[6 + NCAMPAR], v1[6 + NCAMPAR];
    double t[3] = {0, 0, 0}, u[3];
    for (int e = q->pt_start[j]; e < q->pt_start[j + 1]; ++e) {
      const int n = q->pt_obs[e];
      const int m = cam_side(q, n, idx, v0, v1);
      double a0 = 0, a1 = 0;
      for (int k = 0; k < m; ++k) { a0 += v0[k] * p[idx[k]]; a1 += v1[k] * p[idx[k]]; }
      tmp2N[2 * (size_t)n] = a0;
      tmp2N[2 * (size_t)n + 1] = a1;
      const double *Jp = q->Jp + 6 * (size_t)n;
      for (int k = 0; k < 3; ++k) t[k] += Jp[k] * a0 + Jp[3 + k] * a1;
    }
    sym3_mul(Vinv + 6 * (size_t)j, t, u);
    for (int e = q->pt_start[j]; e < q->pt_start[j + 1]; ++e) {
      const int n = q->pt_obs[e];
      const double *Jp = q->Jp + 6 * (size_t)n;
      const double b0 = tmp2N[2 * (size_t)n] - (Jp[0] * u[0] + Jp[1] * u[1] + Jp[2] * u[2]);
      const double b1 = tmp2N[2 * (size_t)n + 1] - (Jp[3] * u[0] + Jp[4] * u[1] + Jp[5] * u[2]);
      const int m = cam_side(q, n, idx, v0, v1);
      for (int k = 0; k < m; ++k) o[idx[k]] += v0[k] * b0 + v1[k] * b1;
    }
  }
  acc_end(&oa);
  for (int k = 0; k < q->ncam; ++k) out[S6 + k] += q->jc[k] * q->jc[k] * p[S6 + k];
  for (int i = 0; i < S6; ++i) out[i] += D2[i] * p[i];
  for (int k = 0; k < q->ncam; ++k) out[S6 + k] += D2[S6 + 3 * q->P + k] * p[S6 + k];
  for (int i = 0; i < q->S; ++i)
    if (!q->sta_used[i])
      for (int k = 0; k < 6; ++k) out[6 * i + k] = p[6 * i + k];
}

static int g_last_pcg_iters;

static int solve_schur_pcg(const prob_t *q, const double *D2, double rtol, int max_iters, double *y) {
  const int nr = q->nr, S6 = 6 * q->S, nc = q->ncam;
  const ba_inputs *in = q->in;
  double *g = calloc(q->nt + 1, sizeof(double));
  double *Vinv = calloc((size_t)6 * q->P + 1, sizeof(double));
  double *b = calloc(nr + 1, sizeof(double)), *x = calloc(nr + 1, sizeof(double));
  double *r = calloc(nr + 1, sizeof(double)), *z = calloc(nr + 1, sizeof(double));
  double *p = calloc(nr + 1, sizeof(double)), *Ap = calloc(nr + 1, sizeof(double));
  double *tmp = calloc((size_t)2 * q->N + 1, sizeof(double));
  double *Mb = calloc((size_t)36 * q->S + 1, sizeof(double)); /* station blocks of S, factored */
  double *Mc = calloc((size_t)nc * nc + 1, sizeof(double));   /* exact camera block of S, factored */
  int rc = 0;
  g_last_pcg_iters = 0;
  normal_rhs(q, g);
  if ((rc = eliminate_points(q, D2, g, Vinv))) goto done;
  /* rhs: b = g_c - sum Jc^T Jp V^-1 g_p  == Jc^T (r - Jp V^-1 g_p) + prior part */
  {
    for (int i = 0; i < S6; ++i) b[i] = g[i];
    for (int k = 0; k < nc; ++k) b[S6 + k] = g[S6 + 3 * q->P + k];
    const int ncm = q->ncmax ? q->ncmax : 1;
    double *Ball = calloc((size_t)3 * (nc + 1) * BAO_NT(), sizeof(double));
    acc_t ba, mba, mca;
    acc_begin(&ba, b, (size_t)nr);
    acc_begin(&mba, Mb, (size_t)36 * q->S);
    acc_begin(&mca, Mc, (size_t)nc * nc);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < q->P; ++j) {
      double *b = acc_mine(&ba), *Mb = acc_mine(&mba), *Mc = acc_mine(&mca); /* this thread's copies */
      double *B = Ball + (size_t)3 * (nc + 1) * BAO_TID();
      int idx[6 + NCAMPAR];
      double v0[6 + NCAMPAR], v1[6 + NCAMPAR];
      double vg[3];
      sym3_mul(Vinv + 6 * (size_t)j, g + S6 + 3 * j, vg);
      const double *Vi = Vinv + 6 * (size_t)j;
      for (int k = 0; k < 3 * nc; ++k) B[k] = 0.0;
      for (int e = q->pt_start[j]; e < q->pt_start[j + 1]; ++e) {
        const int n = q->pt_obs[e];
        const double *Jp = q->Jp + 6 * (size_t)n;
        const double h0 = Jp[0] * vg[0] + Jp[1] * vg[1] + Jp[2] * vg[2];
        const double h1 = Jp[3] * vg[0] + Jp[4] * vg[1] + Jp[5] * vg[2];
        const int m = cam_side(q, n, idx, v0, v1);
        for (int k = 0; k < m; ++k) b[idx[k]] -= v0[k] * h0 + v1[k] * h1;
        /* station block preconditioner: Js^T (I - Jp V^-1 Jp^T) Js */
        double t0[3], t1[3];
        sym3_mul(Vi, Jp, t0);
        sym3_mul(Vi, Jp + 3, t1);
        const double h00 = Jp[0] * t0[0] + Jp[1] * t0[1] + Jp[2] * t0[2];
        const double h01 = Jp[0] * t1[0] + Jp[1] * t1[1] + Jp[2] * t1[2];
        const double h11 = Jp[3] * t1[0] + Jp[4] * t1[1] + Jp[5] * t1[2];
        const double m00 = 1.0 - h00, m01 = -h01, m11 = 1.0 - h11;
        const int i = in->obs_sta[n], c = in->sta_cam[i];
        const double *Js = q->Js + 12 * (size_t)n;
        double *blk = Mb + 36 * (size_t)i;
        for (int a = 0; a < 6; ++a)
          for (int c2 = 0; c2 < 6; ++c2)
            blk[6 * a + c2] += Js[a] * (m00 * Js[c2] + m01 * Js[6 + c2]) + Js[6 + a] * (m01 * Js[c2] + m11 * Js[6 + c2]);
        /* camera block: U_cc and B_j = sum Jc^T Jp */
        const double *Jc = q->Jc + 2 * (size_t)ncm * n;
        for (int a = 0; a < q->cam_nc[c]; ++a) {
          const int ca = q->cam_off[c] + a;
          for (int c2 = 0; c2 < q->cam_nc[c]; ++c2)
            Mc[(size_t)ca * nc + q->cam_off[c] + c2] += Jc[a] * Jc[c2] + Jc[ncm + a] * Jc[ncm + c2];
          for (int k = 0; k < 3; ++k) B[3 * ca + k] += Jc[a] * Jp[k] + Jc[ncm + a] * Jp[3 + k];
        }
      }
      for (int a = 0; a < nc; ++a) {
        double zb[3];
        sym3_mul(Vi, B + 3 * a, zb);
        if (zb[0] == 0.0 && zb[1] == 0.0 && zb[2] == 0.0) continue;
        for (int c2 = 0; c2 < nc; ++c2)
          Mc[(size_t)a * nc + c2] -= zb[0] * B[3 * c2] + zb[1] * B[3 * c2 + 1] + zb[2] * B[3 * c2 + 2];
      }
    }
    acc_end(&ba);
    acc_end(&mba);
    acc_end(&mca);
    free(Ball);
    for (int i = 0; i < q->S; ++i) {
      double *blk = Mb + 36 * (size_t)i;
      for (int k = 0; k < 6; ++k) blk[7 * k] += D2[6 * i + k];
      if (!q->sta_used[i]) { memset(blk, 0, 36 * sizeof(double)); for (int k = 0; k < 6; ++k) blk[7 * k] = 1.0; }
      if (chol_factor(blk, 6)) { snprintf(g_err, sizeof g_err, "station block %d not positive definite", i); rc = BA_ERR_NUMERIC; goto done; }
    }
    for (int k = 0; k < nc; ++k) Mc[(size_t)k * nc + k] += q->jc[k] * q->jc[k] + D2[S6 + 3 * q->P + k];
    if (nc && chol_factor(Mc, nc)) { snprintf(g_err, sizeof g_err, "camera block not positive definite"); rc = BA_ERR_NUMERIC; goto done; }
  }
  for (int i = 0; i < q->S; ++i)
    if (!q->sta_used[i]) for (int k = 0; k < 6; ++k) b[6 * i + k] = 0.0;
  /* PCG, x0 = 0 */
  {
    double rz = 0, rz0;
    memcpy(r, b, sizeof(double) * nr);
    memcpy(z, r, sizeof(double) * nr);
    for (int i = 0; i < q->S; ++i) chol_solve(Mb + 36 * (size_t)i, 6, z + 6 * i);
    if (nc) chol_solve(Mc, nc, z + S6);
    for (int i = 0; i < nr; ++i) rz += r[i] * z[i];
    rz0 = rz;
    memcpy(p, z, sizeof(double) * nr);
    int it = 0;
    while (it < max_iters && rz > 0.0 && sqrt(rz) > rtol * sqrt(rz0)) {
      schur_apply(q, D2, Vinv, p, Ap, tmp);
      double pAp = 0;
      for (int i = 0; i < nr; ++i) pAp += p[i] * Ap[i];
      if (!(pAp > 0.0)) break;
      const double alpha = rz / pAp;
      for (int i = 0; i < nr; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; }
      memcpy(z, r, sizeof(double) * nr);
      for (int i = 0; i < q->S; ++i) chol_solve(Mb + 36 * (size_t)i, 6, z + 6 * i);
      if (nc) chol_solve(Mc, nc, z + S6);
      double rzn = 0;
      for (int i = 0; i < nr; ++i) rzn += r[i] * z[i];
      const double beta = rzn / rz;
      rz = rzn;
      for (int i = 0; i < nr; ++i) p[i] = z[i] + beta * p[i];
      ++it;
    }
    g_last_pcg_iters = it;
  }
  back_substitute(q, Vinv, g, x, y);
done:
  free(g); free(Vinv); free(b); free(x); free(r); free(z); free(p); free(Ap); free(tmp); free(Mb); free(Mc);
  return rc;
}

/* ------------------------------------------------------------------------------------ */
/* Public evaluation entry points                                                         */
/* ------------------------------------------------------------------------------------ */

int bao_residual_jacobian_all(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz,
                              const double *iop, const double *ap, double *r, double *J) {
  for (int n = 0; n < d->n_obs; ++n) {
    const int i = in->obs_sta[n], j = in->obs_pt[n];
    if (i < 0 || i >= d->n_sta || j < 0 || j >= d->n_pts) FAIL(BA_ERR_INVALID, "obs %d: index out of range", n);
    const int c = in->sta_cam[i];
    if (c < 0 || c >= d->n_cam) FAIL(BA_ERR_INVALID, "station %d: camera out of range", i);
    const double obs[6] = {in->obs_x[n], in->obs_y[n], in->obs_sx[n], in->obs_sy[n], in->obs_cx[n], in->obs_cy[n]};
    double rr[2], JJ[56];
    bao_projection(in->model, eop + 6 * i, xyz + 3 * j, iop + 3 * c, ap + 16 * c, obs, rr, J ? JJ : NULL);
    if (r) { r[2 * (size_t)n] = rr[0]; r[2 * (size_t)n + 1] = rr[1]; }
    if (J) memcpy(J + 56 * (size_t)n, JJ, sizeof JJ);
  }
  return 0;
}

/* problem.Evaluate(EvaluateOptions(), &cost, &residuals, NULL, NULL): src/main.cpp:4853 */
int bao_evaluate(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz, const double *iop,
                 const double *ap, int apply_loss, double *cost, double *residuals) {
  prob_t q;
  int rc = prob_init(&q, d, in, 0);
  if (rc) { prob_free(&q); return rc; }
  double fixed = 0.0;
  const double c = eval_cost(&q, eop, xyz, iop, ap, &fixed, apply_loss, residuals);
  if (cost) *cost = c + fixed;
  prob_free(&q);
  return 0;
}

/* Structure of the CRSMatrix of src/main.cpp:6664.  [CERES] CompressedRowJacobianWriter:
 * one column range per parameter block in the order of src/main.cpp:3793-3800, width = the
 * block's tangent size; a constant block keeps its columns but gets no entries. */
int bao_sparsity(const ba_dims *d, const ba_inputs *in, int32_t *num_rows, int32_t *num_cols, int64_t *nnz,
                 int32_t *rows, int32_t *cols) {
  const int N = d->n_obs, S = d->n_sta, P = d->n_pts, C = d->n_cam;
  int *iop_w = malloc(sizeof(int) * (C + 1)), *ap_w = malloc(sizeof(int) * (C + 1));
  int *iop_col = malloc(sizeof(int) * (C + 1)), *ap_col = malloc(sizeof(int) * (C + 1));
  int col = 6 * S + 3 * P;
  for (int c = 0; c < C; ++c) {
    const unsigned m = in->iop_free_mask[c] & 7u;
    iop_w[c] = m ? popcount32(m) : 3;
    iop_col[c] = col;
    col += iop_w[c];
  }
  for (int c = 0; c < C; ++c) {
    const unsigned m = in->ap_free_mask[c];
    ap_w[c] = m ? popcount32(m) : 16;
    ap_col[c] = col;
    col += ap_w[c];
  }
  int64_t z = 0;
  int row = 0;
  if (rows) rows[0] = 0;
  for (int n = 0; n < N; ++n) {
    const int i = in->obs_sta[n], j = in->obs_pt[n];
    if (i < 0 || i >= S || j < 0 || j >= P) { free(iop_w); free(ap_w); free(iop_col); free(ap_col); FAIL(BA_ERR_INVALID, "obs %d: index out of range", n); }
    const int c = in->sta_cam[i];
    for (int rrow = 0; rrow < 2; ++rrow) {
      for (int k = 0; k < 6; ++k, ++z) if (cols) cols[z] = 6 * i + k;
      for (int k = 0; k < 3; ++k, ++z) if (cols) cols[z] = 6 * S + 3 * j + k;
      if (in->iop_free_mask[c] & 7u) for (int k = 0; k < iop_w[c]; ++k, ++z) if (cols) cols[z] = iop_col[c] + k;
      if (in->ap_free_mask[c]) for (int k = 0; k < ap_w[c]; ++k, ++z) if (cols) cols[z] = ap_col[c] + k;
      ++row;
      if (rows) rows[row] = (int32_t)z;
    }
  }
  /* the autodiff 3x3 prior blocks are written as full blocks: 3 structural entries per row */
  for (int j = 0; j < P; ++j)
    for (int k = 0; k < 3; ++k) {
      for (int w = 0; w < 3; ++w, ++z) if (cols) cols[z] = 6 * S + 3 * j + w;
      ++row;
      if (rows) rows[row] = (int32_t)z;
    }
  for (int c = 0; c < C; ++c) {
    /* autodiff 3x3 block: structurally dense over the block's tangent columns; a free
       coordinate of a subset-parameterised block maps to one tangent column */
    for (int k = 0; k < 3; ++k) {
      if (in->iop_free_mask[c] & 7u) for (int w = 0; w < iop_w[c]; ++w, ++z) if (cols) cols[z] = iop_col[c] + w;
      ++row;
      if (rows) rows[row] = (int32_t)z;
    }
  }
  if (num_rows) *num_rows = row;
  if (num_cols) *num_cols = col;
  if (nnz) *nnz = z;
  free(iop_w); free(ap_w); free(iop_col); free(ap_col);
  return 0;
}

double bao_linearize_once(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz,
                          const double *iop, const double *ap) {
  prob_t q;
  if (prob_init(&q, d, in, 1)) { prob_free(&q); return NAN; }
  linearize(&q, eop, xyz, iop, ap);
  double *D2 = calloc(q.nt + 1, sizeof(double)), *Vinv = calloc((size_t)6 * q.P + 1, sizeof(double));
  eliminate_points(&q, D2, NULL, Vinv);
  const double c = q.cost + Vinv[0] * 0.0;
  free(D2); free(Vinv);
  prob_free(&q);
  return c;
}

/* ------------------------------------------------------------------------------------ */
/* Covariance of the estimated parameters                                                 */
/* ------------------------------------------------------------------------------------ */
/* src/main.cpp:5261-5279, 5443-5565: ceres::Covariance with apply_loss_function = true over all
 * parameter blocks; the reference keeps the diagonal (xyzVariance, eopVariance, iopVariance,
 * apVariance) and multiplies by aposterioriVariance, which it has reset to 1 (:4955).
 * [CERES] Covariance: Cx = (J^T J)^-1 of the reduced program at the given point, J with the loss
 * correction, in the tangent space of each block; coordinates held constant by a
 * SubsetParameterization and constant blocks get zero rows and columns.  The reference asks for
 * the DENSE_SVD pseudo-inverse (null_space_rank = -1); with the XYZ and IOP priors in place the
 * matrix has full rank and the pseudo-inverse is the inverse, which is what is computed here (by
 * Cholesky of the symmetrically scaled normal matrix).  A rank-deficient network is reported as an
 * error instead of being truncated. */
/* The full normal matrix at the given point, scaled to a unit diagonal (sc: the scale factors) and replaced by its
 * Cholesky factor (lower triangle, row-major); q is left initialised for the caller (prob_free). */
static int cov_factor(prob_t *q, const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz, const double *iop,
                      const double *ap, double **H_out, double **sc_out) {
  int rc = prob_init(q, d, in, 1);
  if (rc) { prob_free(q); return rc; }
  linearize(q, eop, xyz, iop, ap);
  const int nt = q->nt, ncm = q->ncmax ? q->ncmax : 1;
  if ((double)nt * nt * 8.0 > 6e9) { prob_free(q); FAIL(BA_ERR_INVALID, "covariance: %d unknowns is too many for the dense oracle", nt); }
  double *H = calloc((size_t)nt * nt + 1, sizeof(double)), *sc = calloc(nt + 1, sizeof(double));
  int idx[9 + NCAMPAR];
  double v0[9 + NCAMPAR], v1[9 + NCAMPAR];
  for (int n = 0; n < q->N; ++n) {
    const int i = in->obs_sta[n], j = in->obs_pt[n], c = in->sta_cam[i];
    const double *Js = q->Js + 12 * (size_t)n, *Jp = q->Jp + 6 * (size_t)n, *Jc = q->Jc + 2 * (size_t)ncm * n;
    int m = 0;
    for (int k = 0; k < 6; ++k, ++m) { idx[m] = COL_STA(q, i) + k; v0[m] = Js[k]; v1[m] = Js[6 + k]; }
    for (int k = 0; k < 3; ++k, ++m) { idx[m] = COL_PT(q, j) + k; v0[m] = Jp[k]; v1[m] = Jp[3 + k]; }
    for (int k = 0; k < q->cam_nc[c]; ++k, ++m) { idx[m] = COL_CAM(q, c) + k; v0[m] = Jc[k]; v1[m] = Jc[ncm + k]; }
    for (int a = 0; a < m; ++a)
      for (int b = 0; b < m; ++b) H[(size_t)idx[a] * nt + idx[b]] += v0[a] * v0[b] + v1[a] * v1[b];
  }
  for (int j = 0; j < 3 * q->P; ++j) H[(size_t)(6 * q->S + j) * nt + 6 * q->S + j] += q->jp[j] * q->jp[j];
  for (int k = 0; k < q->ncam; ++k) {
    const size_t cidx = 6 * (size_t)q->S + 3 * (size_t)q->P + k;
    H[cidx * nt + cidx] += q->jc[k] * q->jc[k];
  }
  /* stations without observations are not in the program: keep them out of the inverse */
  for (int i = 0; i < q->S; ++i)
    if (!q->sta_used[i])
      for (int k = 0; k < 6; ++k) H[(size_t)(6 * i + k) * nt + 6 * i + k] = 1.0;
  for (int i = 0; i < nt; ++i) {
    const double dgl = H[(size_t)i * nt + i];
    if (!(dgl > 0.0)) { free(H); free(sc); prob_free(q); FAIL(BA_ERR_NUMERIC, "covariance: unknown %d has no information", i); }
    sc[i] = 1.0 / sqrt(dgl);
  }
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < nt; ++j) H[(size_t)i * nt + j] *= sc[i] * sc[j];
  const int bad = chol_factor(H, nt);
  if (bad) { free(H); free(sc); prob_free(q); FAIL(BA_ERR_NUMERIC, "covariance: normal matrix is rank deficient (column %d of %d)", bad - 1, nt); }
  *H_out = H; *sc_out = sc;
  return 0;
}

int bao_covariance(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz, const double *iop,
                   const double *ap, double *var_eop, double *var_xyz, double *var_iop, double *var_ap) {
  prob_t q;
  double *H = NULL, *sc = NULL;
  const int rc = cov_factor(&q, d, in, eop, xyz, iop, ap, &H, &sc);
  if (rc) return rc;
  const int nt = q.nt;
  /* diag((L L^T)^-1) = squared column norms of L^-1; L^-1 column c by forward substitution */
  double *var = calloc(nt + 1, sizeof(double)), *x = calloc(nt + 1, sizeof(double));
  for (int c = 0; c < nt; ++c) {
    for (int i = c; i < nt; ++i) {
      double sum = (i == c) ? 1.0 : 0.0;
      const double *Li = H + (size_t)i * nt;
      for (int k = c; k < i; ++k) sum -= Li[k] * x[k];
      x[i] = sum / Li[i];
    }
    for (int i = c; i < nt; ++i) var[c] += x[i] * x[i];
  }
  for (int i = 0; i < nt; ++i) var[i] *= sc[i] * sc[i];
  for (int i = 0; i < q.S; ++i)
    for (int k = 0; k < 6; ++k) var_eop[6 * i + k] = q.sta_used[i] ? var[COL_STA(&q, i) + k] : 0.0;
  for (int j = 0; j < 3 * q.P; ++j) var_xyz[j] = var[6 * q.S + j];
  for (int c = 0; c < q.C; ++c) {
    for (int k = 0; k < 3; ++k) var_iop[3 * c + k] = 0.0;
    for (int k = 0; k < 16; ++k) var_ap[16 * c + k] = 0.0;
    for (int k = 0; k < q.cam_nc[c]; ++k) {
      const int par = q.cam_par[c * NCAMPAR + k];
      const double v = var[COL_CAM(&q, c) + k];
      if (par < 3) var_iop[3 * c + par] = v;
      else var_ap[16 * c + par - 3] = v;
    }
  }
  free(H); free(sc); free(var); free(x);
  prob_free(&q);
  return 0;
}

/* Rows of the same covariance matrix: the covariances of every free camera parameter with all parameters, which the
 * reference's correlation analysis reads block by block (src/main.cpp:5720-5778 EOP x IOP and EOP x AP, :6154-6189
 * XYZ x IOP and XYZ x AP, :6423-6436 IOP x AP, :5592-5598 AP x AP).  Row k belongs to the k-th camera column of the
 * program; cam_param[k] = 19 * camera + parameter (0..2 xp yp c, 3..18 AP[0..15]).  cov_eop [ncam][6S],
 * cov_xyz [ncam][3P], cov_cam [ncam][ncam]; any of the three may be NULL (room for 19 rows per camera is always
 * enough); *n_rows = ncam. */
int bao_covariance_rows(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz, const double *iop,
                        const double *ap, int32_t *n_rows, int32_t *cam_param, double *cov_eop, double *cov_xyz, double *cov_cam) {
  prob_t q;
  double *H = NULL, *sc = NULL;
  const int rc = cov_factor(&q, d, in, eop, xyz, iop, ap, &H, &sc);
  if (rc) return rc;
  const int nt = q.nt;
  if (n_rows) *n_rows = q.ncam;
  double *x = calloc(nt + 1, sizeof(double));
  for (int c = 0; c < q.C; ++c)
    for (int k = 0; k < q.cam_nc[c]; ++k) {
      const int col = COL_CAM(&q, c) + k, row = col - COL_CAM(&q, 0);
      if (cam_param) cam_param[row] = NCAMPAR * c + q.cam_par[c * NCAMPAR + k];
      /* L y = e_col, L^T z = y */
      for (int i = 0; i < nt; ++i) {
        double sum = (i == col) ? 1.0 : 0.0;
        const double *Li = H + (size_t)i * nt;
        if (i >= col) for (int m = col; m < i; ++m) sum -= Li[m] * x[m];
        x[i] = i < col ? 0.0 : sum / Li[i];
      }
      for (int i = nt - 1; i >= 0; --i) {
        double sum = x[i];
        for (int m = i + 1; m < nt; ++m) sum -= H[(size_t)m * nt + i] * x[m];
        x[i] = sum / H[(size_t)i * nt + i];
      }
      for (int i = 0; i < nt; ++i) x[i] *= sc[i] * sc[col];
      if (cov_eop)
        for (int i = 0; i < q.S; ++i)
          for (int m = 0; m < 6; ++m) cov_eop[(size_t)row * 6 * q.S + 6 * i + m] = q.sta_used[i] ? x[COL_STA(&q, i) + m] : 0.0;
      if (cov_xyz)
        for (int j = 0; j < 3 * q.P; ++j) cov_xyz[(size_t)row * 3 * q.P + j] = x[6 * q.S + j];
      if (cov_cam)
        for (int m = 0; m < q.ncam; ++m) cov_cam[(size_t)row * q.ncam + m] = x[COL_CAM(&q, 0) + m];
    }
  free(H); free(sc); free(x);
  prob_free(&q);
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Trust-region loop  [CERES] TrustRegionMinimizer::Minimize (trust_region_minimizer.cc)  */
/* ------------------------------------------------------------------------------------ */

typedef struct { double *eop, *xyz, *iop, *ap; } params_t;

static void params_alloc(params_t *x, const prob_t *q) {
  x->eop = calloc((size_t)6 * q->S + 1, sizeof(double));
  x->xyz = calloc((size_t)3 * q->P + 1, sizeof(double));
  x->iop = calloc((size_t)3 * q->C + 1, sizeof(double));
  x->ap = calloc((size_t)16 * q->C + 1, sizeof(double));
}
static void params_free(params_t *x) { free(x->eop); free(x->xyz); free(x->iop); free(x->ap); }
static void params_copy(params_t *dst, const params_t *src, const prob_t *q) {
  memcpy(dst->eop, src->eop, sizeof(double) * 6 * q->S);
  memcpy(dst->xyz, src->xyz, sizeof(double) * 3 * q->P);
  memcpy(dst->iop, src->iop, sizeof(double) * 3 * q->C);
  memcpy(dst->ap, src->ap, sizeof(double) * 16 * q->C);
}

/* is the block part of the reduced program? [CERES] Program::RemoveFixedBlocks */
static int iop_in_program(const prob_t *q, int c) { return (q->in->iop_free_mask[c] & 7u) != 0; }
static int ap_in_program(const prob_t *q, int c) { return q->in->ap_free_mask[c] != 0 && q->cam_used[c]; }

/* candidate = Plus(x, delta): identity plus on free coordinates, 0 on fixed (SubsetParameterization) */
static void plus(const prob_t *q, const params_t *x, const double *delta, params_t *out) {
  params_copy(out, x, q);
  for (int i = 0; i < q->S; ++i)
    if (q->sta_used[i]) for (int k = 0; k < 6; ++k) out->eop[6 * i + k] = x->eop[6 * i + k] + delta[6 * i + k];
  for (int j = 0; j < 3 * q->P; ++j) out->xyz[j] = x->xyz[j] + delta[6 * q->S + j];
  for (int c = 0; c < q->C; ++c)
    for (int k = 0; k < q->cam_nc[c]; ++k) {
      const int par = q->cam_par[c * NCAMPAR + k];
      const double dv = delta[6 * q->S + 3 * q->P + q->cam_off[c] + k];
      if (par < 3) out->iop[3 * c + par] = x->iop[3 * c + par] + dv;
      else out->ap[16 * c + par - 3] = x->ap[16 * c + par - 3] + dv;
    }
  /* [CERES] ParameterBlock::Plus: project the whole block onto its box */
  for (int c = 0; c < q->C; ++c)
    if (iop_in_program(q, c))
      for (int k = 0; k < 3; ++k) {
        if (q->iop_lo) out->iop[3 * c + k] = fmax(out->iop[3 * c + k], q->iop_lo[3 * c + k]);
        if (q->iop_hi) out->iop[3 * c + k] = fmin(out->iop[3 * c + k], q->iop_hi[3 * c + k]);
      }
}

/* max-norm of x - Plus(x, -g)  [CERES] TrustRegionMinimizer::EvaluateGradientAndJacobian */
static double projected_gradient_max(const prob_t *q, const params_t *x, double *neg, params_t *tmp) {
  for (int i = 0; i < q->nt; ++i) neg[i] = -q->grad[i];
  plus(q, x, neg, tmp);
  double m = 0.0;
  for (int i = 0; i < 6 * q->S; ++i) m = fmax(m, fabs(x->eop[i] - tmp->eop[i]));
  for (int i = 0; i < 3 * q->P; ++i) m = fmax(m, fabs(x->xyz[i] - tmp->xyz[i]));
  for (int c = 0; c < q->C; ++c) {
    if (iop_in_program(q, c)) for (int k = 0; k < 3; ++k) m = fmax(m, fabs(x->iop[3 * c + k] - tmp->iop[3 * c + k]));
    if (ap_in_program(q, c)) for (int k = 0; k < 16; ++k) m = fmax(m, fabs(x->ap[16 * c + k] - tmp->ap[16 * c + k]));
  }
  return m;
}

/* norms over the ambient coordinates of the blocks in the reduced program */
static double ambient_norm(const prob_t *q, const params_t *a, const params_t *b) {
  double s = 0.0;
#define ACC(A, B, idx) { const double v = (A)[idx] - ((B) ? (B)[idx] : 0.0); s += v * v; }
  for (int i = 0; i < q->S; ++i)
    if (q->sta_used[i]) for (int k = 0; k < 6; ++k) ACC(a->eop, b ? b->eop : NULL, 6 * i + k);
  for (int j = 0; j < 3 * q->P; ++j) ACC(a->xyz, b ? b->xyz : NULL, j);
  for (int c = 0; c < q->C; ++c) {
    if (iop_in_program(q, c)) for (int k = 0; k < 3; ++k) ACC(a->iop, b ? b->iop : NULL, 3 * c + k);
    if (ap_in_program(q, c)) for (int k = 0; k < 16; ++k) ACC(a->ap, b ? b->ap : NULL, 16 * c + k);
  }
#undef ACC
  return sqrt(s);
}


/* ------------------------------------------------------------------------------------ */
/* Projected line search of bounds-constrained problems                                   */
/* [CERES] line_search.cc ArmijoLineSearch::DoSearch, polynomial.cc                       */
/* ------------------------------------------------------------------------------------ */

typedef struct { double x, value, gradient; int value_ok, gradient_ok; } ls_sample;

/* coefficients highest power first, as in [CERES] polynomial.h */
static double poly_eval(const double *p, int n, double x) {
  double v = 0.0;
  for (int i = 0; i < n; ++i) v = v * x + p[i];
  return v;
}

/* [CERES] FindInterpolatingPolynomial: one row per valid value, one per valid gradient; solved with
 * complete pivoting (Eigen fullPivLu there). Returns the number of coefficients. */
static int poly_fit(const ls_sample *s, int ns, double *coef) {
  int nc = 0;
  for (int i = 0; i < ns; ++i) nc += (s[i].value_ok != 0) + (s[i].gradient_ok != 0);
  const int deg = nc - 1;
  double A[8][9];
  int row = 0;
  for (int i = 0; i < ns; ++i) {
    if (s[i].value_ok) {
      for (int j = 0; j <= deg; ++j) A[row][j] = pow(s[i].x, deg - j);
      A[row][nc] = s[i].value;
      ++row;
    }
    if (s[i].gradient_ok) {
      for (int j = 0; j < deg; ++j) A[row][j] = (deg - j) * pow(s[i].x, deg - j - 1);
      A[row][deg] = 0.0;
      A[row][nc] = s[i].gradient;
      ++row;
    }
  }
  int perm[8];
  for (int j = 0; j < nc; ++j) perm[j] = j;
  for (int k = 0; k < nc; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < nc; ++r)
      for (int c = k; c < nc; ++c)
        if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); pr = r; pc = c; }
    if (best <= 0.0) break; /* singular: remaining unknowns stay zero */
    for (int c = 0; c <= nc; ++c) { const double t = A[k][c]; A[k][c] = A[pr][c]; A[pr][c] = t; }
    for (int r = 0; r < nc; ++r) { const double t = A[r][k]; A[r][k] = A[r][pc]; A[r][pc] = t; }
    { const int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
    for (int r = k + 1; r < nc; ++r) {
      const double f = A[r][k] / A[k][k];
      for (int c = k; c <= nc; ++c) A[r][c] -= f * A[k][c];
    }
  }
  double y[8];
  for (int k = nc - 1; k >= 0; --k) {
    double v = A[k][nc];
    for (int c = k + 1; c < nc; ++c) v -= A[k][c] * y[c];
    y[k] = (A[k][k] != 0.0) ? v / A[k][k] : 0.0;
  }
  for (int k = 0; k < nc; ++k) coef[perm[k]] = y[k];
  return nc;
}

/* real roots of p (n coefficients) inside [lo, hi]: between two neighbouring roots of p' the polynomial is
 * monotone, so every sign change there brackets exactly one root, which bisection pins to the last bit.
 * ([CERES] FindPolynomialRoots takes the eigenvalues of the companion matrix; only the real roots inside the
 * interval can win in MinimizePolynomial.) */
static int poly_real_roots(const double *p, int n, double lo, double hi, double *roots) {
  while (n > 0 && p[0] == 0.0) { ++p; --n; }
  if (n <= 1) return 0;
  int nr = 0;
  if (n == 2) {
    const double r = -p[1] / p[0];
    if (r >= lo && r <= hi) roots[nr++] = r;
    return nr;
  }
  double dp[8], crit[8];
  for (int i = 0; i < n - 1; ++i) dp[i] = (n - 1 - i) * p[i];
  const int ncrit = poly_real_roots(dp, n - 1, lo, hi, crit);
  for (int i = 1; i < ncrit; ++i) /* insertion sort */
    for (int j = i; j > 0 && crit[j] < crit[j - 1]; --j) { const double t = crit[j]; crit[j] = crit[j - 1]; crit[j - 1] = t; }
  double a = lo, fa = poly_eval(p, n, lo);
  for (int i = 0; i <= ncrit; ++i) {
    const double b = (i < ncrit) ? crit[i] : hi, fb = poly_eval(p, n, b);
    if (fa == 0.0) {
      if (nr == 0 || roots[nr - 1] != a) roots[nr++] = a;
    } else if (fb != 0.0 && (fa < 0.0) != (fb < 0.0)) {
      double l = a, h = b, fl = fa;
      for (int it = 0; it < 200; ++it) {
        const double m = 0.5 * (l + h);
        if (m <= l || m >= h) break;
        const double fm = poly_eval(p, n, m);
        if (fm == 0.0) { l = h = m; break; }
        if ((fm < 0.0) == (fl < 0.0)) { l = m; fl = fm; } else h = m;
      }
      roots[nr++] = 0.5 * (l + h);
    }
    a = b; fa = fb;
  }
  if (fa == 0.0 && (nr == 0 || roots[nr - 1] != a)) roots[nr++] = a;
  return nr;
}

/* [CERES] LineSearch::InterpolatingPolynomialMinimizingStepSize (CUBIC) + MinimizeInterpolatingPolynomial */
double bao_interpolating_step(const double *lower, const double *previous, const double *current, double x_min,
                              double x_max) {
  /* each sample: x, value, gradient, value_ok, gradient_ok */
  ls_sample s[3];
  int ns = 0;
  if (!(current[3] != 0.0)) return fmin(fmax(current[0] * 0.5, x_min), x_max);
  const double *src[3] = {lower, current, previous};
  for (int i = 0; i < 3; ++i) {
    if (i == 2 && !(previous && previous[3] != 0.0)) break;
    s[ns].x = src[i][0]; s[ns].value = src[i][1]; s[ns].gradient = src[i][2];
    s[ns].value_ok = src[i][3] != 0.0; s[ns].gradient_ok = src[i][4] != 0.0;
    ++ns;
  }
  double coef[8], der[8], roots[8];
  const int n = poly_fit(s, ns, coef);
  /* MinimizePolynomial: middle, ends, then the stationary points inside */
  double best_x = (x_min + x_max) / 2.0, best = poly_eval(coef, n, best_x);
  const double v_min = poly_eval(coef, n, x_min), v_max = poly_eval(coef, n, x_max);
  if (v_min < best) { best = v_min; best_x = x_min; }
  if (v_max < best) { best = v_max; best_x = x_max; }
  /* ([CERES] also offers the real parts of complex root pairs of the derivative; such a point is not stationary, so its
   *  value cannot lie below the smallest of the ends and the real stationary points inside: it changes nothing) */
  if (n > 2) {
    for (int i = 0; i < n - 1; ++i) der[i] = (n - 1 - i) * coef[i];
    const int nroots = poly_real_roots(der, n - 1, x_min, x_max, roots);
    for (int i = 0; i < nroots; ++i) {
      const double v = poly_eval(coef, n, roots[i]);
      if (v < best) { best = v; best_x = roots[i]; }
    }
  }
  for (int i = 0; i < ns; ++i) {
    if (s[i].x < x_min || s[i].x > x_max) continue;
    const double v = poly_eval(coef, n, s[i].x);
    if (v < best) { best = v; best_x = s[i].x; }
  }
  return best_x;
}

int bao_solve(const ba_dims *d, const ba_inputs *in, const ba_options *opt, int mode, double *eop, double *xyz,
              double *iop, double *ap, ba_summary *summary, ba_iter_row *rows, int32_t rows_cap) {
  return bao_solve_bounded(d, in, opt, mode, NULL, NULL, eop, xyz, iop, ap, summary, rows, rows_cap);
}

int bao_solve_bounded(const ba_dims *d, const ba_inputs *in, const ba_options *opt, int mode, const double *iop_lower,
                      const double *iop_upper, double *eop, double *xyz, double *iop, double *ap, ba_summary *summary,
                      ba_iter_row *rows, int32_t rows_cap) {
  prob_t q;
  int rc = prob_init(&q, d, in, 1);
  if (rc) { prob_free(&q); return rc; }
  q.iop_lo = iop_lower;
  q.iop_hi = iop_upper;
  /* [CERES] Program::IsFeasible: a constant block must lie inside its box, a variable one needs a non-empty box;
   * Program::IsBoundsConstrained: some variable block has a finite bound */
  int constrained = 0;
  for (int c = 0; c < q.C; ++c)
    for (int k = 0; k < 3; ++k) {
      const double lo = iop_lower ? iop_lower[3 * c + k] : -DBL_MAX, hi = iop_upper ? iop_upper[3 * c + k] : DBL_MAX;
      if (iop_in_program(&q, c)) {
        if (lo >= hi) { prob_free(&q); FAIL(BA_ERR_INVALID, "camera %d: IOP[%d] has the empty interval [%g, %g]", c, k, lo, hi); }
        if (lo > -DBL_MAX || hi < DBL_MAX) constrained = 1;
      } else if (iop[3 * c + k] < lo || iop[3 * c + k] > hi) {
        prob_free(&q);
        FAIL(BA_ERR_INVALID, "camera %d: constant IOP[%d] = %g lies outside [%g, %g]", c, k, iop[3 * c + k], lo, hi);
      }
    }
  const double t_begin = now_s();
  const int nt = q.nt;
  params_t x, cand;
  params_alloc(&x, &q);
  params_alloc(&cand, &q);
  memcpy(x.eop, eop, sizeof(double) * 6 * q.S);
  memcpy(x.xyz, xyz, sizeof(double) * 3 * q.P);
  memcpy(x.iop, iop, sizeof(double) * 3 * q.C);
  memcpy(x.ap, ap, sizeof(double) * 16 * q.C);
  double *diag = calloc(nt + 1, sizeof(double)), *D2 = calloc(nt + 1, sizeof(double));
  double *step = calloc(nt + 1, sizeof(double)), *delta = calloc(nt + 1, sizeof(double));
  double *trial = calloc(nt + 1, sizeof(double));
  params_t tmp;
  params_alloc(&tmp, &q);
  int stale = 0; /* the line search left the linearisation of another point in q */
  ba_summary sm;
  memset(&sm, 0, sizeof sm);
  int n_rows = 0;
  double min_iter_cost;

  /* sizes of the reduced program */
  {
    int nsu = 0, npar = 0, neff = 0, nres = 2 * q.N + 3 * q.P, nall = 6 * q.S + 3 * q.P + 19 * q.C;
    for (int i = 0; i < q.S; ++i) nsu += q.sta_used[i];
    npar = 6 * nsu + 3 * q.P;
    neff = 6 * nsu + 3 * q.P + q.ncam;
    for (int c = 0; c < q.C; ++c) {
      if (iop_in_program(&q, c)) { npar += 3; nres += 3; }
      if (ap_in_program(&q, c)) npar += 16;
    }
    sm.num_parameters = nall;
    sm.num_residuals = 2 * q.N + 3 * q.P + 3 * q.C;
    sm.num_parameters_reduced = npar;
    sm.num_effective_parameters_reduced = neff;
    sm.num_residuals_reduced = nres;
  }

  /* ---- IterationZero ---- */
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0, num_invalid = 0;
  if (constrained) {
    /* [CERES] TrustRegionMinimizer::IterationZero: start from the projection of the initial point */
    memset(delta, 0, sizeof(double) * nt);
    plus(&q, &x, delta, &cand);
    params_copy(&x, &cand, &q);
  }
  double x_norm = ambient_norm(&q, &x, NULL);
  double t0 = now_s();
  linearize(&q, x.eop, x.xyz, x.iop, x.ap); /* scale == 1 here */
  if (opt->jacobi_scaling) {
    /* [CERES] jacobian_scaling_ = 1 / (1 + sqrt(SquaredColumnNorm)), computed once */
    column_sqnorm(&q, diag);
    for (int i = 0; i < nt; ++i) q.scale[i] = 1.0 / (1.0 + sqrt(diag[i]));
    linearize(&q, x.eop, x.xyz, x.iop, x.ap);
  }
  double x_cost = q.cost;
  const double fixed_cost = q.fixed_cost;
  double grad_max = projected_gradient_max(&q, &x, trial, &tmp);
  sm.initial_cost = x_cost + fixed_cost;
  sm.fixed_cost = fixed_cost;
  min_iter_cost = sm.initial_cost;
  ba_iter_row it;
  memset(&it, 0, sizeof it);
  it.iteration = 0;
  it.step_is_valid = 1;
  it.step_is_successful = 1;
  it.cost = x_cost + fixed_cost;
  it.gradient_max_norm = grad_max;
  it.trust_region_radius = radius;
  it.t_linearize_s = now_s() - t0;
  sm.termination = BA_NO_CONVERGENCE;

  for (;;) {
    /* ---- FinalizeIterationAndCheckIfMinimizerCanContinue ---- */
    if (it.step_is_successful) sm.num_successful_steps++;
    else sm.num_unsuccessful_steps++;
    it.trust_region_radius = radius;
    if (rows && n_rows < rows_cap) rows[n_rows] = it;
    n_rows++;
    if (it.cost < min_iter_cost) min_iter_cost = it.cost;
    if (it.iteration >= opt->max_num_iterations) { sm.termination = BA_NO_CONVERGENCE; sm.stop_reason = BA_STOP_MAX_ITERATIONS; break; }
    if (it.step_is_successful && it.gradient_max_norm <= opt->gradient_tolerance) { sm.termination = BA_CONVERGENCE; sm.stop_reason = BA_STOP_GRADIENT_TOL; break; }
    if (radius <= opt->min_trust_region_radius) { sm.termination = BA_CONVERGENCE; sm.stop_reason = BA_STOP_MIN_RADIUS; break; }

    const int iter_no = it.iteration + 1;
    const double prev_grad_max = it.gradient_max_norm;
    memset(&it, 0, sizeof it);
    it.iteration = iter_no;
    it.gradient_max_norm = prev_grad_max;

    /* ---- LevenbergMarquardtStrategy::ComputeStep ---- */
    t0 = now_s();
    if (stale) {
      linearize(&q, x.eop, x.xyz, x.iop, x.ap);
      stale = 0;
    }
    if (!reuse_diagonal) {
      column_sqnorm(&q, diag);
      for (int i = 0; i < nt; ++i) diag[i] = fmin(fmax(diag[i], opt->min_lm_diagonal), opt->max_lm_diagonal);
    }
    for (int i = 0; i < nt; ++i) {
      const double lm = sqrt(diag[i] / radius); /* lm_diagonal_ */
      D2[i] = lm * lm;
    }
    int lrc;
    if (mode == BAO_FULL_DENSE) lrc = solve_full_dense(&q, D2, step);
    else if (mode == BAO_SCHUR_DENSE) lrc = solve_schur_dense(&q, D2, step);
    else { lrc = solve_schur_pcg(&q, D2, opt->pcg_rtol, opt->pcg_max_iters, step); it.linear_iterations = g_last_pcg_iters; sm.total_linear_iterations += g_last_pcg_iters; }
    int step_finite = (lrc == 0);
    if (lrc == BA_ERR_INVALID) { rc = lrc; goto out; }
    for (int i = 0; step_finite && i < nt; ++i) if (!isfinite(step[i])) step_finite = 0;
    for (int i = 0; i < nt; ++i) step[i] = -step[i];
    reuse_diagonal = 1;
    it.t_solve_s = now_s() - t0;

    /* ---- ComputeTrustRegionStep ---- */
    double mcc = 0.0;
    it.step_is_valid = 0;
    if (step_finite) {
      mcc = model_cost_change(&q, step);
      it.step_is_valid = (mcc > 0.0);
    }
    if (!it.step_is_valid) {
      /* ---- HandleInvalidStep ---- */
      if (++num_invalid >= opt->max_num_consecutive_invalid_steps) {
        sm.termination = BA_FAILURE;
        sm.stop_reason = BA_STOP_INVALID_STEPS;
        break;
      }
      radius = radius / decrease_factor; /* StepIsInvalid == StepRejected */
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      it.cost = x_cost + fixed_cost;
      it.cost_change = 0.0;
      it.step_norm = 0.0;
      it.relative_decrease = 0.0;
      it.step_is_successful = 0;
      continue;
    }
    num_invalid = 0;
    for (int i = 0; i < nt; ++i) delta[i] = step[i] * q.scale[i];

    if (constrained) {
      /* ---- DoLineSearch: Armijo backtracking along Plus(x, a delta), cubic interpolation ---- */
      /* [CERES] defaults: sufficient decrease 1e-4, contraction within [1e-3, 0.6], 20 iterations, min step 1e-9 */
      double g0 = 0.0, dmax = 0.0;
      for (int i = 0; i < nt; ++i) { g0 += q.grad[i] * delta[i]; dmax = fmax(dmax, fabs(delta[i])); }
      double lower[5] = {0.0, x_cost, g0, 1.0, 1.0}, prev[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, cur[5] = {1.0, 0.0, 0.0, 0.0, 0.0};
      int ls_iters = 0, ls_ok = 0;
      for (;;) {
        for (int i = 0; i < nt; ++i) trial[i] = cur[0] * delta[i];
        plus(&q, &x, trial, &cand);
        linearize(&q, cand.eop, cand.xyz, cand.iop, cand.ap);
        stale = 1;
        double gd = 0.0;
        for (int i = 0; i < nt; ++i) gd += delta[i] * q.grad[i];
        cur[1] = q.cost; cur[2] = gd;
        cur[3] = cur[4] = (isfinite(q.cost) && isfinite(gd)) ? 1.0 : 0.0;
        if (cur[3] != 0.0 && !(cur[1] > x_cost + 1e-4 * g0 * cur[0])) { ls_ok = 1; break; }
        if (++ls_iters >= 20) break;
        const double a = bao_interpolating_step(lower, prev, cur, 1e-3 * cur[0], 0.6 * cur[0]);
        if (a * dmax < 1e-9) break;
        memcpy(prev, cur, sizeof prev);
        cur[0] = a;
      }
      if (ls_ok) for (int i = 0; i < nt; ++i) delta[i] *= cur[0];
    }

    /* ---- ComputeCandidatePointAndEvaluateCost ---- */
    t0 = now_s();
    plus(&q, &x, delta, &cand);
    double cand_cost = eval_cost(&q, cand.eop, cand.xyz, cand.iop, cand.ap, NULL, 0, NULL);
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX;
    it.t_eval_s = now_s() - t0;

    /* ---- ParameterToleranceReached ---- */
    it.step_norm = ambient_norm(&q, &x, &cand);
    if (it.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
      sm.termination = BA_CONVERGENCE;
      sm.stop_reason = BA_STOP_PARAMETER_TOL;
      break;
    }
    /* ---- FunctionToleranceReached ---- */
    it.cost_change = x_cost - cand_cost;
    if (fabs(it.cost_change) <= opt->function_tolerance * x_cost) {
      sm.termination = BA_CONVERGENCE;
      sm.stop_reason = BA_STOP_FUNCTION_TOL;
      break;
    }
    /* ---- IsStepSuccessful: TrustRegionStepEvaluator::StepQuality, monotonic ---- */
    it.relative_decrease = it.cost_change / mcc;
    if (it.relative_decrease > opt->min_relative_decrease) {
      /* ---- HandleSuccessfulStep ---- */
      params_copy(&x, &cand, &q);
      x_norm = ambient_norm(&q, &x, NULL);
      t0 = now_s();
      linearize(&q, x.eop, x.xyz, x.iop, x.ap);
      stale = 0;
      it.t_linearize_s = now_s() - t0;
      x_cost = q.cost;
      grad_max = projected_gradient_max(&q, &x, trial, &tmp);
      it.gradient_max_norm = grad_max;
      it.cost = x_cost + fixed_cost;
      it.step_is_successful = 1;
      /* StepAccepted */
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * it.relative_decrease - 1.0, 3.0));
      radius = fmin(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
    } else {
      /* ---- HandleUnsuccessfulStep ---- */
      it.step_is_successful = 0;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
      it.cost = cand_cost + fixed_cost;
    }
  }
  /* [CERES] solver.cc SetSummaryFinalCost: minimum over the logged iterations */
  sm.final_cost = min_iter_cost;
  sm.num_iterations = n_rows;
  sm.t_total_s = now_s() - t_begin;
  memcpy(eop, x.eop, sizeof(double) * 6 * q.S);
  memcpy(xyz, x.xyz, sizeof(double) * 3 * q.P);
  memcpy(iop, x.iop, sizeof(double) * 3 * q.C);
  memcpy(ap, x.ap, sizeof(double) * 16 * q.C);
  if (summary) *summary = sm;
out:
  free(diag); free(D2); free(step); free(delta); free(trial);
  params_free(&tmp);
  params_free(&x);
  params_free(&cand);
  prob_free(&q);
  return rc;
}
