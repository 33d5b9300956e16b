/*
 * ba_oracle.h -- CPU restatement of the reference's Levenberg-Marquardt path.
 *
 * TEST INFRASTRUCTURE ONLY.  This code is the checker the HIP path is compared against and
 * the CPU baseline that bench.py times beside it.  Nothing in the product (the package, the
 * C-ABI library, the host tools) links, loads or calls it.
 *
 * Pinning status:
 *   - cost model (bao_collinearity): pinned against the reference's own functor source
 *     (src/main.cpp:1315-1401, 2033-2120, 1057-1085) compiled in this container, see
 *     oracle/ref/ and tests/golden/;
 *   - trust-region loop, loss correction, Jacobi scaling, step acceptance (bao_solve):
 *     restated from the published algorithm of Ceres Solver 1.13 (the version named in the
 *     reference's src/.vscode/c_cpp_properties.json:38), which is not vendored and not
 *     installable here.  The reference holds no tests, fixtures or recorded outputs for
 *     this part: PARITY UNPINNED against Ceres itself.
 *
 * The struct types are the ones of include/ba_gpu.h so that the same Python/ctypes
 * descriptions serve both sides.
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H

#include "../include/ba_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

enum bao_mode {
  BAO_FULL_DENSE = 0,  /* normal equations over every unknown, dense Cholesky: the literal
                          counterpart of SPARSE_NORMAL_CHOLESKY (src/main.cpp:4706) */
  BAO_SCHUR_DENSE = 1, /* point elimination, dense Cholesky of the reduced system  */
  BAO_SCHUR_PCG = 2    /* point elimination, preconditioned CG on the reduced system */
};

/* src/main.cpp:1315-1401 (obs[4]=obs[5]=0) and :2033-2120.  obs = x,y,sx,sy,xcorr,ycorr.
 * J (may be NULL) is row-major 2x28: EOP 6 | XYZ 3 | IOP 3 | AP 16. */
void bao_collinearity(const double *eop, const double *xyz, const double *iop, const double *ap,
                      const double *obs, double *r, double *J);
/* the same with the projection of another of the reference's functors: model 0 collinearity, 1 collinearity
 * stereographic, 2..5 fisheye equidistant / equisolid-angle / orthographic / stereographic (src/main.cpp:2133-2571),
 * 6 collinearity orthographic (:1412-1490), 7 omnidirectional (:2700-2790) */
void bao_projection(int model, const double *eop, const double *xyz, const double *iop, const double *ap,
                      const double *obs, double *r, double *J);

/* src/main.cpp:1057-1085.  J is diag(1/sigma). */
void bao_constrain_point(const double *p, const double *p0, const double *sigma, double *r);

/* First-match lookup of src/main.cpp:3821-3831 (std::find + std::distance), O(n_keys*n_ids).
 * A key that is absent yields n_ids, as std::distance(begin, end) does. */
void bao_index_first_match(const int32_t *ids, int32_t n_ids, const int32_t *keys, int32_t n_keys,
                           int32_t *out);

int bao_residual_jacobian_all(const ba_dims *d, const ba_inputs *in, const double *eop,
                              const double *xyz, const double *iop, const double *ap, double *r,
                              double *J);

int bao_evaluate(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz,
                 const double *iop, const double *ap, int apply_loss, double *cost,
                 double *residuals);

int bao_sparsity(const ba_dims *d, const ba_inputs *in, int32_t *num_rows, int32_t *num_cols,
                 int64_t *nnz, int32_t *rows, int32_t *cols);

int bao_solve(const ba_dims *d, const ba_inputs *in, const ba_options *opt, int mode, double *eop,
              double *xyz, double *iop, double *ap, ba_summary *summary, ba_iter_row *rows,
              int32_t rows_cap);

/* bao_solve with the box on the IOP blocks that the reference's fisheye branches set (src/main.cpp:3957-3961,
 * 4252-4256: xp in [xMin, xMax], yp in [-yMax, yMin], c >= 0).  iop_lower / iop_upper are [3*n_cam] or NULL;
 * -DBL_MAX / DBL_MAX mean no bound.  With a finite bound on a variable block the loop takes the bounds-constrained
 * route of Ceres 1.13: the start point is projected, Plus projects, the gradient norm is that of the projected
 * gradient step, and every valid step goes through an Armijo line search along the projected path. */
int bao_solve_bounded(const ba_dims *d, const ba_inputs *in, const ba_options *opt, int mode, const double *iop_lower,
                      const double *iop_upper, double *eop, double *xyz, double *iop, double *ap, ba_summary *summary,
                      ba_iter_row *rows, int32_t rows_cap);

/* Step size of one backtracking move: minimiser, inside [x_min, x_max], of the polynomial through the samples
 * (each: x, value, gradient, value valid, gradient valid; previous may be NULL).  For tests. */
double bao_interpolating_step(const double *lower, const double *previous, const double *current, double x_min,
                              double x_max);

/* One residual+Jacobian+point-elimination pass (what the GPU's fused kernel does), for the
 * Jacobian obs/s CPU figure.  Returns the cost. */
double bao_linearize_once(const ba_dims *d, const ba_inputs *in, const double *eop,
                          const double *xyz, const double *iop, const double *ap);

/* Diagonal of the covariance matrix of all parameters (src/main.cpp:5261-5279, 5443-5565), in the
 * reference's layout: var_eop [6S], var_xyz [3P], var_iop [3C], var_ap [16C]; zero for parameters held
 * constant.  Dense: small networks only. */
int bao_covariance(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz,
                   const double *iop, const double *ap, double *var_eop, double *var_xyz, double *var_iop,
                   double *var_ap);

/* The covariances of every free camera parameter with all parameters (the rows of Cx that the reference's correlation
 * analysis reads, src/main.cpp:5720-5778, 6154-6189, 6423-6436, 5592-5598).  Row k = k-th camera column;
 * cam_param[k] = 19 * camera + parameter (0..2 xp yp c, 3..18 AP[0..15]); cov_eop [ncam][6S], cov_xyz [ncam][3P],
 * cov_cam [ncam][ncam], each may be NULL; *n_rows = ncam (at most 19 per camera).  Dense: small networks only. */
int bao_covariance_rows(const ba_dims *d, const ba_inputs *in, const double *eop, const double *xyz,
                        const double *iop, const double *ap, int32_t *n_rows, int32_t *cam_param, double *cov_eop,
                        double *cov_xyz, double *cov_cam);

const char *bao_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
