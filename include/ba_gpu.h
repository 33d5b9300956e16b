/*
 * ba_gpu.h -- C ABI of the MI355X bundle-adjustment solver library (libba_gpu.so).
 *
 * The reference program (jckchow/BundleAdjustment, src/main.cpp) has no plugin or FFI
 * boundary of its own: the only seam around its Levenberg-Marquardt inner loop is the set
 * of Ceres Solver calls it makes.  Each entry point below replaces one group of those
 * calls; the comment on each names the reference lines it stands in for.
 *
 * Conventions
 *   - plain C, no C++ or torch types in any signature; all pointers are caller-owned host
 *     memory unless a comment says "device";
 *   - every function returns 0 on success and a negative ba_status on error; the text of
 *     the last error on the calling thread is available from ba_last_error();
 *   - nothing throws across the boundary;
 *   - one ba_problem is used from one host thread at a time;
 *   - all floating point data is IEEE fp64, all indices are 32-bit signed.
 */
#ifndef BA_GPU_H
#define BA_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_problem ba_problem; /* opaque */

enum ba_status {
  BA_OK = 0,
  BA_ERR_INVALID = -1,     /* bad argument / inconsistent sizes / index out of range */
  BA_ERR_DEVICE = -2,      /* HIP runtime error */
  BA_ERR_NUMERIC = -3,     /* Cholesky breakdown / non-finite step */
  BA_ERR_UNSUPPORTED = -4, /* option value not implemented */
  BA_ERR_COMM = -5         /* collective failed */
};

/* Termination types; values follow ceres::TerminationType as used by Solver::Summary. */
enum ba_termination {
  BA_CONVERGENCE = 0,
  BA_NO_CONVERGENCE = 1,
  BA_FAILURE = 2
};

/* Which test ended the solve (Ceres only reports this in Summary::message). */
enum ba_stop_reason {
  BA_STOP_NONE = 0,
  BA_STOP_MAX_ITERATIONS = 1,
  BA_STOP_GRADIENT_TOL = 2,
  BA_STOP_PARAMETER_TOL = 3,
  BA_STOP_FUNCTION_TOL = 4,
  BA_STOP_MIN_RADIUS = 5,
  BA_STOP_INVALID_STEPS = 6,
  BA_STOP_LINEAR_SOLVER = 7
};

enum ba_linear_solver {
  BA_SOLVER_DENSE_SCHUR = 0, /* point elimination + dense fp64 Cholesky of the reduced system */
  BA_SOLVER_PCG_SCHUR = 1,   /* point elimination + PCG on the reduced system, held as 6x6 blocks in
                                block-sparse form, with the multilevel (smoothed aggregation) preconditioner
                                unless ba_options.preconditioner asks for block-Jacobi (matrix-free with
                                block-Jacobi when the system is too dense for that form, or when the cameras
                                have more than 79 free parameters between them) */
  BA_SOLVER_PCG_IMPLICIT = 2, /* same PCG, always matrix-free: S p = Jc^T (I - Jp V^-1 Jp^T) Jc p */
  BA_SOLVER_AUTO = 3,         /* the default: DENSE_SCHUR while the reduced system has at most
                                 BA_AUTO_DENSE_ORDER unknowns (6 per station + the free camera
                                 parameters), SPARSE_SCHUR up to BA_AUTO_SPARSE_ORDER when the block-sparse
                                 form exists, PCG_SCHUR above and otherwise.  The reference fixes its solver
                                 at src/main.cpp:4706, so a caller coming from it never chooses one. */
  BA_SOLVER_SPARSE_SCHUR = 4  /* point elimination + sparse direct Cholesky of the block-sparse reduced system:
                                 nested dissection of the station graph, multifrontal factorisation with dense
                                 fronts on the matrix cores -- the counterpart of the reference's
                                 SPARSE_NORMAL_CHOLESKY (src/main.cpp:4706) on the reduced system.  Needs the
                                 block-sparse form (BA_ERR_UNSUPPORTED otherwise). */
};
#define BA_AUTO_DENSE_ORDER 4096
#define BA_AUTO_SPARSE_ORDER 250000

typedef struct {
  int32_t n_obs; /* rows of the .pho file  */
  int32_t n_sta; /* rows of the .eop file  */
  int32_t n_pts; /* rows of the .xyz file  */
  int32_t n_cam; /* rows of the .iop file  */
} ba_dims;

/*
 * Problem definition.  Replaces the problem-assembly calls of the reference:
 *   AddParameterBlock           src/main.cpp:3793-3800  (column order EOP | XYZ | IOP | AP)
 *   AddResidualBlock (image)    src/main.cpp:3845-3848 / 4242-4245  (one 2-row block per .pho row)
 *   AddResidualBlock (XYZ prior) src/main.cpp:4596-4599
 *   AddResidualBlock (IOP prior) src/main.cpp:4626-4629
 *   HuberLoss(1.0)              src/main.cpp:3809-3810
 *   SubsetParameterization      src/main.cpp:4479-4511
 * The index maps obs_pt / obs_sta / sta_cam are the indexPoint / indexPose / indexSensor
 * values of src/main.cpp:3821-3831, already resolved by the host (ba_host.h).
 */
typedef struct {
  const int32_t *obs_pt;  /* [n_obs] row of .xyz for each observation */
  const int32_t *obs_sta; /* [n_obs] row of .eop for each observation */
  const int32_t *sta_cam; /* [n_sta] row of .iop for each station     */
  const double *obs_x, *obs_y;   /* [n_obs] .pho columns 3,4  image coordinates          */
  const double *obs_sx, *obs_sy; /* [n_obs] .pho columns 5,6  standard deviations         */
  const double *obs_cx, *obs_cy; /* [n_obs] .pho columns 7,8  constant correction (xMLP,yMLP
                                    of collinearityMachineLearnedSimple, src/main.cpp:2096);
                                    zeros give the plain collinearity functor :1315-1401)  */
  const double *pt_prior, *pt_sigma;   /* [3*n_pts] .xyz X,Y,Z and their sigmas (prior, :4598) */
  const double *iop_prior, *iop_sigma; /* [3*n_cam] xp,yp,c prior mean and sigma (:4612-4628)  */
  const uint16_t *ap_free_mask; /* [n_cam] bit k set -> AP[k] is estimated; 0 -> block constant */
  const uint8_t *iop_free_mask; /* [n_cam] bits 0..2 -> xp,yp,c estimated; 0 -> block constant  */
  double huber_a;               /* 1.0 in the reference; <= 0 disables the robust loss          */
  int32_t model;                /* projection functor of the image rows (src/main.cpp:3845-3918, :4186-4250):
                                 * 0 collinearity :1315/:2033, 1 collinearity stereographic :1500/:2133,
                                 * 2 fisheye equidistant :1596/:2226, 3 equisolid angle :1680/:2312,
                                 * 4 fisheye orthographic :1765/:2398, 5 fisheye stereographic :1850/:2484,
                                 * 6 collinearity orthographic :1412, 7 omnidirectional :2700 */
  const uint8_t *cam_in_use;    /* NULL, or [n_cam] for a problem that holds one rank's share of a network
                                 * (ba_set_allreduce): non-zero for every camera that some observation of the WHOLE
                                 * network refers to.  The AP block of a camera is part of the program when an image
                                 * row refers to it (src/main.cpp:4242-4245); a rank whose share holds no such row
                                 * still carries the block's columns (with zero sums of its own), so that the reduced
                                 * systems of all ranks have one layout.  ba_net_shard fills it. */
} ba_inputs;

/*
 * Solver options.  Replaces ceres::Solver::Options as set at src/main.cpp:4705-4719; every
 * field not set there carries the Ceres default (ba_default_options fills them in).
 */
typedef struct {
  double min_lm_diagonal;          /* 1e-150 (src/main.cpp:4713) */
  double max_lm_diagonal;          /* 1e-150 (src/main.cpp:4712) */
  double initial_trust_region_radius; /* 1e4  */
  double max_trust_region_radius;  /* 1e16 */
  double min_trust_region_radius;  /* 1e-32 */
  double min_relative_decrease;    /* 1e-3 */
  double function_tolerance;       /* 1e-6 */
  double gradient_tolerance;       /* 1e-10 */
  double parameter_tolerance;      /* 1e-8 */
  int32_t max_num_iterations;      /* 100 (src/main.cpp:4718) */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;          /* 1 */
  int32_t linear_solver;           /* enum ba_linear_solver */
  double pcg_rtol;                 /* PCG stops when |r|_M^-1 <= pcg_rtol * |b|_M^-1 */
  int32_t pcg_max_iters;
  int32_t preconditioner;          /* of BA_SOLVER_PCG_SCHUR: 0 = multilevel (smoothed aggregation over the station
                                      graph, additive, block-Jacobi on every level); 1 = block-Jacobi only           */
} ba_options;

/* One row of the iteration log; same columns as Ceres' minimizer_progress_to_stdout table. */
typedef struct {
  int32_t iteration;
  int32_t step_is_successful;
  int32_t step_is_valid;
  int32_t linear_iterations; /* PCG iterations (0 for the dense solver); equal to ba_options.pcg_max_iters when the
                                PCG stopped there without reaching pcg_rtol (the step is then used as it is) */
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease; /* tr_ratio */
  double trust_region_radius;
  double t_linearize_s; /* wall time: residual/Jacobian + point elimination */
  double t_solve_s;     /* wall time: reduced solve + back substitution     */
  double t_eval_s;      /* wall time: candidate cost evaluation             */
} ba_iter_row;

/* Replaces the fields of ceres::Solver::Summary read at src/main.cpp:4722-4723, 4906-4907. */
typedef struct {
  double initial_cost;
  double final_cost;
  double fixed_cost;
  int32_t num_iterations;  /* rows logged, including iteration 0 */
  int32_t termination;     /* enum ba_termination */
  int32_t stop_reason;     /* enum ba_stop_reason */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_residuals_reduced;
  int32_t num_parameters_reduced;           /* ambient size of the non-constant, referenced blocks */
  int32_t num_effective_parameters_reduced; /* tangent size of the same                             */
  int32_t num_parameters;                   /* ambient size of every block                          */
  int32_t num_residuals;
  int32_t total_linear_iterations;
  int32_t linear_solver_used; /* what BA_SOLVER_AUTO resolved to (else the option as given) */
  double t_total_s;
} ba_summary;

/* Fills *opt with the reference's option values (src/main.cpp:4705-4719 over Ceres defaults). */
void ba_default_options(ba_options *opt);

/*
 * Builds a problem on the current HIP device: validates the index maps, sorts observations
 * by point and by station, and copies the constant data to HBM.
 * Replaces ceres::Problem construction, src/main.cpp:3784-4636.
 */
int ba_create(const ba_dims *dims, const ba_inputs *in, ba_problem **out);

/* Releases everything ba_create allocated.  (ceres::Problem destructor.) */
void ba_destroy(ba_problem *p);

/*
 * Parameter blocks.  In the reference the parameter memory is owned by the caller and
 * updated in place by ceres::Solve (src/main.cpp:3153, 3544, 3657, 4721); here the caller
 * uploads the four arrays, solves on the device, and downloads them.
 *   eop [6*n_sta] = omega,phi,kappa (radians), Xo,Yo,Zo per station (src/main.cpp:3177-3182)
 *   xyz [3*n_pts]; iop [3*n_cam] = xp,yp,c; ap [16*n_cam] (src/main.cpp:3581-3604)
 */
int ba_set_parameters(ba_problem *p, const double *eop, const double *xyz, const double *iop,
                      const double *ap);
int ba_get_parameters(ba_problem *p, double *eop, double *xyz, double *iop, double *ap);

/*
 * New constant image corrections, [n_obs] each in file order, in place of ba_inputs.obs_cx / obs_cy; everything
 * else of the problem stays as it is on the device.  In the reference the outer loop writes the regressor's
 * corrections into columns 7-8 of temp.pho, reads the file again and rebuilds the whole ceres::Problem for the
 * next adjustment (src/main.cpp:3040-3056, :3130, :4186-4187); this call is that hand-over without the rebuild.
 */
int ba_set_corrections(ba_problem *p, const double *obs_cx, const double *obs_cy);

/*
 * Box on the interior orientation blocks.  Replaces the Problem::SetParameterLowerBound / SetParameterUpperBound
 * calls that the reference's stereographic and fisheye branches make (src/main.cpp:3957-3961, 4252-4256:
 * xp in [xMin, xMax], yp in [-yMax, yMin], c >= 0; its collinearity branches set none).
 * lower / upper are [3*n_cam] in the order xp, yp, c; NULL, -DBL_MAX / DBL_MAX and -inf / +inf mean no bound.
 * With a finite bound on a camera whose IOP block is variable, ba_solve_resident follows the bounds-constrained
 * route of Ceres 1.13: the start point is projected onto the box, every Plus projects, gradient_max_norm is the
 * norm of the projected gradient step, and every valid trust-region step is shortened by an Armijo line search
 * along the projected path (cubic interpolation, sufficient decrease 1e-4, contraction within [1e-3, 0.6], at most
 * 20 trials, minimum step 1e-9: the Solver::Options defaults, which the reference leaves alone) before it is tested.
 * Errors as Program::IsFeasible: lower >= upper on a variable block at this call; a constant block outside its box
 * when the solve starts.
 */
int ba_set_iop_bounds(ba_problem *p, const double *lower, const double *upper);

/*
 * One contraction of that line search, exposed so it can be checked without a device: the minimiser over
 * [min_step, max_step] of the polynomial that interpolates the values and gradients of the samples (start = step 0;
 * previous may be NULL or invalid), i.e. LineSearch::InterpolatingPolynomialMinimizingStepSize of Ceres 1.13 with
 * CUBIC interpolation; half the current step, clamped, when the current sample is not valid.
 */
typedef struct {
  double x, value, gradient;
  int32_t value_is_valid, gradient_is_valid;
} ba_ls_sample;
double ba_line_search_contract(const ba_ls_sample *start, const ba_ls_sample *previous, const ba_ls_sample *current,
                               double min_step, double max_step);

/*
 * Runs the trust-region Levenberg-Marquardt loop on the parameters resident on the device
 * and leaves the result there.  Replaces ceres::Solve, src/main.cpp:4721.
 * rows may be NULL; at most rows_cap rows are written.
 */
int ba_solve_resident(ba_problem *p, const ba_options *opt, ba_summary *summary,
                      ba_iter_row *rows, int32_t rows_cap);

/* ba_set_parameters + ba_solve_resident + ba_get_parameters: the in-place behaviour of
 * ceres::Solve on caller-owned arrays (src/main.cpp:4721). */
int ba_solve(ba_problem *p, const ba_options *opt, double *eop, double *xyz, double *iop,
             double *ap, ba_summary *summary, ba_iter_row *rows, int32_t rows_cap);

/*
 * Cost and residual vector at the given parameters.  Replaces
 * problem.Evaluate(EvaluateOptions(), &cost, &residuals, NULL, NULL), src/main.cpp:4853, 6664.
 * residuals (may be NULL) has 2*n_obs + 3*n_pts + 3*n_cam entries in the row order of
 * src/main.cpp:4242-4263, 4596-4603, 4626-4634; with apply_loss != 0 the image rows carry
 * the robust correction sqrt(rho') as Ceres' default EvaluateOptions do.
 */
int ba_evaluate(ba_problem *p, const double *eop, const double *xyz, const double *iop,
                const double *ap, int apply_loss, double *cost, double *residuals);

/*
 * Variances of all parameters at the values the problem currently holds (after ba_set_parameters or
 * ba_solve_resident).  Replaces the COMPUTECX block, src/main.cpp:5261-5279 (ceres::Covariance with
 * apply_loss_function = true over all parameter blocks) and the extraction of the diagonals at
 * :5443-5565 into xyzVariance, eopVariance, iopVariance and apVariance, whose layout the outputs keep:
 * var_eop [6*n_sta], var_xyz [3*n_pts], var_iop [3*n_cam], var_ap [16*n_cam] (any may be NULL);
 * parameters held constant and stations without observations get 0.  The reference multiplies by
 * aposterioriVariance, which it has reset to 1 at :4955, so the values are not scaled here either.
 * Two routes, both on the reduced camera system with the points eliminated.  While the reduced matrix
 * (order 6*n_sta + free camera parameters) and a work matrix of the same size fit the device memory that
 * is free, its dense inverse is formed on the matrix cores.  Beyond that, the station graph is cut into
 * clusters by level-set separators and only the blocks of the inverse that the variances need are formed
 * (cluster inverses, the Schur complement on the separator stations and the camera columns, and the
 * update of the cluster blocks from its inverse); this needs the block-sparse reduced system and a
 * station graph that has separators, BA_ERR_UNSUPPORTED otherwise.  A rank-deficient network is an
 * error (the reference's DENSE_SVD pseudo-inverse would truncate it instead).
 */
int ba_covariance(ba_problem *p, double *var_eop, double *var_xyz, double *var_iop, double *var_ap);

/*
 * Process-wide switches that exist for comparisons and tests, not for production use; they apply to problems created
 * (or, for the PCG ones, solved) afterwards.  Names: "setup" (0 host / 1 device / -1 by size: where ba_create builds
 * its lists), "pcg_graph" (0: PCG iterations queued launch by launch instead of replayed as graphs), "amg_levels"
 * (1..3), "amg_theta" (aggregation threshold, default 0.5), "fused_sums" (0: station sums by separate sweeps),
 * "schur_windows" (0: off-diagonal blocks pair list by pair list), "keyed_tiles" (tiles per segment of the station-sum
 * plan), "rows_solver" (0 dense / 1 sparse direct / 2 PCG / -1 by size: the reduced solver of
 * ba_covariance_camera_rows).  Every setting gives the same results to rounding.  BA_ERR_INVALID for an unknown name.  No counterpart in
 * the reference.
 */
int ba_tuning(const char *name, double value);

/*
 * What BA_SOLVER_SPARSE_SCHUR has built for this problem (after its first solve): the number of fronts of the
 * elimination tree and of its levels, the order of the largest front, the number of 128-column panels on the
 * critical path (the sum over the levels of the most panels any front of the level has), the bytes the fronts
 * take and the floating-point operations of one factorisation.  Any pointer may be NULL.  No counterpart in the
 * reference (Ceres keeps such figures in Solver::Summary::FullReport, which the reference prints at :4726).
 */
int ba_sparse_solver_stats(const ba_problem *p, int32_t *fronts, int32_t *levels, int32_t *largest_front,
                           int32_t *critical_panels, double *front_bytes, double *factor_flops);

/*
 * Which route ba_covariance takes: 0 chooses (the default: the dissection when the dense inverse does not
 * fit the free memory and from order 16384 on, where it is the faster one; the dense inverse otherwise and
 * for station graphs without separators), 1 the dense inverse only, 2 the dissection at any size.  max_cluster_stations bounds the stations per cluster of route 2
 * (0: the default, 1536).  No counterpart in the reference, whose ceres::Covariance call has one route.
 */
int ba_set_covariance_route(ba_problem *p, int32_t route, int32_t max_cluster_stations);

/*
 * What the last ba_covariance call did: the route (1 or 2) and, for route 2, the number of clusters, the
 * order of the boundary system and the order of the largest cluster (any pointer may be NULL).
 */
int ba_covariance_stats(const ba_problem *p, int32_t *route_used, int32_t *clusters, int32_t *boundary_order,
                        int32_t *largest_cluster_order);

/*
 * The camera part of the same: variances of xp, yp, c [3*n_cam] and of the additional parameters [16*n_cam] only
 * (src/main.cpp:5459-5565, the figures a self-calibration is run for), for networks of any size.  One solve of the
 * reduced system per free camera parameter (right-hand side e_k, PCG to 1e-12) instead of the dense inverse.
 */
int ba_covariance_cameras(ba_problem *p, double *var_iop, double *var_ap);

/*
 * The rows of Cx that belong to the free camera parameters: their covariances with every station parameter, every
 * point coordinate and each other.  These are the blocks the reference's correlation analysis asks ceres::Covariance
 * for and reads with GetCovarianceBlock: EOP x IOP and EOP x AP (src/main.cpp:5720-5778), XYZ x IOP and XYZ x AP
 * (:6154-6189), IOP x AP (:6423-6436), AP x AP (:5592-5598); ba_correlations of ba_host.h turns them, with the
 * variances of ba_covariance, into the reference's correlation tables.  Row k belongs to the k-th free camera
 * parameter in the order of the program (camera by camera: xp, yp, c, AP[0..15], free ones only);
 * cam_param[k] = 19 * camera + parameter (0..2 xp yp c, 3..18 AP[0..15]); *n_rows = their number (at most 19 per
 * camera, for sizing call with every array NULL).  cov_eop [n_rows][6*n_sta], cov_xyz [n_rows][3*n_pts] (file order),
 * cov_cam [n_rows][n_rows]; any may be NULL.  One factorisation of the reduced system (dense up to
 * BA_AUTO_DENSE_ORDER unknowns, the sparse direct solver up to BA_AUTO_SPARSE_ORDER) and one pair of triangular
 * solves per row, beyond that one PCG solve per row; the point part is the back substitution of that solution.
 * One rank only.
 */
int ba_covariance_camera_rows(ba_problem *p, int32_t *n_rows, int32_t *cam_param, double *cov_eop, double *cov_xyz,
                              double *cov_cam);

/*
 * Structure of the Jacobian that problem.Evaluate(..., &jacobian) returns at
 * src/main.cpp:6664 as a ceres::CRSMatrix: rows[] has num_rows+1 offsets, cols[] the column
 * of every structural non-zero.  Call with rows == cols == NULL to get the sizes.
 */
int ba_sparsity(ba_problem *p, int32_t *num_rows, int32_t *num_cols, int64_t *nnz,
                int32_t *rows, int32_t *cols);

/*
 * The residual/Jacobian kernel on its own: evaluates every image observation at the
 * parameters resident on the device.  r [2*n_obs] and the dense per-observation Jacobian
 * J [n_obs*2*28] (row-major 2x28: EOP 6 | XYZ 3 | IOP 3 | AP 16, fixed columns included,
 * no robust correction, no column scaling) are written in .pho row order.
 * This is what ceres::AutoDiffCostFunction<collinearity,2,6,3,3,16>::Evaluate produces per
 * block (src/main.cpp:3845-3847).  Either output may be NULL.
 */
int ba_residual_jacobian(ba_problem *p, double *r, double *J);

/*
 * Times `repeats` launches of one kernel at the resident parameters with HIP events on the
 * solver's stream and returns the mean seconds per launch.  Used for the Jacobian obs/s figure
 * and the roofline line of the benchmark.  which:
 *   0  residual + Jacobian + point elimination: the two kernels 7 and 8 back to back
 *   1  point half of the reduced operator  S p   (matrix-free PCG inner loop)
 *   2  station half of the reduced operator      (PCG inner loop)
 *   3  residual-only cost evaluation at a candidate point
 *   4  assembly of the off-diagonal 6x6 blocks of the reduced matrix
 *   5  one PCG iteration on the block-sparse reduced system (SpMV, vector kernel, preconditioner kernels)
 *   6  the block-sparse matrix-vector product of that iteration alone
 *   7  per-point kernel of 0: first evaluation, V, its gradient, the camera coupling, V^-1
 *   8  per-observation kernel of 0: second evaluation, all per-observation rows written once
 *   9  one PCG iteration with time stamps at the phase marks of its kernels, printed to stderr (returns 0)
 *  10  Cholesky factorisation of the dense reduced system (n^3 / 3 multiply-adds, n = 6 n_sta + free camera parameters)
 *  11  the same with the two triangular solves
 *  12  the per-observation kernel that also forms the station sums, with clock counters around its phases (table on stderr)
 *  13  BA_SOLVER_SPARSE_SCHUR: assembly of the fronts and the sparse factorisation (after a solve with that solver)
 *  14  the same with the forward and backward solves on the tree
 * No counterpart in the reference (its timers are wall-clock prints, src/main.cpp:3027, 4703).
 */
int ba_time_kernel(ba_problem *p, int32_t which, int32_t repeats, double *seconds_per_launch);

/* The GPU of the calling thread (one process per GPU picks its own before ba_create), for hosts that do not link the HIP
 * runtime themselves.  No counterpart in the reference. */
int ba_device_count(int32_t *count);
int ba_set_device(int32_t device);

/*
 * Multi-GPU: a problem built from one shard of the points (and all observations of those
 * points) takes part in a sum over ranks of the reduced camera system.  Stations and
 * cameras are replicated, so eop/iop/ap passed to every rank are the full arrays.
 *   - point priors belong to the shard that owns the point;
 *   - IOP priors are added by every rank after the sums over ranks, so they count once.
 * The callback must reduce buf[0..count) over all ranks in place (device memory) with
 * op 0 = sum or op 1 = max and return 0; it is called on the solver's stream, which it
 * receives as an opaque hipStream_t.
 * The reference has no counterpart: it is single-process (src/main.cpp:4705-4721 never sets
 * num_threads).
 */
typedef int (*ba_allreduce_fn)(void *user, double *device_buf, int64_t count, int32_t op,
                               void *hip_stream);
int ba_set_allreduce(ba_problem *p, ba_allreduce_fn fn, void *user);

/* What the reductions cost this rank.  ba_reduction_stats returns, for the calls of the callback since the last
 * ba_reduction_stats: their number, the bytes they carried and -- while ba_reduction_timing is on -- the time they took
 * on the solver's stream between events recorded before and after each (a collective also waits for the slowest rank,
 * so load imbalance shows here); it waits for the stream and resets the three.  A solve makes four calls per LM
 * iteration with the dense and the block-sparse PCG solvers: station sums + camera rows + cost, gradient maximum +
 * not-positive-definite flag, reduced matrix + camera border, step sums + candidate cost.  The matrix-free PCG adds one
 * per inner iteration, a bounded solve those of its line search. */
int ba_reduction_timing(ba_problem *p, int32_t on);
int ba_reduction_stats(ba_problem *p, int64_t *count, double *bytes, double *seconds);

/* Which rank of how many this problem is, and whether the block-sparse PCG forms its product q = S p by rows: every rank
 * then takes a slice of the station rows holding an equal share of the 6x6 blocks, and one more reduction per PCG
 * iteration -- [q | partial sums], 6 n_sta + a few thousand doubles -- completes the vector on all ranks (everything else
 * of the iteration stays replicated: the vectors are bitwise equal on all ranks either way).  mode -1: never; 0: when the
 * time of the product that the other ranks take over exceeds the time of that reduction (a ring over xGMI links of
 * about 150 GB/s plus 30 us: true from about 60 000 stations on, on 8 ranks); 1: always.  Without this call every rank forms
 * the whole product; ba_use_rccl makes it with mode 0.  ba_reduced_solve_rows returns the rows that the last
 * block-sparse PCG solve gave this rank (0, 0: the whole product on every rank).
 * No counterpart in the reference (single-process, src/main.cpp:4705-4721). */
int ba_shard_reduced_solve(ba_problem *p, int32_t rank, int32_t n_ranks, int32_t mode);
int ba_reduced_solve_rows(const ba_problem *p, int32_t *row_begin, int32_t *row_end);

/* Helpers for callbacks that stage through host memory (e.g. a gloo or MPI reduction in tests):
 * synchronous copies ordered after the work already queued on hip_stream. */
int ba_copy_to_host(void *host_dst, const double *device_src, int64_t count, void *hip_stream);
int ba_copy_to_device(double *device_dst, const void *host_src, int64_t count, void *hip_stream);

/* RCCL implementation of the callback above.  unique_id is the 128-byte ncclUniqueId made
 * by ba_rccl_unique_id on rank 0 and distributed by the host. */
int ba_rccl_unique_id(void *unique_id_128);
int ba_use_rccl(ba_problem *p, int32_t n_ranks, int32_t rank, const void *unique_id_128);

/*
 * ba_evaluate for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  No counterpart in the reference (its problem.Evaluate, src/main.cpp:4853, allocates freely).
 * Same cost and residual vector as ba_evaluate, from arrays that are ALL in device memory: the arrays of *in
 * (ap_free_mask / iop_free_mask are not read), the parameters eop [6*n_sta], xyz [3*n_pts], iop [3*n_cam],
 * ap [16*n_cam], and the outputs cost [2] (cost[0] = the cost, cost[1] = number of rows skipped because an index or a
 * sigma was out of range: ba_create would have refused such a problem) and residuals [2*n_obs + 3*n_pts + 3*n_cam]
 * (may be NULL) in file order.  The kernels are queued on hip_stream (a hipStream_t, NULL = the default stream);
 * the call does not synchronise, copies nothing to the host and calls no hipMalloc / hipFree or any other allocating
 * routine.  Its work memory (station records, camera records, per-workgroup partial sums) is one fixed array in the
 * library's device image, about 12.6 MB per GPU, held until the process ends: at most BA_NOALLOC_MAX_STATIONS
 * stations and BA_NOALLOC_MAX_CAMERAS cameras (BA_ERR_UNSUPPORTED beyond; points and observations are not limited).
 * All calls on one GPU share that array: calls on one stream are ordered and safe, calls that overlap in time on one
 * GPU (two streams, two host threads) race.
 * The first call on a device looks the array's address up (hipGetSymbolAddress, which may load the library's code
 * object): make one call on every device outside any graph capture before the first captured one.
 */
#define BA_NOALLOC_MAX_STATIONS 65536
#define BA_NOALLOC_MAX_CAMERAS 64
int ba_evaluate_noalloc(const ba_dims *dims, const ba_inputs *in, const double *eop, const double *xyz,
                        const double *iop, const double *ap, int apply_loss, double *cost, double *residuals,
                        void *hip_stream);

const char *ba_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* BA_GPU_H */
