/*
 * ba_host.h -- host side of the bundle adjustment: the reference's drop-in surface around the
 * solver (libba_host.so, C++ inside, C ABI outside so tests and tools can bind it).
 *
 * What each group restates, in /root/reference/src/main.cpp:
 *   readers      .pho :3046-3095   .eop :3135-3199   .iop :3526-3640   .xyz :3651-3692
 *   index maps   :3821-3831 (first-match std::find)
 *   priors       :4596-4599 (XYZ from the .xyz columns), :4612-4628 (IOP, sigma 10,10,1000)
 *   fixed/free   :4479-4511 (default: every AP fixed except k1)
 *   sigma0       :4905-4912
 *   writers      temp.pho :3127-3132, EOP.jck :7093-7098, temp.eop :4801-4806, iop.jck :7104-7109,
 *                temp.iop :4840-4845, ap.jck :7115-7120, XYZ.jck :7082-7087, image.jck :6993-6998
 * plus a deterministic synthetic-network generator (no counterpart in the reference, which
 * ships no data).
 */
#ifndef BA_HOST_H
#define BA_HOST_H

#include <stdint.h>

#include "ba_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_network ba_network; /* opaque: the four files, the index maps, the parameters */

/* Reads the four text files.  Record-dropping quirk of the reference's
 * `while(true){ in >> ...; push_back; if(eof) break; } pop_back()` idiom is kept: a record
 * whose extraction reaches end-of-file (a last line with no trailing newline) is dropped.
 * Differences, on purpose: an unreadable file or a malformed token is an error here (the
 * reference loops forever), and an observation whose target/station/camera id is absent is
 * an error (the reference indexes out of bounds). */
int ba_net_read(const char *pho, const char *eop, const char *iop, const char *xyz, ba_network **out);

typedef struct {
  int32_t n_cam, n_sta, n_pts;
  int32_t rays_per_point;      /* stations observing each point; n_obs = n_pts * rays_per_point */
  uint64_t seed;
  uint16_t ap_free_mask;       /* bit k -> AP[k] free, same for every camera */
  uint8_t iop_free_mask;       /* bits 0..2 -> xp,yp,c free                   */
  uint8_t empirical_terms;     /* 1: truth has non-zero AP[7] and AP[10]      */
  double noise_sigma;          /* image noise = sigma_x = sigma_y, pixels     */
  double outlier_fraction;     /* share of observations with a gross error    */
  double control_sigma;        /* prior sigma of control points, mm           */
  double free_sigma;           /* prior sigma of every other point, mm        */
  int32_t control_every;       /* every n-th point is a control point         */
  int32_t perfect_start;       /* 1: initial values = truth (for debugging)   */
  int32_t model;               /* projection the image points are made with, 0..7 (ba_inputs.model) */
} ba_synth_config;

void ba_synth_default(ba_synth_config *cfg);
int ba_net_synth(const ba_synth_config *cfg, ba_network **out);

void ba_net_free(ba_network *net);
int ba_net_dims(const ba_network *net, ba_dims *dims);
/* Fills *in with pointers into the network's own storage (valid until ba_net_free). */
int ba_net_inputs(ba_network *net, ba_inputs *in);
/* Partition of a network over the ranks of a multi-GPU solve (one process per GPU).  Points are independent given the
 * station and camera unknowns, so rank r gets the contiguous point range [bounds[r], bounds[r+1]) -- balanced by
 * observations + 1 per point -- with every observation of those points; stations and cameras are replicated and the
 * ranks only sum the reduced camera system (ba_set_allreduce / ba_use_rccl in ba_gpu.h).  No counterpart in the
 * reference, which runs ceres::Solve (src/main.cpp:4721) in one process.  bounds has world + 1 entries. */
int ba_net_point_ranges(const ba_network *net, int32_t world, int64_t *bounds);
/* New network (ba_net_free it) holding rank's points and observations; pt_lo/pt_hi (optional) = its range in `net`. */
int ba_net_shard(const ba_network *net, int32_t rank, int32_t world, ba_network **out, int64_t *pt_lo, int64_t *pt_hi);
/* Parameter arrays, updated in place by the caller after a solve: [6S], [3P], [3C], [16C]. */
double *ba_net_eop(ba_network *net);
double *ba_net_xyz(ba_network *net);
double *ba_net_iop(ba_network *net);
double *ba_net_ap(ba_network *net);
/* Truth of a synthetic network (NULL for a network read from files). */
const double *ba_net_truth_eop(const ba_network *net);
const double *ba_net_truth_xyz(const ba_network *net);
const double *ba_net_truth_iop(const ba_network *net);
const double *ba_net_truth_ap(const ba_network *net);
/* File ids, [n_obs] target / station of each .pho row, [n_sta], [n_pts], [n_cam]. */
const int32_t *ba_net_ids(const ba_network *net, int which /*0 pho target,1 pho station,2 eop station,3 eop camera,4 xyz target,5 iop camera*/);
int ba_net_set_free(ba_network *net, uint16_t ap_free_mask, uint8_t iop_free_mask);
/* Projection handed on by ba_net_inputs (the functor chosen at src/main.cpp:3845-3918, :4186-4250): 0 collinearity,
 * 1 collinearity stereographic, 2..5 fisheye equidistant / equisolid-angle / orthographic / stereographic,
 * 6 collinearity orthographic (:1412), 7 omnidirectional (:2700). */
int ba_net_set_model(ba_network *net, int32_t model);
/* The box the reference's stereographic and fisheye branches put on each IOP block from columns 3-6 of the .iop file
 * (src/main.cpp:3957-3961, 4252-4256): lower = xMin, -yMax, 0; upper = xMax, yMin, none (DBL_MAX); [3*n_cam] each,
 * for ba_set_iop_bounds of ba_gpu.h.  The formula is kept as written there: it presumes image rows counted from
 * yMin = 0 with y negated, and gives an empty yp interval for a sensor format centred on zero. */
int ba_net_iop_bounds(const ba_network *net, double *lower, double *upper);

/* First-match id lookup with a hash map: out[n] = first i with ids[i] == keys[n].
 * Returns BA_ERR_INVALID and sets out[n] = n_ids for every key that is absent. */
int ba_index_first_match(const int32_t *ids, int32_t n_ids, const int32_t *keys, int32_t n_keys, int32_t *out);

/* Writes <dir>/<stem>.pho .eop .iop .xyz in the reference's input format (with a trailing newline). */
int ba_net_write_inputs(const ba_network *net, const char *dir, const char *stem);

/* Writes temp.pho, EOP.jck, temp.eop, iop.jck, temp.iop, ap.jck, XYZ.jck, image.jck into dir.
 * residuals = the 2*n_obs leading entries of ba_evaluate(..., apply_loss=1, ...). The sigma
 * columns carry sqrt(variance) (src/main.cpp:7085-7118) of the variances given to
 * ba_net_set_variances, else the reference's COMPUTECX=0 defaults (sqrt(1e6); redundancy 1e6). */
int ba_net_write_results(const ba_network *net, const char *dir, const double *residuals);

/* Variances of the parameters for the result files, in the layout of ba_covariance (src/main.cpp:5200-5203:
 * eopVariance [6 per station], xyzVariance [3 per point], iopVariance [3 per camera], apVariance [16 per camera]);
 * a NULL array goes back to the default. */
int ba_net_set_variances(ba_network *net, const double *var_eop, const double *var_xyz, const double *var_iop,
                         const double *var_ap);

/* src/main.cpp:4907-4910: redundancy and a-posteriori standard deviation.
 * num_ap_correction = number of AP fixed by the subset parameterisation (:4509). */
double ba_sigma0(const ba_summary *summary, int32_t n_pts, int32_t num_ap_correction, double *redundancy);
int32_t ba_net_num_ap_correction(const ba_network *net);

/* ---- reporting (src/main.cpp:791-1019 helpers; :4756-4782, :4905-5137, :6982-6984, :7242-7346 their use) ---- */
typedef struct { double median, mean, stdev, min, max, rmse; } ba_stats;
double ba_stat_median(const double *v, int64_t n);              /* calcMedian :791                    */
double ba_stat_stddev(const double *v, int64_t n);              /* calcStdDev :811 (divides by n)     */
void ba_stat_all(const double *v, int64_t n, ba_stats *out);    /* calcStatistics :884; zeros if n==0 */
double ba_aic(double num_obs, double mse, double num_unk);      /* calculateAIC :1008                 */
double ba_bic(double num_obs, double mse, double num_unk);      /* calculateBIC :1015                 */
double ba_incidence_angle(const double *eop, const double *xyz);                         /* :923, radians */
double ba_refraction_angle(double x, double y, const double *iop, const double *ap);     /* :947, radians */
double ba_radial_distance(double x, double y, const double *iop, const double *ap);      /* :968          */

/* angles.jck: target, station, camera, refraction and incidence angle in degrees per image point (:4756-4782). */
int ba_net_write_angles(const ba_network *net, const char *dir);

typedef struct {
  double quadratic_error, redundancy, aposteriori_variance, aposteriori_stddev; /* :4906-4912 */
  double aic, bic, ic_n, ic_k;                                                    /* :4914-4915 */
  double vtpv, approx_dof, vtpv_image;                                            /* :4919-4940 */
  int32_t n_frames, n_targets;                                                    /* distinct ids of the .pho file, :3097-3107 */
  double res_x_mean, res_x_min, res_x_max, res_y_mean, res_y_min, res_y_max;      /* :6982-6983 */
  double rmse_x, rmse_y, rmse_xy;                                                 /* :6984      */
  int32_t bins;                                                                   /* QUANTILE_RESIDUALS_BINS :51 */
  int64_t bin_width;                                                              /* :4984      */
} ba_report;
typedef struct { int64_t count; ba_stats obs, res, dist; } ba_bin_stats;          /* one quantile bin, :4987-5062 */

/* The figures the reference prints after the solve.  residuals = all rows of ba_evaluate(..., apply_loss=1, ...)
 * (image rows first); bins = 0 skips the quantile statistics, else bin_stats holds `bins` entries. */
int ba_net_report(const ba_network *net, const ba_summary *summary, const double *residuals, int64_t n_residuals,
                  int32_t bins, ba_report *out, ba_bin_stats *bin_stats);
/* The same as text, line for line as the reference's std::cout (path NULL: stdout). */
int ba_net_print_report(const ba_network *net, const ba_report *report, const ba_bin_stats *bin_stats, const char *path);

/* ---- correlation analysis (the COMPUTECORRELATION blocks, src/main.cpp:5592-5640, 5727-5778, 5801-6142, 6162-6188,
 * 6211-6411, 6430-6495; helpers extractAPCorrelation :837 and calcStatistics :851) ---- */
typedef struct {
  /* first index: 0 max, 1 median, 2 mean, 3 min of |correlation| over the stations (or points); NaN (min: 0, as
   * calcStatistics leaves it) where a parameter is held constant */
  double eop_iop[4][6][3];   /* [omega phi kappa Xo Yo Zo][xp yp c], :5808-5879 */
  double eop_ap[4][6][16];   /* :5979-6050 */
  double xyz_iop[4][3][3];   /* [X Y Z][xp yp c], :6218-6260 */
  double xyz_ap[4][3][16];   /* :6350-6389 */
  double iop_ap[3][16];      /* correlationIOP_AP, :6434-6436 */
  double ap_ap[16][16];      /* correlationAP_AP, :5598 */
} ba_correlations;

/* The correlation tables of one camera from the variances of ba_covariance (var_eop [6*n_sta], var_xyz [3*n_pts],
 * var_iop [3*n_cam], var_ap [16*n_cam]) and the camera rows of ba_covariance_camera_rows (n_rows, cam_param, cov_eop,
 * cov_xyz, cov_cam; cov_xyz may be NULL: the XYZ tables are NaN then).  correlation = covariance / (sqrt(variance) *
 * sqrt(variance)) as at :5731, :5762, :6166, :6184, :6434, :5598; NaN entries (constant parameters, stations outside
 * the program) are left out of the statistics as extractAPCorrelation does.  The reference's matrices have one slot
 * per station and are overwritten camera by camera, so its figures are those of the last camera: camera = n_cam - 1
 * gives the same. */
int ba_correlations_compute(int32_t n_sta, int64_t n_pts, int32_t camera, int32_t n_rows, const int32_t *cam_param,
                            const double *var_eop, const double *var_xyz, const double *var_iop, const double *var_ap,
                            const double *cov_eop, const double *cov_xyz, const double *cov_cam, ba_correlations *out);
/* The same as text in the layout of the reference's std::cout (two decimals, fixed; path NULL: stdout). */
int ba_print_correlations(const ba_correlations *c, const char *path);

typedef struct {
  int32_t n_truth, n_estimated, n_matches;
  double mean[3], min[3], max[3], rmse[3], rmse_average;
} ba_qc;
/* Object-space error against a truth file of rows "id X Y Z sX sY sZ", matched by id (:7242-7346). */
int ba_net_qc(const ba_network *net, const char *truth_path, ba_qc *out);
int ba_net_print_qc(const ba_qc *qc, const char *truth_path, const char *path);

/* ---- the outer loop's hand-over files (src/main.cpp:3040-3056, :7544-7613).  The regression step itself (the
 * reference shells out to python/nearestNeighbour.py) is the caller's. ---- */
/* New corrections for columns 7-8 of temp.pho, [n_obs] in file order (pair with ba_set_corrections of ba_gpu.h). */
int ba_net_set_corrections(ba_network *net, const double *obs_cx, const double *obs_cy);
/* First row of kNNCost.jck / decisionTreeCost.jck: cost, redundancy, three hyper-parameters (:7562-7590). */
int ba_read_ml_cost(const char *path, double *cost, double *redundancy, double *hyper /*[3]*/);
/* One line of costs.jck (2 * lsq_cost first, :7611) and, if hyper != NULL, of hyperparameters.jck (:7599);
 * iteration 0 starts the files, later ones append. */
int ba_append_costs(const char *dir, int32_t iteration, double lsq_cost, double lsq_redundancy, double ml_cost,
                    double ml_redundancy, double reprojection_2d, double reconstruction_3d, const double *hyper);

const char *ba_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
