// ba_kernels.h -- argument blocks and launch wrappers of the kernels in ba_kernels.hip and
// ba_chol.hip.  Host-side code (ba_api.hip) only sees these.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

struct CamRec;

namespace ba {

enum { MODE_APPLY = 0, MODE_BACKSUB = 1 };
enum { SMODE_GATHER = 0, SMODE_GATHER_CAM = 1, SMODE_SETUP = 2, SMODE_CAMNORM = 3, SMODE_CAM2 = 4 };
enum { BA_CHUNK = 1024 };  // observations of one station handled by one workgroup

// Per-station sums formed inside the observation kernel (ba_keyed.hip): the plan, as the kernel sees it
#define BA_KS_NSLOT 24   // rows of a wave's LDS table: stations a segment can work on at once
#define BA_KS_NCB 8      // camera border columns per station row (the single-camera path's limit)
struct KeyedPlanDev {
  const uint8_t *spos, *slot;  // per observation: row in its tile's station-sorted order; per sorted row: table slot
  const uint32_t *umask;       // per tile: bit i set when row i or row 32 + i is the last of a run of one station
  const int *seg_op0;          // per segment: its first entry in ops
  const int2 *ops;             // x = (tile inside the segment << 8) | slot to write out before that tile, y = record
  int tiles_per_seg, n_seg;
  double *part;                // partial records, ncolp doubles each
  // null, or KS_TRACE_SLOTS counters per wave of the grid: shader clocks spent in the phases of a tile, summed over the
  // wave's tiles (ba_time_kernel, which = 12)
  unsigned long long *trace;
};
enum { KS_TRACE_SLOTS = 8 };

struct LinArgs {
  int P, ncmax;
  int64_t N;
  const int *pt_start, *obs_pt;
  const double *obs, *sta, *xyz, *scale_pt, *pt_prior, *pt_isig;
  const CamRec *cam;
  double *rec, *Jp, *Jc, *Mo, *Vinv, *g_pt, *colnorm_pt, *diag_pt, *cost_partial;
  double *Y;  // may be null
  double *Jt, *cam_partial;  // single-camera fast path (null otherwise): projected camera rows, partials of the camera sums
  double *elim;              // per-point record handed from the point kernel to the observation kernel
  unsigned *flags;
  double huber_a, lm_min, lm_max, radius;
  int reuse_diag;
  int model;  // projection, 0..7 (ba_inputs.model)
  int emp;    // 0: no camera has an empirical term AP[7..15] that is free or non-zero (kernels then skip them)
  int lean;   // 1: the rows Jp and Jc are not written (nothing reads them: the back substitution works from Y)
  int fused;  // 1: the observation kernel also forms the per-station sums (kp); it then writes Y only
  KeyedPlanDev kp;
};

struct PointArgs {
  int P, S, ncmax;
  const int *pt_start, *obs_sta, *sta_cam, *obs_pt;  // obs_sta, obs_pt: station and point of each observation in point-sorted order
  const double *obs, *sta, *Jp, *Jc, *Vinv, *vec;
  const CamRec *cam;
  double *rec;
  // back substitution only
  const double *g_pt, *xyz, *pt_prior, *pt_isig, *scale_pt;
  double *y_pt, *partial;
  // back substitution from the whitened coupling blocks (launch_backsub_y)
  const double *Y, *elim, *diag_pt;
  double radius;
  int ncam;  // free camera parameters (single camera, at most lin_ncb()) whose Z columns the elimination records hold
};

struct StationArgs {
  int n_chunks, ncmax, voff, out_stride;
  const int *chunk_sta, *chunk_begin, *sta_list, *sta_cam;
  const double *rec, *Jc, *Mo;
  const CamRec *cam;
  double *out;
};

struct StationFinishArgs {
  int S, reuse_diag;
  const int *sta_used;
  const double *F;  // folded SMODE_SETUP sums, 39 per station
  double lm_min, lm_max, radius;
  double *colnorm_sta, *diag_sta, *d2_sta, *g_sta, *b_sta, *Sdiag;
};

struct CamFinishArgs {
  int C, reuse_diag, need_norms;
  const CamRec *cam;
  const double *colnorm_in;  // image-row column norms (null when not computed)
  double lm_min, lm_max, radius;
  double *jc2, *colnorm_cam, *diag_cam, *d2_cam, *camdiag, *g_cam, *b_cam;
};

struct CostArgs {
  int64_t N;
  const double *obs, *sta, *xyz;
  const int *obs_pt;
  const CamRec *cam;
  double huber_a;
  int apply_loss;
  int model;
  int emp;  // as in LinArgs
  int one_cam;  // every station uses camera 0: its record is read at one address for all lanes
  double *res_out;  // file order, may be null
  double *partial;
};

struct CandArgs {
  int S, P, C;
  const int *sta_used;
  const double *eop, *xyz, *y_cam, *y_pt, *scale_sta, *scale_pt;
  const CamRec *cam;
  double *eop_c, *xyz_c;
  CamRec *cam_c;
  double *partial;
  double alpha;  // the candidate is Plus(x, alpha * delta); 1 outside the line search
};

struct CgArgs {
  int n, S, ncam;
  const double *b, *Minv, *Lcc, *Ssc, *Scc, *d2_sta;
  double *x, *r, *z, *p, *q, *scal;
};

void launch_station_prep(hipStream_t s, int S, const double *eop, const double *scale_sta, const int *sta_cam, double *sta);
int linearize_blocks(int P);
struct KeyedPlanHost {
  std::vector<uint8_t> spos, slot;
  std::vector<uint32_t> umask;
  std::vector<int> seg_op0, sta_rec0, sta_recs;
  std::vector<int2> ops;
  int tiles_per_seg = 0, n_seg = 0, n_rec = 0;
};
// false: some tile of 64 observations sees more than keyed_nslot() stations (the caller keeps the station sweeps)
bool build_keyed_plan(int64_t N, int S, const int *e_sta, int tiles_per_seg, int threads, KeyedPlanHost &out);
int keyed_nslot();
int keyed_ncolp(bool cam1);          // doubles per partial record
int keyed_blocks(int n_seg);         // workgroups of the observation kernel when it forms the sums
void launch_keyed_fold(hipStream_t s, int S, int ncolp, int ncam, const int *sta_rec0, const int *sta_recs, const double *part,
                       double *F, double *Ssc);
void launch_linearize(hipStream_t s, const LinArgs &a, int part = 3);
int lin_ncb();
int elim_stride();
int obs_rows_blocks(int64_t N);
void launch_camera_rows_fold(hipStream_t s, const double *cam_partial, int nblocks, int ncam, double *out_g, double *out_b);
int lin_npart();
void launch_camera_corner_fold(hipStream_t s, const double *cam_partial, int nblocks, double *corner);
void launch_point_pass(hipStream_t s, const PointArgs &a, int mode, bool withcam);
void launch_backsub_y(hipStream_t s, const PointArgs &a);
// out[slot] += g.y - (ytSy - sum d2 y^2) / 2 over the nr camera-side unknowns (one workgroup)
void launch_mcc_reduced(hipStream_t s, int64_t nr, const double *g, const double *y, const double *d2, const double *ytSy, double *out, int slot);
void launch_station_pass(hipStream_t s, const StationArgs &a, int smode);
void launch_station_fold(hipStream_t s, int S, const int *sta_chunk0, const double *part, int stride, int ncols, double *out);
void launch_camera_fold(hipStream_t s, int C, int n_chunks, const int *chunk_sta, const int *sta_cam, const CamRec *cam,
                        const double *part, int stride, int coloff, double *out_cam);
void launch_add_diag_entry(hipStream_t s, double *v, const double *diag, int64_t k);
void launch_camera_prior_cost(hipStream_t s, int C, const CamRec *cam, double *out, int slot, int fixed_slot, double *res_out);
void launch_station_finish(hipStream_t s, const StationFinishArgs &a);
void launch_invert_blocks(hipStream_t s, int S, const double *src, const int *idx, const int *sta_used, double *Minv, unsigned *flags);
int cost_blocks(int64_t n);
void launch_cost_obs(hipStream_t s, const CostArgs &a, int blocks);
void launch_cost_prior(hipStream_t s, int P, const double *xyz, const double *prior, const double *isig, const int *pt_file,
                       double *res_out, double *partial, int blocks);
// ---- index structures of a problem built on the device (ba_setup.hip)
struct SetupArrays {  // device arrays, allocated by the caller
  int *pt_file, *pt_start /* P + 1 */, *obs_pt, *obs_sta, *sta_list;
  double *obs, *pt_prior, *pt_isig;
};
bool setup_on_device(int64_t n_obs);  // BA_SETUP=host|gpu in the environment overrides the size rule
int setup_sort_observations(hipStream_t s, int N, int S, int P, const int *obs_pt, const int *obs_sta, const double *const obs_cols[6],
                            const double *pt_prior, const double *pt_sigma, const int *deg, const int *min_sta, const SetupArrays &out,
                            int *pt_file_host, std::string &err);
int point_order_station_bits(int S, int dmax);
int setup_pairs(hipStream_t s, int S, int P, const int *d_pt_start, const int *d_obs_sta, int64_t pair_cap, int2 **d_pairs, int64_t *n_pairs,
                std::vector<int> &ub_lo, std::vector<int> &ub_hi, std::vector<int> &cstart, std::string &err);

#define BA_PACK_MAX 4
struct PackList {
  double *ptr[BA_PACK_MAX];
  int n[BA_PACK_MAX];
};
// gather 1: the segments, one after another, into packed; 0: back from packed into the segments
void launch_pack_segments(hipStream_t s, const PackList &l, double *packed, int gather);
void launch_flag_to_scal(hipStream_t s, const unsigned *flags, double *out, int slot);
void launch_sum_partials(hipStream_t s, const double *partial, int n, double *out, int slot);
void launch_max_partials(hipStream_t s, const double *partial, int n, double *out, int slot);
void launch_absmax_ratio(hipStream_t s, int64_t n, const double *g, const double *scale, const double *x, double *partial, int blocks);
void launch_camera_gradient_step(hipStream_t s, int C, const CamRec *cam, const double *g_cam, double *out, int slot);
void launch_dot(hipStream_t s, int64_t n, const double *a, const double *b, double *partial, int blocks);
void launch_absmax_prod(hipStream_t s, int64_t n, const double *a, const double *b, double *partial, int blocks);
void launch_make_scale(hipStream_t s, int64_t n, const double *colnorm, double *scale);
void launch_fill(hipStream_t s, int64_t n, double *v, double val);
void launch_candidate(hipStream_t s, const CandArgs &a, int blocks);
void launch_xnorm(hipStream_t s, const CandArgs &a, int blocks);
void launch_camera_finish(hipStream_t s, const CamFinishArgs &a);
void launch_camera_mcc(hipStream_t s, int C, const CamRec *cam, const double *y_cam, double *out, int slot);
void launch_cg_init(hipStream_t s, const CgArgs &a);
void launch_cg_step(hipStream_t s, const CgArgs &a);
void launch_copy(hipStream_t s, int64_t n, const double *src, double *dst);
void launch_set_corrections(hipStream_t s, int64_t N, double *obs, const double *cx, const double *cy);
void launch_unit(hipStream_t s, int64_t n, double *v, int64_t k);
void launch_resjac_dense(hipStream_t s, const CostArgs &a, double *r_out, double *J_out);

// ba_schur.hip: explicit block-sparse reduced system and PCG on it
struct BsrCgArgs {
  int S, ncam;
  const int *rowptr, *col;
  const double *val, *Ssc, *Scc, *Lcc, *Minv, *b;  // Lcc: explicit INVERSE of the camera block of the preconditioner
  double *x, *r, *z, *p, *q, *scal, *part_pq, *part_qc, *part_rz;
  double *r2;  // second residual buffer of the multilevel iteration (iteration it reads r or r2 by parity, writes the other)
  // camera coarse space of the multilevel preconditioner (null when not used): G [6S x ncam], an
  // approximation of S_ss^-1 S_sc; Lcc is then the INVERSE of Q^T S Q, Q = [-G; I]
  const double *G, *GT;  // by rows and by columns
  double *part_gr;       // per-workgroup partials of G^T r_s
  // null, or PCG_TRACE_KERNELS x PCG_TRACE_MAXWG x PCG_TRACE_SLOTS time stamps (100 MHz counter) that thread 0 of every
  // workgroup leaves at the phase marks of the iteration's kernels (ba_time_kernel, which = 9)
  unsigned long long *trace;
  // Several ranks, product sharded by rows: this rank forms rows [row0, row_end) of q = S p (and their share of the
  // partial sums) and the ranks add their pieces up before the next kernel; rows_sharded == 0: all rows.
  int rows_sharded, row0, row_end;
};
enum { PCG_TRACE_KERNELS = 5, PCG_TRACE_MAXWG = 2048, PCG_TRACE_SLOTS = 8 };
// ba_schur_win.hip: the same blocks assembled window by window (every record read once)
#define BA_SW_WCAP 448   // observations per window: their 144-byte records fill 63 KB of LDS (two workgroups share a CU)
#define BA_SW_PCAP 2048  // pairs per window (their indices are staged in LDS too)
struct SchurWinDev {
  int n_win, n_seg, nu;
  const int *win_obs0, *win_seg0, *win_pair0;  // per window: first observation, first (window, block) segment, first pair
  const int *seg_q0;               // per segment: first pair
  const unsigned *pairs16;         // pairs in (window, block) order: observation a | observation b << 16, relative to the window
  const int *blk_seg0, *blk_segs;  // per block: its segments, ascending
  double *part;                    // one partial block (36 doubles) per segment
};
int setup_schur_windows(hipStream_t s, int64_t N, int P, const int *pt_start, const int2 *d_pairs, int64_t n_pairs, const std::vector<int> &cstart,
                        SchurWinDev &out, std::vector<void *> &owned, std::string &err);
void launch_schur_windows(hipStream_t s, const SchurWinDev &a, const double *Y, const int2 *dst, double *val);
int schur_windows_grid(const SchurWinDev &a);  // workgroups of the window kernel
void launch_schur_blocks(hipStream_t s, int nu, const int *cstart, const int2 *dst, const int2 *pairs, const double *Y, double *val);
void launch_bsr_add_diag(hipStream_t s, int S, const int *rowptr, const double *Sdiag, double *val);
void launch_bsr_to_dense(hipStream_t s, int S, int ncam, const int *rowptr, const int *col, const double *val, const double *Ssc,
                         const double *Scc, double *A, int ld);
void launch_store_border_col(hipStream_t s, int S, int ncam, int k, const double *v, double *Ssc, double *Scc);
int border_ncb();   // largest number of free camera parameters the explicit border kernels handle
int border_nacc();  // partial sums per chunk of the border station sweep
void launch_border_station(hipStream_t s, const StationArgs &sa, const double *Jt);
void launch_border_fold(hipStream_t s, int S, int ncam, int n_chunks, const int *sta_chunk0, const double *part, int stride,
                        double *Ssc);
void launch_border_corner(hipStream_t s, int ncam, const double *corner, const double *camdiag, double *Scc);
// general explicit border: any number of cameras and free camera parameters
struct GenBorderArgs {
  int P, S, ncam, ncmax, n_chunks;
  const int *pt_start, *obs_sta, *obs_pt, *sta_cam, *col_cam, *col_pos, *chunk_sta, *chunk_begin, *sta_list, *sta_chunk0;
  const double *rec, *Jp, *Jc, *Vinv;
  const CamRec *cam;
  double *Zc;              // [P x ncam x 3]
  double *out_sc, *out_cc; // per chunk: [6 x ncam], [ncmax x ncam]
  double *Ssc, *corner;    // [6S x ncam], raw corner [ncam x ncam] (before the sum over ranks)
};
void launch_border_general(hipStream_t s, const GenBorderArgs &a);
void launch_border_general_corner(hipStream_t s, int ncam, const double *corner, const double *camdiag, double *Scc);
int cg2_blocks_spmv(int S);
int cg2_blocks_vec(int S);
int cg2_blocks_vec8(int S);  // vector workgroups of the multilevel iteration: 8 lanes per station, 64 stations each
void launch_cg2_init(hipStream_t s, const BsrCgArgs &a, double rtol);
void launch_cg2_iteration(hipStream_t s, const BsrCgArgs &a, int parity);
void launch_bsr_spmv(hipStream_t s, const BsrCgArgs &a, int parity);
void launch_quadratic_form(hipStream_t s, const BsrCgArgs &a, double *out);
int defl_max_cam();  // largest camera block the coarse space handles
int cg2_max_cam();   // most free camera parameters the explicit block-sparse PCG handles
void launch_defl_get_column(hipStream_t s, int S, int ncam, int k, const double *M, double *v);
void launch_defl_set_column(hipStream_t s, int S, int ncam, int k, const double *v, double *M, double *MT);
// E = (Scc - Ssc^T G - G^T Ssc + G^T S_ss G)^-1, Lcc its Cholesky factor before inversion; SG and part are work
// arrays ([6S x ncam], [defl_gram_blocks x 64]); info[0] != 0 when the matrix was not positive definite
int defl_gram_blocks(int S);
void launch_defl_corner(hipStream_t s, const BsrCgArgs &a, const double *G, double *SG, double *part, double *E, double *Lcc, int *info);
void launch_cg2_start(hipStream_t s, const BsrCgArgs &a);                     // x = 0, r = b, z = D^-1 r, partials of r.z
void launch_cg2_spmv_update1(hipStream_t s, const BsrCgArgs &a, int parity);  // first two kernels of an iteration

// ba_amg.hip: multilevel preconditioner of the PCG above
struct Amg;
Amg *amg_create();
void amg_destroy(Amg *a);
const char *amg_error(const Amg *a);
int amg_levels(const Amg *a);
int amg_level_blocks(const Amg *a, int level);
int amg_build(Amg *a, hipStream_t s, int nb0, const std::vector<int> &rowptr, const std::vector<int> &col,
              const std::vector<int> &used0, const int *d_rowptr, const int *d_col, const double *d_val, const double *d_Dinv,
              double theta);
void amg_numeric(Amg *a, hipStream_t s, const double *d_val, const double *d_Dinv);
void launch_amg_cg_init(hipStream_t s, const BsrCgArgs &cg, const Amg *a, double rtol);
void launch_amg_cg_iteration(hipStream_t s, const BsrCgArgs &cg, const Amg *a, int parity);
// the same in two halves, for callers that put a sum over ranks between the product q = S p (launch_bsr_spmv) and the rest
bool amg_cg_iteration_splits(const BsrCgArgs &cg, const Amg *a);
void launch_amg_cg_iteration_after_product(hipStream_t s, const BsrCgArgs &cg, const Amg *a, int parity);

// ba_chol.hip: in-place lower Cholesky of a dense row-major n x n matrix (ld = n) and the
// solve L L^T x = b (b overwritten).  info[0] != 0 after the factorisation when a pivot was
// not positive.
void launch_cholesky(hipStream_t s, double *A, int n, int *info);
// ba_chol_blocked.hip: the same for matrices of order >= chol_blocked_min_order(), row-major with the even leading
// dimension chol_blocked_ld(n) and zeros in the padding; W (chol_blocked_workspace(n) doubles) receives the inverted
// diagonal blocks, which the solves and the inverse use
int chol_blocked_min_order();
int chol_blocked_ld(int n);
size_t chol_blocked_workspace(int n);
void launch_cholesky_blocked(hipStream_t s, double *A, int n, int ld, double *W, int *info);
void launch_chol_solve_blocked(hipStream_t s, const double *L, int n, int ld, const double *W, double *b);
// A: the factor on entry, (L L^T)^-1 as a full symmetric matrix on return; U: n x ld work matrix, T: 128 x ld
void launch_chol_inverse_blocked(hipStream_t s, double *A, int n, int ld, const double *W, double *U, double *T);
// C = (accumulate ? C : 0) + alpha A B^T on the matrix cores: A am x k, B bm x k, C am x bm, all row-major; lda and ldb
// even, the column after an odd k holds zero, A and B 16-byte aligned
void launch_gemm_nt(hipStream_t s, const double *A, int lda, int am, const double *B, int ldb, int bm, int k, double alpha,
                    bool accumulate, double *C, int ldc);
// One front of a sparse factorisation: dense, order m = nvp + nu, the first nv columns eliminated (nvp = nv rounded up to
// the panel width front_panel(), the rows between are padding), W the inverted diagonal blocks of its panels, f its part
// of the right-hand side / solution.
struct FrontDesc {
  double *A, *W, *f;
  double *g;          // [4 x nvp] partial sums of the backward solve
  int nv, nvp, m, ld;
  int parent;         // index of the parent's descriptor, -1 at the root
  const int *up_map;  // [m - nvp] position in the parent's front of every row that stays
  const int *vidx;    // [nvp] index in the global vector of every eliminated row, -1 for padding
};
// panels 0 .. n_panels-1 of every front of the list: diagonal block, rows below, trailing update (max_m: largest order)
void launch_front_panels(hipStream_t s, const FrontDesc *fd, int n_fronts, int n_panels, int max_m, int *info);
int front_panel();
// Process-wide switches for comparisons and tests (ba_tuning in the C-ABI); a name that was never set has its default.
//   "setup"          0: the problem's lists are built on the host, 1: on the device, -1 (default): by size
//   "pcg_graph"      0: the PCG's batches are queued launch by launch, 1 (default): replayed as graphs
//   "amg_levels"     levels of the multilevel preconditioner at most (default and maximum 3)
//   "amg_theta"      strength threshold of its aggregation (default 0.5)
//   "fused_sums"     0: the station sums are formed by separate sweeps, 1 (default): inside the observation kernel
//   "schur_windows"  0: the off-diagonal blocks are summed pair list by pair list, 1 (default): window by window
//   "keyed_tiles"    tiles per segment of the station-sum plan (default 16)
double tuning_value(const char *name, double dflt);
bool tuning_set(const char *name, double value);

// ba_sparse_chol.hip: sparse direct Cholesky of the block-sparse reduced system (nested dissection, dense fronts)
struct SparseChol;
struct SparseCholStats {
  int fronts, levels, leaves, largest_front, critical_panels;
  double arena_bytes, factor_flops;
};
SparseChol *sparse_chol_create();
void sparse_chol_destroy(SparseChol *sc);
// structure, once per network; 0, 1 (does not fit / no valid tree: err says why) or 2 (device failure)
int sparse_chol_analyse(SparseChol *sc, int S, int ncam, const std::vector<int> &rowptr, const std::vector<int> &col,
                        const std::vector<int> &used, int leaf_stations, std::string &err);
void sparse_chol_clear(SparseChol *sc, hipStream_t s);  // zeros in the fronts ahead of the next factorisation (optional)
void sparse_chol_factor(SparseChol *sc, hipStream_t s, const double *bsr_val, const double *Ssc, const double *Scc, int *d_info);
void sparse_chol_solve(SparseChol *sc, hipStream_t s, const double *b, double *x);  // entries of x outside the program are left alone
void sparse_chol_stats(const SparseChol *sc, SparseCholStats *st);
void launch_chol_solve(hipStream_t s, const double *L, int n, double *b);
void launch_chol_inverse(hipStream_t s, const double *L, int n, double *B);  // B = (L L^T)^-1, n x n

// covariance of the parameters from the inverse of the dense reduced matrix
struct CovArgs {
  int P, S, n, ncmax;
  int ld;  // leading dimension of Sinv
  const int *pt_start, *obs_sta, *sta_cam, *sta_used, *pt_file;
  const double *rec, *Jp, *Jc, *Vinv, *Sinv, *scale_pt, *scale_sta, *scale_cam;
  const CamRec *cam;
  double *Tc;       // work: [P x ncam x 3]
  double *var_xyz;  // [3P], file order
  double *var_cs;   // [n]: stations then camera columns
  // Sinv as a local matrix: S local stations (sta_loc: station -> local index), then the camera columns; only the points
  // with pt_home[jj] == home are computed.  Null pointers: Sinv is the whole inverse.
  const int *sta_loc, *pt_home;
  int home;
};
void launch_cov(hipStream_t s, const CovArgs &a);
void launch_cov_points(hipStream_t s, const CovArgs &a);

// ba_cov_select.hip: the same variances from one level of nested dissection of the station graph, for reduced systems
// whose dense inverse does not fit.  Returns 0, 1 (not applicable: no separators / not enough memory), 2 (device
// failure) or 3 (rank deficient); err says why.
struct SelInvArgs {
  hipStream_t stream;
  int S, ncam;
  const int *bsr_rowptr_h, *bsr_col_h, *sta_used_h;  // host copies of the block pattern
  const int *d_bsr_rowptr, *d_bsr_col;
  const double *d_bsr_val, *d_Ssc, *d_Scc;
  int max_cluster;  // stations per cluster at most
  CovArgs cov;      // the point kernel's arguments and the outputs var_xyz / var_cs; Sinv, ld, S, n are set per cluster
  int *d_info;
};
struct SelInvStats {
  int clusters, boundary_stations, boundary_order, largest_cluster_order;
};
int covariance_selected(const SelInvArgs &a, std::string &err, SelInvStats *stats);
void launch_dense_fix_unused(hipStream_t s, int S, int ld, const int *sta_used, double *A);

}  // namespace ba
