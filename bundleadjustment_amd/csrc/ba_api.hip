// ba_api.hip -- the C ABI of include/ba_gpu.h: problem set-up, the trust-region loop that
// drives the kernels, evaluation, sparsity pattern, multi-GPU reduction hooks.
//
// The loop follows Ceres Solver 1.13's TrustRegionMinimizer with the Levenberg-Marquardt
// strategy as configured at /root/reference/src/main.cpp:4705-4719 (Ceres is not part of the
// reference tree; the restated steps are marked [Ceres 1.13] below).  Only scalars cross the
// PCIe bus inside the loop.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "ba_device.h"
#include "ba_gpu.h"
#include "ba_kernels.h"

using namespace ba;

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_OK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(BA_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define BA_TRY(expr)          \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_ != 0) return rc_; \
  } while (0)

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int popcount32(unsigned v) { return __builtin_popcount(v); }

// Slots of the scalar block.  What is summed over the ranks in one reduction sits side by side: SC_MCC .. SC_CCOST
// (the step phase), and for the maximum SC_GMAX_P, SC_NOTPD_P (the linearisation).
enum Scal {
  SC_COST = 0, SC_FIXED, SC_GMAX, SC_STEP2_SC, SC_CNORM2_SC, SC_MCC, SC_STEP2_P, SC_CNORM2_P, SC_CCOST, SC_XNORM2_SC,
  SC_XNORM2_P, SC_GMAX_P, SC_NOTPD_P, SC_GMAX_SC, SC_LS_DOT_P, SC_LS_DOT_SC, SC_LS_MAX_P, SC_LS_MAX_SC, SC_FLAGS,
  SC_CAMINFO,  // what the Cholesky of the camera block of the reduced system reports (0, or 1 + the failing column)
  SC_SPINFO,   // what the dense or sparse Cholesky of the reduced system reports, when its caller reads it with the step's scalars
  SC_COUNT = 32
};

}  // namespace

namespace ba {
// the same for the other translation units of the library
int fail_msg(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace ba

// Most bytes of 6x6 blocks the explicit reduced system may take (beyond: the matrix-free operator): the product kernel of
// the PCG addresses the values with 32-bit byte offsets.  14 million blocks -- 600 000 stations with 22 neighbours each.
static const double BSR_MAX_BYTES = 4.0e9;

struct ba_problem {
  int N = 0, S = 0, P = 0, C = 0;
  int ncam = 0, ncmax = 0, nr = 0, n_chunks = 0;
  double huber_a = 1.0;
  int model = 0;  // projection of the image rows, 0..7
  // an empirical term AP[7..15] is estimated or non-zero in the resident parameters / in the last set uploaded as a
  // candidate: the observation kernels skip those terms when neither is the case
  bool emp = true, emp_c = true;
  hipStream_t stream = nullptr;
  std::vector<void *> allocs;
  // host copies
  std::vector<int> pt_file;  // internal point -> .xyz row
  std::vector<CamRec> cam_h;
  std::vector<int> sta_cam_h, sta_used_h;
  std::vector<int32_t> obs_pt_h, obs_sta_h;  // file order, for ba_sparsity
  std::vector<uint16_t> ap_mask_h;
  std::vector<uint8_t> iop_mask_h;
  // device: structure
  int *d_pt_start = nullptr, *d_obs_pt = nullptr, *d_obs_sta = nullptr, *d_sta_list = nullptr, *d_chunk_sta = nullptr, *d_chunk_begin = nullptr;
  int *d_sta_chunk0 = nullptr, *d_sta_cam = nullptr, *d_pt_file = nullptr, *d_sta_used = nullptr, *d_info = nullptr;
  unsigned *d_flags = nullptr;
  double *d_obs = nullptr, *d_pt_prior = nullptr, *d_pt_isig = nullptr;
  // device: parameters (current and candidate)
  double *d_eop = nullptr, *d_xyz = nullptr, *d_eop_c = nullptr, *d_xyz_c = nullptr;
  CamRec *d_cam = nullptr, *d_cam_c = nullptr;
  double *d_sta = nullptr, *d_sta_c = nullptr;
  // device: scaling, linearisation
  double *d_scale_pt = nullptr, *d_scale_sta = nullptr, *d_scale_cam = nullptr;
  double *d_rec = nullptr, *d_Jp = nullptr, *d_Jc = nullptr, *d_Mo = nullptr, *d_Vinv = nullptr, *d_g_pt = nullptr;
  double *d_y_pt = nullptr, *d_colnorm_pt = nullptr, *d_diag_pt = nullptr;
  double *d_colnorm_sta = nullptr, *d_diag_sta = nullptr, *d_d2_sta = nullptr;
  double *d_colnorm_cam = nullptr, *d_colnorm_cam_in = nullptr, *d_diag_cam = nullptr, *d_d2_cam = nullptr, *d_jc2 = nullptr;
  double *d_diagfull = nullptr;  // [nr] = d2_sta | camdiag
  double *d_F = nullptr;         // folded station sums [S*39]
  double *d_part = nullptr;      // chunk partials [n_chunks*39]
  double *d_partial = nullptr;   // block partials
  int n_partial = 0;
  double *d_scal = nullptr, *h_scal = nullptr;
  // device: reduced system
  double *d_g = nullptr, *d_b = nullptr, *d_x = nullptr, *d_r = nullptr, *d_z = nullptr, *d_p = nullptr, *d_q = nullptr;
  double *d_Minv = nullptr, *d_Ssc = nullptr, *d_Scc = nullptr, *d_Lcc = nullptr, *d_Sccinv = nullptr, *d_cgscal = nullptr, *h_cgscal = nullptr;
  double *d_Sdense = nullptr, *d_cholW = nullptr;  // dense reduced matrix (leading dimension dense_ld), inverted diagonal blocks
  int dense_ld = 0;
  bool dense_blocked = false;  // order >= chol_blocked_min_order(): the blocked factorisation on the matrix cores
  // explicit block-sparse reduced system (null / 0 when the network is too dense for it)
  bool explicit_ok = false;
  int n_ublk = 0, n_bsr = 0;
  int64_t n_pairs = 0;
  int *d_bsr_rowptr = nullptr, *d_bsr_col = nullptr, *d_ublk_cstart = nullptr;
  int2 *d_ublk_dst = nullptr, *d_pairs = nullptr;
  double *d_Y = nullptr, *d_bsr_val = nullptr, *d_Sdiag = nullptr, *d_Jt = nullptr, *d_corner = nullptr, *d_campart = nullptr, *d_elim = nullptr;
  double *d_part_pq = nullptr, *d_part_qc = nullptr, *d_part_rz = nullptr, *d_cg2scal = nullptr;
  std::vector<int> ub_lo_h, ub_hi_h, bsr_rowptr_h, bsr_col_h;  // host copies of the block structure
  SparseChol *sparse = nullptr;  // BA_SOLVER_SPARSE_SCHUR: the tree and the fronts, built at the first solve
  bool sparse_failed = false, sparse_clear_queued = false;
  hipEvent_t ev_sparse_clear = nullptr;
  hipStream_t stream_clear = nullptr;  // the clearing of the fronts: a stream of its own, so that nothing queues behind it
  int cov_route = 0, cov_max_cluster = 1536, cov_route_used = 0;  // ba_set_covariance_route / ba_covariance_stats
  SelInvStats cov_stats{};
  std::vector<int> cam_used_h;
  hipGraphExec_t cg_graph = nullptr;  // one batch of PCG iterations, captured once
  hipGraphExec_t cg_graph_tail = nullptr;  // a short batch for the end of a solve
  int last_pcg_iters = 0;                  // iterations the previous solve of this problem took (0: none yet)
  // several ranks: which of how many this problem is (ba_shard_reduced_solve / ba_use_rccl), whether the product of the
  // block-sparse PCG is formed by rows (-1 never, 0 when the model below says it pays, 1 always), its buffer
  // [q | partials], and the rows the last solve gave this rank (both 0: every rank formed the whole product)
  int shard_rank = 0, shard_world = 1, shard_mode = 0;
  bool shard_layout_checked = false;  // the ranks have compared the layout of the buffer they sum (solve_pcg_explicit)
  double *d_qshard = nullptr;
  int last_row0 = 0, last_row_end = 0;
  bool cg_graph_failed = false;
  int cg_graph_precond = -1;          // preconditioner the captured batch was recorded with
  ba::Amg *amg = nullptr;             // multilevel preconditioner (built at the first PCG solve)
  // camera coarse space of that preconditioner: G ~ S_ss^-1 S_sc, computed once per problem
  double *d_G = nullptr, *d_GT = nullptr, *d_SG = nullptr, *d_defl_part = nullptr, *d_E = nullptr, *d_Lcc_defl = nullptr, *d_part_gr = nullptr;
  double *d_defl_b = nullptr, *d_r2 = nullptr;
  bool have_G = false;
  bool cam_info_pending = false;  // build_reduced has left the camera block's factorisation report in SC_CAMINFO
  int cam_block_failed = 0;       // 1 + the column at which that factorisation broke down (the PCG cannot run then)
  bool defl_failed = false;       // the coarse-space matrix was not positive definite once: solves go without it
  // general explicit border (several cameras or many free camera parameters); null -> one probe per camera column
  double *d_Zc = nullptr, *d_part_cc = nullptr, *d_corner_gen = nullptr;
  int *d_col_cam = nullptr, *d_col_pos = nullptr;
  // state
  bool have_params = false;
  bool rows_lean = false;  // the last linearisation did not write the rows Jp and Jc (the back substitution works from Y)
  // per-station sums formed inside the observation kernel (ba_keyed.hip): plan of this network, null when it has none
  bool fused_ok = false;
  bool rows_fused = false;  // the last linearisation took that route: the station sums and the camera border are done
  ba::SchurWinDev sw{};  // off-diagonal blocks assembled window by window (n_win == 0: block by block)
  ba::KeyedPlanDev kp{};
  int kp_ncolp = 0;
  int *d_kp_sta_rec0 = nullptr, *d_kp_sta_recs = nullptr;
  double lin_radius = 1.0; // trust-region radius that linearisation carried
  double *d_zero = nullptr, *d_ytSy = nullptr;
  // second stream: the numbers of the multilevel preconditioner are set up beside the camera border, both
  // only depend on the station blocks of the reduced matrix
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_blocks = nullptr, ev_amg = nullptr, ev_rows = nullptr, ev_schur = nullptr;
  bool amg_numbers_queued = false;
  bool schur_queued = false;  // the off-diagonal blocks of this linearisation are being assembled on stream2
  bool boxed = false;        // ba_set_iop_bounds gave some camera a finite bound
  bool constrained = false;  // ... and that camera's IOP block is variable: the solve takes Ceres' bounds-constrained route
  // multi-GPU
  ba_allreduce_fn ar = nullptr;
  void *ar_user = nullptr;
  double *d_redsum = nullptr;   // [blocks | camera border | corner sums], see alloc_reduced_sums
  size_t redsum_blocks = 0, redsum_border = 0;
  bool general_border = false;  // several cameras: the border comes from the general point and station sweeps
  ncclComm_t comm = nullptr;

  template <class T>
  int dalloc(T **p, size_t n) {
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(BA_ERR_DEVICE, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    allocs.push_back(q);
    *p = static_cast<T *>(q);
    return 0;
  }
  template <class T>
  int upload(T **p, const std::vector<T> &v) {
    BA_TRY(dalloc(p, v.size()));
    if (!v.empty()) HIP_OK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
  // reductions over the ranks, counted and (on request, ba_reduction_timing) bracketed by events on the stream
  bool time_reductions = false;
  int64_t n_reductions = 0;
  double reduction_bytes = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> reduction_events;
  double reduction_seconds = 0.0;  // of event pairs already folded in (the list is kept short)
  // elapsed time of the pairs recorded so far into reduction_seconds; the events are released
  void fold_reduction_events() {
    for (auto &e : reduction_events) {
      float ms = 0.f;
      if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) reduction_seconds += 1e-3 * ms;
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    reduction_events.clear();
  }
  int allreduce(double *buf, int64_t n, int op) {
    if (!ar || n <= 0) return 0;
    n_reductions++;
    reduction_bytes += 8.0 * (double)n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (time_reductions) {
      if (reduction_events.size() >= 1024) fold_reduction_events();  // a long solve makes one reduction per inner iteration
      HIP_OK(hipEventCreate(&e0));
      HIP_OK(hipEventCreate(&e1));
      HIP_OK(hipEventRecord(e0, stream));
    }
    if (ar(ar_user, buf, n, op, stream) != 0) {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      // a communicator of this library's own is torn down at once: the ranks that are inside the collective get an
      // error from it instead of waiting for this one, and nothing is queued behind a reduction that did not happen
      if (comm) {
        (void)ncclCommAbort(comm);
        comm = nullptr;
        ar = nullptr;
      }
      return fail(BA_ERR_COMM, "reduction over the ranks failed (%lld doubles, %s): the ranks no longer hold the same sums, the solve ends here",
                  (long long)n, op == 1 ? "max" : "sum");
    }
    if (time_reductions) {
      HIP_OK(hipEventRecord(e1, stream));
      reduction_events.emplace_back(e0, e1);
    }
    return 0;
  }
  int read_scal() {
    HIP_OK(hipMemcpyAsync(h_scal, d_scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return 0;
  }
  int check_kernels() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BA_ERR_DEVICE, "kernel launch: %s", hipGetErrorString(e));
    return 0;
  }
};

namespace {

// ---------------------------------------------------------------------------------------------
// set-up
// ---------------------------------------------------------------------------------------------
// Host worker threads for the set-up's memory-bound loops: OMP_NUM_THREADS when set (job schedulers size it to
// the cores a process may use), else the hardware's count, at most 16.
int setup_threads() {
  int n = 0;
  if (const char *e = std::getenv("OMP_NUM_THREADS")) n = std::atoi(e);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(n, 16));
}

// fn(begin, end) over [0, n) split into one contiguous range per thread
template <class F>
void parallel_ranges(int64_t n, F fn) {
  const int T = (n < (1 << 16)) ? 1 : setup_threads();
  if (T == 1) { fn((int64_t)0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back(fn, n * t / T, n * (t + 1) / T);
  for (auto &x : th) x.join();
}

// Wall-clock marks of the set-up phases on stderr when BA_TRACE_SETUP is set in the environment
// (the reference prints its own timers the same way, src/main.cpp:3027, 4703).
struct SetupTrace {
  bool on = std::getenv("BA_TRACE_SETUP") != nullptr;
  double t = now_s();
  void mark(const char *what) {
    if (!on) return;
    const double u = now_s();
    fprintf(stderr, "ba_create: %-28s %8.3f s\n", what, u - t);
    t = u;
  }
};

// One allocation for everything of the reduced system that the ranks sum: the off-diagonal blocks, the camera border
// and the corner sums lie one after another, so that one reduction covers them.  Called again when the block pattern
// changes (the union over the shards).
int alloc_reduced_sums(ba_problem *p) {
  auto pad = [](size_t n) { return (n + 31) / 32 * 32; };
  const size_t nb = p->explicit_ok ? pad(36 * (size_t)p->n_bsr) : 0, nsc = pad(6 * (size_t)p->S * p->ncam);
  const size_t ncorner = pad(std::max<size_t>(64, (size_t)p->ncam * p->ncam));
  if (p->d_redsum) {
    auto it = std::find(p->allocs.begin(), p->allocs.end(), (void *)p->d_redsum);
    if (it != p->allocs.end()) p->allocs.erase(it);
    HIP_OK(hipFree(p->d_redsum));
    p->d_redsum = nullptr;
  }
  BA_TRY(p->dalloc(&p->d_redsum, nb + nsc + ncorner));
  HIP_OK(hipMemset(p->d_redsum, 0, (nb + nsc + ncorner) * sizeof(double)));
  p->d_bsr_val = p->explicit_ok ? p->d_redsum : nullptr;
  p->d_Ssc = p->d_redsum + nb;
  p->d_corner = p->d_Ssc + nsc;
  p->d_corner_gen = p->general_border ? p->d_corner : nullptr;
  p->redsum_blocks = nb;
  p->redsum_border = nsc + ncorner;
  return 0;
}

int build_problem(ba_problem *p, const ba_dims *d, const ba_inputs *in) {
  SetupTrace trace;
  const int N = d->n_obs, S = d->n_sta, P = d->n_pts, C = d->n_cam;
  if (N < 0 || S < 0 || P < 0 || C < 0) return fail(BA_ERR_INVALID, "negative dimension");
  if (in->model < 0 || in->model > 7)
    return fail(BA_ERR_UNSUPPORTED, "projection model %d is not implemented (0 collinearity, 1 stereographic, 2..5 fisheye, 6 orthographic, 7 omnidirectional)", in->model);
  if (N > 0 && (!in->obs_pt || !in->obs_sta || !in->obs_x || !in->obs_y || !in->obs_sx || !in->obs_sy || !in->obs_cx || !in->obs_cy))
    return fail(BA_ERR_INVALID, "null observation array");
  if ((S > 0 && !in->sta_cam) || (P > 0 && (!in->pt_prior || !in->pt_sigma)) ||
      (C > 0 && (!in->iop_prior || !in->iop_sigma || !in->ap_free_mask || !in->iop_free_mask)))
    return fail(BA_ERR_INVALID, "null station / point / camera array");
  p->N = N; p->S = S; p->P = P; p->C = C;
  p->huber_a = in->huber_a;
  p->model = in->model;
  for (int n = 0; n < N; ++n) {
    if (in->obs_pt[n] < 0 || in->obs_pt[n] >= P) return fail(BA_ERR_INVALID, "observation %d: point index %d out of range [0,%d)", n, in->obs_pt[n], P);
    if (in->obs_sta[n] < 0 || in->obs_sta[n] >= S) return fail(BA_ERR_INVALID, "observation %d: station index %d out of range [0,%d)", n, in->obs_sta[n], S);
    if (!(in->obs_sx[n] > 0.0) || !(in->obs_sy[n] > 0.0)) return fail(BA_ERR_INVALID, "observation %d: standard deviation must be positive", n);
  }
  for (int i = 0; i < S; ++i)
    if (in->sta_cam[i] < 0 || in->sta_cam[i] >= C) return fail(BA_ERR_INVALID, "station %d: camera index %d out of range [0,%d)", i, in->sta_cam[i], C);
  for (int64_t k = 0; k < 3 * (int64_t)P; ++k)
    if (!(in->pt_sigma[k] > 0.0)) return fail(BA_ERR_INVALID, "point %lld: prior sigma must be positive", (long long)(k / 3));
  for (int k = 0; k < 3 * C; ++k)
    if (!(in->iop_sigma[k] > 0.0)) return fail(BA_ERR_INVALID, "camera %d: prior sigma must be positive", k / 3);

  trace.mark("validation");
  p->obs_pt_h.assign(in->obs_pt, in->obs_pt + N);
  p->obs_sta_h.assign(in->obs_sta, in->obs_sta + N);
  p->sta_cam_h.assign(in->sta_cam, in->sta_cam + S);
  p->ap_mask_h.assign(in->ap_free_mask, in->ap_free_mask + C);
  p->iop_mask_h.assign(in->iop_free_mask, in->iop_free_mask + C);
  p->sta_used_h.assign(S, 0);
  std::vector<int> &cam_used = p->cam_used_h;
  cam_used.assign(C, 0);
  std::vector<int> deg(P, 0), sdeg(S, 0), min_sta(P, std::max(S - 1, 0));
  for (int n = 0; n < N; ++n) {
    p->sta_used_h[in->obs_sta[n]] = 1;
    cam_used[in->sta_cam[in->obs_sta[n]]] = 1;
    deg[in->obs_pt[n]]++;
    sdeg[in->obs_sta[n]]++;
    min_sta[in->obs_pt[n]] = std::min(min_sta[in->obs_pt[n]], in->obs_sta[n]);
  }
  // a share of a larger network: cameras that only other ranks observe keep their columns here too
  if (in->cam_in_use)
    for (int c = 0; c < C; ++c)
      if (in->cam_in_use[c]) cam_used[c] = 1;
  // cameras: free-parameter lists (src/main.cpp:4479-4511 for the AP; an AP block that no
  // observation references is not part of the reduced program)
  p->cam_h.assign(C, CamRec());
  int off = 0;
  for (int c = 0; c < C; ++c) {
    CamRec &r = p->cam_h[c];
    std::memset(&r, 0, sizeof r);
    unsigned mask = in->iop_free_mask[c] & 7u;
    if (cam_used[c]) mask |= (unsigned)in->ap_free_mask[c] << 3;
    r.mask = mask;
    r.nc = popcount32(mask);
    r.off = off;
    off += r.nc;
    r.in_program = ((in->iop_free_mask[c] & 7u) ? 1 : 0) | ((in->ap_free_mask[c] && cam_used[c]) ? 2 : 0);
    for (int k = 0; k < BA_NCAMPAR; ++k) r.scale[k] = 1.0;
    for (int k = 0; k < 3; ++k) {
      r.prior[k] = in->iop_prior[3 * c + k];
      r.isig[k] = 1.0 / in->iop_sigma[3 * c + k];
      r.lo[k] = -DBL_MAX;
      r.hi[k] = DBL_MAX;
    }
    p->ncmax = std::max(p->ncmax, r.nc);
  }
  p->ncam = off;
  p->nr = 6 * S + p->ncam;

  trace.mark("degrees, cameras");
  // The index structures: on the device for networks of some size (ba_setup.hip), by the host loops below otherwise.
  // Both give the same arrays.
  const bool on_device = setup_on_device(N);
  HIP_OK(hipStreamCreate(&p->stream));
  HIP_OK(hipStreamCreate(&p->stream2));
  // internal point order: by number of observations (descending, stable) so that the lanes of
  // a wave walk segments of similar length
  p->pt_file.resize(P);
  std::vector<int> pt_start, obs_pt, e_sta;
  std::vector<double> obs;
  if (on_device) {
    SetupArrays sa{};
    BA_TRY(p->dalloc(&p->d_pt_file, (size_t)P));
    BA_TRY(p->dalloc(&p->d_pt_start, (size_t)P + 1));
    BA_TRY(p->dalloc(&p->d_obs_pt, (size_t)N));
    BA_TRY(p->dalloc(&p->d_obs_sta, (size_t)N));
    BA_TRY(p->dalloc(&p->d_sta_list, (size_t)N));
    BA_TRY(p->dalloc(&p->d_obs, (size_t)N * BA_OBS_STRIDE));
    BA_TRY(p->dalloc(&p->d_pt_prior, 3 * (size_t)P));
    BA_TRY(p->dalloc(&p->d_pt_isig, 3 * (size_t)P));
    sa.pt_file = p->d_pt_file; sa.pt_start = p->d_pt_start; sa.obs_pt = p->d_obs_pt; sa.obs_sta = p->d_obs_sta; sa.sta_list = p->d_sta_list;
    sa.obs = p->d_obs; sa.pt_prior = p->d_pt_prior; sa.pt_isig = p->d_pt_isig;
    const double *cols[6] = {in->obs_x, in->obs_y, in->obs_sx, in->obs_sy, in->obs_cx, in->obs_cy};
    std::string err;
    if (setup_sort_observations(p->stream, N, S, P, in->obs_pt, in->obs_sta, cols, in->pt_prior, in->pt_sigma, deg.data(), min_sta.data(), sa, p->pt_file.data(), err))
      return fail(BA_ERR_DEVICE, "set-up on the device: %s", err.c_str());
  } else {
  std::iota(p->pt_file.begin(), p->pt_file.end(), 0);
  {
    // more observations first, then by the lowest station that sees the point (points of one neighbourhood end up side
    // by side, so the records a station sweep gathers cluster); the device path sorts by the same key
    int dmax = 0;
    for (int j = 0; j < P; ++j) dmax = std::max(dmax, deg[j]);
    const bool by_station = point_order_station_bits(S, dmax) > 0;
    std::stable_sort(p->pt_file.begin(), p->pt_file.end(), [&](int a, int b) {
      if (deg[a] != deg[b]) return deg[a] > deg[b];
      return by_station && min_sta[a] < min_sta[b];
    });
  }
  std::vector<int> pt_int(P);
  for (int jj = 0; jj < P; ++jj) pt_int[p->pt_file[jj]] = jj;
  pt_start.assign(P + 1, 0);
  for (int jj = 0; jj < P; ++jj) pt_start[jj + 1] = pt_start[jj] + deg[p->pt_file[jj]];
  // observations sorted by internal point, file order inside a point
  std::vector<int> fill(pt_start.begin(), pt_start.end() - 1);
  obs.assign((size_t)N * BA_OBS_STRIDE, 0.0);
  obs_pt.resize(N); e_sta.resize(N);
  std::vector<int> slot(N);
  for (int n = 0; n < N; ++n) slot[n] = fill[pt_int[in->obs_pt[n]]]++;  // the only order-dependent part
  parallel_ranges(N, [&](int64_t n0, int64_t n1) {
    for (int64_t n = n0; n < n1; ++n) {
      const int e = slot[n];
      double *o = &obs[(size_t)e * BA_OBS_STRIDE];
      o[0] = in->obs_x[n]; o[1] = in->obs_y[n]; o[2] = in->obs_sx[n]; o[3] = in->obs_sy[n];
      o[4] = in->obs_cx[n]; o[5] = in->obs_cy[n];
      int32_t w[2] = {in->obs_sta[n], (int32_t)n};
      std::memcpy(&o[6], w, sizeof w);
      obs_pt[e] = pt_int[in->obs_pt[n]];
      e_sta[e] = in->obs_sta[n];
    }
  });
  }
  trace.mark("observations by point");
  // station-sorted index list and its chunks
  std::vector<int> sta_start(S + 1, 0);
  for (int i = 0; i < S; ++i) sta_start[i + 1] = sta_start[i] + sdeg[i];
  std::vector<int> sta_list;
  if (!on_device) {
    std::vector<int> sfill(sta_start.begin(), sta_start.end() - 1);
    sta_list.resize(N);
    for (int e = 0; e < N; ++e) sta_list[sfill[e_sta[e]]++] = e;
  }
  std::vector<int> chunk_sta, chunk_begin, sta_chunk0(S + 1, 0);
  for (int i = 0; i < S; ++i) {
    sta_chunk0[i] = (int)chunk_sta.size();
    for (int b = sta_start[i]; b < sta_start[i + 1]; b += BA_CHUNK) {
      chunk_sta.push_back(i);
      chunk_begin.push_back(b);
    }
  }
  sta_chunk0[S] = (int)chunk_sta.size();
  chunk_begin.push_back(N);
  // a chunk ends where the next begins, except at a station boundary
  p->n_chunks = (int)chunk_sta.size();
  std::vector<int> chunk_range(p->n_chunks + 1);
  for (int ch = 0; ch < p->n_chunks; ++ch) chunk_range[ch] = chunk_begin[ch];
  chunk_range[p->n_chunks] = N;
  // (chunks are laid out in station order and stations are contiguous in sta_list, so
  //  chunk_begin[ch+1] is also the end of chunk ch)

  trace.mark("observations by station");
  // Block structure of the reduced matrix: every unordered pair of observations of one point
  // couples the two stations.  Pairs are grouped by (lo station, hi station); the groups become
  // the upper-triangular blocks, mirrored into a full BSR pattern for the SpMV.
  std::vector<int> bsr_rowptr, bsr_col, ublk_cstart;
  std::vector<int2> ublk_dst, pairs;
  {
    int64_t npairs = 0;
    for (int j = 0; j < P; ++j) npairs += (int64_t)deg[j] * (deg[j] - 1) / 2;
    // pairs of one point that share a station are listed in both orders (diagonal block)
    const int64_t pair_cap = 600000000LL;
    std::vector<int> ub_lo, ub_hi;
    int64_t tot = 0;
    bool have_blocks = false;
    if (npairs <= pair_cap && S > 0 && on_device) {
      std::string err;
      const int rc = setup_pairs(p->stream, S, P, p->d_pt_start, p->d_obs_sta, pair_cap, &p->d_pairs, &tot, ub_lo, ub_hi, ublk_cstart, err);
      if (rc < 0) return fail(BA_ERR_DEVICE, "set-up on the device: %s", err.c_str());
      if (p->d_pairs) p->allocs.push_back(p->d_pairs);
      have_blocks = rc == 0;
      trace.mark("pairs on the device");
    } else if (npairs <= pair_cap && S > 0) {
      std::vector<int64_t> rowcnt(S + 1, 0);
      auto for_pairs = [&](auto &&fn) {
        for (int jj = 0; jj < P; ++jj)
          for (int a = pt_start[jj]; a < pt_start[jj + 1]; ++a)
            for (int b = a + 1; b < pt_start[jj + 1]; ++b) {
              const int ia = e_sta[a], ib = e_sta[b];
              if (ia < ib) fn(ia, ib, a, b);
              else if (ib < ia) fn(ib, ia, b, a);
              else { fn(ia, ia, a, b); fn(ia, ia, b, a); }
            }
      };
      for_pairs([&](int lo, int, int, int) { rowcnt[lo + 1]++; });
      for (int i = 0; i < S; ++i) rowcnt[i + 1] += rowcnt[i];
      tot = rowcnt[S];
      std::vector<int> hi_of(tot), ea_of(tot), eb_of(tot);
      {
        std::vector<int64_t> pos(rowcnt.begin(), rowcnt.end() - 1);
        for_pairs([&](int lo, int hi, int a, int b) {
          const int64_t q = pos[lo]++;
          hi_of[q] = hi; ea_of[q] = a; eb_of[q] = b;
        });
      }
      trace.mark("pairs by low station");
      // inside a row: stable counting sort by hi (keeps point order inside a block)
      pairs.resize(tot);
      std::vector<int> cnt(S + 1), nb_lo(S, 0), nb_hi(S, 0);
      ublk_cstart.push_back(0);
      std::vector<int> touched;
      for (int lo = 0; lo < S; ++lo) {
        const int64_t r0 = rowcnt[lo], r1 = rowcnt[lo + 1];
        if (r0 == r1) continue;
        touched.clear();
        for (int64_t q = r0; q < r1; ++q)
          if (cnt[hi_of[q]]++ == 0) touched.push_back(hi_of[q]);
        std::sort(touched.begin(), touched.end());
        int64_t run = r0;
        for (int h : touched) {
          const int c = cnt[h];
          cnt[h] = (int)(run - r0);  // start offset of this block inside the row
          run += c;
          ub_lo.push_back(lo);
          ub_hi.push_back(h);
          ublk_cstart.push_back((int)run);
        }
        for (int64_t q = r0; q < r1; ++q) {
          const int64_t d = r0 + cnt[hi_of[q]]++;
          pairs[d] = make_int2(ea_of[q], eb_of[q]);
        }
        for (int h : touched) cnt[h] = 0;
      }
      trace.mark("pairs by block");
      have_blocks = true;
    }
    if (have_blocks) {
      const int nu = (int)ub_lo.size();
      // full BSR pattern: diagonal first, then neighbours in ascending column order
      std::vector<int> nnb(S, 1);
      for (int u = 0; u < nu; ++u)
        if (ub_lo[u] != ub_hi[u]) { nnb[ub_lo[u]]++; nnb[ub_hi[u]]++; }
      bsr_rowptr.assign(S + 1, 0);
      for (int i = 0; i < S; ++i) bsr_rowptr[i + 1] = bsr_rowptr[i] + nnb[i];
      const int64_t nb_total = bsr_rowptr[S];
      if ((double)nb_total * 288.0 <= BSR_MAX_BYTES && tot < 2000000000LL) {
        bsr_col.assign(nb_total, 0);
        std::vector<int> fillb(S);
        for (int i = 0; i < S; ++i) { bsr_col[bsr_rowptr[i]] = i; fillb[i] = bsr_rowptr[i] + 1; }
        ublk_dst.resize(nu);
        for (int u = 0; u < nu; ++u) {
          const int lo = ub_lo[u], hi = ub_hi[u];
          if (lo == hi) { ublk_dst[u] = make_int2(bsr_rowptr[lo], -1); continue; }
          const int dlo = fillb[lo]++, dhi = fillb[hi]++;
          bsr_col[dlo] = hi;
          bsr_col[dhi] = lo;
          ublk_dst[u] = make_int2(dlo, dhi);
        }
        p->explicit_ok = true;
        p->n_ublk = nu;
        p->n_bsr = (int)nb_total;
        p->n_pairs = tot;
        p->ub_lo_h = ub_lo;
        p->ub_hi_h = ub_hi;
        p->bsr_rowptr_h = bsr_rowptr;
        p->bsr_col_h = bsr_col;
      }
    }
  }

  trace.mark("block-sparse pattern");
  std::vector<double> prior, isig;
  if (!on_device) {
    prior.resize(3 * (size_t)P); isig.resize(3 * (size_t)P);
    for (int jj = 0; jj < P; ++jj)
      for (int k = 0; k < 3; ++k) {
        prior[3 * (size_t)jj + k] = in->pt_prior[3 * (size_t)p->pt_file[jj] + k];
        isig[3 * (size_t)jj + k] = 1.0 / in->pt_sigma[3 * (size_t)p->pt_file[jj] + k];
      }
  }

  trace.mark("point priors");
  HIP_OK(hipEventCreateWithFlags(&p->ev_blocks, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&p->ev_amg, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&p->ev_rows, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&p->ev_schur, hipEventDisableTiming));
  trace.mark("stream");
  if (!on_device) {
    BA_TRY(p->upload(&p->d_pt_start, pt_start));
    BA_TRY(p->upload(&p->d_obs_pt, obs_pt));
    BA_TRY(p->upload(&p->d_obs_sta, e_sta));
    BA_TRY(p->upload(&p->d_sta_list, sta_list));
    BA_TRY(p->upload(&p->d_pt_file, p->pt_file));
    BA_TRY(p->upload(&p->d_obs, obs));
    BA_TRY(p->upload(&p->d_pt_prior, prior));
    BA_TRY(p->upload(&p->d_pt_isig, isig));
  }
  BA_TRY(p->upload(&p->d_chunk_sta, chunk_sta));
  BA_TRY(p->upload(&p->d_chunk_begin, chunk_range));
  BA_TRY(p->upload(&p->d_sta_chunk0, sta_chunk0));
  BA_TRY(p->upload(&p->d_sta_cam, p->sta_cam_h));
  BA_TRY(p->upload(&p->d_sta_used, p->sta_used_h));
  BA_TRY(p->upload(&p->d_cam, p->cam_h));
  trace.mark("uploads");
  BA_TRY(p->dalloc(&p->d_cam_c, (size_t)C));
  BA_TRY(p->dalloc(&p->d_info, 4));
  BA_TRY(p->dalloc(&p->d_flags, 4));
  BA_TRY(p->dalloc(&p->d_eop, 6 * (size_t)S));
  BA_TRY(p->dalloc(&p->d_eop_c, 6 * (size_t)S));
  BA_TRY(p->dalloc(&p->d_xyz, 3 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_xyz_c, 3 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_sta, (size_t)BA_STA_STRIDE * S));
  BA_TRY(p->dalloc(&p->d_sta_c, (size_t)BA_STA_STRIDE * S));
  BA_TRY(p->dalloc(&p->d_scale_pt, 3 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_scale_sta, 6 * (size_t)S));
  BA_TRY(p->dalloc(&p->d_scale_cam, (size_t)p->ncam));
  BA_TRY(p->dalloc(&p->d_rec, (size_t)BA_REC_STRIDE * N));
  BA_TRY(p->dalloc(&p->d_Jp, 6 * (size_t)N));
  BA_TRY(p->dalloc(&p->d_Jc, 2 * (size_t)p->ncmax * N));
  BA_TRY(p->dalloc(&p->d_Mo, 3 * (size_t)N));
  BA_TRY(p->dalloc(&p->d_Vinv, 6 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_elim, (size_t)elim_stride() * P));
  BA_TRY(p->dalloc(&p->d_g_pt, 3 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_y_pt, 3 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_colnorm_pt, 3 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_diag_pt, 3 * (size_t)P));
  BA_TRY(p->dalloc(&p->d_colnorm_sta, 6 * (size_t)S));
  BA_TRY(p->dalloc(&p->d_diag_sta, 6 * (size_t)S));
  BA_TRY(p->dalloc(&p->d_colnorm_cam, (size_t)p->ncam));
  BA_TRY(p->dalloc(&p->d_colnorm_cam_in, (size_t)p->ncam));
  BA_TRY(p->dalloc(&p->d_diag_cam, (size_t)p->ncam));
  BA_TRY(p->dalloc(&p->d_d2_cam, (size_t)p->ncam));
  BA_TRY(p->dalloc(&p->d_jc2, (size_t)p->ncam));
  BA_TRY(p->dalloc(&p->d_diagfull, (size_t)p->nr));
  p->d_d2_sta = p->d_diagfull;
  BA_TRY(p->dalloc(&p->d_F, 39 * (size_t)S + 3 * (size_t)p->ncam + 8));  // room behind for what is reduced along with it
  BA_TRY(p->dalloc(&p->d_part, (size_t)std::max(std::max(39, border_nacc()), 6 * p->ncam) * p->n_chunks));
  BA_TRY(p->dalloc(&p->d_Sdiag, 36 * (size_t)S));
  trace.mark("allocations");
  if (p->explicit_ok) {
    BA_TRY(p->upload(&p->d_bsr_rowptr, bsr_rowptr));
    BA_TRY(p->upload(&p->d_bsr_col, bsr_col));
    BA_TRY(p->upload(&p->d_ublk_cstart, ublk_cstart));
    BA_TRY(p->upload(&p->d_ublk_dst, ublk_dst));
    if (!on_device) BA_TRY(p->upload(&p->d_pairs, pairs));
    BA_TRY(p->dalloc(&p->d_Y, (size_t)BA_Y_STRIDE * N));
    const size_t npart = (size_t)std::max(std::max(cg2_blocks_spmv(S), cg2_blocks_vec8(S)), 1024) + 1;
    BA_TRY(p->dalloc(&p->d_part_pq, npart));
    BA_TRY(p->dalloc(&p->d_part_qc, npart * std::max(p->ncam, 1)));
    BA_TRY(p->dalloc(&p->d_part_rz, npart));
    BA_TRY(p->dalloc(&p->d_cg2scal, 32));
  }
  if (p->explicit_ok && p->n_pairs > 0 && tuning_value("schur_windows", 1.0) != 0.0) {
    std::vector<int> pts_host;
    const int *pts = pt_start.data();
    if (on_device) {
      pts_host.resize((size_t)P + 1);
      HIP_OK(hipMemcpy(pts_host.data(), p->d_pt_start, ((size_t)P + 1) * sizeof(int), hipMemcpyDeviceToHost));
      pts = pts_host.data();
    }
    std::string err;
    if (setup_schur_windows(p->stream, N, P, pts, p->d_pairs, p->n_pairs, ublk_cstart, p->sw, p->allocs, err) < 0)
      return fail(BA_ERR_DEVICE, "set-up of the window lists: %s", err.c_str());
  }
  trace.mark("pair lists upload");
  const bool single_camera_path = C == 1 && p->ncmax > 0 && p->ncmax <= border_ncb() && border_ncb() == lin_ncb();
  if (p->ncam > 0 && !single_camera_path && (double)P * p->ncam * 24.0 <= 16e9) {
    std::vector<int> col_cam(p->ncam), col_pos(p->ncam);
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < p->cam_h[c].nc; ++k) {
        col_cam[p->cam_h[c].off + k] = c;
        col_pos[p->cam_h[c].off + k] = k;
      }
    BA_TRY(p->upload(&p->d_col_cam, col_cam));
    BA_TRY(p->upload(&p->d_col_pos, col_pos));
    BA_TRY(p->dalloc(&p->d_Zc, (size_t)P * p->ncam * 3));
    BA_TRY(p->dalloc(&p->d_part_cc, (size_t)p->n_chunks * p->ncmax * p->ncam));
    p->general_border = true;
  }
  BA_TRY(alloc_reduced_sums(p));
  int campart_rows = obs_rows_blocks(N);
  if (p->explicit_ok && N > 0 && (single_camera_path || p->ncam == 0) && tuning_value("fused_sums", 1.0) != 0.0) {
    // The station sums and the camera border can come out of the observation kernel when no tile of 64 observations
    // sees more stations than a wave's table has rows, and the partial records stay a small part of the traffic.
    std::vector<int> sta_host;
    const int *sta_of_obs = e_sta.data();
    if (on_device) {
      sta_host.resize((size_t)N);
      HIP_OK(hipMemcpy(sta_host.data(), p->d_obs_sta, (size_t)N * sizeof(int), hipMemcpyDeviceToHost));
      sta_of_obs = sta_host.data();
    }
    int tps = 16;
    tps = std::max(1, (int)tuning_value("keyed_tiles", (double)tps));
    KeyedPlanHost kh;
    const int ncolp = keyed_ncolp(single_camera_path);
    if (build_keyed_plan(N, S, sta_of_obs, tps, setup_threads(), kh) && (double)kh.n_rec * ncolp * 8.0 <= 48.0 * (double)N + 1e6) {
      uint8_t *d_spos = nullptr, *d_slot = nullptr;
      uint32_t *d_um = nullptr;
      int *d_op0 = nullptr;
      int2 *d_ops = nullptr;
      BA_TRY(p->upload(&d_spos, kh.spos));
      BA_TRY(p->upload(&d_slot, kh.slot));
      BA_TRY(p->upload(&d_um, kh.umask));
      BA_TRY(p->upload(&d_op0, kh.seg_op0));
      BA_TRY(p->upload(&d_ops, kh.ops));
      BA_TRY(p->upload(&p->d_kp_sta_rec0, kh.sta_rec0));
      BA_TRY(p->upload(&p->d_kp_sta_recs, kh.sta_recs));
      p->kp.spos = d_spos; p->kp.slot = d_slot; p->kp.umask = d_um; p->kp.seg_op0 = d_op0; p->kp.ops = d_ops;
      p->kp.tiles_per_seg = kh.tiles_per_seg; p->kp.n_seg = kh.n_seg;
      BA_TRY(p->dalloc(&p->kp.part, (size_t)kh.n_rec * ncolp));
      p->kp_ncolp = ncolp;
      p->fused_ok = true;
      campart_rows = std::max(campart_rows, keyed_blocks(kh.n_seg));
    }
    trace.mark("plan of the station sums");
  }
  if (single_camera_path) {
    // single camera with few free parameters: the linearisation kernel also produces the camera border's inputs
    BA_TRY(p->dalloc(&p->d_Jt, 2 * (size_t)border_ncb() * N));
    BA_TRY(p->dalloc(&p->d_campart, (size_t)campart_rows * lin_npart()));
  }
  p->n_partial = 4 * std::max(std::max(linearize_blocks(P), cost_blocks(N)), cost_blocks(6 * (int64_t)S + 3 * (int64_t)P)) + 16;
  BA_TRY(p->dalloc(&p->d_partial, (size_t)p->n_partial));
  BA_TRY(p->dalloc(&p->d_scal, (size_t)SC_COUNT));
  BA_TRY(p->dalloc(&p->d_cgscal, 8));
  BA_TRY(p->dalloc(&p->d_zero, 16));
  BA_TRY(p->dalloc(&p->d_ytSy, 2));
  HIP_OK(hipMemsetAsync(p->d_zero, 0, 16 * sizeof(double), p->stream));
  HIP_OK(hipHostMalloc((void **)&p->h_scal, SC_COUNT * sizeof(double)));
  HIP_OK(hipHostMalloc((void **)&p->h_cgscal, 16 * sizeof(double)));
  const size_t nr = (size_t)p->nr;
  BA_TRY(p->dalloc(&p->d_g, nr)); BA_TRY(p->dalloc(&p->d_b, nr)); BA_TRY(p->dalloc(&p->d_x, nr));
  BA_TRY(p->dalloc(&p->d_r, nr)); BA_TRY(p->dalloc(&p->d_z, nr)); BA_TRY(p->dalloc(&p->d_p, nr));
  BA_TRY(p->dalloc(&p->d_q, nr));
  BA_TRY(p->dalloc(&p->d_r2, nr));
  BA_TRY(p->dalloc(&p->d_Minv, 36 * (size_t)S));
  BA_TRY(p->dalloc(&p->d_Scc, (size_t)p->ncam * p->ncam));
  BA_TRY(p->dalloc(&p->d_Lcc, (size_t)p->ncam * p->ncam));
  BA_TRY(p->dalloc(&p->d_Sccinv, (size_t)p->ncam * p->ncam));
  HIP_OK(hipMemsetAsync(p->d_flags, 0, 4 * sizeof(unsigned), p->stream));
  HIP_OK(hipMemsetAsync(p->d_scal, 0, SC_COUNT * sizeof(double), p->stream));
  HIP_OK(hipStreamSynchronize(p->stream));
  trace.mark("remaining allocations");
  return 0;
}

int set_parameters(ba_problem *p, const double *eop, const double *xyz, const double *iop, const double *ap, bool candidate) {
  if ((p->S && !eop) || (p->P && !xyz) || (p->C && (!iop || !ap))) return fail(BA_ERR_INVALID, "null parameter array");
  std::vector<double> x(3 * (size_t)p->P);
  for (int jj = 0; jj < p->P; ++jj)
    for (int k = 0; k < 3; ++k) x[3 * (size_t)jj + k] = xyz[3 * (size_t)p->pt_file[jj] + k];
  std::vector<CamRec> cam = p->cam_h;
  for (int c = 0; c < p->C; ++c) {
    for (int k = 0; k < 3; ++k) cam[c].iop[k] = iop[3 * c + k];
    for (int k = 0; k < 16; ++k) cam[c].ap[k] = ap[16 * c + k];
  }
  bool emp = false;
  for (int c = 0; c < p->C; ++c) {
    if (cam[c].mask >> BA_NCAMPAR_BROWN) emp = true;
    for (int k = 7; k < 16; ++k)
      if (ap[16 * c + k] != 0.0) emp = true;
  }
  (candidate ? p->emp_c : p->emp) = emp;
  double *de = candidate ? p->d_eop_c : p->d_eop, *dx = candidate ? p->d_xyz_c : p->d_xyz;
  CamRec *dc = candidate ? p->d_cam_c : p->d_cam;
  if (p->S) HIP_OK(hipMemcpyAsync(de, eop, 6 * (size_t)p->S * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (p->P) HIP_OK(hipMemcpyAsync(dx, x.data(), x.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (p->C) HIP_OK(hipMemcpyAsync(dc, cam.data(), cam.size() * sizeof(CamRec), hipMemcpyHostToDevice, p->stream));
  HIP_OK(hipStreamSynchronize(p->stream));
  if (!candidate) {
    p->cam_h = cam;
    p->have_params = true;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// phases of one LM iteration
// ---------------------------------------------------------------------------------------------
struct LmState {
  double radius, lm_min, lm_max;
  bool reuse_diag;
  bool need_norms;  // column norms of the camera block are only needed when they are not clamped away
  int reduced = 0;  // what to build after the elimination: 0 nothing, 1 preconditioner + border (implicit operator),
                    // 2 explicit block-sparse system + border
  bool full_rows = false;  // write the rows Jp and Jc even where the solve itself does not read them (covariance, timing)
};

StationArgs station_args(ba_problem *p, int voff, int stride) {
  StationArgs a{};
  a.n_chunks = p->n_chunks; a.ncmax = p->ncmax; a.voff = voff; a.out_stride = stride;
  a.chunk_sta = p->d_chunk_sta; a.chunk_begin = p->d_chunk_begin; a.sta_list = p->d_sta_list; a.sta_cam = p->d_sta_cam;
  a.rec = p->d_rec; a.Jc = p->d_Jc; a.Mo = p->d_Mo; a.cam = p->d_cam; a.out = p->d_part;
  return a;
}

PointArgs point_args(ba_problem *p, const double *vec) {
  PointArgs a{};
  a.P = p->P; a.S = p->S; a.ncmax = p->ncmax;
  a.pt_start = p->d_pt_start; a.obs_sta = p->d_obs_sta; a.sta_cam = p->d_sta_cam; a.obs_pt = p->d_obs_pt;
  a.obs = p->d_obs; a.sta = p->d_sta; a.Jp = p->d_Jp; a.Jc = p->d_Jc; a.Vinv = p->d_Vinv;
  a.vec = vec; a.cam = p->d_cam; a.rec = p->d_rec;
  a.g_pt = p->d_g_pt; a.xyz = p->d_xyz; a.pt_prior = p->d_pt_prior; a.pt_isig = p->d_pt_isig; a.scale_pt = p->d_scale_pt;
  a.y_pt = p->d_y_pt; a.partial = p->d_partial;
  return a;
}

int build_reduced(ba_problem *p, bool explicit_blocks);

// Residuals, Jacobians, point elimination, station blocks, gradient, reduced right-hand side.
// Leaves cost in h_scal[SC_COST] (+ SC_FIXED) and the max-norm of the gradient in h_scal[SC_GMAX].
int linearize(ba_problem *p, const LmState &st) {
  hipStream_t s = p->stream;
  launch_station_prep(s, p->S, p->d_eop, p->d_scale_sta, p->d_sta_cam, p->d_sta);
  LinArgs la{};
  la.P = p->P; la.ncmax = p->ncmax; la.N = p->N; la.pt_start = p->d_pt_start; la.obs_pt = p->d_obs_pt; la.elim = p->d_elim;
  la.obs = p->d_obs; la.sta = p->d_sta; la.xyz = p->d_xyz;
  la.scale_pt = p->d_scale_pt; la.pt_prior = p->d_pt_prior; la.pt_isig = p->d_pt_isig; la.cam = p->d_cam;
  la.rec = p->d_rec; la.Jp = p->d_Jp; la.Jc = p->d_Jc; la.Mo = p->d_Mo; la.Vinv = p->d_Vinv; la.g_pt = p->d_g_pt;
  la.colnorm_pt = p->d_colnorm_pt; la.diag_pt = p->d_diag_pt; la.cost_partial = p->d_partial; la.flags = p->d_flags;
  la.huber_a = p->huber_a; la.model = p->model; la.emp = p->emp; la.lm_min = st.lm_min; la.lm_max = st.lm_max; la.radius = st.radius; la.reuse_diag = st.reuse_diag;
  la.Y = (st.reduced == 2) ? p->d_Y : nullptr;
  la.Jt = p->d_Jt;
  la.cam_partial = p->d_campart;
  // With the explicit block-sparse system and the border from the projected camera rows, nothing after this kernel
  // reads Jp or Jc: the back substitution and the model cost change work from Y and the elimination records.
  la.lean = (st.reduced == 2 && !st.full_rows && !st.need_norms && (p->d_Jt != nullptr || p->ncam == 0) && p->ncam <= cg2_max_cam()) ? 1 : 0;
  p->rows_lean = la.lean != 0;
  la.fused = (la.lean && p->fused_ok) ? 1 : 0;
  la.kp = p->kp;
  p->rows_fused = la.fused != 0;
  p->lin_radius = st.radius;
  launch_linearize(s, la, 1);
  if (p->sparse && !p->sparse_clear_queued) {
    // Zeros in the fronts of the sparse solver for the factorisation that follows this linearisation, on the second stream
    // beside the observation kernel: that kernel is bound by instruction issue and leaves most of the memory bandwidth
    // unused (beside the back substitution, which streams at the device's ceiling, the same 2 GB cost what they saved).
    if (!p->ev_sparse_clear) HIP_OK(hipEventCreateWithFlags(&p->ev_sparse_clear, hipEventDisableTiming));
    if (!p->stream_clear) HIP_OK(hipStreamCreateWithFlags(&p->stream_clear, hipStreamNonBlocking));
    HIP_OK(hipEventRecord(p->ev_sparse_clear, s));
    HIP_OK(hipStreamWaitEvent(p->stream_clear, p->ev_sparse_clear, 0));
    sparse_chol_clear(p->sparse, p->stream_clear);
    HIP_OK(hipEventRecord(p->ev_sparse_clear, p->stream_clear));
    p->sparse_clear_queued = true;
  }
  launch_linearize(s, la, 2);
  if (st.reduced == 2) {
    // the off-diagonal blocks only need the rows Y: assembled on the second stream while this one sweeps the stations
    // (stream2 is in order, so numbers of the preconditioner still being set up from the old blocks finish first)
    HIP_OK(hipEventRecord(p->ev_rows, s));
    HIP_OK(hipStreamWaitEvent(p->stream2, p->ev_rows, 0));
    HIP_OK(hipMemsetAsync(p->d_bsr_val, 0, 36 * (size_t)p->n_bsr * sizeof(double), p->stream2));
    if (p->sw.n_win > 0) launch_schur_windows(p->stream2, p->sw, p->d_Y, p->d_ublk_dst, p->d_bsr_val);
    else launch_schur_blocks(p->stream2, p->n_ublk, p->d_ublk_cstart, p->d_ublk_dst, p->d_pairs, p->d_Y, p->d_bsr_val);
    HIP_OK(hipEventRecord(p->ev_schur, p->stream2));
    p->schur_queued = true;
  }
  launch_sum_partials(s, p->d_partial, linearize_blocks(p->P), p->d_scal, SC_COST);
  // this rank's share of the station blocks and of the camera rows of the gradient and the reduced right-hand side
  if (la.fused) {
    // the observation kernel left them in partial records: station sums into F, the camera border into Ssc
    launch_keyed_fold(s, p->S, p->kp_ncolp, p->ncam, p->d_kp_sta_rec0, p->d_kp_sta_recs, p->kp.part, p->d_F, p->d_Ssc);
  } else {
    StationArgs sa = station_args(p, 6, 39);
    launch_station_pass(s, sa, SMODE_SETUP);
    launch_station_fold(s, p->S, p->d_sta_chunk0, p->d_part, 39, 39, p->d_F);
  }
  const int rows_blocks = la.fused ? keyed_blocks(p->kp.n_seg) : obs_rows_blocks(p->N);
  double *g_cam = p->d_g + 6 * (size_t)p->S, *b_cam = p->d_b + 6 * (size_t)p->S;
  if (p->ncam > 0) {
    if (p->d_Jt) {
      launch_camera_rows_fold(s, p->d_campart, rows_blocks, p->ncam, g_cam, b_cam);
    } else {
      // camera rows of the gradient and of the reduced right-hand side: one station sweep over the camera columns for both
      StationArgs sc2 = station_args(p, 6, 2 * BA_NCAMPAR);
      launch_station_pass(s, sc2, SMODE_CAM2);
      launch_camera_fold(s, p->C, p->n_chunks, p->d_chunk_sta, p->d_sta_cam, p->d_cam, p->d_part, 2 * BA_NCAMPAR, 0, g_cam);
      launch_camera_fold(s, p->C, p->n_chunks, p->d_chunk_sta, p->d_sta_cam, p->d_cam, p->d_part, 2 * BA_NCAMPAR, BA_NCAMPAR, b_cam);
    }
    if (st.need_norms) {
      StationArgs sn = station_args(p, 6, BA_NCAMPAR);
      launch_station_pass(s, sn, SMODE_CAMNORM);
      launch_camera_fold(s, p->C, p->n_chunks, p->d_chunk_sta, p->d_sta_cam, p->d_cam, p->d_part, BA_NCAMPAR, 0, p->d_colnorm_cam_in);
    }
  }
  if (p->ar) {
    // one sum over the ranks: the small arrays travel behind the station sums
    PackList pl{};
    pl.ptr[0] = g_cam; pl.n[0] = p->ncam;
    pl.ptr[1] = b_cam; pl.n[1] = p->ncam;
    pl.ptr[2] = p->d_colnorm_cam_in; pl.n[2] = (p->ncam > 0 && st.need_norms) ? p->ncam : 0;
    pl.ptr[3] = p->d_scal + SC_COST; pl.n[3] = 1;
    double *tail = p->d_F + 39 * (size_t)p->S;
    launch_pack_segments(s, pl, tail, 1);
    BA_TRY(p->allreduce(p->d_F, 39 * (int64_t)p->S + pl.n[0] + pl.n[1] + pl.n[2] + pl.n[3], 0));
    launch_pack_segments(s, pl, tail, 0);
  }
  launch_camera_prior_cost(s, p->C, p->d_cam, p->d_scal, SC_COST, SC_FIXED, nullptr);
  StationFinishArgs fa{};
  fa.S = p->S; fa.reuse_diag = st.reuse_diag; fa.sta_used = p->d_sta_used; fa.F = p->d_F;
  fa.lm_min = st.lm_min; fa.lm_max = st.lm_max; fa.radius = st.radius;
  fa.colnorm_sta = p->d_colnorm_sta; fa.diag_sta = p->d_diag_sta; fa.d2_sta = p->d_d2_sta; fa.g_sta = p->d_g; fa.b_sta = p->d_b;
  fa.Sdiag = p->d_Sdiag;
  launch_station_finish(s, fa);
  if (p->ncam > 0) {
    CamFinishArgs ca{};
    ca.C = p->C; ca.reuse_diag = st.reuse_diag; ca.need_norms = st.need_norms; ca.cam = p->d_cam;
    ca.colnorm_in = st.need_norms ? p->d_colnorm_cam_in : nullptr;
    ca.lm_min = st.lm_min; ca.lm_max = st.lm_max; ca.radius = st.radius;
    ca.jc2 = p->d_jc2; ca.colnorm_cam = p->d_colnorm_cam; ca.diag_cam = p->d_diag_cam; ca.d2_cam = p->d_d2_cam;
    ca.camdiag = p->d_diagfull + 6 * (size_t)p->S; ca.g_cam = g_cam; ca.b_cam = b_cam;
    launch_camera_finish(s, ca);
  }
  // gradient max norm: |x - Plus(x, -g_scaled / scale)| over points (local), stations and cameras (replicated).  The
  // flag of a point block that is not positive definite shows on the rank that owns the point only; it travels with
  // the maximum, so that every rank takes the same branch below (or the others would wait in the next collective
  // for ever)
  const int bp = cost_blocks(3 * (int64_t)p->P), bs = cost_blocks(6 * (int64_t)p->S);
  launch_absmax_ratio(s, 3 * (int64_t)p->P, p->d_g_pt, p->d_scale_pt, p->d_xyz, p->d_partial, bp);
  launch_max_partials(s, p->d_partial, bp, p->d_scal, SC_GMAX_P);
  if (p->ar) {
    launch_flag_to_scal(s, p->d_flags, p->d_scal, SC_NOTPD_P);
    BA_TRY(p->allreduce(p->d_scal + SC_GMAX_P, 2, 1));
  }
  launch_absmax_ratio(s, 6 * (int64_t)p->S, p->d_g, p->d_scale_sta, p->d_eop, p->d_partial + bp, bs);
  launch_max_partials(s, p->d_partial + bp, bs, p->d_scal, SC_GMAX_SC);
  if (p->ncam > 0) {
    launch_camera_gradient_step(s, p->C, p->d_cam, g_cam, p->d_scal, SC_GMAX);
  } else {
    HIP_OK(hipMemsetAsync(p->d_scal + SC_GMAX, 0, sizeof(double), s));
  }
  p->cam_info_pending = false;
  if (st.reduced) BA_TRY(build_reduced(p, st.reduced == 2));
  launch_flag_to_scal(s, p->d_flags, p->d_scal, SC_FLAGS);  // the flag word comes back with the scalars: one read-back
  BA_TRY(p->check_kernels());
  BA_TRY(p->read_scal());
  p->h_scal[SC_GMAX] = std::max(p->h_scal[SC_GMAX], std::max(p->h_scal[SC_GMAX_P], p->h_scal[SC_GMAX_SC]));
  // The factor of the camera block serves the PCG only (its preconditioner and coarse space): a break-down there is that
  // solver's failure -- an invalid step, as a linear solver that fails is to the trust-region loop -- and nothing to the
  // direct solvers, whose own factorisation of the whole reduced system finds the same columns.
  p->cam_block_failed = (p->cam_info_pending && p->h_scal[SC_CAMINFO] != 0.0) ? (int)p->h_scal[SC_CAMINFO] : 0;
  unsigned flag = (unsigned)p->h_scal[SC_FLAGS];
  if (p->ar && p->h_scal[SC_NOTPD_P] != 0.0 && !flag) flag = 1u;  // another rank's point block
  if (flag) {
    HIP_OK(hipMemset(p->d_flags, 0, 4 * sizeof(unsigned)));
    return fail(BA_ERR_NUMERIC, "%s block is not positive definite", (flag & 1) ? "a point" : "a station");
  }
  return 0;
}

// v = S e_k through the implicit operator: point sweep (with camera columns), station sweep
// (with camera rows), folds, sum over ranks, diagonal entry.  out has nr entries.
int probe_column(ba_problem *p, int64_t k, double *out) {
  hipStream_t s = p->stream;
  launch_unit(s, p->nr, p->d_p, k);
  PointArgs pa = point_args(p, p->d_p);
  launch_point_pass(s, pa, MODE_APPLY, p->ncam > 0);
  if (p->ncam > 0) {
    StationArgs sa = station_args(p, 7, 6 + BA_NCAMPAR);
    launch_station_pass(s, sa, SMODE_GATHER_CAM);
    launch_station_fold(s, p->S, p->d_sta_chunk0, p->d_part, 6 + BA_NCAMPAR, 6, out);
    launch_camera_fold(s, p->C, p->n_chunks, p->d_chunk_sta, p->d_sta_cam, p->d_cam, p->d_part, 6 + BA_NCAMPAR, 6, out + 6 * (size_t)p->S);
  } else {
    StationArgs sa = station_args(p, 7, 6);
    launch_station_pass(s, sa, SMODE_GATHER);
    launch_station_fold(s, p->S, p->d_sta_chunk0, p->d_part, 6, 6, out);
  }
  BA_TRY(p->allreduce(out, p->nr, 0));
  launch_add_diag_entry(s, out, p->d_diagfull, k);
  return 0;
}

// What the reduced solve needs beyond the right-hand side: the 6x6 station blocks (inverted
// for the block-Jacobi preconditioner), the dense camera border with its Cholesky factor and,
// when explicit_blocks, the off-diagonal station blocks in BSR form.
int build_reduced(ba_problem *p, bool explicit_blocks) {
  hipStream_t s = p->stream;
  const size_t ns = 6 * (size_t)p->S;
  // Several ranks: the blocks, the camera border and its corner sums lie one after another (alloc_reduced_sums) and are
  // summed in ONE reduction once the border sweeps are through.  One rank: the blocks are finished first, so that the
  // numbers of the multilevel preconditioner are refreshed on the second stream beside the border sweeps.
  const bool merged = p->ar && explicit_blocks && p->ncam > 0 && (p->d_Jt || p->d_Zc);
  // No hierarchy to refresh (dense solver, or the first solves of a PCG run): the border sweeps go first as well, so
  // that they run beside the block assembly instead of after it.
  const bool border_first = merged || (explicit_blocks && !p->amg && p->ncam > 0 && (p->d_Jt || p->d_Zc));
  bool blocks_pending = false;  // merged: the wait for the block assembly on the second stream comes right before the sum
  auto finish_blocks = [&]() -> int {
    launch_bsr_add_diag(s, p->S, p->d_bsr_rowptr, p->d_Sdiag, p->d_bsr_val);
    launch_invert_blocks(s, p->S, p->d_bsr_val, p->d_bsr_rowptr, p->d_sta_used, p->d_Minv, p->d_flags);
    if (p->amg) {
      // the hierarchy exists (a multilevel PCG solve has run): refresh its numbers on the second stream while this
      // one goes on with the camera border; solve_pcg_explicit waits for them
      HIP_OK(hipEventRecord(p->ev_blocks, s));
      HIP_OK(hipStreamWaitEvent(p->stream2, p->ev_blocks, 0));
      amg_numeric(p->amg, p->stream2, p->d_bsr_val, p->d_Minv);
      HIP_OK(hipEventRecord(p->ev_amg, p->stream2));
      p->amg_numbers_queued = true;
    }
    return 0;
  };
  auto sum_border = [&]() -> int {
    if (merged) {
      if (blocks_pending) HIP_OK(hipStreamWaitEvent(s, p->ev_schur, 0));
      BA_TRY(p->allreduce(p->d_redsum, (int64_t)(p->redsum_blocks + p->redsum_border), 0));
      return finish_blocks();
    }
    BA_TRY(p->allreduce(p->d_Ssc, (int64_t)p->redsum_border, 0));
    if (border_first) {
      if (blocks_pending) HIP_OK(hipStreamWaitEvent(s, p->ev_schur, 0));
      BA_TRY(p->allreduce(p->d_bsr_val, 36 * (int64_t)p->n_bsr, 0));
      return finish_blocks();
    }
    return 0;
  };
  if (explicit_blocks) {
    if (p->schur_queued) {
      // queued by linearize right after the rows were written
      if (border_first) blocks_pending = true;
      else HIP_OK(hipStreamWaitEvent(s, p->ev_schur, 0));
      p->schur_queued = false;
    } else {
      // numbers queued for the previous matrix and never waited for (the solve took another route) still read it
      if (p->amg_numbers_queued) HIP_OK(hipStreamWaitEvent(s, p->ev_amg, 0));
      HIP_OK(hipMemsetAsync(p->d_bsr_val, 0, 36 * (size_t)p->n_bsr * sizeof(double), s));
      if (p->sw.n_win > 0) launch_schur_windows(s, p->sw, p->d_Y, p->d_ublk_dst, p->d_bsr_val);
      else launch_schur_blocks(s, p->n_ublk, p->d_ublk_cstart, p->d_ublk_dst, p->d_pairs, p->d_Y, p->d_bsr_val);
    }
    if (!border_first) {
      BA_TRY(p->allreduce(p->d_bsr_val, 36 * (int64_t)p->n_bsr, 0));
      BA_TRY(finish_blocks());
    }
  } else {
    launch_invert_blocks(s, p->S, p->d_Sdiag, nullptr, p->d_sta_used, p->d_Minv, p->d_flags);
  }
  if (p->ncam == 0) return 0;
  if (p->d_Jt) {
    // one camera, few free parameters: the border comes out of one point sweep and one station sweep
    if (!p->rows_fused) {
      StationArgs sa = station_args(p, 6, border_nacc());
      launch_border_station(s, sa, p->d_Jt);
      launch_border_fold(s, p->S, p->ncam, p->n_chunks, p->d_sta_chunk0, p->d_part, border_nacc(), p->d_Ssc);
    }
    launch_camera_corner_fold(s, p->d_campart, p->rows_fused ? keyed_blocks(p->kp.n_seg) : obs_rows_blocks(p->N), p->d_corner);  // the sums come out of k_obs_rows
    BA_TRY(sum_border());
    launch_border_corner(s, p->ncam, p->d_corner, p->d_diagfull + ns, p->d_Scc);
  } else if (p->d_Zc) {
    // several cameras / many free parameters: one point sweep for Z = V^-1 B^T, one station sweep for the sums
    GenBorderArgs ga{};
    ga.P = p->P; ga.S = p->S; ga.ncam = p->ncam; ga.ncmax = p->ncmax; ga.n_chunks = p->n_chunks;
    ga.pt_start = p->d_pt_start; ga.obs_sta = p->d_obs_sta; ga.obs_pt = p->d_obs_pt; ga.sta_cam = p->d_sta_cam;
    ga.col_cam = p->d_col_cam; ga.col_pos = p->d_col_pos; ga.chunk_sta = p->d_chunk_sta; ga.chunk_begin = p->d_chunk_begin;
    ga.sta_list = p->d_sta_list; ga.sta_chunk0 = p->d_sta_chunk0; ga.rec = p->d_rec; ga.Jp = p->d_Jp; ga.Jc = p->d_Jc;
    ga.Vinv = p->d_Vinv; ga.cam = p->d_cam; ga.Zc = p->d_Zc; ga.out_sc = p->d_part; ga.out_cc = p->d_part_cc;
    ga.Ssc = p->d_Ssc; ga.corner = p->d_corner_gen;
    launch_border_general(s, ga);
    BA_TRY(sum_border());
    launch_border_general_corner(s, p->ncam, p->d_corner_gen, p->d_diagfull + ns, p->d_Scc);
  } else {
    // fallback: one pass of the implicit operator per camera column
    for (int k = 0; k < p->ncam; ++k) {
      BA_TRY(probe_column(p, (int64_t)ns + k, p->d_q));
      launch_store_border_col(s, p->S, p->ncam, k, p->d_q, p->d_Ssc, p->d_Scc);
    }
  }
  launch_copy(s, (int64_t)p->ncam * p->ncam, p->d_Scc, p->d_Lcc);
  launch_cholesky(s, p->d_Lcc, p->ncam, p->d_info);
  launch_chol_inverse(s, p->d_Lcc, p->ncam, p->d_Sccinv);  // the block-sparse PCG applies the camera block as a product
  // the factorisation's report travels with the scalars that linearize reads back: no wait of its own
  launch_flag_to_scal(s, reinterpret_cast<const unsigned *>(p->d_info), p->d_scal, SC_CAMINFO);
  p->cam_info_pending = true;
  return 0;
}

// Whether forming q = S p by rows pays on `world` ranks: the time of the product that the other ranks take over against
// the time of the sum that completes q.  The product streams 292 B per block and 96 B per station at the 4.8 TB/s the
// kernel reaches; the sum is a ring all-reduce over xGMI links of about 150 GB/s (2 (W-1)/W of the buffer crosses each
// link) plus 30 us for a small collective and for queueing the iteration launch by launch instead of as a graph.
// 8 ranks: 10 000 stations / 220 000 blocks: 12 us saved, 36 us spent -- every rank keeps the whole product;
// 50 000 stations / 1.1 M blocks: 58 us saved, 59 us spent -- the break-even; 100 000 stations: 116 against 87 -- by rows.
bool row_sharding_pays(const ba_problem *p, int nbs) {
  const double W = p->shard_world;
  const double t_product = (292.0 * (double)p->n_bsr + 96.0 * (double)p->S) / 4.8e12;
  const double bytes = 8.0 * (6.0 * (double)p->S + (double)nbs * (1 + p->ncam));
  const double t_sum = 30e-6 + 2.0 * (W - 1.0) / W * bytes / 150e9;
  return t_product * (1.0 - 1.0 / W) > t_sum;
}

// PCG on the explicit block-sparse system: the whole iteration stays on the device, the host
// only looks at the stopping flag once per batch.
int solve_pcg_explicit(ba_problem *p, const ba_options *opt, int *iters) {
  hipStream_t s = p->stream;
  if (p->cam_block_failed)
    return fail(BA_ERR_NUMERIC, "camera block of the reduced system is not positive definite (column %d)", p->cam_block_failed - 1);
  BsrCgArgs a{};
  a.S = p->S; a.ncam = p->ncam; a.rowptr = p->d_bsr_rowptr; a.col = p->d_bsr_col; a.val = p->d_bsr_val; a.Ssc = p->d_Ssc;
  a.Scc = p->d_Scc; a.Lcc = p->d_Sccinv; a.Minv = p->d_Minv; a.b = p->d_b; a.x = p->d_x; a.r = p->d_r; a.z = p->d_z; a.p = p->d_p;
  a.q = p->d_q; a.scal = p->d_cg2scal; a.part_pq = p->d_part_pq; a.part_qc = p->d_part_qc; a.part_rz = p->d_part_rz;
  a.r2 = p->d_r2;
  // multilevel preconditioner: structure once (it needs the values of a first linearisation),
  // numbers after every change of the matrix
  bool use_amg = opt->preconditioner == 0 && p->S > 1;
  if (use_amg) {
    if (!p->amg) {
      p->amg = amg_create();
      const double theta = tuning_value("amg_theta", 0.5);  // strength threshold of the aggregation
      if (amg_build(p->amg, s, p->S, p->bsr_rowptr_h, p->bsr_col_h, p->sta_used_h, p->d_bsr_rowptr, p->d_bsr_col, p->d_bsr_val,
                    p->d_Minv, theta) != 0) {
        const int rc = fail(BA_ERR_DEVICE, "%s", amg_error(p->amg));
        amg_destroy(p->amg);
        p->amg = nullptr;
        return rc;
      }
    } else if (p->amg_numbers_queued) {
      HIP_OK(hipStreamWaitEvent(s, p->ev_amg, 0));  // queued by build_reduced for this matrix
    } else {
      amg_numeric(p->amg, s, p->d_bsr_val, p->d_Minv);
    }
    p->amg_numbers_queued = false;
    if (amg_levels(p->amg) < 2) use_amg = false;  // nothing to coarsen: plain block-Jacobi
  }
  // Camera coarse space: the few camera columns couple to every station, which a block-diagonal
  // preconditioner cannot see.  With G ~ S_ss^-1 S_sc (a dozen PCG iterations per column, once per
  // problem: it only has to be roughly right) the vectors Q = [-G; I] are added as a coarse space,
  //   z = [M_ss^-1 r_s; 0] + Q (Q^T S Q)^-1 Q^T r,
  // which is symmetric positive definite for any G.
  bool use_defl = use_amg && p->ncam > 0 && p->ncam <= defl_max_cam() && !p->defl_failed;
  if (use_defl) {
    const size_t ns = 6 * (size_t)p->S;
    if (!p->d_G) {
      BA_TRY(p->dalloc(&p->d_G, ns * p->ncam));
      BA_TRY(p->dalloc(&p->d_GT, ns * p->ncam));
      BA_TRY(p->dalloc(&p->d_SG, ns * p->ncam));
      BA_TRY(p->dalloc(&p->d_defl_part, (size_t)defl_gram_blocks(p->S) * 64));
      BA_TRY(p->dalloc(&p->d_E, 64));
      BA_TRY(p->dalloc(&p->d_Lcc_defl, 64));
      BA_TRY(p->dalloc(&p->d_part_gr, ((size_t)cg2_blocks_vec8(p->S) + 1) * defl_max_cam()));
      BA_TRY(p->dalloc(&p->d_defl_b, ns));
    }
    if (!p->have_G) {
      BsrCgArgs g = a;
      g.ncam = 0;
      g.b = p->d_defl_b;
      for (int k = 0; k < p->ncam; ++k) {
        launch_defl_get_column(s, p->S, p->ncam, k, p->d_Ssc, p->d_defl_b);
        launch_amg_cg_init(s, g, p->amg, 0.0);
        for (int itg = 0; itg < 12; ++itg) launch_amg_cg_iteration(s, g, p->amg, itg & 1);
        launch_defl_set_column(s, p->S, p->ncam, k, p->d_x, p->d_G, p->d_GT);
      }
      p->have_G = true;
    }
    launch_defl_corner(s, a, p->d_G, p->d_SG, p->d_defl_part, p->d_E, p->d_Lcc_defl, p->d_info);
    // its factorisation's report rides in a free slot of the PCG's scalars and is looked at with the first stopping flag
    launch_flag_to_scal(s, reinterpret_cast<const unsigned *>(p->d_info), p->d_cg2scal, 7);
    a.G = p->d_G;
    a.GT = p->d_GT;
    a.Lcc = p->d_E;  // the explicit inverse
    a.part_gr = p->d_part_gr;
  }
  const int precond = use_amg ? (use_defl ? 2 : 0) : 1;
  // Several ranks: every rank runs the whole iteration on the summed system (all scalars and vectors are bitwise equal
  // on all of them, so no rank needs anything from another), unless the product q = S p is large enough to be worth
  // forming by rows: each rank then takes a slice of the station rows with an equal share of the blocks, leaves zeros
  // in the rest of q, and one sum over the ranks of [q | partial sums] completes it -- adding zeros is exact, so the
  // vectors stay equal on all ranks.  Such an iteration holds a collective and is queued launch by launch, not as a graph.
  const int nbs = cg2_blocks_spmv(p->S);
  const bool shard_rows = p->ar && p->shard_world > 1 && p->shard_mode >= 0 && use_amg && amg_cg_iteration_splits(a, p->amg) &&
                          (p->shard_mode > 0 || row_sharding_pays(p, nbs));
  const int64_t n_shard_buf = 6 * (int64_t)p->S + (int64_t)nbs * (1 + p->ncam);
  p->last_row0 = p->last_row_end = 0;
  if (shard_rows) {
    if (!p->shard_layout_checked) {
      // the buffer that the ranks sum is [q | one partial per workgroup of the product kernel]: its layout follows from
      // the device the rank runs on, so the ranks make sure once that they all have the same one
      double h[2] = {(double)nbs, -(double)nbs};
      HIP_OK(hipMemcpyAsync(p->d_cg2scal + 24, h, sizeof h, hipMemcpyHostToDevice, s));
      BA_TRY(p->allreduce(p->d_cg2scal + 24, 2, 1));
      HIP_OK(hipMemcpyAsync(h, p->d_cg2scal + 24, sizeof h, hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      if (h[0] != (double)nbs || h[1] != -(double)nbs)
        return fail(BA_ERR_COMM, "the ranks run on different devices (%d workgroups of the product kernel here, %d to %d over the ranks): "
                    "the product cannot be split by rows (ba_shard_reduced_solve mode -1 keeps it whole)", nbs, (int)-h[1], (int)h[0]);
      p->shard_layout_checked = true;
    }
    if (!p->d_qshard) BA_TRY(p->dalloc(&p->d_qshard, (size_t)n_shard_buf));
    a.q = p->d_qshard;
    a.part_pq = a.q + 6 * (size_t)p->S;
    a.part_qc = a.part_pq + nbs;
    const std::vector<int> &rp = p->bsr_rowptr_h;
    auto cut = [&](int r) {  // first row of rank r: the blocks before it are r / world of all
      if (r <= 0) return 0;
      if (r >= p->shard_world) return p->S;
      const int64_t target = (int64_t)rp[p->S] * r / p->shard_world;
      return (int)(std::lower_bound(rp.begin(), rp.begin() + p->S, (int)target) - rp.begin());
    };
    a.rows_sharded = 1;
    a.row0 = p->last_row0 = cut(p->shard_rank);
    a.row_end = p->last_row_end = cut(p->shard_rank + 1);
  }
  auto iteration = [&](int parity) -> int {
    if (shard_rows) {
      // the whole buffer: the partial sums too (an iteration that runs after convergence leaves them unwritten, and
      // what the previous reduction left there must not be summed again)
      HIP_OK(hipMemsetAsync(a.q, 0, (size_t)n_shard_buf * sizeof(double), s));
      launch_bsr_spmv(s, a, parity);
      BA_TRY(p->allreduce(a.q, n_shard_buf, 0));
      launch_amg_cg_iteration_after_product(s, a, p->amg, parity);
    } else if (use_amg) {
      launch_amg_cg_iteration(s, a, p->amg, parity);
    } else {
      launch_cg2_iteration(s, a, parity);
    }
    return 0;
  };
  if (use_amg) launch_amg_cg_init(s, a, p->amg, opt->pcg_rtol);
  else launch_cg2_init(s, a, opt->pcg_rtol);
  // Batches of iterations between two looks at the stopping flag (both sizes even, so that the parity pattern of a
  // batch repeats).  An iteration that runs after convergence still costs its launches and one load each, so the long
  // batch is only used while the previous solve of this problem says that it will be used up; the rest goes in short
  // batches (successive solves of one adjustment need almost the same number of iterations).
  const int batch = 32, tail = 8;
  if (p->cg_graph && p->cg_graph_precond != precond) {
    hipGraphExecDestroy(p->cg_graph);
    if (p->cg_graph_tail) hipGraphExecDestroy(p->cg_graph_tail);
    p->cg_graph = p->cg_graph_tail = nullptr;
  }
  if (tuning_value("pcg_graph", 1.0) == 0.0) p->cg_graph_failed = true;  // queue the iterations launch by launch (for comparisons)
  if (!p->cg_graph && !p->cg_graph_failed && !shard_rows) {
    p->cg_graph_precond = precond;
    // the launch sequence of a batch never changes for this problem: capture it into a graph once
    auto capture = [&](int n, hipGraphExec_t *exec) {
      hipGraph_t graph = nullptr;
      bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        for (int k = 0; k < n; ++k) (void)iteration(k & 1);
        ok = hipStreamEndCapture(s, &graph) == hipSuccess && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) hipGraphDestroy(graph);
      if (!ok) *exec = nullptr;
      return ok;
    };
    if (!capture(batch, &p->cg_graph) || !capture(tail, &p->cg_graph_tail)) {
      if (p->cg_graph) hipGraphExecDestroy(p->cg_graph);
      if (p->cg_graph_tail) hipGraphExecDestroy(p->cg_graph_tail);
      p->cg_graph = p->cg_graph_tail = nullptr;
      p->cg_graph_failed = true;
      (void)hipGetLastError();
    }
  }
  int it = 0;
  double done = 0.0;
  const int expect = p->last_pcg_iters;
  auto launch_batch = [&](int at, int n) -> int {
    if (!shard_rows && p->cg_graph && n == batch && (at & 1) == 0) {
      HIP_OK(hipGraphLaunch(p->cg_graph, s));
    } else if (!shard_rows && p->cg_graph_tail && n == tail && (at & 1) == 0) {
      HIP_OK(hipGraphLaunch(p->cg_graph_tail, s));
    } else {
      for (int k = 0; k < n; ++k) BA_TRY(iteration((at + k) & 1));
    }
    return 0;
  };
  // What the previous solve says will be needed is queued without a look at the flag in between (the device is never
  // left waiting for the host's answer and the next launch); only the last few iterations go one short batch at a time.
  int queued = 0;
  if (expect > 0 && ((p->cg_graph && p->cg_graph_tail) || shard_rows || p->cg_graph_failed)) {
    const int sure = std::min(expect - tail / 2, opt->pcg_max_iters);
    while (queued + batch <= sure) {
      BA_TRY(launch_batch(queued, batch));
      queued += batch;
    }
    while (queued + tail <= sure) {
      BA_TRY(launch_batch(queued, tail));
      queued += tail;
    }
  }
  for (;;) {
    HIP_OK(hipMemcpyAsync(p->h_cgscal, p->d_cg2scal, 10 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    if (use_defl && p->h_cgscal[7] != 0.0) {
      // cannot happen for a positive definite S; be safe: drop the coarse space and start the solve again
      p->defl_failed = true;
      return solve_pcg_explicit(p, opt, iters);
    }
    it = (int)p->h_cgscal[3];
    done = std::max(p->h_cgscal[8], p->h_cgscal[9]);
    if (done != 0.0 || it >= opt->pcg_max_iters) break;
    const bool long_batch = expect <= 0 || it + batch <= expect - tail / 2;
    const int n = std::min(long_batch ? batch : tail, opt->pcg_max_iters - it);
    BA_TRY(launch_batch(it, n));
  }
  p->last_pcg_iters = it;
  *iters = it;
  BA_TRY(p->check_kernels());
  if (done == 2.0) return fail(BA_ERR_NUMERIC, "PCG breakdown at iteration %d (p.Sp = %g)", it, p->h_cgscal[4]);
  if (!std::isfinite(p->h_cgscal[2])) return fail(BA_ERR_NUMERIC, "PCG: non-finite right-hand side");
  return 0;
}

// block-Jacobi preconditioned conjugate gradients on the reduced system; solution in d_x
int solve_pcg(ba_problem *p, const ba_options *opt, int *iters) {
  hipStream_t s = p->stream;
  if (p->cam_block_failed)
    return fail(BA_ERR_NUMERIC, "camera block of the reduced system is not positive definite (column %d)", p->cam_block_failed - 1);
  CgArgs ca{};
  ca.n = p->nr; ca.S = p->S; ca.ncam = p->ncam; ca.b = p->d_b; ca.Minv = p->d_Minv; ca.Lcc = p->d_Lcc; ca.Ssc = p->d_Ssc;
  ca.Scc = p->d_Scc; ca.d2_sta = p->d_d2_sta; ca.x = p->d_x; ca.r = p->d_r; ca.z = p->d_z; ca.p = p->d_p; ca.q = p->d_q;
  ca.scal = p->d_cgscal;
  launch_cg_init(s, ca);
  HIP_OK(hipMemcpyAsync(p->h_cgscal, p->d_cgscal, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  const double rz0 = p->h_cgscal[1];
  int it = 0;
  if (!(rz0 > 0.0)) {
    *iters = 0;
    return std::isfinite(rz0) ? 0 : fail(BA_ERR_NUMERIC, "PCG: non-finite right-hand side");
  }
  const double target = opt->pcg_rtol * std::sqrt(rz0);
  StationArgs sa = station_args(p, 7, 6);
  PointArgs pa = point_args(p, p->d_p);
  while (it < opt->pcg_max_iters) {
    launch_point_pass(s, pa, MODE_APPLY, false);
    launch_station_pass(s, sa, SMODE_GATHER);
    launch_station_fold(s, p->S, p->d_sta_chunk0, p->d_part, 6, 6, p->d_q);
    BA_TRY(p->allreduce(p->d_q, 6 * (int64_t)p->S, 0));
    launch_cg_step(s, ca);
    HIP_OK(hipMemcpyAsync(p->h_cgscal, p->d_cgscal, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    ++it;
    const double rz = p->h_cgscal[0], pq = p->h_cgscal[3];
    if (!std::isfinite(rz) || !(pq > 0.0)) {
      *iters = it;
      return fail(BA_ERR_NUMERIC, "PCG breakdown at iteration %d (p.Sp = %g)", it, pq);
    }
    if (std::sqrt(std::max(rz, 0.0)) <= target) break;
  }
  *iters = it;
  return p->check_kernels();
}

// dense reduced matrix (scattered from the block-sparse form, or one implicit-operator probe per
// column when that form is not available), Cholesky on the matrix cores
int build_dense(ba_problem *p, bool explicit_blocks) {
  hipStream_t s = p->stream;
  const size_t nr = (size_t)p->nr;
  if (!p->d_Sdense) {
    p->dense_blocked = p->nr >= chol_blocked_min_order();
    p->dense_ld = p->dense_blocked ? chol_blocked_ld(p->nr) : p->nr;
    const size_t ld = (size_t)p->dense_ld;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if ((double)nr * ld * 8.0 > 0.9 * (double)free_b)
      return fail(BA_ERR_INVALID, "dense reduced system of order %zu (%.1f GB) does not fit the device memory that is free (%.1f GB); use the PCG solver",
                  nr, (double)nr * ld * 8e-9, (double)free_b * 1e-9);
    BA_TRY(p->dalloc(&p->d_Sdense, nr * ld));
    if (p->dense_blocked) BA_TRY(p->dalloc(&p->d_cholW, chol_blocked_workspace(p->nr)));
  }
  const size_t ld = (size_t)p->dense_ld;
  if (explicit_blocks) {
    HIP_OK(hipMemsetAsync(p->d_Sdense, 0, nr * ld * sizeof(double), s));
    launch_bsr_to_dense(s, p->S, p->ncam, p->d_bsr_rowptr, p->d_bsr_col, p->d_bsr_val, p->d_Ssc, p->d_Scc, p->d_Sdense, p->dense_ld);
  } else {
    if (ld != nr) HIP_OK(hipMemsetAsync(p->d_Sdense, 0, nr * ld * sizeof(double), s));  // zeros in the padding
    for (size_t k = 0; k < nr; ++k) BA_TRY(probe_column(p, (int64_t)k, p->d_Sdense + k * ld));
  }
  return 0;
}

// Cholesky of the dense reduced matrix in place; info != 0: not positive definite
int factor_dense(ba_problem *p) {
  if (p->dense_blocked) launch_cholesky_blocked(p->stream, p->d_Sdense, p->nr, p->dense_ld, p->d_cholW, p->d_info);
  else launch_cholesky(p->stream, p->d_Sdense, p->nr, p->d_info);
  return 0;
}

int solve_dense(ba_problem *p, bool explicit_blocks, bool defer_report = false) {  // defer_report: as solve_sparse
  hipStream_t s = p->stream;
  BA_TRY(build_dense(p, explicit_blocks));
  BA_TRY(factor_dense(p));
  launch_copy(s, p->nr, p->d_b, p->d_x);
  if (p->dense_blocked) launch_chol_solve_blocked(s, p->d_Sdense, p->nr, p->dense_ld, p->d_cholW, p->d_x);
  else launch_chol_solve(s, p->d_Sdense, p->nr, p->d_x);
  if (defer_report) {
    launch_flag_to_scal(s, reinterpret_cast<const unsigned *>(p->d_info), p->d_scal, SC_SPINFO);
    return p->check_kernels();
  }
  int info = 0;
  HIP_OK(hipMemcpyAsync(&info, p->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  BA_TRY(p->check_kernels());
  if (info) return fail(BA_ERR_NUMERIC, "reduced Cholesky failed at column %d of %d", info - 1, p->nr);
  return 0;
}

// Sparse direct solve of the block-sparse reduced system (its blocks carry the LM diagonal already).  defer_report: the
// factorisation's report is left in the scalar SC_SPINFO for the caller's next read-back instead of being waited for
// here (a failed factorisation leaves a step that is not finite, which the caller tests anyway).  What the wait cost was
// the copy of one int into pageable host memory: ~0.1 ms per LM iteration.
int solve_sparse(ba_problem *p, bool defer_report = false) {
  hipStream_t s = p->stream;
  if (!p->sparse) {
    p->sparse = sparse_chol_create();
    std::string why;
    int leaf = 63;  // stations per leaf: 378 unknowns, three panels
    const int rc = sparse_chol_analyse(p->sparse, p->S, p->ncam, p->bsr_rowptr_h, p->bsr_col_h, p->sta_used_h, leaf, why);
    if (rc) {
      sparse_chol_destroy(p->sparse);
      p->sparse = nullptr;
      return fail(rc == 1 ? BA_ERR_UNSUPPORTED : BA_ERR_DEVICE, "sparse Cholesky of the reduced system: %s", why.c_str());
    }
  }
  if (p->sparse_clear_queued) {
    HIP_OK(hipStreamWaitEvent(s, p->ev_sparse_clear, 0));  // the fronts were zeroed on the second stream after the last solve
    p->sparse_clear_queued = false;
  }
  sparse_chol_factor(p->sparse, s, p->d_bsr_val, p->d_Ssc, p->d_Scc, p->d_info);
  HIP_OK(hipMemsetAsync(p->d_x, 0, (size_t)p->nr * sizeof(double), s));  // stations outside the program keep a zero step
  sparse_chol_solve(p->sparse, s, p->d_b, p->d_x);
  if (defer_report) {
    launch_flag_to_scal(s, reinterpret_cast<const unsigned *>(p->d_info), p->d_scal, SC_SPINFO);
    return p->check_kernels();
  }
  int info = 0;
  HIP_OK(hipMemcpyAsync(&info, p->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  BA_TRY(p->check_kernels());
  if (info) return fail(BA_ERR_NUMERIC, "sparse Cholesky of the reduced system: a pivot is not positive");
  return 0;
}

int make_candidate(ba_problem *p, double alpha, bool reduce = true);
int cost_local(ba_problem *p, const double *eop, const double *xyz, const CamRec *cam, double *sta_buf, int slot, int apply_loss,
               double *res_out);

// y_p = V^-1 (g_p - W^T y_c); model cost change; candidate = x - y*scale; norms; with_cost: the cost at the candidate as
// well (slot SC_CCOST), so that the phase has ONE sum over the ranks: model cost change, step and candidate norms of the
// points, candidate cost -- the host decides afterwards whether the step is valid at all.  Reads d_x.
int step_and_candidate(ba_problem *p, bool with_cost) {
  hipStream_t s = p->stream;
  const size_t ns = 6 * (size_t)p->S;
  PointArgs pa = point_args(p, p->d_x);
  if (p->rows_lean) {
    pa.Y = p->d_Y; pa.elim = p->d_elim; pa.diag_pt = p->d_diag_pt; pa.radius = p->lin_radius; pa.ncam = p->ncam;
    launch_backsub_y(s, pa);
    launch_sum_partials(s, p->d_partial, linearize_blocks(p->P), p->d_scal, SC_MCC);
    // y_c^T S y_c with the kernels of the PCG (the system is replicated on every rank)
    BsrCgArgs a{};
    a.S = p->S; a.ncam = p->ncam; a.rowptr = p->d_bsr_rowptr; a.col = p->d_bsr_col; a.val = p->d_bsr_val; a.Ssc = p->d_Ssc;
    a.Scc = p->d_Scc; a.p = p->d_x; a.q = p->d_q; a.scal = p->d_zero; a.part_pq = p->d_part_pq; a.part_qc = p->d_part_qc;
    launch_quadratic_form(s, a, p->d_ytSy);
  } else {
    launch_point_pass(s, pa, MODE_BACKSUB, p->ncam > 0);
    launch_sum_partials(s, p->d_partial, linearize_blocks(p->P), p->d_scal, SC_MCC);
  }
  BA_TRY(make_candidate(p, 1.0, false));
  if (with_cost) BA_TRY(cost_local(p, p->d_eop_c, p->d_xyz_c, p->d_cam_c, p->d_sta_c, SC_CCOST, 0, nullptr));
  static_assert(SC_STEP2_P == SC_MCC + 1 && SC_CNORM2_P == SC_MCC + 2 && SC_CCOST == SC_MCC + 3, "slots summed together");
  BA_TRY(p->allreduce(p->d_scal + SC_MCC, with_cost ? 4 : 3, 0));
  // what every rank holds in full comes on top after the sum
  if (p->rows_lean) {
    launch_mcc_reduced(s, (int64_t)ns, p->d_g, p->d_x, p->d_d2_sta, p->d_ytSy, p->d_scal, SC_MCC);
    if (p->ncam > 0) launch_mcc_reduced(s, p->ncam, p->d_g + ns, p->d_x + ns, p->d_d2_cam, nullptr, p->d_scal, SC_MCC);
  } else {
    launch_camera_mcc(s, p->C, p->d_cam, p->d_x + ns, p->d_scal, SC_MCC);
  }
  if (with_cost) launch_camera_prior_cost(s, p->C, p->d_cam_c, p->d_scal, SC_CCOST, -1, nullptr);
  return p->check_kernels();
}

// candidate = Plus(x, alpha * delta), its norm and the norm of the step taken
int make_candidate(ba_problem *p, double alpha, bool reduce) {
  hipStream_t s = p->stream;
  CandArgs ca{};
  ca.alpha = alpha;
  ca.S = p->S; ca.P = p->P; ca.C = p->C; ca.sta_used = p->d_sta_used; ca.eop = p->d_eop; ca.xyz = p->d_xyz; ca.y_cam = p->d_x;
  ca.y_pt = p->d_y_pt; ca.scale_sta = p->d_scale_sta; ca.scale_pt = p->d_scale_pt; ca.cam = p->d_cam;
  ca.eop_c = p->d_eop_c; ca.xyz_c = p->d_xyz_c; ca.cam_c = p->d_cam_c; ca.partial = p->d_partial;
  const int nb = cost_blocks(6 * (int64_t)p->S + 3 * (int64_t)p->P);
  launch_candidate(s, ca, nb);
  launch_sum_partials(s, p->d_partial, nb, p->d_scal, SC_STEP2_SC);
  launch_sum_partials(s, p->d_partial + nb, nb, p->d_scal, SC_CNORM2_SC);
  launch_sum_partials(s, p->d_partial + 2 * nb, nb, p->d_scal, SC_STEP2_P);
  launch_sum_partials(s, p->d_partial + 3 * nb, nb, p->d_scal, SC_CNORM2_P);
  if (reduce) BA_TRY(p->allreduce(p->d_scal + SC_STEP2_P, 2, 0));
  return p->check_kernels();
}

// this rank's share of the cost (image rows and point priors of its own points) into d_scal[slot]
int cost_local(ba_problem *p, const double *eop, const double *xyz, const CamRec *cam, double *sta_buf, int slot, int apply_loss,
               double *res_out) {
  hipStream_t s = p->stream;
  launch_station_prep(s, p->S, eop, p->d_scale_sta, p->d_sta_cam, sta_buf);
  CostArgs ca{};
  ca.N = p->N; ca.obs = p->d_obs; ca.sta = sta_buf; ca.xyz = xyz; ca.obs_pt = p->d_obs_pt; ca.cam = cam; ca.huber_a = p->huber_a; ca.model = p->model;
  ca.emp = p->emp || (cam != p->d_cam && p->emp_c);  // candidates made on the device share the resident set's terms
  ca.apply_loss = apply_loss; ca.res_out = res_out; ca.partial = p->d_partial;
  ca.one_cam = p->C == 1;
  const int b1 = cost_blocks(p->N), b2 = cost_blocks(3 * (int64_t)p->P);
  launch_cost_obs(s, ca, b1);
  launch_cost_prior(s, p->P, xyz, p->d_pt_prior, p->d_pt_isig, p->d_pt_file, res_out ? res_out + 2 * (size_t)p->N : nullptr,
                    p->d_partial + b1, b2);
  launch_sum_partials(s, p->d_partial, b1 + b2, p->d_scal, slot);
  return 0;
}

int cost_at(ba_problem *p, const double *eop, const double *xyz, const CamRec *cam, double *sta_buf, int slot, int fixed_slot,
            int apply_loss, double *res_out) {
  BA_TRY(cost_local(p, eop, xyz, cam, sta_buf, slot, apply_loss, res_out));
  BA_TRY(p->allreduce(p->d_scal + slot, 1, 0));
  launch_camera_prior_cost(p->stream, p->C, cam, p->d_scal, slot, fixed_slot,
                           res_out ? res_out + 2 * (size_t)p->N + 3 * (size_t)p->P : nullptr);
  return p->check_kernels();
}

int xnorm(ba_problem *p, double *out) {
  hipStream_t s = p->stream;
  CandArgs ca{};
  ca.S = p->S; ca.P = p->P; ca.C = p->C; ca.sta_used = p->d_sta_used; ca.eop = p->d_eop; ca.xyz = p->d_xyz; ca.cam = p->d_cam;
  ca.partial = p->d_partial;
  const int nb = cost_blocks(6 * (int64_t)p->S + 3 * (int64_t)p->P);
  launch_xnorm(s, ca, nb);
  launch_sum_partials(s, p->d_partial, nb, p->d_scal, SC_XNORM2_SC);
  launch_sum_partials(s, p->d_partial + nb, nb, p->d_scal, SC_XNORM2_P);
  BA_TRY(p->allreduce(p->d_scal + SC_XNORM2_P, 1, 0));
  BA_TRY(p->read_scal());
  *out = std::sqrt(p->h_scal[SC_XNORM2_SC] + p->h_scal[SC_XNORM2_P]);
  return 0;
}

// Jacobi scaling from the column norms of the unscaled Jacobian at the start point
// [Ceres 1.13 TrustRegionMinimizer::EvaluateGradientAndJacobian, iteration 0]
int init_scaling(ba_problem *p, const ba_options *opt) {
  hipStream_t s = p->stream;
  launch_fill(s, 3 * (int64_t)p->P, p->d_scale_pt, 1.0);
  launch_fill(s, 6 * (int64_t)p->S, p->d_scale_sta, 1.0);
  launch_fill(s, p->ncam, p->d_scale_cam, 1.0);
  for (auto &c : p->cam_h)
    for (int k = 0; k < BA_NCAMPAR; ++k) c.scale[k] = 1.0;
  if (p->C) HIP_OK(hipMemcpyAsync(p->d_cam, p->cam_h.data(), p->cam_h.size() * sizeof(CamRec), hipMemcpyHostToDevice, s));
  HIP_OK(hipStreamSynchronize(s));
  if (!opt->jacobi_scaling) return 0;
  LmState st{opt->initial_trust_region_radius, opt->min_lm_diagonal, opt->max_lm_diagonal, false, true, 0};
  BA_TRY(linearize(p, st));
  launch_make_scale(s, 3 * (int64_t)p->P, p->d_colnorm_pt, p->d_scale_pt);
  launch_make_scale(s, 6 * (int64_t)p->S, p->d_colnorm_sta, p->d_scale_sta);
  if (p->ncam > 0) {
    launch_make_scale(s, p->ncam, p->d_colnorm_cam, p->d_scale_cam);
    std::vector<double> sc(p->ncam);
    HIP_OK(hipMemcpyAsync(sc.data(), p->d_scale_cam, sc.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    for (auto &c : p->cam_h)
      for (int k = 0; k < c.nc; ++k) c.scale[k] = sc[c.off + k];
    HIP_OK(hipMemcpyAsync(p->d_cam, p->cam_h.data(), p->cam_h.size() * sizeof(CamRec), hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));
  }
  return p->check_kernels();
}

void fill_sizes(const ba_problem *p, ba_summary *sm) {
  // [Ceres 1.13] reduced program: constant and unreferenced parameter blocks are removed,
  // and with them the residual blocks that depend on constant blocks only
  int nsu = 0;
  for (int i = 0; i < p->S; ++i) nsu += p->sta_used_h[i];
  int npar = 6 * nsu + 3 * p->P, nres = 2 * p->N + 3 * p->P;
  for (int c = 0; c < p->C; ++c) {
    if (p->cam_h[c].in_program & 1) { npar += 3; nres += 3; }
    if (p->cam_h[c].in_program & 2) npar += 16;
  }
  sm->num_parameters = 6 * p->S + 3 * p->P + 19 * p->C;
  sm->num_residuals = 2 * p->N + 3 * p->P + 3 * p->C;
  sm->num_parameters_reduced = npar;
  sm->num_effective_parameters_reduced = 6 * nsu + 3 * p->P + p->ncam;
  sm->num_residuals_reduced = nres;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *ba_last_error(void) { return g_err.c_str(); }

void ba_default_options(ba_options *o) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->min_lm_diagonal = 1.0E-150;  // src/main.cpp:4713
  o->max_lm_diagonal = 1.0E-150;  // src/main.cpp:4712
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->max_num_iterations = 100;  // src/main.cpp:4718
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->linear_solver = BA_SOLVER_AUTO;
  o->pcg_rtol = 1e-12;
  o->pcg_max_iters = 2000;
}

int ba_create(const ba_dims *dims, const ba_inputs *in, ba_problem **out) {
  if (!dims || !in || !out) return fail(BA_ERR_INVALID, "ba_create: null argument");
  int ndev = 0;
  const hipError_t de = hipGetDeviceCount(&ndev);
  if (de != hipSuccess || ndev < 1)
    return fail(BA_ERR_DEVICE, "no HIP device is visible (%s, %d devices): this library has no CPU path",
                hipGetErrorString(de), ndev);
  ba_problem *p = new ba_problem();
  const int rc = build_problem(p, dims, in);
  if (rc) {
    ba_destroy(p);
    return rc;
  }
  *out = p;
  return 0;
}

void ba_destroy(ba_problem *p) {
  if (!p) return;
  if (p->stream) hipStreamSynchronize(p->stream);
  if (p->stream2) hipStreamSynchronize(p->stream2);
  if (p->stream_clear) hipStreamSynchronize(p->stream_clear);
  if (p->comm) ncclCommDestroy(p->comm);
  if (p->cg_graph) hipGraphExecDestroy(p->cg_graph);
  if (p->cg_graph_tail) hipGraphExecDestroy(p->cg_graph_tail);
  if (p->amg) ba::amg_destroy(p->amg);
  if (p->sparse) ba::sparse_chol_destroy(p->sparse);
  if (p->stream_clear) {
    hipStreamSynchronize(p->stream_clear);
    hipStreamDestroy(p->stream_clear);
  }
  if (p->ev_sparse_clear) hipEventDestroy(p->ev_sparse_clear);
  for (void *q : p->allocs) hipFree(q);
  if (p->h_scal) hipHostFree(p->h_scal);
  if (p->h_cgscal) hipHostFree(p->h_cgscal);
  for (auto &e : p->reduction_events) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (p->ev_blocks) hipEventDestroy(p->ev_blocks);
  if (p->ev_amg) hipEventDestroy(p->ev_amg);
  if (p->ev_rows) hipEventDestroy(p->ev_rows);
  if (p->ev_schur) hipEventDestroy(p->ev_schur);
  if (p->stream2) hipStreamDestroy(p->stream2);
  if (p->stream) hipStreamDestroy(p->stream);
  delete p;
}

int ba_set_parameters(ba_problem *p, const double *eop, const double *xyz, const double *iop, const double *ap) {
  if (!p) return fail(BA_ERR_INVALID, "ba_set_parameters: null problem");
  return set_parameters(p, eop, xyz, iop, ap, false);
}

int ba_set_corrections(ba_problem *p, const double *obs_cx, const double *obs_cy) {
  if (!p) return fail(BA_ERR_INVALID, "ba_set_corrections: null problem");
  if (p->N > 0 && (!obs_cx || !obs_cy)) return fail(BA_ERR_INVALID, "ba_set_corrections: null array");
  if (p->N == 0) return 0;
  for (int n = 0; n < p->N; ++n)
    if (!std::isfinite(obs_cx[n]) || !std::isfinite(obs_cy[n]))
      return fail(BA_ERR_INVALID, "ba_set_corrections: observation %d has a correction that is not finite", n);
  double *d_c = nullptr;
  const size_t bytes = (size_t)p->N * sizeof(double);
  if (hipMalloc((void **)&d_c, 2 * bytes) != hipSuccess) return fail(BA_ERR_DEVICE, "ba_set_corrections: hipMalloc of %zu bytes failed", 2 * bytes);
  int rc = 0;
  if (hipMemcpyAsync(d_c, obs_cx, bytes, hipMemcpyHostToDevice, p->stream) != hipSuccess ||
      hipMemcpyAsync(d_c + p->N, obs_cy, bytes, hipMemcpyHostToDevice, p->stream) != hipSuccess)
    rc = fail(BA_ERR_DEVICE, "ba_set_corrections: upload failed");
  if (!rc) {
    launch_set_corrections(p->stream, p->N, p->d_obs, d_c, d_c + p->N);
    rc = p->check_kernels();
  }
  if (hipStreamSynchronize(p->stream) != hipSuccess && !rc) rc = fail(BA_ERR_DEVICE, "ba_set_corrections: kernel failed");
  (void)hipFree(d_c);
  return rc;
}

int ba_set_iop_bounds(ba_problem *p, const double *lower, const double *upper) {
  if (!p) return fail(BA_ERR_INVALID, "ba_set_iop_bounds: null problem");
  std::vector<double> lo(3 * (size_t)p->C, -DBL_MAX), hi(3 * (size_t)p->C, DBL_MAX);
  bool boxed = false, constrained = false;
  for (int c = 0; c < p->C; ++c)
    for (int k = 0; k < 3; ++k) {
      double &l = lo[3 * c + k], &h = hi[3 * c + k];
      if (lower) l = lower[3 * c + k];
      if (upper) h = upper[3 * c + k];
      if (std::isnan(l) || std::isnan(h)) return fail(BA_ERR_INVALID, "ba_set_iop_bounds: camera %d, IOP[%d]: bound is not a number", c, k);
      l = std::max(l, -DBL_MAX);  // -inf / +inf: no bound
      h = std::min(h, DBL_MAX);
      const bool finite = l > -DBL_MAX || h < DBL_MAX;
      boxed = boxed || finite;
      if (p->cam_h[c].in_program & 1) {
        // [Ceres 1.13 Program::IsFeasible] a variable block needs a box with an interior
        if (l >= h) return fail(BA_ERR_INVALID, "ba_set_iop_bounds: camera %d, IOP[%d]: empty interval [%g, %g]", c, k, l, h);
        constrained = constrained || finite;  // [Program::IsBoundsConstrained]
      }
    }
  for (int c = 0; c < p->C; ++c)
    for (int k = 0; k < 3; ++k) {
      p->cam_h[c].lo[k] = lo[3 * c + k];
      p->cam_h[c].hi[k] = hi[3 * c + k];
    }
  p->boxed = boxed;
  p->constrained = constrained;
  return 0;
}

int ba_get_parameters(ba_problem *p, double *eop, double *xyz, double *iop, double *ap) {
  if (!p) return fail(BA_ERR_INVALID, "ba_get_parameters: null problem");
  if (!p->have_params) return fail(BA_ERR_INVALID, "ba_get_parameters: no parameters were set");
  HIP_OK(hipStreamSynchronize(p->stream));
  if (eop && p->S) HIP_OK(hipMemcpy(eop, p->d_eop, 6 * (size_t)p->S * sizeof(double), hipMemcpyDeviceToHost));
  if (xyz && p->P) {
    std::vector<double> x(3 * (size_t)p->P);
    HIP_OK(hipMemcpy(x.data(), p->d_xyz, x.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int jj = 0; jj < p->P; ++jj)
      for (int k = 0; k < 3; ++k) xyz[3 * (size_t)p->pt_file[jj] + k] = x[3 * (size_t)jj + k];
  }
  if ((iop || ap) && p->C) {
    std::vector<CamRec> cam(p->C);
    HIP_OK(hipMemcpy(cam.data(), p->d_cam, cam.size() * sizeof(CamRec), hipMemcpyDeviceToHost));
    for (int c = 0; c < p->C; ++c) {
      if (iop) for (int k = 0; k < 3; ++k) iop[3 * c + k] = cam[c].iop[k];
      if (ap) for (int k = 0; k < 16; ++k) ap[16 * c + k] = cam[c].ap[k];
    }
  }
  return 0;
}

// g . delta for the gradient now on the device and the step of the last solve: delta = -y * scale and
// g = g_scaled / scale, so the scales cancel
static int gradient_dot_step(ba_problem *p, double *out) {
  hipStream_t s = p->stream;
  const int bp = cost_blocks(3 * (int64_t)p->P), bs = cost_blocks(p->nr);
  launch_dot(s, 3 * (int64_t)p->P, p->d_g_pt, p->d_y_pt, p->d_partial, bp);
  launch_sum_partials(s, p->d_partial, bp, p->d_scal, SC_LS_DOT_P);
  BA_TRY(p->allreduce(p->d_scal + SC_LS_DOT_P, 1, 0));
  launch_dot(s, p->nr, p->d_g, p->d_x, p->d_partial + bp, bs);
  launch_sum_partials(s, p->d_partial + bp, bs, p->d_scal, SC_LS_DOT_SC);
  BA_TRY(p->check_kernels());
  BA_TRY(p->read_scal());
  *out = -(p->h_scal[SC_LS_DOT_P] + p->h_scal[SC_LS_DOT_SC]);
  return 0;
}

// max |delta|  [Ceres 1.13 LineSearchFunction::DirectionInfinityNorm]
static int step_max_norm(ba_problem *p, double *out) {
  hipStream_t s = p->stream;
  const int bp = cost_blocks(3 * (int64_t)p->P), bs = cost_blocks(6 * (int64_t)p->S);
  launch_absmax_prod(s, 3 * (int64_t)p->P, p->d_y_pt, p->d_scale_pt, p->d_partial, bp);
  launch_max_partials(s, p->d_partial, bp, p->d_scal, SC_LS_MAX_P);
  BA_TRY(p->allreduce(p->d_scal + SC_LS_MAX_P, 1, 1));
  launch_absmax_prod(s, 6 * (int64_t)p->S, p->d_x, p->d_scale_sta, p->d_partial + bp, bs);
  if (p->ncam > 0) launch_absmax_prod(s, p->ncam, p->d_x + 6 * (size_t)p->S, p->d_scale_cam, p->d_partial + bp + bs, 1);
  launch_max_partials(s, p->d_partial + bp, bs + (p->ncam > 0 ? 1 : 0), p->d_scal, SC_LS_MAX_SC);
  BA_TRY(p->check_kernels());
  BA_TRY(p->read_scal());
  *out = std::max(p->h_scal[SC_LS_MAX_P], p->h_scal[SC_LS_MAX_SC]);
  return 0;
}

// [Ceres 1.13 TrustRegionMinimizer::DoLineSearch + ArmijoLineSearch::DoSearch] Backtracking along
// a -> Plus(x, a * delta) from a = 1 until cost(a) <= cost(0) + 1e-4 * a * g.delta; every trial is a full
// evaluation (cost and gradient, for the cubic interpolation).  Leaves the candidate buffers at the accepted
// step, or at the full step when the search gives up, and the row buffers at the last trial point.
static int projected_line_search(ba_problem *p, const LmState &st, double x_cost) {
  const double sufficient_decrease = 1e-4, max_contraction = 1e-3, min_contraction = 0.6, min_step = 1e-9;
  const int max_trials = 20;
  ba_ls_sample start{0.0, x_cost, 0.0, 1, 1}, previous{0.0, 0.0, 0.0, 0, 0}, current{1.0, 0.0, 0.0, 0, 0};
  BA_TRY(gradient_dot_step(p, &start.gradient));
  LmState trial = st;
  trial.reuse_diag = false;
  trial.reduced = 0;  // cost and gradient only: no reduced system at a trial point
  double dmax = -1.0;
  int contractions = 0;
  bool found = false;
  for (;;) {
    if (current.x != 1.0) BA_TRY(make_candidate(p, current.x));  // the full step is already in the candidate buffers
    std::swap(p->d_eop, p->d_eop_c);
    std::swap(p->d_xyz, p->d_xyz_c);
    std::swap(p->d_cam, p->d_cam_c);
    int rc = linearize(p, trial);
    if (!rc) {
      current.value = p->h_scal[SC_COST];
      rc = gradient_dot_step(p, &current.gradient);
    }
    std::swap(p->d_eop, p->d_eop_c);
    std::swap(p->d_xyz, p->d_xyz_c);
    std::swap(p->d_cam, p->d_cam_c);
    if (rc && rc != BA_ERR_NUMERIC) return rc;
    const bool valid = !rc && std::isfinite(current.value) && std::isfinite(current.gradient);
    current.value_is_valid = current.gradient_is_valid = valid ? 1 : 0;
    if (valid && !(current.value > x_cost + sufficient_decrease * start.gradient * current.x)) { found = true; break; }
    if (++contractions >= max_trials) break;
    const double next = ba_line_search_contract(&start, &previous, &current, max_contraction * current.x, min_contraction * current.x);
    if (dmax < 0.0) BA_TRY(step_max_norm(p, &dmax));
    if (next * dmax < min_step) break;
    previous = current;
    current.x = next;
  }
  if (!found && current.x != 1.0) BA_TRY(make_candidate(p, 1.0));
  BA_TRY(p->read_scal());
  return 0;
}

// [Ceres 1.13] TrustRegionMinimizer::Minimize with LevenbergMarquardtStrategy and the monotonic
// TrustRegionStepEvaluator; options as at src/main.cpp:4705-4719.
int ba_solve_resident(ba_problem *p, const ba_options *opt, ba_summary *summary, ba_iter_row *rows, int32_t rows_cap) {
  if (!p || !opt) return fail(BA_ERR_INVALID, "ba_solve_resident: null argument");
  if (!p->have_params) return fail(BA_ERR_INVALID, "ba_solve_resident: call ba_set_parameters first");
  int linear_solver = opt->linear_solver;
  // The default: dense Cholesky for small reduced systems; the sparse direct solver where it is the faster of the two
  // sparse routes (measured on the benchmark geometry, ms per solve against the PCG's: 3.0 / 5.9 at order 10 800, 8.2 / 10.1
  // at 60 000, 14.7 / 14.9 at 120 000, 38 / 37 at 300 000); the PCG beyond, and wherever the block-sparse form or a tree is
  // not to be had.
  bool auto_sparse = false;
  if (linear_solver == BA_SOLVER_AUTO) {
    if (p->nr <= BA_AUTO_DENSE_ORDER) linear_solver = BA_SOLVER_DENSE_SCHUR;
    else if (p->nr <= BA_AUTO_SPARSE_ORDER && p->explicit_ok && !p->sparse_failed) { linear_solver = BA_SOLVER_SPARSE_SCHUR; auto_sparse = true; }
    else linear_solver = BA_SOLVER_PCG_SCHUR;
  }
  if (linear_solver != BA_SOLVER_DENSE_SCHUR && linear_solver != BA_SOLVER_PCG_SCHUR && linear_solver != BA_SOLVER_PCG_IMPLICIT &&
      linear_solver != BA_SOLVER_SPARSE_SCHUR)
    return fail(BA_ERR_UNSUPPORTED, "linear_solver %d", opt->linear_solver);
  if (linear_solver == BA_SOLVER_SPARSE_SCHUR && !p->explicit_ok)
    return fail(BA_ERR_UNSUPPORTED, "BA_SOLVER_SPARSE_SCHUR needs the reduced system in block-sparse form (this network's is too dense for it)");
  // The kernels of the explicit block-sparse PCG hold the camera columns in LDS arrays of fixed size: a network with
  // more free camera parameters than they take (many cameras) runs the same PCG on the matrix-free operator instead.
  const bool pcg_too_wide = linear_solver == BA_SOLVER_PCG_SCHUR && p->ncam > cg2_max_cam();
  const bool use_explicit = p->explicit_ok && linear_solver != BA_SOLVER_PCG_IMPLICIT && !pcg_too_wide;
  if (!(opt->initial_trust_region_radius > 0.0)) return fail(BA_ERR_INVALID, "initial_trust_region_radius must be positive");
  const double t_begin = now_s();
  ba_summary sm;
  std::memset(&sm, 0, sizeof sm);
  fill_sizes(p, &sm);
  sm.linear_solver_used = linear_solver;
  int n_rows = 0;
  LmState st{opt->initial_trust_region_radius, opt->min_lm_diagonal, opt->max_lm_diagonal, false,
             opt->min_lm_diagonal != opt->max_lm_diagonal, use_explicit ? 2 : 1};
  double decrease_factor = 2.0;
  int num_invalid = 0;

  if (p->boxed) {
    // [Ceres 1.13 Program::IsFeasible] a constant block must start inside its box: nothing can move it there
    for (int c = 0; c < p->C; ++c)
      for (int k = 0; k < 3; ++k) {
        const CamRec &r = p->cam_h[c];
        if (!(r.in_program & 1) && (r.iop[k] < r.lo[k] || r.iop[k] > r.hi[k]))
          return fail(BA_ERR_INVALID, "camera %d: constant IOP[%d] = %g lies outside [%g, %g]", c, k, r.iop[k], r.lo[k], r.hi[k]);
      }
  }
  const bool constrained = p->constrained;
  if (constrained) {
    // [Ceres 1.13 TrustRegionMinimizer::IterationZero] x = Plus(x, 0): the start point projected onto the box
    // (init_scaling uploads the camera records)
    for (auto &r : p->cam_h)
      if (r.in_program & 1)
        for (int k = 0; k < 3; ++k) r.iop[k] = std::min(std::max(r.iop[k], r.lo[k]), r.hi[k]);
  }
  bool rows_hold_x = true;  // the per-observation rows and the gradient on the device belong to x (not to a trial point)
  // Whatever way the solve ends, the host copy of the camera records follows the device: the next call uploads it
  // (init_scaling), and a stale copy would put the cameras back while stations and points stay where they are.
  struct CamSync {
    ba_problem *p;
    ~CamSync() {
      if (p->C) (void)hipMemcpy(p->cam_h.data(), p->d_cam, p->cam_h.size() * sizeof(CamRec), hipMemcpyDeviceToHost);
    }
  } cam_sync{p};

  // ---- IterationZero ----
  double t0 = now_s();
  BA_TRY(init_scaling(p, opt));
  double x_norm = 0.0;
  BA_TRY(xnorm(p, &x_norm));
  BA_TRY(linearize(p, st));
  double x_cost = p->h_scal[SC_COST];
  const double fixed_cost = p->h_scal[SC_FIXED];
  sm.initial_cost = x_cost + fixed_cost;
  sm.fixed_cost = fixed_cost;
  double min_iter_cost = sm.initial_cost;
  ba_iter_row it;
  std::memset(&it, 0, sizeof it);
  it.step_is_valid = 1;
  it.step_is_successful = 1;
  it.cost = x_cost + fixed_cost;
  it.gradient_max_norm = p->h_scal[SC_GMAX];
  it.t_linearize_s = now_s() - t0;
  sm.termination = BA_NO_CONVERGENCE;
  bool linearized_for_radius = true;  // point/station blocks carry the LM diagonal of st.radius

  for (;;) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
    if (it.step_is_successful) sm.num_successful_steps++; else sm.num_unsuccessful_steps++;
    it.trust_region_radius = st.radius;
    if (rows && n_rows < rows_cap) rows[n_rows] = it;
    n_rows++;
    if (it.cost < min_iter_cost) min_iter_cost = it.cost;
    if (it.iteration >= opt->max_num_iterations) { sm.termination = BA_NO_CONVERGENCE; sm.stop_reason = BA_STOP_MAX_ITERATIONS; break; }
    if (it.step_is_successful && it.gradient_max_norm <= opt->gradient_tolerance) { sm.termination = BA_CONVERGENCE; sm.stop_reason = BA_STOP_GRADIENT_TOL; break; }
    if (st.radius <= opt->min_trust_region_radius) { sm.termination = BA_CONVERGENCE; sm.stop_reason = BA_STOP_MIN_RADIUS; break; }

    const int iter_no = it.iteration + 1;
    const double prev_gmax = it.gradient_max_norm;
    std::memset(&it, 0, sizeof it);
    it.iteration = iter_no;
    it.gradient_max_norm = prev_gmax;

    // ---- LevenbergMarquardtStrategy::ComputeStep ----
    t0 = now_s();
    if (!linearized_for_radius) {
      // same point, smaller radius: the eliminated blocks depend on the radius through D.  After a line search the
      // rows are those of its last trial point: then everything is computed again, to the same values as before
      st.reuse_diag = rows_hold_x;
      rows_hold_x = true;
      BA_TRY(linearize(p, st));
      it.t_linearize_s += now_s() - t0;
      t0 = now_s();
      linearized_for_radius = true;
    }
    int lin_iters = 0;
    int lrc;
    if (linear_solver == BA_SOLVER_DENSE_SCHUR) lrc = solve_dense(p, use_explicit, true);
    else if (linear_solver == BA_SOLVER_SPARSE_SCHUR) {
      lrc = solve_sparse(p, true);
      if (lrc == BA_ERR_UNSUPPORTED && auto_sparse) {  // no tree for this network (or no room for its fronts): the PCG after all
        p->sparse_failed = true;
        linear_solver = BA_SOLVER_PCG_SCHUR;
        sm.linear_solver_used = linear_solver;
        lrc = solve_pcg_explicit(p, opt, &lin_iters);
      }
    }
    else if (use_explicit) lrc = solve_pcg_explicit(p, opt, &lin_iters);
    else lrc = solve_pcg(p, opt, &lin_iters);
    it.linear_iterations = lin_iters;
    sm.total_linear_iterations += lin_iters;
    if (lrc != 0 && lrc != BA_ERR_NUMERIC) return lrc;
    bool step_ok = (lrc == 0);
    double mcc = 0.0;
    // Without a box the candidate is the full step, and its cost is evaluated in the same phase (one sum over the ranks,
    // one read-back); it is simply not looked at when the step turns out to be invalid.
    const bool cost_with_step = !constrained;
    if (step_ok) {
      BA_TRY(step_and_candidate(p, cost_with_step));
      BA_TRY(p->read_scal());
      mcc = -p->h_scal[SC_MCC];
      it.step_norm = std::sqrt(p->h_scal[SC_STEP2_SC] + p->h_scal[SC_STEP2_P]);
      step_ok = std::isfinite(mcc) && std::isfinite(it.step_norm);
      if ((linear_solver == BA_SOLVER_SPARSE_SCHUR || linear_solver == BA_SOLVER_DENSE_SCHUR) && p->h_scal[SC_SPINFO] != 0.0)
        step_ok = false;  // a pivot was not positive
    }
    it.t_solve_s = now_s() - t0;
    // ---- ComputeTrustRegionStep: the step is valid when the model predicts a decrease ----
    it.step_is_valid = step_ok && (mcc > 0.0);
    if (!it.step_is_valid) {
      // ---- HandleInvalidStep ----
      if (++num_invalid >= opt->max_num_consecutive_invalid_steps) {
        sm.termination = BA_FAILURE;
        sm.stop_reason = BA_STOP_INVALID_STEPS;
        break;
      }
      st.radius = st.radius / decrease_factor;
      decrease_factor *= 2.0;
      linearized_for_radius = false;
      it.cost = x_cost + fixed_cost;
      it.step_norm = 0.0;
      it.step_is_successful = 0;
      continue;
    }
    num_invalid = 0;

    if (constrained) {
      // ---- DoLineSearch ----
      t0 = now_s();
      rows_hold_x = false;
      linearized_for_radius = false;
      BA_TRY(projected_line_search(p, st, x_cost));
      it.step_norm = std::sqrt(p->h_scal[SC_STEP2_SC] + p->h_scal[SC_STEP2_P]);
      it.t_linearize_s += now_s() - t0;
    }

    // ---- ComputeCandidatePointAndEvaluateCost ----
    t0 = now_s();
    if (!cost_with_step) {
      BA_TRY(cost_at(p, p->d_eop_c, p->d_xyz_c, p->d_cam_c, p->d_sta_c, SC_CCOST, -1, 0, nullptr));
      BA_TRY(p->read_scal());
    }
    double cand_cost = p->h_scal[SC_CCOST];
    if (!std::isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
    it.t_eval_s = now_s() - t0;

    // ---- ParameterToleranceReached ----
    if (it.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
      sm.termination = BA_CONVERGENCE;
      sm.stop_reason = BA_STOP_PARAMETER_TOL;
      break;
    }
    // ---- FunctionToleranceReached ----
    it.cost_change = x_cost - cand_cost;
    if (std::fabs(it.cost_change) <= opt->function_tolerance * x_cost) {
      sm.termination = BA_CONVERGENCE;
      sm.stop_reason = BA_STOP_FUNCTION_TOL;
      break;
    }
    // ---- IsStepSuccessful ----
    it.relative_decrease = it.cost_change / mcc;
    if (it.relative_decrease > opt->min_relative_decrease) {
      // ---- HandleSuccessfulStep ----
      std::swap(p->d_eop, p->d_eop_c);
      std::swap(p->d_xyz, p->d_xyz_c);
      std::swap(p->d_cam, p->d_cam_c);
      x_norm = std::sqrt(p->h_scal[SC_CNORM2_SC] + p->h_scal[SC_CNORM2_P]);
      // LevenbergMarquardtStrategy::StepAccepted
      st.radius = st.radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3.0));
      st.radius = std::min(opt->max_trust_region_radius, st.radius);
      decrease_factor = 2.0;
      st.reuse_diag = false;
      t0 = now_s();
      BA_TRY(linearize(p, st));
      it.t_linearize_s += now_s() - t0;
      linearized_for_radius = true;
      rows_hold_x = true;
      x_cost = p->h_scal[SC_COST];
      it.gradient_max_norm = p->h_scal[SC_GMAX];
      it.cost = x_cost + fixed_cost;
      it.step_is_successful = 1;
    } else {
      // ---- HandleUnsuccessfulStep / StepRejected ----
      it.step_is_successful = 0;
      st.radius = st.radius / decrease_factor;
      decrease_factor *= 2.0;
      linearized_for_radius = false;
      it.cost = cand_cost + fixed_cost;
    }
  }
  // [Ceres 1.13 solver.cc SetSummaryFinalCost] minimum over the logged iterations
  sm.final_cost = min_iter_cost;
  sm.num_iterations = n_rows;
  sm.t_total_s = now_s() - t_begin;
  if (summary) *summary = sm;
  return 0;
}

int ba_solve(ba_problem *p, const ba_options *opt, double *eop, double *xyz, double *iop, double *ap, ba_summary *summary,
             ba_iter_row *rows, int32_t rows_cap) {
  BA_TRY(ba_set_parameters(p, eop, xyz, iop, ap));
  BA_TRY(ba_solve_resident(p, opt, summary, rows, rows_cap));
  return ba_get_parameters(p, eop, xyz, iop, ap);
}

int ba_tuning(const char *name, double value) {
  if (!name || !tuning_set(name, value)) return fail(BA_ERR_INVALID, "ba_tuning: no switch named '%s'", name ? name : "(null)");
  return 0;
}

int ba_sparse_solver_stats(const ba_problem *p, int32_t *fronts, int32_t *levels, int32_t *largest_front, int32_t *critical_panels,
                           double *front_bytes, double *factor_flops) {
  if (!p) return fail(BA_ERR_INVALID, "ba_sparse_solver_stats: null problem");
  if (!p->sparse) return fail(BA_ERR_INVALID, "ba_sparse_solver_stats: no solve with BA_SOLVER_SPARSE_SCHUR has run on this problem");
  SparseCholStats st;
  sparse_chol_stats(p->sparse, &st);
  if (fronts) *fronts = st.fronts;
  if (levels) *levels = st.levels;
  if (largest_front) *largest_front = st.largest_front;
  if (critical_panels) *critical_panels = st.critical_panels;
  if (front_bytes) *front_bytes = st.arena_bytes;
  if (factor_flops) *factor_flops = st.factor_flops;
  return 0;
}

int ba_set_covariance_route(ba_problem *p, int32_t route, int32_t max_cluster_stations) {
  if (!p) return fail(BA_ERR_INVALID, "ba_set_covariance_route: null problem");
  if (route < 0 || route > 2) return fail(BA_ERR_INVALID, "ba_set_covariance_route: route %d (0 automatic, 1 dense inverse, 2 dissection)", route);
  if (max_cluster_stations < 0) return fail(BA_ERR_INVALID, "ba_set_covariance_route: negative cluster size");
  p->cov_route = route;
  p->cov_max_cluster = max_cluster_stations > 0 ? max_cluster_stations : 1536;
  return 0;
}

int ba_covariance_stats(const ba_problem *p, int32_t *route_used, int32_t *clusters, int32_t *boundary_order, int32_t *largest_cluster_order) {
  if (!p) return fail(BA_ERR_INVALID, "ba_covariance_stats: null problem");
  if (route_used) *route_used = p->cov_route_used;
  if (clusters) *clusters = p->cov_stats.clusters;
  if (boundary_order) *boundary_order = p->cov_stats.boundary_order;
  if (largest_cluster_order) *largest_cluster_order = p->cov_stats.largest_cluster_order;
  return 0;
}

int ba_covariance(ba_problem *p, double *var_eop, double *var_xyz, double *var_iop, double *var_ap) {
  if (!p) return fail(BA_ERR_INVALID, "ba_covariance: null problem");
  if (!p->have_params) return fail(BA_ERR_INVALID, "ba_covariance: no parameters set");
  const int n = p->nr;
  // Route 1 needs the dense reduced matrix, a work matrix of the same size for the inverse and the per-point work array;
  // when they do not fit, the blocks of the inverse that the variances need come from a dissection of the station graph.
  bool selected = p->cov_route == 2, dense_fits = true;
  {
    const double ld = n >= chol_blocked_min_order() ? chol_blocked_ld(n) : n;
    const double need = 2.0 * n * ld * 8.0 + (double)p->P * p->ncam * 24.0 + 1e9;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const double have = (double)free_b + (p->d_Sdense ? (double)n * ld * 8.0 : 0.0);
    dense_fits = need <= have;
    if (!dense_fits && p->cov_route == 1)
      return fail(BA_ERR_UNSUPPORTED, "ba_covariance: reduced system of order %d: the dense inverse needs %.1f GB, %.1f GB are free", n,
                  need * 1e-9, have * 1e-9);
    // left to choose: the dissection when the dense inverse does not fit, and from order 16384 on, where it is the
    // faster of the two (0.5 s against 5.7 s at order 60 007); the dense inverse is the fallback while it fits
    if (p->cov_route == 0 && (!dense_fits || (n >= 16384 && p->explicit_ok && p->S > 0))) selected = true;
  }
  if (selected && (!p->explicit_ok || p->S < 1))
    return fail(BA_ERR_UNSUPPORTED, "ba_covariance: the dissection route needs the reduced system in blocks (a network with stations whose "
                "camera blocks fit the block kernels); ba_covariance_cameras gives the camera parameters at any size");
  hipStream_t s = p->stream;
  ba_options opt;
  ba_default_options(&opt);
  // Jacobi scaling for the conditioning of the factorisation (undone below); no LM damping
  BA_TRY(init_scaling(p, &opt));
  const bool use_explicit = p->explicit_ok;
  LmState st{1e300, opt.min_lm_diagonal, opt.max_lm_diagonal, false, false, use_explicit ? 2 : 1};
  st.full_rows = true;  // the point variances are formed from the rows
  BA_TRY(linearize(p, st));
  std::vector<double> cs((size_t)std::max(n, 1), 0.0), vx(3 * (size_t)std::max(p->P, 1), 0.0);
  double *d_inv = nullptr, *d_tc = nullptr, *d_vx = nullptr, *d_cs = nullptr, *d_T = nullptr;
  auto cleanup = [&]() {
    if (d_inv) (void)hipFree(d_inv);
    if (d_tc) (void)hipFree(d_tc);
    if (d_vx) (void)hipFree(d_vx);
    if (d_cs) (void)hipFree(d_cs);
    if (d_T) (void)hipFree(d_T);
  };
  const size_t nn = (size_t)std::max(n, 1);
  CovArgs ca{};
  ca.P = p->P; ca.S = p->S; ca.n = n; ca.ncmax = p->ncmax;
  ca.pt_start = p->d_pt_start; ca.obs_sta = p->d_obs_sta; ca.sta_cam = p->d_sta_cam; ca.sta_used = p->d_sta_used; ca.pt_file = p->d_pt_file;
  ca.rec = p->d_rec; ca.Jp = p->d_Jp; ca.Jc = p->d_Jc; ca.Vinv = p->d_Vinv;
  ca.scale_pt = p->d_scale_pt; ca.scale_sta = p->d_scale_sta; ca.scale_cam = p->d_scale_cam; ca.cam = p->d_cam;
  p->cov_route_used = selected ? 2 : 1;
  p->cov_stats = SelInvStats{};
  if (selected) {
    if (hipMalloc((void **)&d_tc, std::max<size_t>((size_t)p->P * p->ncam * 3, 1) * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&d_vx, vx.size() * sizeof(double)) != hipSuccess || hipMalloc((void **)&d_cs, nn * sizeof(double)) != hipSuccess) {
      cleanup();
      return fail(BA_ERR_DEVICE, "ba_covariance: out of device memory");
    }
    ca.Tc = d_tc; ca.var_xyz = d_vx; ca.var_cs = d_cs;
    SelInvArgs sa{};
    sa.stream = s; sa.S = p->S; sa.ncam = p->ncam;
    sa.bsr_rowptr_h = p->bsr_rowptr_h.data(); sa.bsr_col_h = p->bsr_col_h.data(); sa.sta_used_h = p->sta_used_h.data();
    sa.d_bsr_rowptr = p->d_bsr_rowptr; sa.d_bsr_col = p->d_bsr_col; sa.d_bsr_val = p->d_bsr_val; sa.d_Ssc = p->d_Ssc; sa.d_Scc = p->d_Scc;
    sa.max_cluster = p->cov_max_cluster; sa.cov = ca; sa.d_info = p->d_info;
    std::string why;
    const int rc = covariance_selected(sa, why, &p->cov_stats);
    if (rc == 1 && p->cov_route == 0 && dense_fits) {
      selected = false;  // a station graph without separators: the dense inverse after all
      p->cov_route_used = 1;
    } else if (rc) {
      (void)hipStreamSynchronize(s);
      cleanup();
      if (rc == 1) return fail(BA_ERR_UNSUPPORTED, "ba_covariance: reduced system of order %d, dissection route: %s "
                               "(ba_covariance_cameras gives the camera parameters at any size)", n, why.c_str());
      if (rc == 3) return fail(BA_ERR_NUMERIC, "ba_covariance: %s", why.c_str());
      return fail(BA_ERR_DEVICE, "ba_covariance: %s", why.c_str());
    }
  }
  if (!selected) {
    if (d_tc) { (void)hipFree(d_tc); d_tc = nullptr; }
    if (d_vx) { (void)hipFree(d_vx); d_vx = nullptr; }
    if (d_cs) { (void)hipFree(d_cs); d_cs = nullptr; }
    if (n > 0) {
      BA_TRY(build_dense(p, use_explicit));
      launch_dense_fix_unused(s, p->S, p->dense_ld, p->d_sta_used, p->d_Sdense);
      BA_TRY(factor_dense(p));
      int info = 0;
      HIP_OK(hipMemcpyAsync(&info, p->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      if (info) return fail(BA_ERR_NUMERIC, "ba_covariance: the reduced normal matrix is rank deficient (column %d of %d)", info - 1, n);
    }
    const size_t ld = (size_t)std::max(p->dense_ld, 1);
    if (hipMalloc((void **)&d_inv, nn * ld * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&d_tc, std::max<size_t>((size_t)p->P * p->ncam * 3, 1) * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&d_vx, vx.size() * sizeof(double)) != hipSuccess || hipMalloc((void **)&d_cs, nn * sizeof(double)) != hipSuccess ||
        (p->dense_blocked && hipMalloc((void **)&d_T, 128 * ld * sizeof(double)) != hipSuccess)) {
      cleanup();
      return fail(BA_ERR_DEVICE, "ba_covariance: out of device memory");
    }
    const double *d_Sinv = d_inv;
    if (n > 0 && p->dense_blocked) {
      // the inverse replaces the factor in place; d_inv is the work matrix U = L^-T
      launch_chol_inverse_blocked(s, p->d_Sdense, n, p->dense_ld, p->d_cholW, d_inv, d_T);
      d_Sinv = p->d_Sdense;
    } else {
      launch_chol_inverse(s, p->d_Sdense, n, d_inv);
    }
    ca.ld = p->dense_ld; ca.Sinv = d_Sinv;
    ca.Tc = d_tc; ca.var_xyz = d_vx; ca.var_cs = d_cs;
    launch_cov(s, ca);
  }
  hipError_t e1 = hipMemcpyAsync(vx.data(), d_vx, vx.size() * sizeof(double), hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipMemcpyAsync(cs.data(), d_cs, nn * sizeof(double), hipMemcpyDeviceToHost, s);
  hipError_t e3 = hipStreamSynchronize(s);
  cleanup();
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(BA_ERR_DEVICE, "ba_covariance: download failed");
  BA_TRY(p->check_kernels());
  if (var_xyz) std::copy(vx.begin(), vx.begin() + 3 * (size_t)p->P, var_xyz);
  if (var_eop) std::copy(cs.begin(), cs.begin() + 6 * (size_t)p->S, var_eop);
  for (int c = 0; c < p->C; ++c) {
    const CamRec &r = p->cam_h[c];
    if (var_iop) var_iop[3 * c] = var_iop[3 * c + 1] = var_iop[3 * c + 2] = 0.0;
    if (var_ap) std::fill(var_ap + 16 * (size_t)c, var_ap + 16 * (size_t)c + 16, 0.0);
    int pos = 0;
    for (int q = 0; q < BA_NCAMPAR; ++q) {
      if (!(r.mask & (1u << q))) continue;
      const double v = cs[6 * (size_t)p->S + r.off + pos];
      if (q < 3) { if (var_iop) var_iop[3 * c + q] = v; }
      else if (var_ap) var_ap[16 * (size_t)c + q - 3] = v;
      ++pos;
    }
  }
  return 0;
}

int ba_covariance_cameras(ba_problem *p, double *var_iop, double *var_ap) {
  if (!p) return fail(BA_ERR_INVALID, "ba_covariance_cameras: null problem");
  if (!p->have_params) return fail(BA_ERR_INVALID, "ba_covariance_cameras: no parameters set");
  if (p->ncam > 0 && !p->explicit_ok)
    return fail(BA_ERR_UNSUPPORTED, "ba_covariance_cameras: needs the explicit block-sparse reduced system (pattern too dense for it)");
  if (p->ncam > cg2_max_cam())
    return fail(BA_ERR_UNSUPPORTED, "ba_covariance_cameras: %d free camera parameters, the block-sparse PCG takes at most %d (use ba_covariance)",
                p->ncam, cg2_max_cam());
  hipStream_t s = p->stream;
  ba_options opt;
  ba_default_options(&opt);
  std::vector<double> var((size_t)std::max(p->ncam, 1), 0.0), scale((size_t)std::max(p->ncam, 1), 1.0);
  if (p->ncam > 0) {
    // as ba_covariance: Jacobi scaling for the conditioning (undone below), no damping; then column k of the
    // camera corner of S^-1 is the camera part of the solution of S x = e_k
    BA_TRY(init_scaling(p, &opt));
    LmState st{1e300, opt.min_lm_diagonal, opt.max_lm_diagonal, false, false, 2};
    BA_TRY(linearize(p, st));
    HIP_OK(hipMemcpyAsync(scale.data(), p->d_scale_cam, p->ncam * sizeof(double), hipMemcpyDeviceToHost, s));
    const int64_t o = 6 * (int64_t)p->S;
    for (int k = 0; k < p->ncam; ++k) {
      launch_unit(s, p->nr, p->d_b, o + k);
      int iters = 0;
      BA_TRY(solve_pcg_explicit(p, &opt, &iters));
      HIP_OK(hipMemcpyAsync(&var[k], p->d_x + o + k, sizeof(double), hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      if (!(var[k] > 0.0) || !std::isfinite(var[k]))
        return fail(BA_ERR_NUMERIC, "ba_covariance_cameras: camera column %d: the reduced normal matrix is not positive definite", k);
    }
    BA_TRY(p->check_kernels());
  }
  for (int c = 0; c < p->C; ++c) {
    const CamRec &r = p->cam_h[c];
    if (var_iop) var_iop[3 * c] = var_iop[3 * c + 1] = var_iop[3 * c + 2] = 0.0;
    if (var_ap) std::fill(var_ap + 16 * (size_t)c, var_ap + 16 * (size_t)c + 16, 0.0);
    int pos = 0;
    for (int q = 0; q < BA_NCAMPAR; ++q) {
      if (!(r.mask & (1u << q))) continue;
      const double v = var[r.off + pos] * scale[r.off + pos] * scale[r.off + pos];
      if (q < 3) { if (var_iop) var_iop[3 * c + q] = v; }
      else if (var_ap) var_ap[16 * (size_t)c + q - 3] = v;
      ++pos;
    }
  }
  return 0;
}

int ba_covariance_camera_rows(ba_problem *p, int32_t *n_rows, int32_t *cam_param, double *cov_eop, double *cov_xyz, double *cov_cam) {
  if (!p) return fail(BA_ERR_INVALID, "ba_covariance_camera_rows: null problem");
  if (!p->have_params) return fail(BA_ERR_INVALID, "ba_covariance_camera_rows: no parameters set");
  if (p->ar) return fail(BA_ERR_UNSUPPORTED, "ba_covariance_camera_rows: one rank only (the points of a sharded problem lie on several)");
  const int n = p->nr, ncam = p->ncam;
  const size_t ns = 6 * (size_t)p->S, np3 = 3 * (size_t)p->P;
  if (n_rows) *n_rows = ncam;
  if (cam_param)
    for (int c = 0; c < p->C; ++c) {
      const CamRec &r = p->cam_h[c];
      int pos = 0;
      for (int q = 0; q < BA_NCAMPAR; ++q)
        if (r.mask & (1u << q)) cam_param[r.off + pos++] = BA_NCAMPAR * c + q;
    }
  if (ncam == 0) return 0;
  hipStream_t s = p->stream;
  ba_options opt;
  ba_default_options(&opt);
  // as ba_covariance: Jacobi scaling for the conditioning (undone below), no damping.  Row k of the camera part of Cx is
  // the solution of S x = e_k on the reduced unknowns and -V^-1 W^T x on the points: the back substitution of a step
  // with the point gradient set to zero.
  BA_TRY(init_scaling(p, &opt));
  LmState st{1e300, opt.min_lm_diagonal, opt.max_lm_diagonal, false, false, p->explicit_ok ? 2 : 1};
  st.full_rows = true;  // the back substitution below reads the rows
  BA_TRY(linearize(p, st));
  int how = 0;  // 0 dense factor, 1 sparse factor, 2 one PCG solve per row
  if (p->explicit_ok && n > BA_AUTO_DENSE_ORDER) how = (n <= BA_AUTO_SPARSE_ORDER && !p->sparse_failed) ? 1 : 2;
  const int forced = (int)tuning_value("rows_solver", -1.0);  // for comparisons of the three routes
  if (forced >= 0 && forced <= 2 && (forced == 0 || p->explicit_ok)) how = forced;
  if (how == 2 && ncam > cg2_max_cam())
    return fail(BA_ERR_UNSUPPORTED, "ba_covariance_camera_rows: %d free camera parameters, the block-sparse PCG takes at most %d", ncam, cg2_max_cam());
  std::vector<double> sc_sta(std::max<size_t>(ns, 1)), sc_cam((size_t)ncam), sc_pt, x((size_t)n), y;
  HIP_OK(hipMemcpyAsync(sc_sta.data(), p->d_scale_sta, ns * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(sc_cam.data(), p->d_scale_cam, ncam * sizeof(double), hipMemcpyDeviceToHost, s));
  if (cov_xyz) {
    sc_pt.resize(std::max<size_t>(np3, 1));
    y.resize(std::max<size_t>(np3, 1));
    HIP_OK(hipMemcpyAsync(sc_pt.data(), p->d_scale_pt, np3 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemsetAsync(p->d_g_pt, 0, np3 * sizeof(double), s));
  }
  bool factored = false;
  for (int k = 0; k < ncam; ++k) {
    launch_unit(s, n, p->d_b, (int64_t)ns + k);
    if (how == 1) {
      if (!factored) {
        const int rc = solve_sparse(p);
        if (rc == BA_ERR_UNSUPPORTED) {  // no tree for this network: the PCG after all
          p->sparse_failed = true;
          if (ncam > cg2_max_cam()) return rc;
          how = 2;
        } else if (rc) {
          return rc;
        }
        factored = true;
      } else {
        HIP_OK(hipMemsetAsync(p->d_x, 0, (size_t)n * sizeof(double), s));
        sparse_chol_solve(p->sparse, s, p->d_b, p->d_x);
      }
    }
    if (how == 0) {
      if (!factored) {
        BA_TRY(build_dense(p, p->explicit_ok));
        launch_dense_fix_unused(s, p->S, p->dense_ld, p->d_sta_used, p->d_Sdense);
        BA_TRY(factor_dense(p));
        int info = 0;
        HIP_OK(hipMemcpyAsync(&info, p->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (info) return fail(BA_ERR_NUMERIC, "ba_covariance_camera_rows: the reduced normal matrix is rank deficient (column %d of %d)", info - 1, n);
        factored = true;
      }
      launch_copy(s, n, p->d_b, p->d_x);
      if (p->dense_blocked) launch_chol_solve_blocked(s, p->d_Sdense, n, p->dense_ld, p->d_cholW, p->d_x);
      else launch_chol_solve(s, p->d_Sdense, n, p->d_x);
    } else if (how == 2) {
      int iters = 0;
      BA_TRY(solve_pcg_explicit(p, &opt, &iters));
    }
    if (cov_xyz) {
      PointArgs pa = point_args(p, p->d_x);
      launch_point_pass(s, pa, MODE_BACKSUB, true);
      HIP_OK(hipMemcpyAsync(y.data(), p->d_y_pt, np3 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIP_OK(hipMemcpyAsync(x.data(), p->d_x, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    BA_TRY(p->check_kernels());
    const double sk = sc_cam[k];
    if (!(x[ns + k] > 0.0) || !std::isfinite(x[ns + k]))
      return fail(BA_ERR_NUMERIC, "ba_covariance_camera_rows: camera column %d: the reduced normal matrix is not positive definite", k);
    if (cov_eop)
      for (size_t i = 0; i < ns; ++i) cov_eop[(size_t)k * ns + i] = p->sta_used_h[i / 6] ? x[i] * sc_sta[i] * sk : 0.0;
    if (cov_cam)
      for (int m = 0; m < ncam; ++m) cov_cam[(size_t)k * ncam + m] = x[ns + m] * sc_cam[m] * sk;
    if (cov_xyz)
      for (size_t jj = 0; jj < (size_t)p->P; ++jj)
        for (int m = 0; m < 3; ++m) cov_xyz[(size_t)k * np3 + 3 * (size_t)p->pt_file[jj] + m] = y[3 * jj + m] * sc_pt[3 * jj + m] * sk;
  }
  return 0;
}

int ba_evaluate(ba_problem *p, const double *eop, const double *xyz, const double *iop, const double *ap, int apply_loss,
                double *cost, double *residuals) {
  if (!p) return fail(BA_ERR_INVALID, "ba_evaluate: null problem");
  BA_TRY(set_parameters(p, eop, xyz, iop, ap, true));
  double *d_res = nullptr;
  const size_t nres = 2 * (size_t)p->N + 3 * (size_t)p->P + 3 * (size_t)p->C;
  if (residuals) HIP_OK(hipMalloc((void **)&d_res, std::max<size_t>(nres, 1) * sizeof(double)));
  // column scales do not enter the residuals; the station records only need valid memory
  int rc = cost_at(p, p->d_eop_c, p->d_xyz_c, p->d_cam_c, p->d_sta_c, SC_CCOST, SC_FIXED, apply_loss, d_res);
  if (!rc) rc = p->read_scal();
  if (!rc && residuals && nres) {
    if (hipMemcpy(residuals, d_res, nres * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(BA_ERR_DEVICE, "residual download failed");
  }
  if (d_res) hipFree(d_res);
  if (rc) return rc;
  if (cost) *cost = p->h_scal[SC_CCOST] + p->h_scal[SC_FIXED];
  return 0;
}

int ba_residual_jacobian(ba_problem *p, double *r, double *J) {
  if (!p) return fail(BA_ERR_INVALID, "ba_residual_jacobian: null problem");
  if (!p->have_params) return fail(BA_ERR_INVALID, "ba_residual_jacobian: call ba_set_parameters first");
  double *d_r = nullptr, *d_J = nullptr;
  const size_t N = (size_t)p->N;
  if (r) HIP_OK(hipMalloc((void **)&d_r, std::max<size_t>(2 * N, 1) * sizeof(double)));
  if (J) {
    hipError_t e = hipMalloc((void **)&d_J, std::max<size_t>(56 * N, 1) * sizeof(double));
    if (e != hipSuccess) {
      if (d_r) hipFree(d_r);
      return fail(BA_ERR_DEVICE, "hipMalloc for the dense Jacobian (%zu bytes): %s", 56 * N * 8, hipGetErrorString(e));
    }
  }
  launch_fill(p->stream, 6 * (int64_t)p->S, p->d_scale_sta, 1.0);
  launch_station_prep(p->stream, p->S, p->d_eop, p->d_scale_sta, p->d_sta_cam, p->d_sta_c);
  CostArgs ca{};
  ca.N = p->N; ca.obs = p->d_obs; ca.sta = p->d_sta_c; ca.xyz = p->d_xyz; ca.obs_pt = p->d_obs_pt; ca.cam = p->d_cam;
  ca.huber_a = p->huber_a; ca.model = p->model; ca.emp = p->emp;
  launch_resjac_dense(p->stream, ca, d_r, d_J);
  int rc = p->check_kernels();
  if (!rc && hipStreamSynchronize(p->stream) != hipSuccess) rc = fail(BA_ERR_DEVICE, "residual/Jacobian kernel failed");
  if (!rc && r && N && hipMemcpy(r, d_r, 2 * N * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(BA_ERR_DEVICE, "download failed");
  if (!rc && J && N && hipMemcpy(J, d_J, 56 * N * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(BA_ERR_DEVICE, "download failed");
  if (d_r) hipFree(d_r);
  if (d_J) hipFree(d_J);
  return rc;
}

int ba_time_kernel(ba_problem *p, int32_t which, int32_t repeats, double *seconds_per_launch) {
  if (!p || !seconds_per_launch || repeats < 1) return fail(BA_ERR_INVALID, "ba_time_kernel: bad argument");
  if (!p->have_params) return fail(BA_ERR_INVALID, "ba_time_kernel: call ba_set_parameters first");
  hipStream_t s = p->stream;
  ba_options opt;
  ba_default_options(&opt);
  // a valid linearisation (scales, station records, Jacobians) for the kernels to run on
  BA_TRY(init_scaling(p, &opt));
  LmState st{opt.initial_trust_region_radius, opt.min_lm_diagonal, opt.max_lm_diagonal, false, false, p->explicit_ok ? 2 : 1};
  st.full_rows = true;  // the kernels of the matrix-free operator read all rows
  BA_TRY(linearize(p, st));
  launch_copy(s, p->nr, p->d_b, p->d_p);
  LinArgs la{};
  la.P = p->P; la.ncmax = p->ncmax; la.N = p->N; la.pt_start = p->d_pt_start; la.obs_pt = p->d_obs_pt; la.elim = p->d_elim;
  la.obs = p->d_obs; la.sta = p->d_sta; la.xyz = p->d_xyz;
  la.scale_pt = p->d_scale_pt; la.pt_prior = p->d_pt_prior; la.pt_isig = p->d_pt_isig; la.cam = p->d_cam;
  la.rec = p->d_rec; la.Jp = p->d_Jp; la.Jc = p->d_Jc; la.Mo = p->d_Mo; la.Vinv = p->d_Vinv; la.g_pt = p->d_g_pt;
  la.colnorm_pt = p->d_colnorm_pt; la.diag_pt = p->d_diag_pt; la.cost_partial = p->d_partial; la.flags = p->d_flags;
  la.huber_a = p->huber_a; la.model = p->model; la.emp = p->emp; la.lm_min = st.lm_min; la.lm_max = st.lm_max; la.radius = st.radius; la.reuse_diag = 0;
  la.Y = p->explicit_ok ? p->d_Y : nullptr;
  la.Jt = p->d_Jt;
  la.cam_partial = p->d_campart;
  // timed as a solve runs it: without the rows that nothing reads
  la.lean = (p->explicit_ok && (p->d_Jt != nullptr || p->ncam == 0) && p->ncam <= cg2_max_cam()) ? 1 : 0;
  la.fused = (la.lean && p->fused_ok) ? 1 : 0;
  la.kp = p->kp;
  BsrCgArgs ba{};
  ba.S = p->S; ba.ncam = p->ncam; ba.rowptr = p->d_bsr_rowptr; ba.col = p->d_bsr_col; ba.val = p->d_bsr_val; ba.Ssc = p->d_Ssc;
  ba.Scc = p->d_Scc; ba.Lcc = p->d_Sccinv; ba.Minv = p->d_Minv; ba.b = p->d_b; ba.x = p->d_x; ba.r = p->d_r; ba.z = p->d_z; ba.p = p->d_p;
  ba.q = p->d_q; ba.scal = p->d_cg2scal; ba.part_pq = p->d_part_pq; ba.part_qc = p->d_part_qc; ba.part_rz = p->d_part_rz;
  ba.r2 = p->d_r2;
  if ((which == 4 || which == 5 || which == 6) && !p->explicit_ok) return fail(BA_ERR_INVALID, "ba_time_kernel: the block-sparse system was not built for this network");
  if ((which == 5 || which == 6) && p->ncam > cg2_max_cam()) return fail(BA_ERR_UNSUPPORTED, "ba_time_kernel: too many free camera parameters for the block-sparse PCG");
  // the PCG iteration is timed with the preconditioner the last solve used (multilevel once it has been built)
  const bool with_amg = p->amg && amg_levels(p->amg) >= 2;
  if ((which == 5 || which == 9) && with_amg && p->have_G && p->ncam > 0 && p->ncam <= defl_max_cam()) {
    // the camera coarse space of the last solve: its matrix belongs to that solve's last linearisation, which is all the
    // timing needs
    ba.G = p->d_G;
    ba.GT = p->d_GT;
    ba.Lcc = p->d_E;
    ba.part_gr = p->d_part_gr;
  }
  if (which == 5 || which == 6) {
    if (with_amg) {
      amg_numeric(p->amg, s, p->d_bsr_val, p->d_Minv);
      launch_amg_cg_init(s, ba, p->amg, 0.0);
    } else {
      launch_cg2_init(s, ba, 0.0);
    }
  }
  PointArgs pa = point_args(p, p->d_p);
  StationArgs sa = station_args(p, 7, 6);
  CostArgs ca{};
  ca.N = p->N; ca.obs = p->d_obs; ca.sta = p->d_sta; ca.xyz = p->d_xyz; ca.obs_pt = p->d_obs_pt; ca.cam = p->d_cam;
  ca.huber_a = p->huber_a; ca.model = p->model; ca.emp = p->emp; ca.partial = p->d_partial;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  auto run = [&]() {
    switch (which) {
      case 0: launch_linearize(s, la); break;
      case 1: launch_point_pass(s, pa, MODE_APPLY, false); break;
      case 2: launch_station_pass(s, sa, SMODE_GATHER); break;
      case 3: launch_cost_obs(s, ca, cost_blocks(p->N)); break;
      case 4:
        if (p->sw.n_win > 0) launch_schur_windows(s, p->sw, p->d_Y, p->d_ublk_dst, p->d_bsr_val);
        else launch_schur_blocks(s, p->n_ublk, p->d_ublk_cstart, p->d_ublk_dst, p->d_pairs, p->d_Y, p->d_bsr_val);
        break;
      case 5:  // two iterations, as the parity of the scalar slots alternates
        for (int par = 0; par < 2; ++par) {
          if (with_amg) launch_amg_cg_iteration(s, ba, p->amg, par);
          else launch_cg2_iteration(s, ba, par);
        }
        break;
      case 6: launch_bsr_spmv(s, ba, 0); break;
      case 7: launch_linearize(s, la, 1); break;
      case 8: launch_linearize(s, la, 2); break;
      default: break;
    }
  };
  if (which == 10 || which == 11) {
    // 10: Cholesky factorisation of the dense reduced matrix alone; 11: factorisation and the two triangular solves.
    // The factorisation is in place, so the matrix is rebuilt (untimed) before every repeat.
    double total_ms = 0.0;
    hipEvent_t f0, f1;
    HIP_OK(hipEventCreate(&f0));
    HIP_OK(hipEventCreate(&f1));
    for (int k = 0; k < repeats + 1; ++k) {
      BA_TRY(build_dense(p, p->explicit_ok));
      HIP_OK(hipEventRecord(f0, s));
      BA_TRY(factor_dense(p));
      if (which == 11) {
        launch_copy(s, p->nr, p->d_b, p->d_x);
        if (p->dense_blocked) launch_chol_solve_blocked(s, p->d_Sdense, p->nr, p->dense_ld, p->d_cholW, p->d_x);
        else launch_chol_solve(s, p->d_Sdense, p->nr, p->d_x);
      }
      HIP_OK(hipEventRecord(f1, s));
      HIP_OK(hipEventSynchronize(f1));
      float ms = 0.f;
      HIP_OK(hipEventElapsedTime(&ms, f0, f1));
      if (k > 0) total_ms += ms;  // the first pass warms up
    }
    hipEventDestroy(f0);
    hipEventDestroy(f1);
    *seconds_per_launch = 1e-3 * total_ms / repeats;
    return p->check_kernels();
  }
  if (which == 13 || which == 14) {
    // 13: assembly of the fronts and the sparse factorisation; 14: the same with the solve on the tree
    if (!p->explicit_ok) return fail(BA_ERR_INVALID, "ba_time_kernel: the block-sparse system was not built for this network");
    if (!p->sparse) return fail(BA_ERR_INVALID, "ba_time_kernel: no solve with BA_SOLVER_SPARSE_SCHUR has run on this problem");
    if (p->sparse_clear_queued) {
      HIP_OK(hipStreamWaitEvent(s, p->ev_sparse_clear, 0));
      p->sparse_clear_queued = false;
    }
    double total_ms = 0.0;
    hipEvent_t f0, f1;
    HIP_OK(hipEventCreate(&f0));
    HIP_OK(hipEventCreate(&f1));
    for (int k = 0; k < repeats + 1; ++k) {
      sparse_chol_clear(p->sparse, s);  // untimed, as in a solve (there it runs beside the evaluation of the step)
      HIP_OK(hipEventRecord(f0, s));
      sparse_chol_factor(p->sparse, s, p->d_bsr_val, p->d_Ssc, p->d_Scc, p->d_info);
      if (which == 14) sparse_chol_solve(p->sparse, s, p->d_b, p->d_x);
      HIP_OK(hipEventRecord(f1, s));
      HIP_OK(hipEventSynchronize(f1));
      float ms = 0.f;
      HIP_OK(hipEventElapsedTime(&ms, f0, f1));
      if (k > 0) total_ms += ms;
    }
    hipEventDestroy(f0);
    hipEventDestroy(f1);
    *seconds_per_launch = 1e-3 * total_ms / repeats;
    return p->check_kernels();
  }
  if (which == 9) {
    // one PCG iteration with time stamps at the phase marks of its kernels, table on stderr
    if (!p->explicit_ok || !with_amg) return fail(BA_ERR_INVALID, "ba_time_kernel: the traced iteration needs the multilevel PCG (solve once first)");
    const size_t nt = (size_t)PCG_TRACE_KERNELS * PCG_TRACE_MAXWG * PCG_TRACE_SLOTS;
    unsigned long long *d_tr = nullptr;
    HIP_OK(hipMalloc((void **)&d_tr, nt * sizeof(unsigned long long)));
    amg_numeric(p->amg, s, p->d_bsr_val, p->d_Minv);
    launch_amg_cg_init(s, ba, p->amg, 0.0);
    for (int k = 0; k < 2 * repeats; ++k) launch_amg_cg_iteration(s, ba, p->amg, k & 1);
    HIP_OK(hipMemsetAsync(d_tr, 0, nt * sizeof(unsigned long long), s));
    HIP_OK(hipStreamSynchronize(s));
    BsrCgArgs bt = ba;
    bt.trace = d_tr;
    launch_amg_cg_iteration(s, bt, p->amg, 0);
    launch_amg_cg_iteration(s, ba, p->amg, 1);
    HIP_OK(hipStreamSynchronize(s));
    std::vector<unsigned long long> tr(nt);
    HIP_OK(hipMemcpy(tr.data(), d_tr, nt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_tr);
    unsigned long long t0 = ~0ULL;
    for (size_t k = 0; k < nt; ++k)
      if (tr[k] && tr[k] < t0) t0 = tr[k];
    static const char *names[PCG_TRACE_KERNELS] = {"spmv", "update1_restrict", "restrict", "prolong", "update2"};
    fprintf(stderr, "PCG iteration trace (us since the first stamp; min / median / max over workgroups)\n");
    const int nbv = cg2_blocks_vec8(p->S);  // the second kernel: vector workgroups first, then the restricting ones
    for (int kk = 0; kk < PCG_TRACE_KERNELS; ++kk)
      for (int part = 0; part < (kk == 1 ? 2 : 1); ++part)
        for (int m = 0; m < PCG_TRACE_SLOTS; ++m) {
          std::vector<double> v;
          const int w0 = (kk == 1 && part == 1) ? nbv : 0, w1 = (kk == 1 && part == 0) ? std::min(nbv, (int)PCG_TRACE_MAXWG) : (int)PCG_TRACE_MAXWG;
          for (int w = w0; w < w1; ++w) {
            const unsigned long long x = tr[((size_t)kk * PCG_TRACE_MAXWG + w) * PCG_TRACE_SLOTS + m];
            if (x) v.push_back(0.01 * (double)(x - t0));
          }
          if (v.empty()) continue;
          std::sort(v.begin(), v.end());
          fprintf(stderr, "  %-18s mark %d  n=%4zu  %7.2f %7.2f %7.2f\n", kk == 1 ? (part ? "update1 (restrict)" : "update1 (vectors)") : names[kk], m,
                  v.size(), v.front(), v[v.size() / 2], v.back());
        }
    *seconds_per_launch = 0.0;
    return p->check_kernels();
  }
  if (which == 12) {
    // the observation kernel that also forms the station sums, with clock counters around the phases of a tile
    if (!la.fused) return fail(BA_ERR_INVALID, "ba_time_kernel: this network has no plan for the station sums");
    const int nw = 4 * keyed_blocks(p->kp.n_seg);
    unsigned long long *d_tr = nullptr;
    HIP_OK(hipMalloc((void **)&d_tr, (size_t)nw * KS_TRACE_SLOTS * sizeof(unsigned long long)));
    HIP_OK(hipMemsetAsync(d_tr, 0, (size_t)nw * KS_TRACE_SLOTS * sizeof(unsigned long long), s));
    LinArgs lt = la;
    lt.kp.trace = d_tr;
    launch_linearize(s, lt, 2);
    HIP_OK(hipStreamSynchronize(s));
    std::vector<unsigned long long> tr((size_t)nw * KS_TRACE_SLOTS);
    HIP_OK(hipMemcpy(tr.data(), d_tr, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_tr);
    static const char *names[KS_TRACE_SLOTS] = {"load + evaluate", "camera columns", "elimination, requests", "Y: compute + store", "camera rows + sums, final write-out", "products, staging writes", "row reads", "adds, table"};
    double tot = 0.0;
    std::vector<double> sum(KS_TRACE_SLOTS, 0.0);
    for (int wv = 0; wv < nw; ++wv)
      for (int k = 0; k < KS_TRACE_SLOTS; ++k) sum[k] += (double)tr[(size_t)wv * KS_TRACE_SLOTS + k];
    for (double v : sum) tot += v;
    fprintf(stderr, "k_obs_rows with station sums: clocks per wave and tile (%d waves, %d tiles each)\n", nw, p->kp.tiles_per_seg);
    for (int k = 0; k < KS_TRACE_SLOTS; ++k)
      fprintf(stderr, "  %-20s %9.0f  %5.1f %%\n", names[k], sum[k] / nw / p->kp.tiles_per_seg, 100.0 * sum[k] / std::max(tot, 1.0));
    *seconds_per_launch = 0.0;
    return p->check_kernels();
  }
  if (which < 0 || which > 8) return fail(BA_ERR_INVALID, "ba_time_kernel: unknown kernel %d", which);
  run();  // warm-up
  HIP_OK(hipEventRecord(e0, s));
  for (int k = 0; k < repeats; ++k) run();
  HIP_OK(hipEventRecord(e1, s));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *seconds_per_launch = 1e-3 * ms / repeats / (which == 5 ? 2 : 1);
  return p->check_kernels();
}

// Structure of the Jacobian that problem.Evaluate returns (src/main.cpp:6664).  [Ceres 1.13
// CompressedRowJacobianWriter]: one column range per parameter block in the order the blocks
// were added (src/main.cpp:3793-3800), as wide as the block's tangent space; every
// (residual block, variable parameter block) pair is written as a full block; a constant
// parameter block keeps its columns and gets no entries.
int ba_sparsity(ba_problem *p, int32_t *num_rows, int32_t *num_cols, int64_t *nnz, int32_t *rows, int32_t *cols) {
  if (!p) return fail(BA_ERR_INVALID, "ba_sparsity: null problem");
  const int N = p->N, S = p->S, P = p->P, C = p->C;
  std::vector<int> iop_w(C), ap_w(C), iop_col(C), ap_col(C);
  int col = 6 * S + 3 * P;
  for (int c = 0; c < C; ++c) {
    const unsigned m = p->iop_mask_h[c] & 7u;
    iop_w[c] = m ? popcount32(m) : 3;
    iop_col[c] = col;
    col += iop_w[c];
  }
  for (int c = 0; c < C; ++c) {
    const unsigned m = p->ap_mask_h[c];
    ap_w[c] = m ? popcount32(m) : 16;
    ap_col[c] = col;
    col += ap_w[c];
  }
  int64_t z = 0;
  int row = 0;
  auto put = [&](int c0, int w) {
    if (cols) for (int k = 0; k < w; ++k) cols[z + k] = c0 + k;
    z += w;
  };
  auto end_row = [&]() {
    ++row;
    if (rows) rows[row] = (int32_t)z;
  };
  if (rows) rows[0] = 0;
  for (int n = 0; n < N; ++n) {
    const int i = p->obs_sta_h[n], j = p->obs_pt_h[n], c = p->sta_cam_h[i];
    for (int r = 0; r < 2; ++r) {
      put(6 * i, 6);
      put(6 * S + 3 * j, 3);
      if (p->iop_mask_h[c] & 7u) put(iop_col[c], iop_w[c]);
      if (p->ap_mask_h[c]) put(ap_col[c], ap_w[c]);
      end_row();
    }
  }
  for (int j = 0; j < P; ++j)
    for (int r = 0; r < 3; ++r) {
      put(6 * S + 3 * j, 3);
      end_row();
    }
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < 3; ++r) {
      if (p->iop_mask_h[c] & 7u) put(iop_col[c], iop_w[c]);
      end_row();
    }
  if (num_rows) *num_rows = row;
  if (num_cols) *num_cols = col;
  if (nnz) *nnz = z;
  return 0;
}

// ---- multi-GPU ------------------------------------------------------------------------------
// Reduces a small host vector over the ranks through the device-side callback (op 0 sum, 1 max).
static int host_allreduce(ba_problem *p, std::vector<double> &v, int op) {
  if (v.empty()) return 0;
  if (v.size() > 39 * (size_t)std::max(p->S, 1)) return fail(BA_ERR_INVALID, "host_allreduce: vector too long");
  HIP_OK(hipMemcpy(p->d_F, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
  BA_TRY(p->allreduce(p->d_F, (int64_t)v.size(), op));
  HIP_OK(hipStreamSynchronize(p->stream));
  HIP_OK(hipMemcpy(v.data(), p->d_F, v.size() * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// What every rank must agree on before a sharded solve: which stations are observed at all,
// whether the block-sparse form is used, and the block pattern itself (the union over the shards,
// so that the value arrays of all ranks have the same layout and can be summed element-wise).
static int sync_shards(ba_problem *p) {
  if (!p->ar) return 0;
  const int S = p->S;
  if (S > 0) {
    std::vector<double> u(S);
    for (int i = 0; i < S; ++i) u[i] = p->sta_used_h[i];
    BA_TRY(host_allreduce(p, u, 1));
    for (int i = 0; i < S; ++i) p->sta_used_h[i] = u[i] > 0.0;
    HIP_OK(hipMemcpy(p->d_sta_used, p->sta_used_h.data(), p->sta_used_h.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  if (p->C > 0 && S > 0) {
    std::vector<double> c(std::min(p->C, 39 * S));
    for (size_t k = 0; k < c.size(); ++k) c[k] = p->cam_used_h[k];
    BA_TRY(host_allreduce(p, c, 1));
    for (size_t k = 0; k < c.size(); ++k)
      if ((c[k] > 0.0) != (p->cam_used_h[k] != 0))
        return fail(BA_ERR_INVALID, "camera %zu is observed on another rank but this rank's problem does not carry its columns: "
                    "create the problems of all ranks with ba_inputs.cam_in_use (the cameras that the whole network observes; "
                    "ba_net_shard fills it)", k);
  }
  if (S == 0) return 0;
  std::vector<double> flag(1, p->explicit_ok ? 0.0 : 1.0);
  BA_TRY(host_allreduce(p, flag, 1));
  if (flag[0] > 0.0) {
    p->explicit_ok = false;
    return 0;
  }
  // union of the neighbour lists, one allreduce(min) round per list position
  std::vector<int> cur(S, -1), pos(S);
  for (int i = 0; i < S; ++i) pos[i] = p->bsr_rowptr_h[i] + 1;  // skip the diagonal block
  std::vector<std::vector<int>> nb(S);
  std::vector<double> cand(S);
  for (;;) {
    for (int i = 0; i < S; ++i) {
      while (pos[i] < p->bsr_rowptr_h[i + 1] && p->bsr_col_h[pos[i]] <= cur[i]) ++pos[i];
      const int c = pos[i] < p->bsr_rowptr_h[i + 1] ? p->bsr_col_h[pos[i]] : S;
      cand[i] = -(double)c;
    }
    BA_TRY(host_allreduce(p, cand, 1));
    bool any = false;
    for (int i = 0; i < S; ++i) {
      const int c = (int)(-cand[i]);
      if (c < S) {
        nb[i].push_back(c);
        cur[i] = c;
        any = true;
      } else {
        cur[i] = S;
      }
    }
    if (!any) break;
  }
  std::vector<int> rowptr(S + 1, 0), col;
  for (int i = 0; i < S; ++i) rowptr[i + 1] = rowptr[i] + 1 + (int)nb[i].size();
  col.resize(rowptr[S]);
  for (int i = 0; i < S; ++i) {
    col[rowptr[i]] = i;
    std::copy(nb[i].begin(), nb[i].end(), col.begin() + rowptr[i] + 1);
  }
  std::vector<int2> dst(p->n_ublk);
  auto find = [&](int row, int c) {
    const int *b = col.data() + rowptr[row] + 1, *e = col.data() + rowptr[row + 1];
    return (int)(std::lower_bound(b, e, c) - col.data());
  };
  for (int u = 0; u < p->n_ublk; ++u) {
    const int lo = p->ub_lo_h[u], hi = p->ub_hi_h[u];
    dst[u] = (lo == hi) ? make_int2(rowptr[lo], -1) : make_int2(find(lo, hi), find(hi, lo));
  }
  if ((double)rowptr[S] * 288.0 > BSR_MAX_BYTES) {  // the same on every rank: the pattern is the union
    p->explicit_ok = false;
    return 0;
  }
  p->bsr_rowptr_h = rowptr;
  p->bsr_col_h = col;
  p->n_bsr = rowptr[S];
  BA_TRY(p->upload(&p->d_bsr_rowptr, rowptr));
  BA_TRY(p->upload(&p->d_bsr_col, col));
  BA_TRY(p->upload(&p->d_ublk_dst, dst));
  BA_TRY(alloc_reduced_sums(p));
  if (p->cg_graph) {
    hipGraphExecDestroy(p->cg_graph);
    if (p->cg_graph_tail) hipGraphExecDestroy(p->cg_graph_tail);
    p->cg_graph = p->cg_graph_tail = nullptr;
  }
  if (p->amg) {  // built on the old pattern
    amg_destroy(p->amg);
    p->amg = nullptr;
  }
  if (p->sparse) {
    if (p->stream_clear) (void)hipStreamSynchronize(p->stream_clear);
    sparse_chol_destroy(p->sparse);
    p->sparse = nullptr;
    p->sparse_clear_queued = false;
  }
  p->have_G = false;
  return 0;
}

int ba_reduction_timing(ba_problem *p, int32_t on) {
  if (!p) return fail(BA_ERR_INVALID, "ba_reduction_timing: null problem");
  p->time_reductions = on != 0;
  return 0;
}

int ba_reduction_stats(ba_problem *p, int64_t *count, double *bytes, double *seconds) {
  if (!p) return fail(BA_ERR_INVALID, "ba_reduction_stats: null problem");
  HIP_OK(hipStreamSynchronize(p->stream));
  p->fold_reduction_events();
  if (count) *count = p->n_reductions;
  if (bytes) *bytes = p->reduction_bytes;
  if (seconds) *seconds = p->reduction_seconds;
  p->n_reductions = 0;
  p->reduction_bytes = 0.0;
  p->reduction_seconds = 0.0;
  return 0;
}

int ba_device_count(int32_t *count) {
  if (!count) return fail(BA_ERR_INVALID, "ba_device_count: null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return 0;
}

int ba_set_device(int32_t device) {
  const hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(BA_ERR_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
  return 0;
}

int ba_shard_reduced_solve(ba_problem *p, int32_t rank, int32_t n_ranks, int32_t mode) {
  if (!p || n_ranks < 1 || rank < 0 || rank >= n_ranks || mode < -1 || mode > 1)
    return fail(BA_ERR_INVALID, "ba_shard_reduced_solve: bad argument");
  p->shard_rank = rank;
  p->shard_world = n_ranks;
  p->shard_mode = mode;
  return 0;
}

int ba_reduced_solve_rows(const ba_problem *p, int32_t *row_begin, int32_t *row_end) {
  if (!p || !row_begin || !row_end) return fail(BA_ERR_INVALID, "ba_reduced_solve_rows: null argument");
  *row_begin = p->last_row0;
  *row_end = p->last_row_end;
  return 0;
}

int ba_set_allreduce(ba_problem *p, ba_allreduce_fn fn, void *user) {
  if (!p) return fail(BA_ERR_INVALID, "ba_set_allreduce: null problem");
  p->ar = fn;
  p->ar_user = user;
  return sync_shards(p);
}

int ba_copy_to_host(void *host_dst, const double *device_src, int64_t count, void *hip_stream) {
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  HIP_OK(hipMemcpyAsync(host_dst, device_src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}

int ba_copy_to_device(double *device_dst, const void *host_src, int64_t count, void *hip_stream) {
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  HIP_OK(hipMemcpyAsync(device_dst, host_src, (size_t)count * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_OK(hipStreamSynchronize(s));
  return 0;
}

static int rccl_allreduce(void *user, double *buf, int64_t count, int32_t op, void *stream) {
  ba_problem *p = static_cast<ba_problem *>(user);
  if (!p->comm) return -1;
  ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, p->comm,
                                 static_cast<hipStream_t>(stream));
  // an error that an earlier collective ran into shows here as well (a peer that is gone, a failed transfer)
  ncclResult_t async_err = ncclSuccess;
  if (r == ncclSuccess && ncclCommGetAsyncError(p->comm, &async_err) == ncclSuccess && async_err != ncclSuccess) r = async_err;
  if (r != ncclSuccess) fprintf(stderr, "ba_gpu: ncclAllReduce: %s\n", ncclGetErrorString(r));
  return r == ncclSuccess ? 0 : -1;
}

int ba_rccl_unique_id(void *unique_id_128) {
  if (!unique_id_128) return fail(BA_ERR_INVALID, "ba_rccl_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return fail(BA_ERR_COMM, "ncclGetUniqueId failed");
  std::memcpy(unique_id_128, &id, sizeof id);
  return 0;
}

int ba_use_rccl(ba_problem *p, int32_t n_ranks, int32_t rank, const void *unique_id_128) {
  if (!p || !unique_id_128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(BA_ERR_INVALID, "ba_use_rccl: bad argument");
  ncclUniqueId id;
  std::memcpy(&id, unique_id_128, sizeof id);
  const ncclResult_t r = ncclCommInitRank(&p->comm, n_ranks, id, rank);
  if (r != ncclSuccess) return fail(BA_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
  p->shard_rank = rank;
  p->shard_world = n_ranks;
  return ba_set_allreduce(p, rccl_allreduce, p);
}

}  // extern "C"
