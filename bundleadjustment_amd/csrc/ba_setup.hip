// ba_setup.hip -- the index structures of a problem, built on the GPU.
//
// What the reference leaves to Ceres when it adds its residual blocks one by one (src/main.cpp:3830-4640) and what
// Ceres' ordering and symbolic factorisation then derive: here the observations sorted by object point, the same
// observations listed by station, and the pairs of observations of one point grouped by station pair (= the blocks of
// the reduced camera system).  All of it is sorting, counting and gathering, done with device-wide radix sorts and scans
// (hipCUB) and a few gather kernels.  Every sort is stable, so the order inside a group is the file order -- exactly the
// order the host-side construction in ba_api.hip produces, which stays in use for small networks.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "ba_device.h"
#include "ba_kernels.h"

namespace ba {

namespace {

#define SETUP_HIP(call)                                                                            \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      err = std::string(#call) + ": " + hipGetErrorString(e_);                                     \
      return -1;                                                                                   \
    }                                                                                              \
  } while (0)

inline int nblocks(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }
inline int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

// device memory of the set-up itself, released when it goes out of scope
struct Scratch {
  std::vector<void *> ptrs;
  ~Scratch() {
    for (void *q : ptrs) (void)hipFree(q);
  }
  template <class T>
  hipError_t get(T **p, size_t n) {
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(q);
    *p = static_cast<T *>(q);
    return e;
  }
};

__global__ void k_iota(int n, int *v) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) v[t] = t;
}

// key of a point for the internal order: more observations first, then by the lowest station that sees the point
// (points of one neighbourhood end up side by side, so the records a station sweep gathers cluster)
__global__ void k_degree_key(int P, const int *__restrict__ deg, const int *__restrict__ min_sta, int dmax, int sbits,
                             unsigned *__restrict__ key) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < P) key[j] = ((unsigned)(dmax - deg[j]) << sbits) | (sbits ? (unsigned)min_sta[j] : 0u);
}

// pt_int = inverse of pt_file; degrees in internal order (entry P = 0 closes the scan)
__global__ void k_point_order(int P, const int *__restrict__ pt_file, const int *__restrict__ deg, int *__restrict__ pt_int,
                              int *__restrict__ deg_sorted) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj < P) {
    const int j = pt_file[jj];
    pt_int[j] = jj;
    deg_sorted[jj] = deg[j];
  } else if (jj == P) {
    deg_sorted[P] = 0;
  }
}

__global__ void k_obs_key(int N, const int *__restrict__ obs_pt_file, const int *__restrict__ pt_int, unsigned *__restrict__ key) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) key[n] = (unsigned)pt_int[obs_pt_file[n]];
}

// observation records in point-sorted order: slot e holds file row perm[e]
__global__ void k_gather_obs(int N, const int *__restrict__ perm, const unsigned *__restrict__ key_sorted, const int *__restrict__ obs_sta_file,
                             const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ sx,
                             const double *__restrict__ sy, const double *__restrict__ cx, const double *__restrict__ cy,
                             double *__restrict__ obs, int *__restrict__ obs_pt, int *__restrict__ obs_sta, unsigned *__restrict__ sta_key) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const int n = perm[e];
  const int sta = obs_sta_file[n];
  double2 *o = reinterpret_cast<double2 *>(obs + (size_t)e * BA_OBS_STRIDE);
  o[0] = make_double2(x[n], y[n]);
  o[1] = make_double2(sx[n], sy[n]);
  o[2] = make_double2(cx[n], cy[n]);
  o[3] = make_double2(__hiloint2double(n, sta), 0.0);  // {station, file row} in the first eight bytes
  obs_pt[e] = (int)key_sorted[e];
  obs_sta[e] = sta;
  sta_key[e] = (unsigned)sta;
}

__global__ void k_gather_priors(int P, const int *__restrict__ pt_file, const double *__restrict__ prior_file,
                                const double *__restrict__ sigma_file, double *__restrict__ prior, double *__restrict__ isig) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= 3 * (int64_t)P) return;
  const int64_t src = 3 * (int64_t)pt_file[t / 3] + t % 3;
  prior[t] = prior_file[src];
  isig[t] = 1.0 / sigma_file[src];
}

// pairs of observations of one point: one entry per unordered pair, two (both orders) when the two share a station
__global__ void k_pair_count(int P, const int *__restrict__ pt_start, const int *__restrict__ obs_sta, long long *__restrict__ cnt) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj > P) return;
  long long c = 0;
  if (jj < P) {
    const int b0 = pt_start[jj], b1 = pt_start[jj + 1];
    for (int a = b0; a < b1; ++a)
      for (int b = a + 1; b < b1; ++b) c += (obs_sta[a] == obs_sta[b]) ? 2 : 1;
  }
  cnt[jj] = c;  // entry P = 0 closes the scan
}

__global__ void k_pair_emit(int P, int S, const int *__restrict__ pt_start, const int *__restrict__ obs_sta, const long long *__restrict__ off,
                            unsigned long long *__restrict__ key, unsigned long long *__restrict__ val) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj >= P) return;
  const int b0 = pt_start[jj], b1 = pt_start[jj + 1];
  long long q = off[jj];
  auto put = [&](int lo, int hi, int a, int b) {
    key[q] = (unsigned long long)lo * (unsigned long long)S + (unsigned long long)hi;
    val[q] = (unsigned long long)(unsigned)a | ((unsigned long long)(unsigned)b << 32);  // reads back as int2 {a, b}
    ++q;
  };
  for (int a = b0; a < b1; ++a)
    for (int b = a + 1; b < b1; ++b) {
      const int ia = obs_sta[a], ib = obs_sta[b];
      if (ia < ib) put(ia, ib, a, b);
      else if (ib < ia) put(ib, ia, b, a);
      else { put(ia, ia, a, b); put(ia, ia, b, a); }
    }
}

}  // namespace

// bits of the station index in the sort key of the point order (0: the key would not fit 32 bits, order by count only)
int point_order_station_bits(int S, int dmax) {
  const int sbits = bits_for((uint64_t)std::max(S, 1));
  return bits_for((uint64_t)dmax) + sbits <= 32 ? sbits : 0;
}

namespace {
struct Tuning {
  const char *name;
  double value;
  bool set;
};
Tuning g_tuning[] = {{"setup", -1.0, false},     {"pcg_graph", 1.0, false},     {"amg_levels", 3.0, false},  {"amg_theta", 0.5, false},
                     {"fused_sums", 1.0, false}, {"schur_windows", 1.0, false}, {"keyed_tiles", 16.0, false},
                     {"rows_solver", -1.0, false}};
}  // namespace

double tuning_value(const char *name, double dflt) {
  for (const Tuning &t : g_tuning)
    if (std::string(t.name) == name) return t.set ? t.value : dflt;
  return dflt;
}
bool tuning_set(const char *name, double value) {
  for (Tuning &t : g_tuning)
    if (std::string(t.name) == name) {
      t.value = value;
      t.set = true;
      return true;
    }
  return false;
}

bool setup_on_device(int64_t n_obs) {
  const int mode = (int)tuning_value("setup", -1.0);
  if (mode == 0) return false;
  if (mode == 1) return n_obs > 0;
  return n_obs >= (1 << 16);  // below that the transfers and launches cost more than the host loops
}

// Observations by point and by station, point records.  deg [P]: observations per point (host).  The arrays of `out`
// are allocated by the caller; pt_file_host [P] receives the internal point order.
int setup_sort_observations(hipStream_t s, int N, int S, int P, const int *obs_pt, const int *obs_sta, const double *const obs_cols[6],
                            const double *pt_prior, const double *pt_sigma, const int *deg, const int *min_sta, const SetupArrays &out,
                            int *pt_file_host, std::string &err) {
  Scratch sc;
  int *d_deg = nullptr, *d_min_sta = nullptr, *d_iota = nullptr, *d_pt_int = nullptr, *d_deg_sorted = nullptr, *d_obs_pt_file = nullptr, *d_obs_sta_file = nullptr,
      *d_perm = nullptr;
  unsigned *d_key = nullptr, *d_key2 = nullptr;
  double *d_col[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *d_prior_file = nullptr, *d_sigma_file = nullptr;
  const int M = std::max(N, P) + 1;
  SETUP_HIP(sc.get(&d_deg, (size_t)P));
  SETUP_HIP(sc.get(&d_min_sta, (size_t)P));
  SETUP_HIP(sc.get(&d_iota, (size_t)M));
  SETUP_HIP(sc.get(&d_pt_int, (size_t)P));
  SETUP_HIP(sc.get(&d_deg_sorted, (size_t)P + 1));
  SETUP_HIP(sc.get(&d_obs_pt_file, (size_t)N));
  SETUP_HIP(sc.get(&d_obs_sta_file, (size_t)N));
  SETUP_HIP(sc.get(&d_perm, (size_t)N));
  SETUP_HIP(sc.get(&d_key, (size_t)M));
  SETUP_HIP(sc.get(&d_key2, (size_t)M));
  for (int k = 0; k < 6; ++k) SETUP_HIP(sc.get(&d_col[k], (size_t)N));
  SETUP_HIP(sc.get(&d_prior_file, 3 * (size_t)P));
  SETUP_HIP(sc.get(&d_sigma_file, 3 * (size_t)P));
  SETUP_HIP(hipMemcpyAsync(d_deg, deg, (size_t)P * sizeof(int), hipMemcpyHostToDevice, s));
  SETUP_HIP(hipMemcpyAsync(d_min_sta, min_sta, (size_t)P * sizeof(int), hipMemcpyHostToDevice, s));
  SETUP_HIP(hipMemcpyAsync(d_obs_pt_file, obs_pt, (size_t)N * sizeof(int), hipMemcpyHostToDevice, s));
  SETUP_HIP(hipMemcpyAsync(d_obs_sta_file, obs_sta, (size_t)N * sizeof(int), hipMemcpyHostToDevice, s));
  for (int k = 0; k < 6; ++k) SETUP_HIP(hipMemcpyAsync(d_col[k], obs_cols[k], (size_t)N * sizeof(double), hipMemcpyHostToDevice, s));
  SETUP_HIP(hipMemcpyAsync(d_prior_file, pt_prior, 3 * (size_t)P * sizeof(double), hipMemcpyHostToDevice, s));
  SETUP_HIP(hipMemcpyAsync(d_sigma_file, pt_sigma, 3 * (size_t)P * sizeof(double), hipMemcpyHostToDevice, s));
  int dmax = 0;
  for (int j = 0; j < P; ++j) dmax = std::max(dmax, deg[j]);

  // one temporary buffer for all sorts and scans of this function
  size_t tmp_bytes = 0, need = 0;
  const int pbits = bits_for((uint64_t)std::max(P, 1)), sbits = bits_for((uint64_t)std::max(S, 1)), dbits = bits_for((uint64_t)dmax);
  const int obits = point_order_station_bits(S, dmax);  // station bits inside the key of the point order
  SETUP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_key, d_key2, d_iota, d_perm, P, 0, dbits + obits, s));
  tmp_bytes = std::max(tmp_bytes, need);
  SETUP_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_key, d_key2, d_iota, d_perm, N, 0, std::max(pbits, sbits), s));
  tmp_bytes = std::max(tmp_bytes, need);
  SETUP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_deg_sorted, out.pt_start, P + 1, s));
  tmp_bytes = std::max(tmp_bytes, need);
  void *d_tmp = nullptr;
  SETUP_HIP(sc.get(reinterpret_cast<char **>(&d_tmp), tmp_bytes));

  // ---- internal point order: by number of observations, descending, stable
  hipLaunchKernelGGL(k_iota, dim3(nblocks(M, 256)), dim3(256), 0, s, M, d_iota);
  if (P > 0) {
    hipLaunchKernelGGL(k_degree_key, dim3(nblocks(P, 256)), dim3(256), 0, s, P, d_deg, d_min_sta, dmax, obits, d_key);
    SETUP_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_key2, d_iota, out.pt_file, P, 0, dbits + obits, s));
  }
  hipLaunchKernelGGL(k_point_order, dim3(nblocks(P + 1, 256)), dim3(256), 0, s, P, out.pt_file, d_deg, d_pt_int, d_deg_sorted);
  SETUP_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_deg_sorted, out.pt_start, P + 1, s));
  if (P > 0) {
    SETUP_HIP(hipMemcpyAsync(pt_file_host, out.pt_file, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(k_gather_priors, dim3(nblocks(3 * (int64_t)P, 256)), dim3(256), 0, s, P, out.pt_file, d_prior_file, d_sigma_file,
                       out.pt_prior, out.pt_isig);
  }
  // ---- observations by internal point, file order inside a point
  if (N > 0) {
    hipLaunchKernelGGL(k_obs_key, dim3(nblocks(N, 256)), dim3(256), 0, s, N, d_obs_pt_file, d_pt_int, d_key);
    SETUP_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_key2, d_iota, d_perm, N, 0, pbits, s));
    hipLaunchKernelGGL(k_gather_obs, dim3(nblocks(N, 256)), dim3(256), 0, s, N, d_perm, d_key2, d_obs_sta_file, d_col[0], d_col[1], d_col[2],
                       d_col[3], d_col[4], d_col[5], out.obs, out.obs_pt, out.obs_sta, d_key);
    // ---- the same observations listed by station (ascending slot inside a station)
    SETUP_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_key2, d_iota, out.sta_list, N, 0, sbits, s));
  }
  SETUP_HIP(hipStreamSynchronize(s));
  SETUP_HIP(hipGetLastError());
  return 0;
}

// Pairs of observations grouped by station pair.  Returns 1 when there are more than pair_cap pairs (nothing built), 0 when
// built: *d_pairs (hipMalloc, the caller owns it) holds the pairs block after block, ub_lo/ub_hi the station pair of every
// block in ascending (lo, hi) order, cstart [blocks + 1] the first pair of every block.
int setup_pairs(hipStream_t s, int S, int P, const int *d_pt_start, const int *d_obs_sta, int64_t pair_cap, int2 **d_pairs, int64_t *n_pairs,
                std::vector<int> &ub_lo, std::vector<int> &ub_hi, std::vector<int> &cstart, std::string &err) {
  Scratch sc;
  *d_pairs = nullptr;
  *n_pairs = 0;
  ub_lo.clear(); ub_hi.clear(); cstart.assign(1, 0);
  if (S <= 0 || P <= 0) return 0;
  long long *d_cnt = nullptr, *d_off = nullptr;
  SETUP_HIP(sc.get(&d_cnt, (size_t)P + 1));
  SETUP_HIP(sc.get(&d_off, (size_t)P + 1));
  hipLaunchKernelGGL(k_pair_count, dim3(nblocks(P + 1, 256)), dim3(256), 0, s, P, d_pt_start, d_obs_sta, d_cnt);
  size_t need = 0;
  SETUP_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_cnt, d_off, P + 1, s));
  void *d_tmp0 = nullptr;
  SETUP_HIP(sc.get(reinterpret_cast<char **>(&d_tmp0), need));
  SETUP_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp0, need, d_cnt, d_off, P + 1, s));
  long long tot = 0;
  SETUP_HIP(hipMemcpyAsync(&tot, d_off + P, sizeof tot, hipMemcpyDeviceToHost, s));
  SETUP_HIP(hipStreamSynchronize(s));
  if (tot > pair_cap || tot >= 2000000000LL) return 1;
  *n_pairs = tot;
  if (tot == 0) return 0;
  unsigned long long *d_key = nullptr, *d_key2 = nullptr, *d_val = nullptr, *d_val2 = nullptr, *d_uniq = nullptr;
  int *d_runs = nullptr, *d_nruns = nullptr;
  SETUP_HIP(sc.get(&d_key, (size_t)tot));
  SETUP_HIP(sc.get(&d_key2, (size_t)tot));
  SETUP_HIP(sc.get(&d_val, (size_t)tot));
  SETUP_HIP(hipMalloc(reinterpret_cast<void **>(&d_val2), (size_t)tot * sizeof(unsigned long long)));  // becomes *d_pairs
  hipLaunchKernelGGL(k_pair_emit, dim3(nblocks(P, 256)), dim3(256), 0, s, P, S, d_pt_start, d_obs_sta, d_off, d_key, d_val);
  const int kbits = bits_for((uint64_t)S * (uint64_t)S);
  size_t tmp_bytes = 0;
  auto fail_free = [&]() { (void)hipFree(d_val2); };
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, d_key2, d_val, d_val2, (int)tot, 0, kbits, s);
  void *d_tmp = nullptr;
  if (e == hipSuccess) e = sc.get(reinterpret_cast<char **>(&d_tmp), tmp_bytes);
  if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_key2, d_val, d_val2, (int)tot, 0, kbits, s);
  // blocks = runs of equal keys
  const size_t max_runs = (size_t)std::min<long long>(tot, (long long)S * S);
  if (e == hipSuccess) e = sc.get(&d_uniq, max_runs);
  if (e == hipSuccess) e = sc.get(&d_runs, max_runs);
  if (e == hipSuccess) e = sc.get(&d_nruns, 1);
  size_t rle_bytes = 0;
  void *d_tmp2 = nullptr;
  if (e == hipSuccess) e = hipcub::DeviceRunLengthEncode::Encode(nullptr, rle_bytes, d_key2, d_uniq, d_runs, d_nruns, (int)tot, s);
  if (e == hipSuccess) e = sc.get(reinterpret_cast<char **>(&d_tmp2), rle_bytes);
  if (e == hipSuccess) e = hipcub::DeviceRunLengthEncode::Encode(d_tmp2, rle_bytes, d_key2, d_uniq, d_runs, d_nruns, (int)tot, s);
  int nu = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&nu, d_nruns, sizeof nu, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  std::vector<unsigned long long> uniq((size_t)std::max(nu, 0));
  std::vector<int> runs((size_t)std::max(nu, 0));
  if (e == hipSuccess && nu > 0) e = hipMemcpy(uniq.data(), d_uniq, (size_t)nu * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e == hipSuccess && nu > 0) e = hipMemcpy(runs.data(), d_runs, (size_t)nu * sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    fail_free();
    err = std::string("pair set-up: ") + hipGetErrorString(e);
    return -1;
  }
  ub_lo.resize(nu); ub_hi.resize(nu); cstart.resize((size_t)nu + 1);
  long long run = 0;
  for (int u = 0; u < nu; ++u) {
    ub_lo[u] = (int)(uniq[u] / (unsigned long long)S);
    ub_hi[u] = (int)(uniq[u] % (unsigned long long)S);
    run += runs[u];
    cstart[u + 1] = (int)run;
  }
  *d_pairs = reinterpret_cast<int2 *>(d_val2);
  return 0;
}

}  // namespace ba
