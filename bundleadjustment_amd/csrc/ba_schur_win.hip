// ba_schur_win.hip -- assembly of the off-diagonal blocks of the reduced camera system, window by window.
//
// S_ss' = - sum over the points seen by stations s and s' of Y_a Y_b^T  (Y = Js^T Jp L per observation, 6 x 3).
// k_schur_blocks (ba_schur.hip) gives every block to one wave, which gathers the records of its pairs from wherever they
// lie: a record takes part in one pair per other observation of its point, so it is fetched several times, by waves that
// run far apart.  Here the observations are cut into windows of consecutive points whose records fit the LDS of one
// workgroup.  A workgroup loads a window's records once, as one contiguous run, and forms all pairs of the window from
// LDS.  The pairs of a window are sorted by block; a (window, block) segment is summed by one wave in a fixed order
// and leaves as one partial block (36 doubles).  A second kernel adds the partial blocks of every block in ascending
// window order and writes the block and its mirror image.  Every record is read from memory once; what is added is
// the partial blocks (a tenth of the records' bytes on a network whose points are ordered by neighbourhood).
// All lists are built once per problem on the device (sorts, scans); nothing depends on the numbers, and every sum has
// one fixed order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "ba_device.h"
#include "ba_kernels.h"

namespace ba {

namespace {

#define SW_HIP(call)                                                      \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) {                                               \
      err = std::string(#call) + ": " + hipGetErrorString(e_);            \
      return -1;                                                          \
    }                                                                     \
  } while (0)

inline int nblocks(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }
inline int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

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

// index of the last entry of the ascending list v[0..n) that is <= x (v[0] <= x is given)
__device__ __forceinline__ int last_le(const int *__restrict__ v, int n, int x) {
  int lo = 0, hi = n;  // v[lo] <= x < v[hi] (v[n] = +inf)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (v[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}
// first index k in [0, n] with v[k] >= x
__device__ __forceinline__ int first_ge(const int *__restrict__ v, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// window and block of every pair of the block-sorted list
__global__ void k_sw_keys(int64_t n_pairs, const int2 *__restrict__ pairs, int n_win, const int *__restrict__ win_obs0, int nu,
                          const int *__restrict__ cstart, unsigned *__restrict__ wkey, int *__restrict__ ublk, int *__restrict__ iota) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= n_pairs) return;
  wkey[q] = (unsigned)last_le(win_obs0, n_win, pairs[q].x);
  ublk[q] = last_le(cstart, nu, (int)q);
  iota[q] = (int)q;
}

// the pairs in (window, block) order with indices relative to the window's first observation; `first` marks where
// a new (window, block) segment begins
__global__ void k_sw_gather(int64_t n_pairs, const int *__restrict__ perm, const unsigned *__restrict__ wkey_sorted, const int2 *__restrict__ pairs,
                            const int *__restrict__ ublk, const int *__restrict__ win_obs0, unsigned *__restrict__ pairs16,
                            int *__restrict__ useq) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= n_pairs) return;
  const int q = perm[k];
  const int2 pr = pairs[q];
  const int o0 = win_obs0[wkey_sorted[k]];
  pairs16[k] = (unsigned)(pr.x - o0) | ((unsigned)(pr.y - o0) << 16);
  useq[k] = ublk[q];
}
__global__ void k_sw_flags(int64_t n_pairs, const unsigned *__restrict__ wkey_sorted, const int *__restrict__ useq, int *__restrict__ first) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= n_pairs) return;
  first[k] = (k == 0 || wkey_sorted[k] != wkey_sorted[k - 1] || useq[k] != useq[k - 1]) ? 1 : 0;
}
__global__ void k_sw_segments(int64_t n_pairs, const int *__restrict__ first, const int *__restrict__ scan, const unsigned *__restrict__ wkey_sorted,
                              const int *__restrict__ useq, int *__restrict__ seg_q0, int *__restrict__ seg_blk, int *__restrict__ seg_win,
                              unsigned *__restrict__ seg_key, int *__restrict__ seg_iota) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= n_pairs) return;
  if (first[k]) {
    const int sgm = scan[k] - 1;  // inclusive scan of the flags
    seg_q0[sgm] = (int)k;
    seg_blk[sgm] = useq[k];
    seg_win[sgm] = (int)wkey_sorted[k];
    seg_key[sgm] = (unsigned)useq[k];
    seg_iota[sgm] = sgm;
  }
}
__global__ void k_sw_win_pairs(int n, const int *__restrict__ win_seg0, const int *__restrict__ seg_q0, int *__restrict__ win_pair0) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n) win_pair0[w] = seg_q0[win_seg0[w]];
}
// out[x] = first index with v[.] >= x, for x = 0 .. n_out - 1 (v ascending, n entries)
__global__ void k_sw_bounds(int n_out, const int *__restrict__ v, int n, int *__restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < n_out) out[x] = first_ge(v, n, x);
}
__global__ void k_sw_bounds_u(int n_out, const unsigned *__restrict__ v, int n, int *__restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= n_out) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v[mid] < (unsigned)x) lo = mid + 1; else hi = mid;
  }
  out[x] = lo;
}

// ------------------------------------------------------------------------------------------
// the assembly kernel
// ------------------------------------------------------------------------------------------
#define SW_THREADS 512
#define SW_WAVES (SW_THREADS / 64)
#define SW_NPRE ((BA_SW_WCAP * BA_Y_STRIDE / 2 + SW_THREADS - 1) / SW_THREADS)  // double2 loads per thread and window

// Rows 3h .. 3h+2 of a 6 x 3 record in LDS (18 doubles, 16-byte aligned): nine doubles that start on a 16-byte
// boundary (h = 0) or 8 bytes past one (h = 1): four 16-byte reads and one 8-byte read either way
__device__ __forceinline__ void lds_half(const double *rec, int h, double (&v)[3][3]) {
  const double *p = rec + 9 * h;
  const double2 *p2 = reinterpret_cast<const double2 *>(p + h);
  const double2 d0 = p2[0], d1 = p2[1], d2 = p2[2], d3 = p2[3];
  const double s = p[h ? 0 : 8];
  const double f[9] = {h ? s : d0.x, h ? d0.x : d0.y, h ? d0.y : d1.x, h ? d1.x : d1.y, h ? d1.y : d2.x,
                       h ? d2.x : d2.y, h ? d2.y : d3.x, h ? d3.x : d3.y, h ? d3.y : s};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) v[i][j] = f[3 * i + j];
}

// Sum over the sixteen quads of a wave of up to sixteen values per lane, the lanes of a quad kept apart: on return
// lane l holds the total, over the lanes with the same l & 3, of value number 8 b5 + 4 b4 + 2 b3 + b2 (b = bits of l).
// Each step halves what a lane holds and doubles the lanes summed, so sixteen values cost fifteen exchanges.
__device__ __forceinline__ double quads_sum16(const double (&v)[16], int lane) {
  double a[8], b[4], c[2];
  const bool h3 = lane & 8, h2 = lane & 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = swap_add32(v[k], v[8 + k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = swap_add16(a[k], a[4 + k]);
#pragma unroll
  for (int k = 0; k < 2; ++k) c[k] = (h3 ? b[2 + k] : b[k]) + __shfl_xor(h3 ? b[k] : b[2 + k], 8, 64);
  return (h2 ? c[1] : c[0]) + __shfl_xor(h2 ? c[0] : c[1], 4, 64);
}

#define SW_PCAP BA_SW_PCAP
#define SW_SEGCAP 447  // segment offsets of a window kept in LDS (a window with more reads the rest from memory)

// One workgroup per window, two workgroups per CU (their LDS allows it): while one waits for its window's records the
// other multiplies.  A workgroup copies the window's records, pair list and segment offsets -- each one contiguous
// run -- into LDS, and the pair loop reads nothing else.
__global__ void __launch_bounds__(SW_THREADS)
k_schur_windows(SchurWinDev a, const double *__restrict__ Y) {
  __shared__ __attribute__((aligned(16))) double ly[BA_SW_WCAP * BA_Y_STRIDE];
  __shared__ __attribute__((aligned(16))) double lblk[SW_WAVES][36];
  __shared__ unsigned lpairs[SW_PCAP];
  __shared__ int lseg[SW_SEGCAP + 1];
  __shared__ int lnext;  // the next segment of the window that no wave has taken yet
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane >> 2, qa = (lane >> 1) & 1, qb = lane & 1;
  const int w = blockIdx.x;
  const int o0 = a.win_obs0[w], o1 = a.win_obs0[w + 1];
  const int s0 = a.win_seg0[w], s1 = a.win_seg0[w + 1];
  const int p0 = a.win_pair0[w], p1 = a.win_pair0[w + 1];
  {
    const int n2 = (o1 - o0) * (BA_Y_STRIDE / 2);
    const double2 *src = reinterpret_cast<const double2 *>(Y + (int64_t)BA_Y_STRIDE * o0);
    double2 *dst = reinterpret_cast<double2 *>(ly);
    double2 pre[SW_NPRE];
#pragma unroll
    for (int k = 0; k < SW_NPRE; ++k) {
      const int i = threadIdx.x + k * SW_THREADS;
      pre[k] = i < n2 ? src[i] : make_double2(0.0, 0.0);
    }
    unsigned prp[SW_PCAP / SW_THREADS];
#pragma unroll
    for (int k = 0; k < SW_PCAP / SW_THREADS; ++k) {
      const int i = threadIdx.x + k * SW_THREADS;
      prp[k] = p0 + i < p1 ? a.pairs16[p0 + i] : 0xFFFFFFFFu;
    }
    int prs = 0;
    if ((int)threadIdx.x <= SW_SEGCAP) prs = (s0 + (int)threadIdx.x <= s1) ? a.seg_q0[s0 + threadIdx.x] : 0;
#pragma unroll
    for (int k = 0; k < SW_NPRE; ++k) {
      const int i = threadIdx.x + k * SW_THREADS;
      if (i < n2) dst[i] = pre[k];
    }
#pragma unroll
    for (int k = 0; k < SW_PCAP / SW_THREADS; ++k) lpairs[threadIdx.x + k * SW_THREADS] = prp[k];
    if ((int)threadIdx.x <= SW_SEGCAP) lseg[threadIdx.x] = prs;
    if (threadIdx.x == 0) lnext = SW_WAVES;
  }
  __syncthreads();
  // The waves take the window's segments as they get free (the segments differ in length): a segment is summed by
  // whichever wave takes it, in the order of its pairs, so the result does not depend on who did.
  for (int r = wv; s0 + r < s1;) {
    const int sgm = s0 + r;
    const int q0 = (r <= SW_SEGCAP ? lseg[r] : a.seg_q0[sgm]) - p0, q1 = (r + 1 <= SW_SEGCAP ? lseg[r + 1] : a.seg_q0[sgm + 1]) - p0;
    double acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = 0.0;
    // sixteen pairs per step, four lanes per pair (each owns a 3 x 3 quadrant of Y_a Y_b^T)
    for (int q = q0; q < q1; q += 16) {
      const unsigned pr = (q + g < q1) ? lpairs[q + g] : 0xFFFFFFFFu;
      if (pr != 0xFFFFFFFFu) {
        double av[3][3], bv[3][3];
        lds_half(ly + BA_Y_STRIDE * (int)(pr & 0xFFFFu), qa, av);
        lds_half(ly + BA_Y_STRIDE * (int)(pr >> 16), qb, bv);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][j] += av[i][0] * bv[j][0] + av[i][1] * bv[j][1] + av[i][2] * bv[j][2];
      }
    }
    // fold the sixteen partial quadrants (lanes of equal (qa, qb) across the wave): afterwards the quad number k < 9
    // holds entry k of its lanes' quadrants
    double *sb = lblk[wv];
    {
      double v16[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v16[k] = k < 9 ? acc[k < 9 ? k / 3 : 0][k < 9 ? k % 3 : 0] : 0.0;
      const double tot = quads_sum16(v16, lane);
      const int k = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
      if (k < 9) sb[6 * (3 * qa + k / 3) + 3 * qb + k % 3] = tot;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 18) *reinterpret_cast<double2 *>(a.part + 36 * (int64_t)sgm + 2 * lane) = *reinterpret_cast<const double2 *>(sb + 2 * lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int grabbed = 0;
    if (lane == 0) grabbed = atomicAdd(&lnext, 1);
    r = __builtin_amdgcn_readfirstlane(grabbed);
  }
}

// block u = - sum of its partial blocks, in ascending window order; the block and its mirror image
__global__ void __launch_bounds__(256)
k_schur_win_fold(int nu, const int *__restrict__ blk_seg0, const int *__restrict__ blk_segs, const double *__restrict__ part,
                 const int2 *__restrict__ dst, double *__restrict__ val) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= 36 * (int64_t)nu) return;
  const int u = (int)(t / 36), e = (int)(t - 36 * (int64_t)u);
  const int q0 = blk_seg0[u], q1 = blk_seg0[u + 1];
  double s = 0.0;
  int sg_next = q0 < q1 ? blk_segs[q0] : 0;
  for (int q = q0; q < q1; ++q) {
    const int sg = sg_next;
    if (q + 1 < q1) sg_next = blk_segs[q + 1];
    s += part[36 * (int64_t)sg + e];
  }
  const int2 d = dst[u];
  if (d.x >= 0) val[36 * (int64_t)d.x + e] = -s;
  if (d.y >= 0) val[36 * (int64_t)d.y + 6 * (e % 6) + e / 6] = -s;
}

}  // namespace

int schur_win_wcap() { return BA_SW_WCAP; }

// Builds the lists.  pt_start: host copy of the first observation of every point (P + 1 entries); d_pairs / cstart: the
// pairs grouped by block and the first pair of every block, as the block-by-block kernel uses them.
// Returns 0 when built (the device arrays of `out` are appended to `owned`), 1 when this network is better served block
// by block (a point with more observations than a window holds, or partial blocks that would outweigh what they
// save), -1 on an error (err).
int setup_schur_windows(hipStream_t s, int64_t N, int P, const int *pt_start, const int2 *d_pairs, int64_t n_pairs, const std::vector<int> &cstart,
                        SchurWinDev &out, std::vector<void *> &owned, std::string &err) {
  out = SchurWinDev{};
  const int nu = (int)cstart.size() - 1;
  if (N <= 0 || P <= 0 || n_pairs <= 0 || nu <= 0 || n_pairs >= 0x7FFFFFFF) return 1;
  // windows of whole points, at most BA_SW_WCAP observations each
  std::vector<int> win_obs0(1, 0);
  {
    // (and at most BA_SW_PCAP pairs: d (d - 1) per point bounds them, two observations of one station counting twice)
    int count = 0;
    int64_t npair = 0;
    for (int jj = 0; jj < P; ++jj) {
      const int d = pt_start[jj + 1] - pt_start[jj];
      const int64_t dp = (int64_t)d * (d - 1);
      if (d > BA_SW_WCAP || dp > BA_SW_PCAP) return 1;
      if (count + d > BA_SW_WCAP || npair + dp > BA_SW_PCAP) {
        win_obs0.push_back(pt_start[jj]);
        count = 0;
        npair = 0;
      }
      count += d;
      npair += dp;
    }
    win_obs0.push_back(pt_start[P]);
  }
  const int n_win = (int)win_obs0.size() - 1;
  Scratch sc;
  int *d_win_obs0 = nullptr, *d_cstart = nullptr, *d_ublk = nullptr, *d_iota = nullptr, *d_perm = nullptr, *d_useq = nullptr, *d_first = nullptr,
      *d_scan = nullptr;
  unsigned *d_wkey = nullptr, *d_wkey2 = nullptr, *d_pairs16 = nullptr;
  SW_HIP(hipMalloc(reinterpret_cast<void **>(&d_win_obs0), (size_t)(n_win + 1) * sizeof(int)));
  SW_HIP(hipMalloc(reinterpret_cast<void **>(&d_pairs16), (size_t)n_pairs * sizeof(unsigned)));
  std::vector<void *> mine = {d_win_obs0, d_pairs16};
  auto give_up = [&](int rc) {
    for (void *q : mine) (void)hipFree(q);
    return rc;
  };
#define SW_TRY(call)                                                      \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) {                                               \
      err = std::string(#call) + ": " + hipGetErrorString(e_);            \
      return give_up(-1);                                                 \
    }                                                                     \
  } while (0)
  SW_TRY(sc.get(&d_cstart, (size_t)nu + 1));
  SW_TRY(sc.get(&d_ublk, (size_t)n_pairs));
  SW_TRY(sc.get(&d_iota, (size_t)n_pairs));
  SW_TRY(sc.get(&d_perm, (size_t)n_pairs));
  SW_TRY(sc.get(&d_useq, (size_t)n_pairs));
  SW_TRY(sc.get(&d_first, (size_t)n_pairs));
  SW_TRY(sc.get(&d_scan, (size_t)n_pairs));
  SW_TRY(sc.get(&d_wkey, (size_t)n_pairs));
  SW_TRY(sc.get(&d_wkey2, (size_t)n_pairs));
  SW_TRY(hipMemcpyAsync(d_win_obs0, win_obs0.data(), (size_t)(n_win + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  SW_TRY(hipMemcpyAsync(d_cstart, cstart.data(), (size_t)(nu + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  const int np = (int)n_pairs;
  hipLaunchKernelGGL(k_sw_keys, dim3(nblocks(n_pairs, 256)), dim3(256), 0, s, n_pairs, d_pairs, n_win, d_win_obs0, nu, d_cstart, d_wkey, d_ublk, d_iota);
  size_t tmp_bytes = 0, need = 0;
  SW_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_wkey, d_wkey2, d_iota, d_perm, np, 0, bits_for((uint64_t)n_win), s));
  tmp_bytes = std::max(tmp_bytes, need);
  SW_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, need, d_first, d_scan, np, s));
  tmp_bytes = std::max(tmp_bytes, need);
  void *d_tmp = nullptr;
  SW_TRY(sc.get(reinterpret_cast<char **>(&d_tmp), tmp_bytes));
  SW_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_wkey, d_wkey2, d_iota, d_perm, np, 0, bits_for((uint64_t)n_win), s));
  hipLaunchKernelGGL(k_sw_gather, dim3(nblocks(n_pairs, 256)), dim3(256), 0, s, n_pairs, d_perm, d_wkey2, d_pairs, d_ublk, d_win_obs0, d_pairs16, d_useq);
  hipLaunchKernelGGL(k_sw_flags, dim3(nblocks(n_pairs, 256)), dim3(256), 0, s, n_pairs, d_wkey2, d_useq, d_first);
  SW_TRY(hipcub::DeviceScan::InclusiveSum(d_tmp, tmp_bytes, d_first, d_scan, np, s));
  int n_seg = 0;
  SW_TRY(hipMemcpyAsync(&n_seg, d_scan + (np - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  SW_TRY(hipStreamSynchronize(s));
  // partial blocks are written and read once each: they must stay a small part of what the windows save (the records
  // are read once instead of several times)
  if (n_seg <= 0 || 288.0 * (double)n_seg > 0.30 * 144.0 * (double)N) return give_up(1);
  int *d_seg_q0 = nullptr, *d_seg_blk = nullptr, *d_seg_win = nullptr, *d_seg_iota = nullptr, *d_win_seg0 = nullptr, *d_blk_seg0 = nullptr,
      *d_blk_segs = nullptr, *d_win_pair0 = nullptr;
  unsigned *d_seg_key = nullptr, *d_seg_key2 = nullptr;
  double *d_part = nullptr;
  auto own = [&](void **p, size_t bytes) -> hipError_t {
    const hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
    if (e == hipSuccess) mine.push_back(*p);
    return e;
  };
  SW_TRY(own(reinterpret_cast<void **>(&d_seg_q0), (size_t)(n_seg + 1) * sizeof(int)));
  SW_TRY(own(reinterpret_cast<void **>(&d_win_seg0), (size_t)(n_win + 1) * sizeof(int)));
  SW_TRY(own(reinterpret_cast<void **>(&d_win_pair0), (size_t)(n_win + 1) * sizeof(int)));
  SW_TRY(own(reinterpret_cast<void **>(&d_blk_seg0), (size_t)(nu + 1) * sizeof(int)));
  SW_TRY(own(reinterpret_cast<void **>(&d_blk_segs), (size_t)n_seg * sizeof(int)));
  SW_TRY(own(reinterpret_cast<void **>(&d_part), 36 * (size_t)n_seg * sizeof(double)));
  SW_TRY(sc.get(&d_seg_blk, (size_t)n_seg));
  SW_TRY(sc.get(&d_seg_win, (size_t)n_seg));
  SW_TRY(sc.get(&d_seg_iota, (size_t)n_seg));
  SW_TRY(sc.get(&d_seg_key, (size_t)n_seg));
  SW_TRY(sc.get(&d_seg_key2, (size_t)n_seg));
  hipLaunchKernelGGL(k_sw_segments, dim3(nblocks(n_pairs, 256)), dim3(256), 0, s, n_pairs, d_first, d_scan, d_wkey2, d_useq, d_seg_q0, d_seg_blk, d_seg_win,
                     d_seg_key, d_seg_iota);
  SW_TRY(hipMemcpyAsync(d_seg_q0 + n_seg, &np, sizeof(int), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_sw_bounds, dim3(nblocks(n_win + 1, 256)), dim3(256), 0, s, n_win + 1, d_seg_win, n_seg, d_win_seg0);
  hipLaunchKernelGGL(k_sw_win_pairs, dim3(nblocks(n_win + 1, 256)), dim3(256), 0, s, n_win + 1, d_win_seg0, d_seg_q0, d_win_pair0);
  // the segments of every block, in ascending (window) order
  size_t need2 = 0;
  SW_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need2, d_seg_key, d_seg_key2, d_seg_iota, d_blk_segs, n_seg, 0, bits_for((uint64_t)nu), s));
  void *d_tmp2 = nullptr;
  SW_TRY(sc.get(reinterpret_cast<char **>(&d_tmp2), need2));
  SW_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp2, need2, d_seg_key, d_seg_key2, d_seg_iota, d_blk_segs, n_seg, 0, bits_for((uint64_t)nu), s));
  hipLaunchKernelGGL(k_sw_bounds_u, dim3(nblocks(nu + 1, 256)), dim3(256), 0, s, nu + 1, d_seg_key2, n_seg, d_blk_seg0);
  SW_TRY(hipStreamSynchronize(s));
  SW_TRY(hipGetLastError());
#undef SW_TRY
  out.n_win = n_win; out.n_seg = n_seg; out.nu = nu;
  out.win_obs0 = d_win_obs0; out.win_seg0 = d_win_seg0; out.win_pair0 = d_win_pair0; out.seg_q0 = d_seg_q0; out.pairs16 = d_pairs16;
  out.blk_seg0 = d_blk_seg0; out.blk_segs = d_blk_segs; out.part = d_part;
  for (void *q : mine) owned.push_back(q);
  return 0;
}

int schur_windows_grid(const SchurWinDev &a) { return a.n_win; }
void launch_schur_windows(hipStream_t s, const SchurWinDev &a, const double *Y, const int2 *dst, double *val) {
  if (a.n_win <= 0) return;
  const int grid = schur_windows_grid(a);
  hipLaunchKernelGGL(k_schur_windows, dim3(grid), dim3(SW_THREADS), 0, s, a, Y);
  hipLaunchKernelGGL(k_schur_win_fold, dim3(nblocks(36 * (int64_t)a.nu, 256)), dim3(256), 0, s, a.nu, a.blk_seg0, a.blk_segs, a.part, dst, val);
}

}  // namespace ba
