// ba_cov_select.hip -- a-posteriori variances of the stations, cameras and points when the reduced camera system is too
// large to be inverted as one dense matrix.
//
// The reference takes the covariance blocks it prints from Ceres (src/main.cpp:5261-5279, 5443-5565): only the diagonal
// blocks of the inverse are ever asked for.  With the points eliminated, those blocks are entries of S^-1 that sit on the
// block pattern of S itself (station pairs that share a point) plus the camera border.  They are computed here by one
// level of nested dissection with dense blocks, everything on the matrix cores:
//   * the station graph (the block pattern of S) is cut by breadth-first level sets into clusters of at most
//     max_cluster stations; the stations of the cuts and the camera columns form the boundary B;
//   * per cluster k with interior I_k:  A_kk^-1 (dense Cholesky and inverse),  X_k = A_kk^-1 A_kB,  and its share of the
//     boundary complement  T = A_BB - sum_k A_Bk X_k;
//   * T^-1 (dense) is S^-1 on the boundary;
//   * per cluster again:  S^-1[I_k, B_k] = -X_k T^-1[B_k, B_k],   S^-1[I_k, I_k] = A_kk^-1 + X_k T^-1[B_k, B_k] X_k^T
//     (only the boundary stations B_k next to the cluster have non-zero columns in A_kB).
// A point's stations lie in one cluster's I_k and B_k, or all in B, so the local matrix of that cluster (or T^-1) holds
// every block its variance needs; the point kernel is the one of the dense route with a station -> local index table.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "ba_kernels.h"

namespace ba {

namespace {

// ------------------------------------------------------------------------------------------------------------------
// host: the dissection
// ------------------------------------------------------------------------------------------------------------------
struct Dissection {
  std::vector<int> part;                   // per station: cluster, -1 boundary, -2 not in use
  std::vector<std::vector<int>> interior;  // stations of each cluster
  std::vector<int> boundary;               // boundary stations, position = index in T
  std::vector<std::vector<int>> next_to;   // per cluster: positions (in boundary) of the boundary stations adjacent to it
};

bool dissect(int S, const std::vector<int> &rowptr, const std::vector<int> &col, const std::vector<int> &used, int max_cluster,
             Dissection &d) {
  d.part.assign((size_t)S, -2);
  std::vector<int> mark((size_t)S, -1), level((size_t)S, 0), queue;
  std::vector<std::vector<int>> todo;
  {
    std::vector<int> all;
    for (int i = 0; i < S; ++i)
      if (used[i]) all.push_back(i);
    if (!all.empty()) todo.push_back(std::move(all));
  }
  int token = 0;
  // breadth-first search inside the set marked with `tok`, from `start`; returns the visiting order in queue, levels in level
  auto bfs = [&](int start, int tok, int seen_tok) {
    queue.clear();
    queue.push_back(start);
    mark[start] = seen_tok;
    level[start] = 0;
    for (size_t h = 0; h < queue.size(); ++h) {
      const int i = queue[h];
      for (int b = rowptr[i]; b < rowptr[i + 1]; ++b) {
        const int j = col[b];
        if (mark[j] != tok) continue;
        mark[j] = seen_tok;
        level[j] = level[i] + 1;
        queue.push_back(j);
      }
    }
  };
  while (!todo.empty()) {
    std::vector<int> V = std::move(todo.back());
    todo.pop_back();
    if ((int)V.size() <= max_cluster) {
      const int k = (int)d.interior.size();
      for (int i : V) d.part[i] = k;
      d.interior.push_back(std::move(V));
      continue;
    }
    // connected component of the first station
    const int tok = token, seen = token + 1, again = token + 2;
    token += 3;
    for (int i : V) mark[i] = tok;
    bfs(V[0], tok, seen);
    if (queue.size() < V.size()) {
      // not connected: the component and the rest go on separately, no cut needed
      std::vector<int> comp(queue), rest;
      for (int i : V)
        if (mark[i] == tok) rest.push_back(i);
      todo.push_back(std::move(comp));
      todo.push_back(std::move(rest));
      continue;
    }
    // level sets from a station far from the first one
    const int far = queue.back();
    bfs(far, seen, again);
    const int n_lev = level[queue.back()] + 1;
    if (n_lev < 3) return false;  // no level set separates anything
    std::vector<int> count((size_t)n_lev, 0);
    for (int i : queue) count[level[i]]++;
    int best = -1;
    {
      int64_t before = 0;
      const int64_t n = (int64_t)V.size();
      for (int l = 0; l < n_lev; ++l) {
        const bool balanced = 10 * before >= 3 * n && 10 * before <= 7 * n && l >= 1 && l <= n_lev - 2;
        if (balanced && (best < 0 || count[l] < count[best])) best = l;
        before += count[l];
      }
      if (best < 0) {
        // no level inside the balance window (a few very large levels): take the level where the running count crosses the half
        before = 0;
        for (int l = 0; l < n_lev; ++l) {
          if (l >= 1 && l <= n_lev - 2 && 2 * (before + count[l]) >= n) { best = l; break; }
          before += count[l];
        }
        if (best < 0) best = n_lev - 2;
      }
    }
    std::vector<int> lo, hi;
    for (int i : queue) {
      if (level[i] < best) lo.push_back(i);
      else if (level[i] > best) hi.push_back(i);
      else {
        // a station of the cut level that touches nothing beyond it stays with the near side
        bool touches = false;
        for (int b = rowptr[i]; b < rowptr[i + 1] && !touches; ++b) {
          const int j = col[b];
          touches = mark[j] == again && level[j] > best;
        }
        if (touches) d.part[i] = -1;
        else lo.push_back(i);
      }
    }
    if (lo.empty() || hi.empty()) return false;
    todo.push_back(std::move(lo));
    todo.push_back(std::move(hi));
  }
  std::vector<int> bpos((size_t)S, -1);
  for (int i = 0; i < S; ++i)
    if (d.part[i] == -1) { bpos[i] = (int)d.boundary.size(); d.boundary.push_back(i); }
  d.next_to.assign(d.interior.size(), {});
  std::vector<int> stamp(d.boundary.size(), -1);
  for (size_t k = 0; k < d.interior.size(); ++k) {
    for (int i : d.interior[k])
      for (int b = rowptr[i]; b < rowptr[i + 1]; ++b) {
        const int j = col[b];
        if (d.part[j] == -1 && stamp[bpos[j]] != (int)k) { stamp[bpos[j]] = (int)k; d.next_to[k].push_back(bpos[j]); }
        // an interior station of another cluster next to this one would break the block structure
        if (d.part[j] >= 0 && d.part[j] != (int)k) return false;
      }
    std::sort(d.next_to[k].begin(), d.next_to[k].end());
  }
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// device: gathering dense blocks out of the block-sparse reduced matrix
// ------------------------------------------------------------------------------------------------------------------
// Rows: the stations rlist[0 .. nrs); columns: the stations j with lo <= loc[j] < hi at 6 (loc[j] - lo), then (cam_cols)
// the camera columns at 6 ncs.  One workgroup per row station.
__global__ void __launch_bounds__(64)
k_sel_gather_rows(int nrs, const int *__restrict__ rlist, const int *__restrict__ loc, int lo, int hi, int ncs, int cam_cols, int ncam,
                  const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                  const double *__restrict__ Ssc, double *__restrict__ D, int ld) {
  const int ri = blockIdx.x;
  if (ri >= nrs) return;
  const int i = rlist[ri];
  for (int b = rowptr[i]; b < rowptr[i + 1]; ++b) {
    const int l = loc[col[b]];
    if (l < lo || l >= hi) continue;
    for (int t = threadIdx.x; t < 36; t += 64) D[(6 * (int64_t)ri + t / 6) * ld + 6 * (l - lo) + t % 6] = val[36 * (int64_t)b + t];
  }
  if (cam_cols)
    for (int t = threadIdx.x; t < 6 * ncam; t += 64) {
      const int r = t / ncam, k = t - r * ncam;
      D[(6 * (int64_t)ri + r) * ld + 6 * ncs + k] = Ssc[(6 * (int64_t)i + r) * ncam + k];
    }
}

// Camera rows at row0: columns of the stations clist[0 .. ncs), then (corner) the camera block.
__global__ void k_sel_gather_cam_rows(int ncam, int row0, const int *__restrict__ clist, int ncs, int corner, const double *__restrict__ Ssc,
                                      const double *__restrict__ Scc, double *__restrict__ D, int ld) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t per = 6 * (int64_t)ncs + (corner ? ncam : 0);
  if (t >= per * ncam) return;
  const int k = (int)(t / per);
  const int64_t c = t - k * per;
  double v;
  if (c < 6 * (int64_t)ncs) v = Ssc[(6 * (int64_t)clist[c / 6] + c % 6) * ncam + k];
  else v = Scc[(int64_t)k * ncam + (c - 6 * (int64_t)ncs)];
  D[(int64_t)(row0 + k) * ld + c] = v;
}

__global__ void k_sel_unit_diagonal(int r0, int r1, double *__restrict__ D, int ld) {
  const int t = r0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (t < r1) D[(int64_t)t * ld + t] = 1.0;
}

// T[g[r], g[c]] -= C[r, c]
__global__ void k_sel_scatter_sub(int m, const int *__restrict__ g, const double *__restrict__ C, int ldc, double *__restrict__ T, int ldt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < m) T[(int64_t)g[r] * ldt + g[c]] -= C[(int64_t)r * ldc + c];
}

// C[r, c] = T[g[r], g[c]]
__global__ void k_sel_gather_sub(int m, const int *__restrict__ g, const double *__restrict__ T, int ldt, double *__restrict__ C, int ldc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < m) C[(int64_t)r * ldc + c] = T[(int64_t)g[r] * ldt + g[c]];
}

// M = [Skk  -W; -W^T  Tk], Skk n x n, W n x m, Tk m x m
__global__ void k_sel_local_matrix(int n, int m, const double *__restrict__ Skk, int lds, const double *__restrict__ W, int ldw,
                                   const double *__restrict__ Tk, int ldtk, double *__restrict__ M, int ldm) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= n + m) return;
  double v;
  if (r < n) v = c < n ? Skk[(int64_t)r * lds + c] : -W[(int64_t)r * ldw + (c - n)];
  else v = c < n ? -W[(int64_t)c * ldw + (r - n)] : Tk[(int64_t)(r - n) * ldtk + (c - n)];
  M[(int64_t)r * ldm + c] = v;
}

// var_cs[g[r]] = M[r, r] scale^2, g: index among [6S station parameters | camera columns], -1: skip
__global__ void k_sel_variances(int rows, const int *__restrict__ g, const double *__restrict__ M, int ldm, int ns,
                                const double *__restrict__ scale_sta, const double *__restrict__ scale_cam, double *__restrict__ var_cs) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int t = g[r];
  if (t < 0) return;
  const double sc = t < ns ? scale_sta[t] : scale_cam[t - ns];
  var_cs[t] = M[(int64_t)r * ldm + r] * sc * sc;
}

__global__ void k_sel_fill_loc(int n, const int *__restrict__ list, int base, int *__restrict__ loc) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) loc[list[t]] = base + t;
}

// the cluster a point belongs to: that of its first interior station; n_clusters when all its stations are on the boundary
__global__ void k_sel_point_home(int P, const int *__restrict__ pt_start, const int *__restrict__ obs_sta, const int *__restrict__ part,
                                 int n_clusters, int *__restrict__ home) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj >= P) return;
  int h = n_clusters;
  for (int e = pt_start[jj]; e < pt_start[jj + 1]; ++e) {
    const int k = part[obs_sta[e]];
    if (k >= 0) { h = k; break; }
  }
  home[jj] = h;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) {
    if (p) { (void)hipFree(p); p = nullptr; }
    return hipMalloc(&p, std::max<size_t>(bytes, 16)) == hipSuccess;
  }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

}  // namespace

// Cholesky and inverse of the n x n matrix in A (ld = chol_blocked_ld(n), zeros in the padding), in place.
static int spd_inverse(hipStream_t s, double *A, int n, int ld, double *W, double *U, double *T, int *d_info) {
  launch_cholesky_blocked(s, A, n, ld, W, d_info);
  int info = 0;
  if (hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1;
  if (info) return info;
  launch_chol_inverse_blocked(s, A, n, ld, W, U, T);
  return 0;
}

int covariance_selected(const SelInvArgs &a, std::string &err, SelInvStats *stats) {
  hipStream_t s = a.stream;
  const int S = a.S, ncam = a.ncam, ns = 6 * S;
  const int NMIN = chol_blocked_min_order();
  char msg[256];
  Dissection d;
  std::vector<int> rp(a.bsr_rowptr_h, a.bsr_rowptr_h + S + 1), cl(a.bsr_col_h, a.bsr_col_h + rp[S]), used(a.sta_used_h, a.sta_used_h + S);
  if (!dissect(S, rp, cl, used, std::max(a.max_cluster, 8), d)) {
    err = "the station graph has no level-set separators (every station shares points with almost every other)";
    return 1;
  }
  const int K = (int)d.interior.size(), nB = (int)d.boundary.size();
  // orders: every dense matrix has order >= NMIN (identity rows fill up a smaller one) so that one set of routines serves all
  const int tB = 6 * nB + ncam, tB_eff = std::max(tB, NMIN), ldB = chol_blocked_ld(tB_eff);
  int n_max = 0, m_max = 0, loc_max = 0;
  for (int k = 0; k < K; ++k) {
    const int n = std::max(6 * (int)d.interior[k].size(), NMIN), m = 6 * (int)d.next_to[k].size() + ncam;
    n_max = std::max(n_max, n);
    m_max = std::max(m_max, m);
    loc_max = std::max(loc_max, n + m);
  }
  if (stats) { stats->clusters = K; stats->boundary_stations = nB; stats->boundary_order = tB; stats->largest_cluster_order = n_max; }
  const int ldn_max = chol_blocked_ld(n_max), ldm_max = chol_blocked_ld(std::max(m_max, 2)), ldl_max = chol_blocked_ld(loc_max);
  // memory: T and its work matrix; per cluster the inverse and X (kept for the second pass); work of the largest cluster
  {
    double need = 2.0 * tB_eff * ldB + 128.0 * ldB + (double)chol_blocked_workspace(tB_eff);
    for (int k = 0; k < K; ++k) {
      const double n = std::max(6 * (int)d.interior[k].size(), NMIN), m = 6 * (int)d.next_to[k].size() + ncam;
      need += n * chol_blocked_ld((int)n) + n * chol_blocked_ld((int)m);
    }
    need += (double)n_max * ldn_max + 128.0 * ldn_max + 3.0 * m_max * ldn_max + 2.0 * m_max * ldm_max + (double)n_max * ldm_max +
            (double)loc_max * ldl_max;
    need = need * 8.0 + 1e9;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (need > (double)free_b) {
      snprintf(msg, sizeof msg, "%d clusters, boundary of order %d: %.1f GB needed, %.1f GB free", K, tB, need * 1e-9, (double)free_b * 1e-9);
      err = msg;
      return 1;
    }
  }
  DevBuf bT, bTU, bTT, bTW, bU, bTw, bW, bG, bXT, bCk, bTk, bWk, bM, bLoc, bBloc, bPart, bHome, bList, bGidx, bBlist;
  std::vector<DevBuf> bAinv((size_t)K), bX((size_t)K);
  bool ok = bT.alloc((size_t)tB_eff * ldB * 8) && bTU.alloc((size_t)tB_eff * ldB * 8) && bTT.alloc((size_t)128 * ldB * 8) &&
            bTW.alloc(chol_blocked_workspace(tB_eff) * 8) && bU.alloc((size_t)n_max * ldn_max * 8) && bTw.alloc((size_t)128 * ldn_max * 8) &&
            bW.alloc(chol_blocked_workspace(n_max) * 8) && bG.alloc((size_t)m_max * ldn_max * 8) && bXT.alloc((size_t)m_max * ldn_max * 8) &&
            bCk.alloc((size_t)m_max * ldm_max * 8) && bTk.alloc((size_t)m_max * ldm_max * 8) && bWk.alloc((size_t)n_max * ldm_max * 8) &&
            bM.alloc((size_t)loc_max * ldl_max * 8) && bLoc.alloc((size_t)S * 4) && bBloc.alloc((size_t)S * 4) && bPart.alloc((size_t)S * 4) &&
            bHome.alloc((size_t)std::max(a.cov.P, 1) * 4) && bList.alloc((size_t)(n_max / 6 + m_max / 6 + 8) * 4) &&
            bGidx.alloc((size_t)(loc_max + 8) * 4) && bBlist.alloc((size_t)std::max(nB, 1) * 4);
  for (int k = 0; k < K && ok; ++k) {
    const int n = std::max(6 * (int)d.interior[k].size(), NMIN), m = 6 * (int)d.next_to[k].size() + ncam;
    ok = bAinv[k].alloc((size_t)n * chol_blocked_ld(n) * 8) && bX[k].alloc((size_t)n * chol_blocked_ld(m) * 8);
  }
  if (!ok) { err = "out of device memory"; return 2; }
  int *loc = bLoc.as<int>(), *bloc = bBloc.as<int>(), *d_list = bList.as<int>(), *d_gidx = bGidx.as<int>();
  double *T = bT.as<double>();
#define SEL_OK(x) do { if ((x) != hipSuccess) { err = "device call failed: " #x; return 2; } } while (0)
  SEL_OK(hipMemcpyAsync(bPart.p, d.part.data(), (size_t)S * 4, hipMemcpyHostToDevice, s));
  SEL_OK(hipMemsetAsync(bloc, 0xFF, (size_t)S * 4, s));
  if (nB > 0) {
    SEL_OK(hipMemcpyAsync(bBlist.p, d.boundary.data(), (size_t)nB * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sel_fill_loc, dim3(nblk(nB, 256)), dim3(256), 0, s, nB, bBlist.as<int>(), 0, bloc);
  }
  // T = A_BB (boundary stations, then the camera columns, then identity rows up to the minimum order)
  SEL_OK(hipMemsetAsync(T, 0, (size_t)tB_eff * ldB * 8, s));
  if (nB > 0)
    hipLaunchKernelGGL(k_sel_gather_rows, dim3(nB), dim3(64), 0, s, nB, bBlist.as<int>(), bloc, 0, nB, nB, 1, ncam, a.d_bsr_rowptr, a.d_bsr_col,
                       a.d_bsr_val, a.d_Ssc, T, ldB);
  if (ncam > 0)
    hipLaunchKernelGGL(k_sel_gather_cam_rows, dim3(nblk((6 * (int64_t)nB + ncam) * ncam, 256)), dim3(256), 0, s, ncam, 6 * nB, bBlist.as<int>(), nB, 1,
                       a.d_Ssc, a.d_Scc, T, ldB);
  if (tB_eff > tB) hipLaunchKernelGGL(k_sel_unit_diagonal, dim3(nblk(tB_eff - tB, 256)), dim3(256), 0, s, tB, tB_eff, T, ldB);

  std::vector<int> list, gidx;
  // first pass: cluster inverses and the boundary complement
  for (int k = 0; k < K; ++k) {
    const int nI = (int)d.interior[k].size(), nBk = (int)d.next_to[k].size();
    const int n = 6 * nI, n_eff = std::max(n, NMIN), ldn = chol_blocked_ld(n_eff), m = 6 * nBk + ncam, ldm = chol_blocked_ld(m);
    double *Ainv = bAinv[k].as<double>(), *X = bX[k].as<double>(), *G = bG.as<double>(), *XT = bXT.as<double>(), *Ck = bCk.as<double>();
    list.assign(d.interior[k].begin(), d.interior[k].end());
    for (int q : d.next_to[k]) list.push_back(d.boundary[q]);
    SEL_OK(hipMemcpyAsync(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice, s));
    SEL_OK(hipMemsetAsync(loc, 0xFF, (size_t)S * 4, s));
    hipLaunchKernelGGL(k_sel_fill_loc, dim3(nblk(nI, 256)), dim3(256), 0, s, nI, d_list, 0, loc);
    SEL_OK(hipMemsetAsync(Ainv, 0, (size_t)n_eff * ldn * 8, s));
    hipLaunchKernelGGL(k_sel_gather_rows, dim3(nI), dim3(64), 0, s, nI, d_list, loc, 0, nI, nI, 0, ncam, a.d_bsr_rowptr, a.d_bsr_col, a.d_bsr_val,
                       a.d_Ssc, Ainv, ldn);
    if (n_eff > n) hipLaunchKernelGGL(k_sel_unit_diagonal, dim3(nblk(n_eff - n, 256)), dim3(256), 0, s, n, n_eff, Ainv, ldn);
    // G = A_Bk: rows the boundary stations next to the cluster, then the camera rows
    SEL_OK(hipMemsetAsync(G, 0, (size_t)m * ldn * 8, s));
    if (nBk > 0)
      hipLaunchKernelGGL(k_sel_gather_rows, dim3(nBk), dim3(64), 0, s, nBk, d_list + nI, loc, 0, nI, nI, 0, ncam, a.d_bsr_rowptr, a.d_bsr_col,
                         a.d_bsr_val, a.d_Ssc, G, ldn);
    if (ncam > 0)
      hipLaunchKernelGGL(k_sel_gather_cam_rows, dim3(nblk(6 * (int64_t)nI * ncam, 256)), dim3(256), 0, s, ncam, 6 * nBk, d_list, nI, 0, a.d_Ssc,
                         a.d_Scc, G, ldn);
    SEL_OK(hipStreamSynchronize(s));  // list is reused by the next cluster
    const int info = spd_inverse(s, Ainv, n_eff, ldn, bW.as<double>(), bU.as<double>(), bTw.as<double>(), a.d_info);
    if (info < 0) { err = "device call failed in the cluster factorisation"; return 2; }
    if (info > 0) {
      const int r = info - 1;
      snprintf(msg, sizeof msg, "the reduced normal matrix is rank deficient (station %d, parameter %d)", r / 6 < nI ? d.interior[k][r / 6] : -1, r % 6);
      err = msg;
      return 3;
    }
    SEL_OK(hipMemsetAsync(X, 0, (size_t)n_eff * ldm * 8, s));
    SEL_OK(hipMemsetAsync(XT, 0, (size_t)m * ldn * 8, s));
    launch_gemm_nt(s, Ainv, ldn, n, G, ldn, m, n, 1.0, false, X, ldm);   // X  = A^-1 G^T   (n x m)
    launch_gemm_nt(s, G, ldn, m, Ainv, ldn, n, n, 1.0, false, XT, ldn);  // XT = G A^-1     (m x n)
    launch_gemm_nt(s, G, ldn, m, XT, ldn, m, n, 1.0, false, Ck, ldm);    // G A^-1 G^T     (m x m)
    gidx.resize((size_t)m);
    for (int q = 0; q < nBk; ++q)
      for (int c = 0; c < 6; ++c) gidx[6 * (size_t)q + c] = 6 * d.next_to[k][q] + c;
    for (int c = 0; c < ncam; ++c) gidx[6 * (size_t)nBk + c] = 6 * nB + c;
    SEL_OK(hipMemcpyAsync(d_gidx, gidx.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sel_scatter_sub, dim3(nblk(m, 128), m), dim3(128), 0, s, m, d_gidx, Ck, ldm, T, ldB);
    SEL_OK(hipStreamSynchronize(s));
  }
  // S^-1 on the boundary
  {
    const int info = spd_inverse(s, T, tB_eff, ldB, bTW.as<double>(), bTU.as<double>(), bTT.as<double>(), a.d_info);
    if (info < 0) { err = "device call failed in the boundary factorisation"; return 2; }
    if (info > 0) {
      const int r = info - 1;
      if (r < 6 * nB) snprintf(msg, sizeof msg, "the reduced normal matrix is rank deficient (station %d, parameter %d)", d.boundary[r / 6], r % 6);
      else snprintf(msg, sizeof msg, "the reduced normal matrix is rank deficient (camera column %d)", r - 6 * nB);
      err = msg;
      return 3;
    }
  }
  SEL_OK(hipMemsetAsync(a.cov.var_cs, 0, (size_t)(ns + ncam) * 8, s));
  if (a.cov.P > 0)
    hipLaunchKernelGGL(k_sel_point_home, dim3(nblk(a.cov.P, 256)), dim3(256), 0, s, a.cov.P, a.cov.pt_start, a.cov.obs_sta, bPart.as<int>(), K,
                       bHome.as<int>());
  CovArgs ca = a.cov;
  ca.pt_home = bHome.as<int>();
  // boundary: variances of its stations and of the cameras; points seen from boundary stations only
  {
    gidx.assign((size_t)tB, -1);
    for (int q = 0; q < nB; ++q)
      for (int c = 0; c < 6; ++c) gidx[6 * (size_t)q + c] = 6 * d.boundary[q] + c;
    for (int c = 0; c < ncam; ++c) gidx[6 * (size_t)nB + c] = ns + c;
    DevBuf bGB;
    if (!bGB.alloc((size_t)tB * 4)) { err = "out of device memory"; return 2; }
    SEL_OK(hipMemcpyAsync(bGB.p, gidx.data(), (size_t)tB * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sel_variances, dim3(nblk(tB, 256)), dim3(256), 0, s, tB, bGB.as<int>(), T, ldB, ns, a.cov.scale_sta, a.cov.scale_cam,
                       a.cov.var_cs);
    ca.S = nB; ca.n = tB; ca.ld = ldB; ca.Sinv = T; ca.sta_loc = bloc; ca.home = K;
    launch_cov_points(s, ca);
    SEL_OK(hipStreamSynchronize(s));
  }
  // second pass: the clusters' blocks of S^-1
  for (int k = 0; k < K; ++k) {
    const int nI = (int)d.interior[k].size(), nBk = (int)d.next_to[k].size();
    const int n = 6 * nI, n_eff = std::max(n, NMIN), ldn = chol_blocked_ld(n_eff), m = 6 * nBk + ncam, ldm = chol_blocked_ld(m);
    const int nl = n + m, ldl = chol_blocked_ld(nl);
    double *Ainv = bAinv[k].as<double>(), *X = bX[k].as<double>(), *Tk = bTk.as<double>(), *Wk = bWk.as<double>(), *M = bM.as<double>();
    gidx.resize((size_t)m);
    for (int q = 0; q < nBk; ++q)
      for (int c = 0; c < 6; ++c) gidx[6 * (size_t)q + c] = 6 * d.next_to[k][q] + c;
    for (int c = 0; c < ncam; ++c) gidx[6 * (size_t)nBk + c] = 6 * nB + c;
    SEL_OK(hipMemcpyAsync(d_gidx, gidx.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
    SEL_OK(hipMemsetAsync(Tk, 0, (size_t)m * ldm * 8, s));
    hipLaunchKernelGGL(k_sel_gather_sub, dim3(nblk(m, 128), m), dim3(128), 0, s, m, d_gidx, T, ldB, Tk, ldm);
    SEL_OK(hipMemsetAsync(Wk, 0, (size_t)n_eff * ldm * 8, s));
    launch_gemm_nt(s, X, ldm, n, Tk, ldm, m, m, 1.0, false, Wk, ldm);   // W = X T^-1[Bk, Bk]   (n x m)
    launch_gemm_nt(s, Wk, ldm, n, X, ldm, n, m, 1.0, true, Ainv, ldn);  // S^-1[Ik, Ik] = A^-1 + W X^T
    hipLaunchKernelGGL(k_sel_local_matrix, dim3(nblk(nl, 128), nl), dim3(128), 0, s, n, m, Ainv, ldn, Wk, ldm, Tk, ldm, M, ldl);
    SEL_OK(hipStreamSynchronize(s));  // d_gidx and d_list are rewritten below
    list.assign(d.interior[k].begin(), d.interior[k].end());
    for (int q : d.next_to[k]) list.push_back(d.boundary[q]);
    SEL_OK(hipMemcpyAsync(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice, s));
    SEL_OK(hipMemsetAsync(loc, 0xFF, (size_t)S * 4, s));
    hipLaunchKernelGGL(k_sel_fill_loc, dim3(nblk((int)list.size(), 256)), dim3(256), 0, s, (int)list.size(), d_list, 0, loc);
    gidx.assign((size_t)n, -1);
    for (int q = 0; q < nI; ++q)
      for (int c = 0; c < 6; ++c) gidx[6 * (size_t)q + c] = 6 * d.interior[k][q] + c;
    SEL_OK(hipMemcpyAsync(d_gidx, gidx.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sel_variances, dim3(nblk(n, 256)), dim3(256), 0, s, n, d_gidx, M, ldl, ns, a.cov.scale_sta, a.cov.scale_cam, a.cov.var_cs);
    ca.S = nI + nBk; ca.n = nl; ca.ld = ldl; ca.Sinv = M; ca.sta_loc = loc; ca.home = k;
    launch_cov_points(s, ca);
    SEL_OK(hipStreamSynchronize(s));
  }
#undef SEL_OK
  return 0;
}

}  // namespace ba
