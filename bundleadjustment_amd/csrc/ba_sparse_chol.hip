// ba_sparse_chol.hip -- sparse direct Cholesky of the block-sparse reduced camera system: nested dissection of the
// station graph, multifrontal factorisation with dense fronts on the matrix cores.
//
// The reference solves its normal equations with Ceres' SPARSE_NORMAL_CHOLESKY (src/main.cpp:4706): a fill-reducing
// order and a sparse factorisation.  This is the same for the reduced system S (points eliminated): the station graph
// (stations that share a point) is cut recursively by breadth-first level sets; the tree of separators with the leaf
// pieces below is the elimination tree.  Every tree node owns a front: a dense symmetric matrix over its own stations V
// and the stations U of its ancestors that it is (or, through its children, becomes) coupled to.  Bottom-up, level by
// level and all fronts of a level in one launch per step:
//   front = entries of S of its columns  +  the update matrices of its children            (assembly, extend-add)
//   L_VV L_VV^T = F_VV,  L_UV = F_UV L_VV^-T,  F_UU -= L_UV L_UV^T                          (ba_chol_blocked.hip)
// and F_UU is the update matrix handed to the parent.  The camera columns (dense: every station couples to them) are the
// root's own unknowns and part of every U.  Solves run up and down the same tree on per-front vectors.
// The structure depends on the network alone and is built once; every number has one fixed order of operations.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "ba_kernels.h"

namespace ba {

namespace {

inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

// ------------------------------------------------------------------------------------------------------------------
// kernels: assembly
// ------------------------------------------------------------------------------------------------------------------
// block b of the reduced matrix into its front (dst: offset of its first entry in the arena, -1: not stored -- the
// transposed twin is)
__global__ void k_sc_gather_blocks(int64_t n36, const int64_t *__restrict__ dst, const int *__restrict__ ld, const double *__restrict__ val,
                                   double *__restrict__ arena) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= n36) return;
  const int64_t b = t / 36;
  const int e = (int)(t - 36 * b);
  const int64_t d = dst[b];
  if (d >= 0) arena[d + (int64_t)(e / 6) * ld[b] + e % 6] = val[t];
}

// camera rows of station i: rows (camera columns) c, columns the six parameters of the station
__global__ void k_sc_gather_cams(int S, int ncam, const int64_t *__restrict__ dst, const int *__restrict__ ld, const double *__restrict__ Ssc,
                                 double *__restrict__ arena) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= 6 * (int64_t)S * ncam) return;
  const int i = (int)(t / (6 * ncam));
  const int rem = (int)(t - (int64_t)i * 6 * ncam), r = rem / ncam, c = rem - r * ncam;
  const int64_t d = dst[i];
  if (d >= 0) arena[d + (int64_t)c * ld[i] + r] = Ssc[t];
}

__global__ void k_sc_gather_corner(int ncam, const double *__restrict__ Scc, double *__restrict__ A, int ld) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ncam * ncam) A[(int64_t)(t / ncam) * ld + t % ncam] = Scc[t];
}

// parent += update matrix of the child (lower triangle), for the children of one list (at most one per parent); a
// workgroup per row
__global__ void __launch_bounds__(128)
k_sc_extend_add(const FrontDesc *__restrict__ fd, const int *__restrict__ children) {
  const FrontDesc c = fd[children[blockIdx.y]];
  const int a = blockIdx.x;
  if (a >= c.m - c.nvp) return;
  const FrontDesc p = fd[c.parent];
  const double *src = c.A + (int64_t)(c.nvp + a) * c.ld + c.nvp;
  double *dst = p.A + (int64_t)c.up_map[a] * p.ld;
  for (int b = threadIdx.x; b <= a; b += 128) dst[c.up_map[b]] += src[b];
}

// ------------------------------------------------------------------------------------------------------------------
// kernels: solves on the fronts' vectors
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_sc_fwd_init(const FrontDesc *__restrict__ fd, const double *__restrict__ b) {
  const FrontDesc f = fd[blockIdx.y];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= f.m) return;
  double v = 0.0;
  if (r < f.nvp) {
    const int g = f.vidx[r];
    if (g >= 0) v = b[g];
  }
  f.f[r] = v;
}

__global__ void k_sc_child_add(const FrontDesc *__restrict__ fd, const int *__restrict__ children) {
  const FrontDesc c = fd[children[blockIdx.y]];
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= c.m - c.nvp) return;
  fd[c.parent].f[c.up_map[a]] += c.f[c.nvp + a];
}

// y = W_kp v for one 128-block: thread pair (row t, half h of the columns); v in xs, result of the pair summed through hs
__device__ __forceinline__ double sc_apply_w(const double *__restrict__ W, int t, int h, const double *xs, double *hs) {
  const double2 *wr = reinterpret_cast<const double2 *>(W + (int64_t)t * 128 + 64 * h);
  double2 wv[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) wv[c] = wr[c];
  __syncthreads();  // xs is complete
  double y = 0.0;
#pragma unroll
  for (int c = 0; c < 32; ++c) y += wv[c].x * xs[64 * h + 2 * c] + wv[c].y * xs[64 * h + 2 * c + 1];
  if (h == 1) hs[t] = y;
  __syncthreads();
  return y + hs[t];
}

// x_0 = W_0 f_0 of every front of the list
__global__ void __launch_bounds__(256)
k_sc_fwd_first(const FrontDesc *__restrict__ fd) {
  __shared__ double xs[128], hs[128];
  const FrontDesc f = fd[blockIdx.x];
  if (f.nvp == 0) return;
  const int t = threadIdx.x & 127, h = threadIdx.x >> 7;
  if (h == 0) xs[t] = f.f[t];
  const double y = sc_apply_w(f.W, t, h, xs, hs);
  if (h == 0) f.f[t] = y;
}

// forward step kp: the workgroup of block row kp + 1 + bx subtracts L[rows, panel kp] x_kp from its part of f; the one of
// block row kp + 1, when that is still one of the eliminated panels, goes on to x_{kp+1} = W_{kp+1} f_{kp+1}
__global__ void __launch_bounds__(256)
k_sc_fwd_step(const FrontDesc *__restrict__ fd, int kp) {
  __shared__ double xs[128], hs[128];
  const FrontDesc f = fd[blockIdx.y];
  const int k0 = 128 * kp, ib = kp + 1 + blockIdx.x, r0 = 128 * ib;
  if (k0 >= f.nvp || r0 >= f.m) return;
  const int t = threadIdx.x & 127, h = threadIdx.x >> 7;
  const int row = r0 + t;
  const bool on = row < f.m;
  if (h == 0) xs[t] = f.f[k0 + t];
  const double2 *lr = reinterpret_cast<const double2 *>(f.A + (int64_t)(on ? row : r0) * f.ld + k0 + 64 * h);
  double2 l[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) l[c] = lr[c];
  const double own = (on && h == 0) ? f.f[row] : 0.0;
  __syncthreads();
  double sum = 0.0;
#pragma unroll
  for (int c = 0; c < 32; ++c) sum += l[c].x * xs[64 * h + 2 * c] + l[c].y * xs[64 * h + 2 * c + 1];
  if (h == 1) hs[t] = sum;
  __syncthreads();
  const double v = own - (sum + hs[t]);
  if (blockIdx.x != 0 || r0 >= f.nvp) {
    if (h == 0 && on) f.f[row] = v;
    return;
  }
  __syncthreads();  // xs and hs are free again
  if (h == 0) xs[t] = v;
  const double y = sc_apply_w(f.W + (int64_t)ib * 128 * 128, t, h, xs, hs);
  if (h == 0) f.f[row] = y;
}

__global__ void k_sc_bwd_gather(const FrontDesc *__restrict__ fd, int first) {
  const FrontDesc c = fd[first + blockIdx.y];
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (c.parent < 0 || a >= c.m - c.nvp) return;
  c.f[c.nvp + a] = fd[c.parent].f[c.up_map[a]];
}

// y[col] = sum_r W[r][col] v[r] for one 128-block: thread (col, half h of the rows)
__device__ __forceinline__ double sc_apply_wt(const double *__restrict__ W, int col, int h, const double *xs, double *hs) {
  const double *wc = W + col + (int64_t)64 * h * 128;
  double w[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) w[r] = wc[(int64_t)r * 128];
  __syncthreads();  // xs is complete
  double y = 0.0;
#pragma unroll
  for (int r = 0; r < 64; ++r) y += w[r] * xs[64 * h + r];
  if (h == 1) hs[col] = y;
  __syncthreads();
  return y + hs[col];
}

// What the rows that stay contribute to the eliminated ones, L_UV^T x_U, in SC_UCH partial sums per column (the rows cut
// into that many runs): workgroup (column block, run), thread (column, one of eight slices of the run)
#define SC_UCH 4
__global__ void __launch_bounds__(1024)
k_sc_bwd_u(const FrontDesc *__restrict__ fd) {
  __shared__ double hs[8][128];
  const FrontDesc f = fd[blockIdx.z];
  const int k0 = 128 * blockIdx.x;
  if (k0 >= f.nvp) return;
  const int col = threadIdx.x & 127, h = threadIdx.x >> 7;
  const int nu = f.m - f.nvp, run = (nu + SC_UCH - 1) / SC_UCH, slice = (run + 7) / 8;
  const int run_a = f.nvp + run * blockIdx.y, run_b = min(f.m, run_a + run);
  const int ra = min(run_b, run_a + slice * h), rb = min(run_b, ra + slice);
  double s = 0.0;
  const double *lc = f.A + k0 + col;
  for (int r = ra; r < rb; r += 8) {
    double l[8], x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int rr = min(r + u, rb - 1);
      l[u] = lc[(int64_t)rr * f.ld];
      x[u] = (r + u < rb) ? f.f[rr] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += l[u] * x[u];
  }
  hs[h][col] = s;
  __syncthreads();
  if (h == 0)
    f.g[(int64_t)blockIdx.y * f.nvp + k0 + col] =
        ((hs[0][col] + hs[1][col]) + (hs[2][col] + hs[3][col])) + ((hs[4][col] + hs[5][col]) + (hs[6][col] + hs[7][col]));
}

// the last eliminated panel of every front: x_p = W_p^T (f_p - its share of L_UV^T x_U)
__global__ void __launch_bounds__(256)
k_sc_bwd_first(const FrontDesc *__restrict__ fd) {
  __shared__ double xs[128], hs[128];
  const FrontDesc f = fd[blockIdx.x];
  if (f.nvp == 0) return;
  const int col = threadIdx.x & 127, h = threadIdx.x >> 7, k0 = f.nvp - 128;
  if (h == 0) {
    double v = f.f[k0 + col];
    if (f.m > f.nvp)
      for (int j = 0; j < SC_UCH; ++j) v -= f.g[(int64_t)j * f.nvp + k0 + col];
    xs[col] = v;
  }
  const double y = sc_apply_wt(f.W + (int64_t)(k0 / 128) * 128 * 128, col, h, xs, hs);
  if (h == 0) f.f[k0 + col] = y;
}

// backward step i (panels from the last one up): workgroup q < i subtracts L[panel i, panel q]^T x_i from f_q; the one of
// panel i - 1 then goes on to x_{i-1} = W_{i-1}^T (f_{i-1} - its share of L_UV^T x_U)
__global__ void __launch_bounds__(256)
k_sc_bwd_step(const FrontDesc *__restrict__ fd, int i) {
  __shared__ double xs[128], hs[128];
  const FrontDesc f = fd[blockIdx.y];
  const int q = blockIdx.x;
  if (128 * i >= f.nvp || q >= i) return;
  const int col = threadIdx.x & 127, h = threadIdx.x >> 7, i0 = 128 * i, q0 = 128 * q;
  if (h == 0) xs[col] = f.f[i0 + col];
  const double *lc = f.A + (int64_t)(i0 + 64 * h) * f.ld + q0 + col;
  double l[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) l[r] = lc[(int64_t)r * f.ld];
  const double own = h == 0 ? f.f[q0 + col] : 0.0;
  __syncthreads();
  double sum = 0.0;
#pragma unroll
  for (int r = 0; r < 64; ++r) sum += l[r] * xs[64 * h + r];
  if (h == 1) hs[col] = sum;
  __syncthreads();
  double v = own - (sum + hs[col]);
  if (q != i - 1) {
    if (h == 0) f.f[q0 + col] = v;
    return;
  }
  __syncthreads();
  if (h == 0) {
    if (f.m > f.nvp)
      for (int j = 0; j < SC_UCH; ++j) v -= f.g[(int64_t)j * f.nvp + q0 + col];
    xs[col] = v;
  }
  const double y = sc_apply_wt(f.W + (int64_t)q * 128 * 128, col, h, xs, hs);
  if (h == 0) f.f[q0 + col] = y;
}

__global__ void k_sc_bwd_store(const FrontDesc *__restrict__ fd, double *__restrict__ x) {
  const FrontDesc f = fd[blockIdx.y];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= f.nvp) return;
  const int g = f.vidx[r];
  if (g >= 0) x[g] = f.f[r];
}

// ------------------------------------------------------------------------------------------------------------------
// host: the tree
// ------------------------------------------------------------------------------------------------------------------
struct Node {
  int parent = -1;
  std::vector<int> V;  // own stations (the root: none, it owns the camera columns)
  std::vector<int> U;  // stations of ancestors that the front reaches, in elimination order
  std::vector<int> children;
  int level = 0;
};

}  // namespace

struct SparseChol {
  int S = 0, ncam = 0, n_fronts = 0, n_levels = 0;
  int64_t n_blocks = 0, arena_doubles = 0;
  std::vector<int> lvl_ptr, lvl_maxpv, lvl_max_m, lvl_max_nu;
  // per level: lists of children (descriptor indices) of its fronts, first children, second children, ...
  std::vector<std::vector<std::pair<int, int>>> lvl_slots;  // (offset in d_children, count)
  std::vector<int> lvl_slot_max_nu;
  std::vector<std::vector<int>> slot_max_nu;
  FrontDesc *d_desc = nullptr;
  double *d_arena = nullptr, *d_W = nullptr, *d_vec = nullptr;
  int64_t *d_blk_dst = nullptr, *d_sta_dst = nullptr;
  int *d_blk_ld = nullptr, *d_sta_ld = nullptr, *d_up_map = nullptr, *d_vidx = nullptr, *d_children = nullptr;
  int root_ld = 0;
  double *d_root_A = nullptr;
  std::vector<void *> allocs;
  bool zeroed = false;  // the fronts hold zeros already (sparse_chol_clear has run since the last factorisation)
  // statistics
  int n_leaves = 0, largest_front = 0, critical_panels = 0;
  double factor_flops = 0.0;

  template <class T>
  bool dalloc(T **p, size_t n) {
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return false;
    allocs.push_back(q);
    *p = static_cast<T *>(q);
    return true;
  }
  template <class T>
  bool upload(T **p, const std::vector<T> &v) {
    if (!dalloc(p, v.size())) return false;
    return v.empty() || hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
  }
  void release() {
    for (void *q : allocs) (void)hipFree(q);
    allocs.clear();
  }
};

SparseChol *sparse_chol_create() { return new SparseChol(); }
void sparse_chol_destroy(SparseChol *sc) {
  if (!sc) return;
  sc->release();
  delete sc;
}
void sparse_chol_stats(const SparseChol *sc, SparseCholStats *st) {
  st->fronts = sc->n_fronts; st->levels = sc->n_levels; st->leaves = sc->n_leaves; st->largest_front = sc->largest_front;
  st->critical_panels = sc->critical_panels; st->arena_bytes = 8.0 * (double)sc->arena_doubles; st->factor_flops = sc->factor_flops;
}

// The tree, the fronts' index sets and every list the kernels read.  Returns 0, 1 (not worth it / does not fit: err
// says why) or 2 (device failure).
int sparse_chol_analyse(SparseChol *sc, int S, int ncam, const std::vector<int> &rowptr, const std::vector<int> &col,
                        const std::vector<int> &used, int leaf_stations, std::string &err) {
  sc->release();
  sc->S = S;
  sc->ncam = ncam;
  const int CBp = front_panel();
  std::vector<Node> nodes(1);  // node 0: the root
  std::vector<int> node_of((size_t)S, -1);
  {
    std::vector<int> mark((size_t)S, -1), level((size_t)S, 0), queue;
    std::vector<std::pair<std::vector<int>, int>> todo;
    {
      std::vector<int> all;
      for (int i = 0; i < S; ++i)
        if (used[i]) all.push_back(i);
      if (!all.empty()) todo.emplace_back(std::move(all), 0);
    }
    int token = 0;
    // A cut may leave at most 56 % of the stations on one side.  On the 10 000-station benchmark network: 30 % / 70 % gives
    // 15 levels and 57 panels on the critical path, 44 % / 56 % 13 and 53 (leaves of 21 stations), 10 and 45 with leaves of
    // 63 -- the factorisation is a chain of panel steps of about 110 us each, so the shorter chain wins although it costs
    // 17 % more operations.
    const int64_t bal_lo = 44;
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
    auto make_node = [&](std::vector<int> &&V, int parent) {
      const int id = (int)nodes.size();
      nodes.emplace_back();
      nodes[id].parent = parent;
      for (int i : V) node_of[i] = id;
      nodes[id].V = std::move(V);
      nodes[parent].children.push_back(id);
      return id;
    };
    while (!todo.empty()) {
      std::vector<int> V = std::move(todo.back().first);
      const int parent = todo.back().second;
      todo.pop_back();
      if ((int)V.size() <= leaf_stations) {
        make_node(std::move(V), parent);
        continue;
      }
      const int tok = token, seen = token + 1, again = token + 2;
      token += 3;
      for (int i : V) mark[i] = tok;
      bfs(V[0], tok, seen);
      if (queue.size() < V.size()) {  // not connected: the pieces hang side by side under the same parent
        std::vector<int> comp(queue), rest;
        for (int i : V)
          if (mark[i] == tok) rest.push_back(i);
        todo.emplace_back(std::move(comp), parent);
        todo.emplace_back(std::move(rest), parent);
        continue;
      }
      const int far = queue.back();
      bfs(far, seen, again);
      const int n_lev = level[queue.back()] + 1;
      if (n_lev < 3) {  // nothing to cut: one dense front
        make_node(std::move(V), parent);
        continue;
      }
      std::vector<int> count((size_t)n_lev, 0);
      for (int i : queue) count[level[i]]++;
      int best = -1;
      {
        int64_t before = 0;
        const int64_t n = (int64_t)V.size();
        for (int l = 0; l < n_lev; ++l) {
          const bool balanced = 100 * before >= bal_lo * n && 100 * before <= (100 - bal_lo) * n && l >= 1 && l <= n_lev - 2;
          if (balanced && (best < 0 || count[l] < count[best])) best = l;
          before += count[l];
        }
        if (best < 0) {
          before = 0;
          for (int l = 0; l < n_lev; ++l) {
            if (l >= 1 && l <= n_lev - 2 && 2 * (before + count[l]) >= n) { best = l; break; }
            before += count[l];
          }
          if (best < 0) best = n_lev - 2;
        }
      }
      std::vector<int> lo, hi, sep;
      for (int i : queue) {
        if (level[i] < best) lo.push_back(i);
        else if (level[i] > best) hi.push_back(i);
        else {
          bool touches = false;
          for (int b = rowptr[i]; b < rowptr[i + 1] && !touches; ++b) {
            const int j = col[b];
            touches = mark[j] == again && level[j] > best;
          }
          if (touches) sep.push_back(i);
          else lo.push_back(i);
        }
      }
      if (lo.empty() || hi.empty() || sep.empty()) {
        std::vector<int> all(queue);
        make_node(std::move(all), parent);
        continue;
      }
      const int id = make_node(std::move(sep), parent);
      todo.emplace_back(std::move(lo), id);
      todo.emplace_back(std::move(hi), id);
    }
  }
  const int nn = (int)nodes.size();
  // levels (children first), descriptor order by level
  {
    // nodes were created parents first: walk backwards
    for (int j = nn - 1; j >= 1; --j) nodes[nodes[j].parent].level = std::max(nodes[nodes[j].parent].level, nodes[j].level + 1);
  }
  std::vector<int> order((size_t)nn), desc_of((size_t)nn);
  for (int j = 0; j < nn; ++j) order[j] = j;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nodes[a].level < nodes[b].level; });
  for (int k = 0; k < nn; ++k) desc_of[order[k]] = k;
  sc->n_fronts = nn;
  sc->n_levels = nodes[0].level + 1;
  sc->lvl_ptr.assign((size_t)sc->n_levels + 1, 0);
  for (int j = 0; j < nn; ++j) sc->lvl_ptr[nodes[j].level + 1]++;
  for (int l = 0; l < sc->n_levels; ++l) sc->lvl_ptr[l + 1] += sc->lvl_ptr[l];
  // elimination position of a station: (descriptor of its node, place in V)
  std::vector<int64_t> ord((size_t)S, -1);
  std::vector<int> place((size_t)S, -1);
  for (int j = 1; j < nn; ++j)
    for (size_t k = 0; k < nodes[j].V.size(); ++k) {
      place[nodes[j].V[k]] = (int)k;
      ord[nodes[j].V[k]] = ((int64_t)desc_of[j] << 32) | (int64_t)k;
    }
  // U sets, children before parents
  for (int k = 0; k < nn; ++k) {
    const int j = order[k];
    if (j == 0) continue;
    Node &nd = nodes[j];
    std::vector<int> u;
    for (int s : nd.V)
      for (int b = rowptr[s]; b < rowptr[s + 1]; ++b) {
        const int t = col[b];
        if (node_of[t] < 0 || node_of[t] == j) continue;
        if (desc_of[node_of[t]] > k) u.push_back(t);
      }
    for (int c : nd.children)
      for (int t : nodes[c].U)
        if (node_of[t] != j) u.push_back(t);
    std::sort(u.begin(), u.end(), [&](int a, int b) { return ord[a] < ord[b]; });
    u.erase(std::unique(u.begin(), u.end()), u.end());
    // every one of them must sit in an ancestor
    for (int t : u) {
      int a = nd.parent;
      while (a > 0 && a != node_of[t]) a = nodes[a].parent;
      if (a != node_of[t]) { err = "the dissection left an edge between two branches of the tree"; return 1; }
    }
    nd.U = std::move(u);
  }
  // sizes and offsets
  struct Lay { int nv, nvp, nu, m, ld; int64_t a_off, w_off, f_off, up_off, v_off; };
  std::vector<Lay> lay((size_t)nn);
  int64_t a_tot = 0, w_tot = 0, f_tot = 0, up_tot = 0, v_tot = 0;
  sc->lvl_maxpv.assign((size_t)sc->n_levels, 0);
  sc->lvl_max_m.assign((size_t)sc->n_levels, 0);
  sc->lvl_max_nu.assign((size_t)sc->n_levels, 0);
  sc->n_leaves = 0; sc->largest_front = 0; sc->factor_flops = 0.0;
  for (int k = 0; k < nn; ++k) {
    const int j = order[k];
    Lay &L = lay[j];
    L.nv = j == 0 ? ncam : 6 * (int)nodes[j].V.size();
    L.nvp = (L.nv + CBp - 1) / CBp * CBp;
    L.nu = j == 0 ? 0 : 6 * (int)nodes[j].U.size() + ncam;
    L.m = L.nvp + L.nu;
    L.ld = (L.m + 15) & ~15;
    L.a_off = a_tot; a_tot += (int64_t)L.m * L.ld;
    L.w_off = w_tot; w_tot += (int64_t)(L.nvp / CBp) * CBp * CBp;
    L.f_off = f_tot; f_tot += ((L.m + 1) & ~1) + SC_UCH * L.nvp;  // the vector, then the partial sums of k_sc_bwd_u
    L.up_off = up_tot; up_tot += L.nu;
    L.v_off = v_tot; v_tot += L.nvp;
    const int lv = nodes[j].level;
    sc->lvl_maxpv[lv] = std::max(sc->lvl_maxpv[lv], L.nvp / CBp);
    sc->lvl_max_m[lv] = std::max(sc->lvl_max_m[lv], L.m);
    sc->lvl_max_nu[lv] = std::max(sc->lvl_max_nu[lv], L.nu);
    sc->largest_front = std::max(sc->largest_front, L.m);
    if (nodes[j].children.empty() && j != 0) sc->n_leaves++;
    const double nv = L.nv, nu = L.nu;
    sc->factor_flops += nv * nv * nv / 3.0 + nv * nv * nu + nv * nu * nu;
  }
  sc->critical_panels = 0;
  for (int l = 0; l < sc->n_levels; ++l) sc->critical_panels += sc->lvl_maxpv[l];
  sc->arena_doubles = a_tot;
  {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const double need = 8.0 * ((double)a_tot + (double)w_tot + (double)f_tot) + 1e9;
    if (need > 0.5 * (double)free_b) {
      char msg[200];
      snprintf(msg, sizeof msg, "the fronts of the sparse factorisation take %.1f GB, %.1f GB are free", need * 1e-9, (double)free_b * 1e-9);
      err = msg;
      return 1;
    }
  }
  // lists: where the rows that stay sit in the parent, the global index of the own rows
  std::vector<int> up_map((size_t)up_tot), vidx((size_t)v_tot, -1);
  auto pos_in = [&](int j, int station) -> int {  // position of a station's first parameter in front j
    if (node_of[station] == j) return 6 * place[station];
    const std::vector<int> &U = nodes[j].U;
    const auto it = std::lower_bound(U.begin(), U.end(), station, [&](int a, int b) { return ord[a] < ord[b]; });
    return lay[j].nvp + 6 * (int)(it - U.begin());
  };
  auto cam_pos = [&](int j) { return j == 0 ? 0 : lay[j].nvp + 6 * (int)nodes[j].U.size(); };
  for (int j = 0; j < nn; ++j) {
    const Lay &L = lay[j];
    if (j == 0) {
      for (int c = 0; c < ncam; ++c) vidx[L.v_off + c] = 6 * S + c;
      continue;
    }
    for (size_t k = 0; k < nodes[j].V.size(); ++k)
      for (int r = 0; r < 6; ++r) vidx[L.v_off + 6 * k + r] = 6 * nodes[j].V[k] + r;
    const int pj = nodes[j].parent;
    for (size_t k = 0; k < nodes[j].U.size(); ++k) {
      const int pp = pos_in(pj, nodes[j].U[k]);
      for (int r = 0; r < 6; ++r) up_map[L.up_off + 6 * k + r] = pp + r;
    }
    for (int c = 0; c < ncam; ++c) up_map[L.up_off + 6 * nodes[j].U.size() + c] = cam_pos(pj) + c;
  }
  // where the blocks of S go: block (i, t) with i eliminated no earlier than t sits in the front of t's node
  sc->n_blocks = (int64_t)col.size();
  std::vector<int64_t> blk_dst(col.size(), -1), sta_dst((size_t)S, -1);
  std::vector<int> blk_ld(col.size(), 0), sta_ld((size_t)S, 0);
  for (int i = 0; i < S; ++i) {
    if (node_of[i] < 0) continue;
    for (int b = rowptr[i]; b < rowptr[i + 1]; ++b) {
      const int t = col[b];
      if (node_of[t] < 0 || ord[i] < ord[t]) continue;
      const int j = node_of[t];
      blk_dst[b] = lay[j].a_off + (int64_t)pos_in(j, i) * lay[j].ld + pos_in(j, t);
      blk_ld[b] = lay[j].ld;
    }
    const int j = node_of[i];
    sta_dst[i] = lay[j].a_off + (int64_t)cam_pos(j) * lay[j].ld + pos_in(j, i);
    sta_ld[i] = lay[j].ld;
  }
  // children lists per level of the parent: first children, second children, ...
  std::vector<int> children;
  sc->lvl_slots.assign((size_t)sc->n_levels, {});
  sc->slot_max_nu.assign((size_t)sc->n_levels, {});
  for (int l = 0; l < sc->n_levels; ++l) {
    size_t most = 0;
    for (int k = sc->lvl_ptr[l]; k < sc->lvl_ptr[l + 1]; ++k) most = std::max(most, nodes[order[k]].children.size());
    for (size_t q = 0; q < most; ++q) {
      const int off = (int)children.size();
      int mx = 0;
      for (int k = sc->lvl_ptr[l]; k < sc->lvl_ptr[l + 1]; ++k) {
        const Node &nd = nodes[order[k]];
        if (q < nd.children.size()) {
          children.push_back(desc_of[nd.children[q]]);
          mx = std::max(mx, lay[nd.children[q]].nu);
        }
      }
      sc->lvl_slots[l].emplace_back(off, (int)children.size() - off);
      sc->slot_max_nu[l].push_back(mx);
    }
  }
  // device
  if (!sc->dalloc(&sc->d_arena, (size_t)a_tot) || !sc->dalloc(&sc->d_W, (size_t)w_tot) || !sc->dalloc(&sc->d_vec, (size_t)f_tot) ||
      !sc->upload(&sc->d_up_map, up_map) || !sc->upload(&sc->d_vidx, vidx) || !sc->upload(&sc->d_blk_dst, blk_dst) ||
      !sc->upload(&sc->d_blk_ld, blk_ld) || !sc->upload(&sc->d_sta_dst, sta_dst) || !sc->upload(&sc->d_sta_ld, sta_ld) ||
      !sc->upload(&sc->d_children, children)) {
    err = "out of device memory";
    sc->release();
    return 2;
  }
  std::vector<FrontDesc> desc((size_t)nn);
  for (int k = 0; k < nn; ++k) {
    const int j = order[k];
    const Lay &L = lay[j];
    FrontDesc &d = desc[k];
    d.A = sc->d_arena + L.a_off; d.W = sc->d_W + L.w_off; d.f = sc->d_vec + L.f_off;
    d.g = d.f + ((L.m + 1) & ~1);
    d.nv = L.nv; d.nvp = L.nvp; d.m = L.m; d.ld = L.ld;
    d.parent = j == 0 ? -1 : desc_of[nodes[j].parent];
    d.up_map = sc->d_up_map + L.up_off;
    d.vidx = sc->d_vidx + L.v_off;
  }
  if (!sc->upload(&sc->d_desc, desc)) {
    err = "out of device memory";
    sc->release();
    return 2;
  }
  sc->d_root_A = sc->d_arena + lay[0].a_off;
  sc->root_ld = lay[0].ld;
  sc->zeroed = false;
  return 0;
}

// Numbers: the blocks of S (with whatever diagonal the caller has added) into the fronts, then the factorisation.
// info (device) receives a non-zero value when a pivot is not positive.
// Zeros in the fronts ahead of the next factorisation, on any stream (the caller orders it after the last solve and
// before the next sparse_chol_factor): 1 GB at the 10 000-station network.
// Only the lower triangles are ever read: half the bytes of a plain fill.  Sixteen rows of one front per workgroup.
__global__ void __launch_bounds__(256)
k_sc_zero_lower(const FrontDesc *__restrict__ fd) {
  const FrontDesc f = fd[blockIdx.y];
  const int r0 = 16 * blockIdx.x;
  if (r0 >= f.m) return;
  const double2 z = make_double2(0.0, 0.0);
  for (int r = r0 + (threadIdx.x >> 4); r < min(f.m, r0 + 16); r += 16) {
    double2 *row = reinterpret_cast<double2 *>(f.A + (int64_t)r * f.ld);
    const int n2 = min(f.ld, r + 2) / 2;  // columns 0 .. r, rounded up to a pair
    for (int c = threadIdx.x & 15; c < n2; c += 16) row[c] = z;
  }
}

void sparse_chol_clear(SparseChol *sc, hipStream_t s) {
  if (sc->n_fronts > 0 && sc->largest_front > 0)
    hipLaunchKernelGGL(k_sc_zero_lower, dim3(nblk(sc->largest_front, 16), sc->n_fronts), dim3(256), 0, s, sc->d_desc);
  sc->zeroed = true;
}

void sparse_chol_factor(SparseChol *sc, hipStream_t s, const double *bsr_val, const double *Ssc, const double *Scc, int *d_info) {
  (void)hipMemsetAsync(d_info, 0, sizeof(int), s);
  if (!sc->zeroed) sparse_chol_clear(sc, s);
  sc->zeroed = false;
  const int64_t n36 = 36 * sc->n_blocks;
  if (n36 > 0) hipLaunchKernelGGL(k_sc_gather_blocks, dim3(nblk(n36, 256)), dim3(256), 0, s, n36, sc->d_blk_dst, sc->d_blk_ld, bsr_val, sc->d_arena);
  if (sc->ncam > 0) {
    hipLaunchKernelGGL(k_sc_gather_cams, dim3(nblk(6 * (int64_t)sc->S * sc->ncam, 256)), dim3(256), 0, s, sc->S, sc->ncam, sc->d_sta_dst, sc->d_sta_ld,
                       Ssc, sc->d_arena);
    hipLaunchKernelGGL(k_sc_gather_corner, dim3(nblk(sc->ncam * sc->ncam, 256)), dim3(256), 0, s, sc->ncam, Scc, sc->d_root_A, sc->root_ld);
  }
  for (int l = 0; l < sc->n_levels; ++l) {
    for (size_t q = 0; q < sc->lvl_slots[l].size(); ++q) {
      const auto sl = sc->lvl_slots[l][q];
      if (sl.second > 0 && sc->slot_max_nu[l][q] > 0)
        hipLaunchKernelGGL(k_sc_extend_add, dim3(sc->slot_max_nu[l][q], sl.second), dim3(128), 0, s, sc->d_desc, sc->d_children + sl.first);
    }
    const int nf = sc->lvl_ptr[l + 1] - sc->lvl_ptr[l];
    launch_front_panels(s, sc->d_desc + sc->lvl_ptr[l], nf, sc->lvl_maxpv[l], sc->lvl_max_m[l], d_info);
  }
}

// x = S^-1 b with the factor (b and x: the reduced vector, stations then camera columns; may be the same array)
void sparse_chol_solve(SparseChol *sc, hipStream_t s, const double *b, double *x) {
  const int CBp = front_panel();
  for (int l = 0; l < sc->n_levels; ++l) {
    const FrontDesc *fd = sc->d_desc + sc->lvl_ptr[l];
    const int nf = sc->lvl_ptr[l + 1] - sc->lvl_ptr[l];
    if (nf <= 0 || sc->lvl_max_m[l] <= 0) continue;
    hipLaunchKernelGGL(k_sc_fwd_init, dim3(nblk(sc->lvl_max_m[l], 256), nf), dim3(256), 0, s, fd, b);
    for (size_t q = 0; q < sc->lvl_slots[l].size(); ++q) {
      const auto sl = sc->lvl_slots[l][q];
      if (sl.second > 0 && sc->slot_max_nu[l][q] > 0)
        hipLaunchKernelGGL(k_sc_child_add, dim3(nblk(sc->slot_max_nu[l][q], 256), sl.second), dim3(256), 0, s, sc->d_desc, sc->d_children + sl.first);
    }
    if (sc->lvl_maxpv[l] > 0) hipLaunchKernelGGL(k_sc_fwd_first, dim3(nf), dim3(256), 0, s, fd);
    for (int kp = 0; kp < sc->lvl_maxpv[l]; ++kp) {
      const int T = nblk(sc->lvl_max_m[l], 128) - kp - 1;  // block rows below the panel, of the largest front
      if (T > 0) hipLaunchKernelGGL(k_sc_fwd_step, dim3(T, nf), dim3(256), 0, s, fd, kp);
    }
  }
  for (int l = sc->n_levels - 1; l >= 0; --l) {
    const FrontDesc *fd = sc->d_desc + sc->lvl_ptr[l];
    const int nf = sc->lvl_ptr[l + 1] - sc->lvl_ptr[l];
    if (nf <= 0 || sc->lvl_max_m[l] <= 0) continue;
    if (sc->lvl_max_nu[l] > 0) hipLaunchKernelGGL(k_sc_bwd_gather, dim3(nblk(sc->lvl_max_nu[l], 256), nf), dim3(256), 0, s, sc->d_desc, sc->lvl_ptr[l]);
    const int pv = sc->lvl_maxpv[l];
    if (pv > 0 && sc->lvl_max_nu[l] > 0) hipLaunchKernelGGL(k_sc_bwd_u, dim3(pv, SC_UCH, nf), dim3(1024), 0, s, fd);
    if (pv > 0) hipLaunchKernelGGL(k_sc_bwd_first, dim3(nf), dim3(256), 0, s, fd);
    for (int i = pv - 1; i >= 1; --i) hipLaunchKernelGGL(k_sc_bwd_step, dim3(i, nf), dim3(256), 0, s, fd, i);
    if (sc->lvl_maxpv[l] > 0) hipLaunchKernelGGL(k_sc_bwd_store, dim3(nblk(CBp * sc->lvl_maxpv[l], 256), nf), dim3(256), 0, s, fd, x);
  }
}

}  // namespace ba
