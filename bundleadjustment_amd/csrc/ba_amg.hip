// ba_amg.hip -- multilevel preconditioner for the PCG on the block-sparse reduced camera system.
//
// The reference hands the normal equations to a sparse direct factorisation inside Ceres
// (src/main.cpp:4679-4721, SPARSE_NORMAL_CHOLESKY); here the reduced system is solved by PCG, and
// block-Jacobi alone needs on the order of a thousand iterations on large strip/block networks.
// This file builds a smoothed-aggregation hierarchy over the station graph and applies it
// additively:
//
//     z = D0^-1 r + P0 ( D1^-1 P0^T r + P1 ( D2^-1 P1^T P0^T r + ... ) )
//
// with D_l the 6x6 diagonal blocks of the Galerkin matrices A_{l+1} = P_l^T A_l P_l and
// P_l = (I - w_l D_l^-1 A_l) T_l the damped-Jacobi smoothed piecewise-constant prolongator of the
// aggregates.  Every term is symmetric positive semidefinite and the first is definite, so z is a
// valid CG preconditioner.  All sums run in a fixed order (host-built gather lists, no atomics).
//
// Structure (aggregates, sparsity patterns, gather lists) is built once per problem on the host
// from the block pattern and the block norms of the first linearisation; the numeric part
// (P_l, A_l P_l, A_{l+1}, D_{l+1}^-1) is recomputed on the device after every linearisation.
#include <hip/hip_runtime.h>

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ba_cg_device.h"
#include "ba_device.h"
#include "ba_kernels.h"

namespace ba {

// ------------------------------------------------------------------------------------------
// device kernels: numeric set-up
// ------------------------------------------------------------------------------------------
__global__ void k_amg_block_norms(int n, const double *__restrict__ val, double *__restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const double *v = val + 36 * (int64_t)b;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 36; ++k) s += v[k] * v[k];
  out[b] = sqrt(s);
}

// P[i,c] = [agg(i) == c] I - omega Dinv_i sum_{j in row i, agg(j) == c} A_ij ; one wave per block of P
__global__ void __launch_bounds__(256)
k_amg_build_P(int nP, const int *__restrict__ P_row, const int *__restrict__ P_own, const int *__restrict__ Pa_ptr,
              const int *__restrict__ Pa_idx, const double *__restrict__ A_val, const double *__restrict__ Dinv, double omega,
              double *__restrict__ P_val, const int *__restrict__ P_pos, double *__restrict__ Pc_val) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= nP) return;
  const int e = lane < 36 ? lane : 0, r = e / 6, c = e % 6;
  double sum = 0.0;
  // the list's indices by one load per 64 entries, its blocks eight at a time: a loop of single loads would wait for
  // each index and each block in turn
  const int beg = Pa_ptr[t], end = Pa_ptr[t + 1];
  for (int base = beg; base < end; base += 64) {
    const int n = min(64, end - base);
    const int mine = lane < n ? Pa_idx[base + lane] : 0;
    for (int u0 = 0; u0 < n; u0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = __shfl(mine, (u0 + u) & 63, 64);
        v[u] = (u0 + u < n) ? A_val[36 * (int64_t)idx + e] : 0.0;
      }
      sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  const double *D = Dinv + 36 * (int64_t)P_row[t];
  double acc = 0.0;
#pragma unroll
  for (int m = 0; m < 6; ++m) acc += D[6 * r + m] * __shfl(sum, 6 * m + c, 64);
  const double own = (P_own[t] && r == c) ? 1.0 : 0.0;
  if (lane < 36) {
    P_val[36 * (int64_t)t + lane] = own - omega * acc;
    Pc_val[36 * (int64_t)P_pos[t] + lane] = own - omega * acc;  // the same block where its column's blocks lie side by side
  }
}

// out[o] = sum_k op(X[la[k]]) Y[lb[k]], k in [ptr[o], ptr[o+1]) ; op = transpose when TRANS.
// One wave per PAIR of output blocks, on the matrix cores: the two 6x6 products of a step sit on the diagonal of one
// 16x16 tile of v_mfma_f64_16x16x4_f64 (rows/columns 0-5 and 6-11; the summation index 6 takes two instructions), so a
// lane fetches single matrix entries straight from memory -- no shuffles -- and four steps' entries are in flight at
// once, with the indices of the next four already requested.  The lists are short (a dozen to a hundred products) and
// every product is two dependent loads (index, block): the kernel is bounded by how many of them overlap.
typedef double amg_d4 __attribute__((ext_vector_type(4)));
template <int TRANS>
__global__ void __launch_bounds__(256)
k_amg_products(int n_out, const int *__restrict__ ptr, const int *__restrict__ la, const int *__restrict__ lb,
               const double *__restrict__ X, const double *__restrict__ Y, double *__restrict__ out) {
  constexpr int U = 4;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (2 * w >= n_out) return;
  const int c16 = lane & 15, kq = lane >> 4, g = c16 / 6, e = c16 % 6;
  const int o = 2 * w + g;
  const bool active = g < 2 && o < n_out;
  const int p0 = ptr[2 * w], p1 = ptr[2 * w + 1], p2 = (2 * w + 1 < n_out) ? ptr[2 * w + 2] : p1;
  const int beg = g == 0 ? p0 : p1, len = active ? (g == 0 ? p1 - p0 : p2 - p1) : 0;
  const int nsteps = max(p1 - p0, p2 - p1);
  // entry offsets inside a block: operand A = op(X)[e][kk], operand B = Y[kk][e], kk = kq + 4 c
  int offa[2], offb[2];
  bool okk[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int kk = kq + 4 * c;
    okk[c] = kk < 6;
    offa[c] = okk[c] ? (TRANS ? 6 * kk + e : 6 * e + kk) : 0;
    offb[c] = okk[c] ? 6 * kk + e : 0;
  }
  amg_d4 acc = {0.0, 0.0, 0.0, 0.0};
  int ia[U], ib[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool on = u < len;
    ia[u] = on ? la[beg + u] : -1;
    ib[u] = on ? lb[beg + u] : -1;
  }
  for (int s = 0; s < nsteps; s += U) {
    int na[U], nb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool on = s + U + u < len;
      na[u] = on ? la[beg + s + U + u] : -1;
      nb[u] = on ? lb[beg + s + U + u] : -1;
    }
    double xa[U][2], yb[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bool on = ia[u] >= 0 && okk[c];
        xa[u][c] = on ? X[36 * (int64_t)ia[u] + offa[c]] : 0.0;
        yb[u][c] = on ? Y[36 * (int64_t)ib[u] + offb[c]] : 0.0;
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u][c], yb[u][c], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ia[u] = na[u];
      ib[u] = nb[u];
    }
  }
  // D layout: column = lane & 15, row = (lane >> 4) + 4 reg ; the block of this lane's column group is rows 6g .. 6g+5
  if (active) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = kq + 4 * reg - 6 * g;
      if (row >= 0 && row < 6) out[36 * (int64_t)o + 6 * row + e] = acc[reg];
    }
  }
}

// dst[pos[t]] = src[t], 6x6 blocks: the column-ordered copy of a block list
__global__ void k_amg_permute_blocks(int n, const double *__restrict__ src, const int *__restrict__ pos, double *__restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 36 * (int64_t)n) return;
  const int t = (int)(idx / 36), e = (int)(idx - 36 * (int64_t)t);
  dst[36 * (int64_t)pos[t] + e] = src[idx];
}

// x_out = Dinv A x_in ; one wave per block row (power iteration for the damping factor)
__global__ void __launch_bounds__(256)
k_amg_power_step(int nb, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                 const double *__restrict__ Dinv, const double *__restrict__ xin, double *__restrict__ xout) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= nb) return;
  const int g = lane >> 3, r = lane & 7;
  double s = 0.0;
  if (r < 6)
    for (int b = rowptr[i] + g; b < rowptr[i + 1]; b += 8) {
      const double *B = val + 36 * (int64_t)b + 6 * r, *x = xin + 6 * (int64_t)col[b];
#pragma unroll
      for (int m = 0; m < 6; ++m) s += B[m] * x[m];
    }
  s += __shfl_xor(s, 8, 64);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  double y = 0.0;
  const double *D = Dinv + 36 * (int64_t)i;
#pragma unroll
  for (int m = 0; m < 6; ++m) y += D[6 * (lane < 6 ? lane : 0) + m] * __shfl(s, m, 64);
  if (lane < 6) xout[6 * (int64_t)i + lane] = y;
}

__global__ void k_amg_seed(int n, double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = 0.5 + (double)((unsigned)i * 2654435761u >> 8) * (1.0 / 16777216.0);
}

__global__ void __launch_bounds__(1024)
k_amg_norm2(int n, const double *__restrict__ x, double *__restrict__ out) {
  __shared__ double lds[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i] * x[i];
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0) out[0] = sqrt(t);
}

// ------------------------------------------------------------------------------------------
// device kernels: application inside the PCG iteration
// ------------------------------------------------------------------------------------------
struct AmgTransfer {  // between a fine level and the next coarser one
  int nb, nc;
  const int *P_rowptr, *P_col;
  const double *P_val;
  const double *Pc_val;  // the blocks once more, ordered by column: a restriction reads a column as one contiguous stream
  const int *Pt_ptr, *Pt_blk, *Pt_row;  // blocks of P by column (Pt_blk: index in P_val of the block at that position of Pc_val)
  // the same lists padded to a fixed width, so that a lane finds its entry without reading a pointer array first
  // (one dependent load fewer per kernel): the first AMG_PT_ELL blocks of column c as (block, row) at Pt_ell[c * AMG_PT_ELL + k],
  // the columns of the first AMG_P_ELL blocks of row i at P_col_ell[i * AMG_P_ELL + k]; -1 = no entry.  Longer lists
  // continue in the pointer-based arrays.
  const int2 *Pt_ell;
  const int *P_col_ell;
  const double *Dinv_c;                 // inverse diagonal blocks of the coarse matrix
  double *rc, *y;                       // coarse vectors: restricted residual, correction
};
// at most AMG_MAX_LEVELS levels; a level needs AMG_MIN_BLOCKS blocks, one beyond the third AMG_DEEP_BLOCKS
// (more additive levels than that cost iterations instead of saving them)
enum { AMG_MAX_LEVELS = 3, AMG_MIN_BLOCKS = 4, AMG_DEEP_BLOCKS = 48 };
enum { AMG_PT_ELL = 64, AMG_P_ELL = 8 };
// Weights of the additive combination  z = D0^-1 r + w1 P0 (D1^-1 + w2 P1 D2^-1 P1^T) P0^T r.
// Any positive weights keep the preconditioner symmetric positive definite; what they change is how much the levels'
// overlapping corrections overshoot.  Measured on the 10M-observation network (PCG iterations per LM iteration):
// (1, 1) 195, (0.65, 1) 188, (0.5, 1.5) 185, (0.4, 2) 187, (0.3, 1) 217, (1.7, 1) 220 -- flat around (0.5, 1.5).
// A fourth level wants a weight near zero (50 000 stations: 234 iterations with it at this weight, 207 at 0.7 of it,
// 198 at 0.2, 199 without the level and two launches fewer per iteration), hence three levels at most.
#define AMG_COARSE_WEIGHT 0.5
#define AMG_DEEP_WEIGHT 1.5
struct AmgApplyArgs {
  int n_transfers;
  AmgTransfer t[AMG_MAX_LEVELS - 1];
  double *part;  // per-workgroup partials of rc . y from the restrictions that run as kernels of their own
  int n_part;
  double *totals;  // 1 + AMG_DEFL_NC sums for the last kernel (k_amg_prolong's folding workgroup)
};

// The kernels below are bounded by the latency of dependent loads (index list -> values), not by bytes: each issues
// every load that does not depend on a scalar of the iteration before it waits for anything.

// row r (= lane & 7, < 6) of the inverse diagonal block of coarse block c
__device__ __forceinline__ void load_dinv_row(const double *__restrict__ D, int r, double (&d)[6]) {
  const double2 *p = reinterpret_cast<const double2 *>(D + 6 * (r < 6 ? r : 0));
  const double2 a = p[0], b = p[1], c = p[2];
  d[0] = a.x; d[1] = a.y; d[2] = b.x; d[3] = b.y; d[4] = c.x; d[5] = c.y;
}
// lanes 0..5 of a group of 8 get y = Dinv rc (rc[k] sits in lane base + k of the same group);
// returns rc[r] * y[r] of this lane (0 for r >= 6)
__device__ __forceinline__ double coarse_jacobi(const double (&d)[6], double s, int lane, int r, double *y_out) {
  const int base = lane & ~7;
  double y = 0.0;
#pragma unroll
  for (int m = 0; m < 6; ++m) y += d[m] * __shfl(s, base + m, 64);
  *y_out = y;
  return r < 6 ? s * y : 0.0;
}

// sv = (P^T v)[c][r] (and sq likewise for a second vector q when WITHQ) in every lane with lane & 7 == r < 6.
// One wave per coarse block c.  Lane (g, r) = (lane >> 3, lane & 7) takes block g, g + 8, ... of the column and of each
// its row r: three 16-byte loads of P and one entry of v (and q); P[r][m] v[r] is summed per m in registers and the
// twelve sums are folded over the wave at the end (one transposing reduction).
struct RestrictIdx {
  int blk, row, k0, k1;
};
__device__ __forceinline__ RestrictIdx restrict_issue(const AmgTransfer &t, int c, int lane) {
  const int2 e = t.Pt_ell[(int64_t)c * AMG_PT_ELL + lane];
  RestrictIdx x;
  x.blk = e.x;
  x.row = e.y;
  x.k0 = t.Pt_ptr[c];      // only for columns longer than AMG_PT_ELL
  x.k1 = t.Pt_ptr[c + 1];
  return x;
}
template <bool WITHQ>
__device__ __forceinline__ void restrict_finish(const AmgTransfer &t, RestrictIdx x, const double *__restrict__ v, const double *__restrict__ q,
                                                int lane, double &sv, double &sq) {
  const int g = lane >> 3, r = lane & 7;
  const bool rl = r < 6;
  const int k1 = x.k1;
  double av[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, aq[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int my_blk = x.blk, my_row = x.row;
  for (int kb = x.k0;;) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double2 pv[4][3];
      double vv[4], qv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = g + 8 * (4 * h + u);
        const int blk = __shfl(my_blk, e, 64), row = __shfl(my_row, e, 64);
        const bool on = blk >= 0 && rl;
        const double2 *P = reinterpret_cast<const double2 *>(t.Pc_val + 36 * (int64_t)(on ? blk : 0) + 6 * (rl ? r : 0));
#pragma unroll
        for (int m = 0; m < 3; ++m) pv[u][m] = on ? P[m] : make_double2(0.0, 0.0);
        vv[u] = on ? v[6 * (int64_t)row + r] : 0.0;
        if (WITHQ) qv[u] = on ? q[6 * (int64_t)row + r] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          av[2 * m] += pv[u][m].x * vv[u];
          av[2 * m + 1] += pv[u][m].y * vv[u];
          if (WITHQ) {
            aq[2 * m] += pv[u][m].x * qv[u];
            aq[2 * m + 1] += pv[u][m].y * qv[u];
          }
        }
    }
    kb += 64;
    if (kb >= k1) break;  // uniform over the wave
    const int n = min(64, k1 - kb);
    my_blk = lane < n ? kb + lane : -1;  // position in the column-ordered copy
    my_row = lane < n ? t.Pt_row[kb + lane] : 0;
  }
  double w[16];
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    w[m] = av[m];
    w[6 + m] = WITHQ ? aq[m] : 0.0;
  }
  w[12] = w[13] = w[14] = w[15] = 0.0;
  const double tot = wave_sum16(w, lane);  // the total of value k sits in the lanes whose bits 5..2 spell k backwards
  const int kv = rl ? r : 0, kq = 6 + kv;
  sv = __shfl(tot, 32 * ((kv >> 3) & 1) + 16 * ((kv >> 2) & 1) + 8 * ((kv >> 1) & 1) + 4 * (kv & 1), 64);
  sq = WITHQ ? __shfl(tot, 32 * ((kq >> 3) & 1) + 16 * ((kq >> 2) & 1) + 8 * ((kq >> 1) & 1) + 4 * (kq & 1), 64) : 0.0;
}

// restriction: rc = P^T v, y = Dc^-1 rc ; one wave per coarse block; per-workgroup partials of rc . y
__global__ void __launch_bounds__(256)
k_amg_restrict(AmgTransfer t, const double *__restrict__ v, double *__restrict__ part, unsigned long long *trace) {
  __shared__ double lds[4];
  pcg_mark(trace, 2, 0, blockIdx.x);
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = lane & 7;
  double dot = 0.0;
  if (c < t.nc) {
    double d[6], s, unused;
    load_dinv_row(t.Dinv_c + 36 * (int64_t)c, r, d);
    restrict_finish<false>(t, restrict_issue(t, c, lane), v, nullptr, lane, s, unused);
    double y;
    const double dd = coarse_jacobi(d, s, lane, r, &y);
    if (lane < 6) {
      t.rc[6 * (int64_t)c + lane] = s;
      t.y[6 * (int64_t)c + lane] = y;
      dot = dd;
    }
  }
  pcg_mark(trace, 2, 1, blockIdx.x);
  const double tot = block_sum(dot, lds);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
  pcg_mark(trace, 2, 2, blockIdx.x);
}

// Second kernel of a multilevel PCG iteration: the vector update and the first restriction in one
// launch.  Workgroups [0, nbv) update x, r, z -- 8 lanes per station, lane r < 6 holding entry r, so that every load of
// the role is independent and issued at once (a thread per station needs 800 bytes in four dependent batches, on forty
// CUs) --; the others restrict the NEW residual r_in - alpha q: P^T r_in and P^T q are summed before alpha is known, so
// nothing but arithmetic waits for the fold.  The residual is double buffered by iteration parity because the two
// groups run concurrently.
__global__ void __launch_bounds__(256, 2)
k_amg_update1_restrict(BsrCgArgs a, AmgTransfer t, int parity, int nblk_spmv, int nbv, double *__restrict__ part) {
  __shared__ Cg2Shared sh;
  pcg_mark(a.trace, 1, 0, blockIdx.x);
  const double done = a.scal[8 + parity], rz = a.scal[parity];
  const double *rin = parity ? a.r2 : a.r;
  double *rout = parity ? a.r : a.r2;
  FoldLoads f;
  fold_issue(a, nblk_spmv, 0, f);
  if ((int)blockIdx.x < nbv) {
    // 64 stations per workgroup, two per group of 8 lanes (so that the vector and the restricting workgroups are all
    // resident at once: two per CU at this kernel's register count)
    const int r = threadIdx.x & 7;
    const int64_t o = 6 * (int64_t)a.S;
    bool mine[2];
    int64_t e[2];
    double2 mv[2][3], rv[2][3], qv[2][3];
    double r_own[2], q_own[2], x_own[2], p_own[2], g[2][DEFL_NC];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = blockIdx.x * 64 + 32 * h + (threadIdx.x >> 3);
      mine[h] = i < a.S && r < 6;
      const int64_t e0 = 6 * (int64_t)(i < a.S ? i : 0);
      e[h] = e0 + (r < 6 ? r : 0);
      const double2 *M = reinterpret_cast<const double2 *>(a.Minv + 6 * e[h]);  // row r of the station's inverse block
      const double2 *R = reinterpret_cast<const double2 *>(rin + e0), *Q = reinterpret_cast<const double2 *>(a.q + e0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        mv[h][k] = M[k];
        rv[h][k] = R[k];
        qv[h][k] = Q[k];
      }
      r_own[h] = rin[e[h]];
      q_own[h] = a.q[e[h]];
      x_own[h] = a.x[e[h]];
      p_own[h] = a.p[e[h]];
#pragma unroll
      for (int k = 0; k < DEFL_NC; ++k) g[h][k] = (a.G && mine[h] && k < a.ncam) ? a.GT[k * o + e[h]] : 0.0;  // G by columns
    }
    const bool last = (int)blockIdx.x == nbv - 1;
    const double rc = (last && (int)threadIdx.x < a.ncam) ? rin[o + threadIdx.x] : 0.0;
    if (done != 0.0) return;  // waits for the flag alone: the loads above stay in flight
    pcg_mark(a.trace, 1, 1, blockIdx.x);
    double u[2], w[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u[h] = mv[h][0].x * rv[h][0].x + mv[h][0].y * rv[h][0].y + mv[h][1].x * rv[h][1].x + mv[h][1].y * rv[h][1].y + mv[h][2].x * rv[h][2].x + mv[h][2].y * rv[h][2].y;
      w[h] = mv[h][0].x * qv[h][0].x + mv[h][0].y * qv[h][0].y + mv[h][1].x * qv[h][1].x + mv[h][1].y * qv[h][1].y + mv[h][2].x * qv[h][2].x + mv[h][2].y * qv[h][2].y;
    }
    const double alpha = cg2_alpha(a, rz, nblk_spmv, f, sh);
    pcg_mark(a.trace, 1, 2, blockIdx.x);
    double dot = 0.0;
    double gr[DEFL_NC];
#pragma unroll
    for (int k = 0; k < DEFL_NC; ++k) gr[k] = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double rn = r_own[h] - alpha * q_own[h], z = u[h] - alpha * w[h];
      if (mine[h]) {
        a.x[e[h]] = x_own[h] + alpha * p_own[h];
        rout[e[h]] = rn;
        a.z[e[h]] = z;
        dot += z * rn;
      }
#pragma unroll
      for (int k = 0; k < DEFL_NC; ++k) gr[k] += g[h][k] * rn;  // g is zero in the lanes that hold nothing
    }
    if (a.G) defl_partials(a, gr, blockIdx.x);
    if (last) {
      // the camera entries (ncam <= CG2_MAX_CAM): z_c = Scc^-1 r_c with the explicit inverse, one thread per row; with
      // the camera coarse space only x and r here, the camera part of z follows in the last kernel
      if ((int)threadIdx.x < a.ncam) {
        a.x[o + threadIdx.x] += alpha * sh.pcs[threadIdx.x];
        const double rk = rc - alpha * sh.red[1 + threadIdx.x];
        rout[o + threadIdx.x] = rk;
        sh.rcs[threadIdx.x] = rk;
      }
      __syncthreads();
      if (!a.G && (int)threadIdx.x < a.ncam) {
        double zc = 0.0;
        for (int m = 0; m < a.ncam; ++m) zc += a.Lcc[threadIdx.x * a.ncam + m] * sh.rcs[m];
        a.z[o + threadIdx.x] = zc;
        dot += sh.rcs[threadIdx.x] * zc;
      }
      if (threadIdx.x == 0) a.scal[4] = sh.pq;
    }
    pcg_mark(a.trace, 1, 3, blockIdx.x);
    const double tot = block_sum(dot, sh.lds);
    if (threadIdx.x == 0) a.part_rz[blockIdx.x] = tot;
    pcg_mark(a.trace, 1, 4, blockIdx.x);
    return;
  }
  const int cb = blockIdx.x - nbv;
  const int c = cb * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 7;
  double d[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, sv = 0.0, sq = 0.0;
  RestrictIdx ri{-1, 0, 0, 0};
  if (c < t.nc) {
    load_dinv_row(t.Dinv_c + 36 * (int64_t)c, r, d);
    ri = restrict_issue(t, c, lane);
  }
  if (done != 0.0) return;  // an iteration after convergence costs one trip, not the whole kernel
  if (c < t.nc) restrict_finish<true>(t, ri, rin, a.q, lane, sv, sq);
  pcg_mark(a.trace, 1, 1, blockIdx.x);
  const double alpha = cg2_alpha(a, rz, nblk_spmv, f, sh);
  pcg_mark(a.trace, 1, 2, blockIdx.x);
  double dot = 0.0;
  if (c < t.nc) {
    const double s = sv - alpha * sq;
    double y;
    const double dd = coarse_jacobi(d, s, lane, r, &y);
    if (lane < 6) {
      t.rc[6 * (int64_t)c + lane] = s;
      t.y[6 * (int64_t)c + lane] = y;
      dot = dd;
    }
  }
  pcg_mark(a.trace, 1, 3, blockIdx.x);
  const double tot = block_sum(dot, sh.lds);
  if (threadIdx.x == 0) part[cb] = tot;
  pcg_mark(a.trace, 1, 4, blockIdx.x);
}

// (P yc)[i][r] for the 8 lanes of a group (r = lane & 7 < 6): the column indices of the row's first AMG_P_ELL blocks
// come from the padded list (no pointer needed), so they travel together with the row pointer (prolong_issue); the
// value loads follow (prolong_finish)
struct ProlongIdx {
  int col, k0, k1;
};
__device__ __forceinline__ ProlongIdx prolong_issue(const AmgTransfer &t, int i, int lane) {
  ProlongIdx x;
  x.col = t.P_col_ell[(int64_t)i * AMG_P_ELL + (lane & 7)];
  x.k0 = t.P_rowptr[i];
  x.k1 = t.P_rowptr[i + 1];
  return x;
}
__device__ __forceinline__ double prolong_finish(const AmgTransfer &t, ProlongIdx x, int lane) {
  const int r = lane & 7, rr = r < 6 ? r : 0, base = lane & ~7;
  int my_col = x.col;
  const int k0 = x.k0, k1 = x.k1;
  double s = 0.0;
  for (int kb = k0;;) {
    double2 pv[8][3], yv[8][3];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int col = __shfl(my_col, base + u, 64);
      const bool on = col >= 0;
      const double2 *P = reinterpret_cast<const double2 *>(t.P_val + 36 * (int64_t)(on ? kb + u : k0) + 6 * rr);
      const double2 *yc = reinterpret_cast<const double2 *>(t.y + 6 * (int64_t)(on ? col : 0));
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        pv[u][m] = on ? P[m] : make_double2(0.0, 0.0);
        yv[u][m] = on ? yc[m] : make_double2(0.0, 0.0);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int m = 0; m < 3; ++m) s += pv[u][m].x * yv[u][m].x + pv[u][m].y * yv[u][m].y;
    kb += 8;
    if (kb >= k1) break;  // not uniform over the wave: the shuffles only read lanes of the own group
    const int n = min(8, k1 - kb);
    my_col = r < n ? t.P_col[kb + r] : -1;
  }
  return s;
}

// What the last kernel of an iteration needs summed: the partials of r.z of the vector kernel (part_rz) and of the
// restrictions (part, level 0 up to np0 with weight sigma, the deeper ones with sigma tau), and the partials of G^T r
// (part_gr, n_part_rz rows of ncam; null without the camera coarse space).  totals: [0] r.z, [1 + k] (G^T r)_k.
struct AmgFoldArgs {
  const double *part_rz, *part, *part_gr;
  int n_part_rz, n_part, np0, ncam;
  double sigma, tau;
  double *totals;
};
// The sums, in a fixed order, by one workgroup of 256 threads (the loads eight / four at a time)
__device__ __forceinline__ void amg_fold(const AmgFoldArgs &fa) {
  __shared__ double lds[4];
  __shared__ double gr4[4][16];
  double g = 0.0;
  if (fa.part_gr) {
    // 16 lanes per sum, the partials dealt round-robin; then across the wave and the waves
    const int k = threadIdx.x & 15, j = threadIdx.x >> 4;
    if (k < fa.ncam)
      for (int m = j; m < fa.n_part_rz; m += 128) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (m + 16 * u < fa.n_part_rz) ? fa.part_gr[(int64_t)(m + 16 * u) * fa.ncam + k] : 0.0;
        g += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
  }
  double s = 0.0;
  for (int k = threadIdx.x; k < fa.n_part_rz; k += 4 * blockDim.x) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (k + u * (int)blockDim.x < fa.n_part_rz) ? fa.part_rz[k + u * blockDim.x] : 0.0;
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  for (int k = threadIdx.x; k < fa.n_part; k += 4 * blockDim.x) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k + u * (int)blockDim.x;
      v[u] = (kk < fa.n_part) ? (kk < fa.np0 ? fa.sigma : fa.sigma * fa.tau) * fa.part[kk] : 0.0;
    }
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  g += __shfl_xor(g, 16, 64);
  g += __shfl_xor(g, 32, 64);
  if ((threadIdx.x & 63) < 16) gr4[threadIdx.x >> 6][threadIdx.x & 15] = g;
  const double tot = block_sum(s, lds);  // has the barriers that publish gr4
  if (threadIdx.x == 0) fa.totals[0] = tot;
  if (fa.part_gr && (int)threadIdx.x < fa.ncam)
    fa.totals[1 + threadIdx.x] = (gr4[0][threadIdx.x] + gr4[1][threadIdx.x]) + (gr4[2][threadIdx.x] + gr4[3][threadIdx.x]);
}

// prolongation: yf += tau P yc ; 8 lanes per fine block row (whole waves stay converged for the shuffles).  With
// fold.totals set, one more workgroup (the last) forms the sums for the last kernel of the iteration: every partial is
// final by now, and summed here once instead of by each of that kernel's workgroups -- their number and the number of
// partials both grow with the stations.
__global__ void __launch_bounds__(256)
k_amg_prolong(AmgTransfer t, double *__restrict__ yf, double tau, AmgFoldArgs fold, unsigned long long *trace) {
  if (fold.totals && blockIdx.x == gridDim.x - 1) {
    amg_fold(fold);
    return;
  }
  pcg_mark(trace, 3, 0, blockIdx.x);
  const int i = blockIdx.x * 32 + (threadIdx.x >> 3), r = threadIdx.x & 7, lane = threadIdx.x & 63;
  const bool on = i < t.nb;
  const double y0 = (on && r < 6) ? yf[6 * (int64_t)i + r] : 0.0;
  const double s = prolong_finish(t, prolong_issue(t, on ? i : 0, lane), lane);
  pcg_mark(trace, 3, 1, blockIdx.x);
  if (on && r < 6) yf[6 * (int64_t)i + r] = y0 + tau * s;
  pcg_mark(trace, 3, 2, blockIdx.x);
}

#define AMG_DEFL_NC 8

// Last kernel of an iteration.  Every workgroup folds the partials of r.z = r.D0^-1 r + sum_l rc_l.y_l
// in the same order (identical beta everywhere), each level's share with its weight (sigma, then sigma tau; np0: where
// the partials of the first level end); z = D0^-1 r (already in cg.z) + sigma P0 y1;
// p = z + beta p (mode 1) or p = z (mode 0, start of the solve); 8 lanes per station.
// Workgroup 0 publishes the scalars for the next iteration in the other parity slot.
__global__ void __launch_bounds__(256)
k_amg_update2(BsrCgArgs cg, AmgTransfer t, int has_coarse, const double *__restrict__ part, int n_part, int n_part_rz,
              int parity, int mode, double rtol, double sigma, double tau, int np0, const double *__restrict__ totals) {
  __shared__ double lds[4];
  __shared__ double rzn_s;
  __shared__ double Ls[AMG_DEFL_NC * AMG_DEFL_NC], gr4[4][16], zcs[AMG_DEFL_NC], tcs[AMG_DEFL_NC], tot_s;
  pcg_mark(cg.trace, 4, 0, blockIdx.x);
  // ---- loads: none of them depends on beta
  const double done = (mode == 1) ? cg.scal[8 + parity] : 0.0;
  const double rz_old = (mode == 1) ? cg.scal[parity] : 1.0;
  const double it_old = cg.scal[3], pq_last = cg.scal[4], target = cg.scal[6];
  const int64_t o = 6 * (int64_t)cg.S;
  const bool defl = cg.G != nullptr;  // then ncam <= AMG_DEFL_NC
  const int i = blockIdx.x * 32 + (threadIdx.x >> 3), r = threadIdx.x & 7;
  const bool on = i < cg.S, mine = on && r < 6;
  const double z0 = mine ? cg.z[6 * (int64_t)i + r] : 0.0;
  const double p0 = (mine && mode == 1) ? cg.p[6 * (int64_t)i + r] : 0.0;
  double gi[AMG_DEFL_NC];
#pragma unroll
  for (int k = 0; k < AMG_DEFL_NC; ++k) gi[k] = (defl && mine && k < cg.ncam) ? cg.G[(6 * (int64_t)i + r) * cg.ncam + k] : 0.0;
  const double lsv = (defl && (int)threadIdx.x < cg.ncam * cg.ncam) ? cg.Lcc[threadIdx.x] : 0.0;
  const double rcv = (defl && (int)threadIdx.x < cg.ncam) ? ((mode == 1 && !parity) ? cg.r2 : cg.r)[o + threadIdx.x] : 0.0;
  ProlongIdx pi{-1, 0, 0};
  if (has_coarse) pi = prolong_issue(t, on ? i : 0, threadIdx.x & 63);
  if (done != 0.0) {  // waits for the flag alone; an iteration after convergence costs one trip
    if (blockIdx.x == 0 && threadIdx.x == 0) cg.scal[8 + (parity ^ 1)] = done;  // keep both slots set
    return;
  }
  double g = 0.0, s = 0.0;
  if (totals) {
    // summed already (k_amg_prolong's last workgroup): placed where the reductions below expect partial sums
    if (threadIdx.x == 0) s = totals[0];
    if (defl && (int)threadIdx.x < cg.ncam) g = totals[1 + threadIdx.x];
  } else {
    if (defl) {
      // G^T r_s: 16 lanes per sum, the partials dealt round-robin; then across the wave and the waves
      const int k = threadIdx.x & 15, j = threadIdx.x >> 4;
      // eight loads at a time: a loop of single loads would wait for each in turn
      if (k < cg.ncam)
        for (int m = j; m < n_part_rz; m += 128) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = (m + 16 * u < n_part_rz) ? cg.part_gr[(int64_t)(m + 16 * u) * cg.ncam + k] : 0.0;
          g += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    for (int k = threadIdx.x; k < n_part_rz; k += 4 * blockDim.x) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (k + u * (int)blockDim.x < n_part_rz) ? cg.part_rz[k + u * blockDim.x] : 0.0;
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (int k = threadIdx.x; k < n_part; k += 4 * blockDim.x) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k + u * (int)blockDim.x;
        v[u] = (kk < n_part) ? (kk < np0 ? sigma : sigma * tau) * part[kk] : 0.0;
      }
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  }
  const double corr = has_coarse ? prolong_finish(t, pi, threadIdx.x & 63) : 0.0;
  pcg_mark(cg.trace, 4, 1, blockIdx.x);
  // ---- scalars
  if (defl) {
    if ((int)threadIdx.x < cg.ncam * cg.ncam) Ls[threadIdx.x] = lsv;
    g += __shfl_xor(g, 16, 64);
    g += __shfl_xor(g, 32, 64);
    if ((threadIdx.x & 63) < 16) gr4[threadIdx.x >> 6][threadIdx.x & 15] = g;
  }
  const double tot = block_sum(s, lds);  // has the barriers that publish Ls and gr4
  if (threadIdx.x == 0) tot_s = tot;
  if (defl && threadIdx.x < cg.ncam)  // t_c = r_c - G^T r_s
    tcs[threadIdx.x] = rcv - ((gr4[0][threadIdx.x] + gr4[1][threadIdx.x]) + (gr4[2][threadIdx.x] + gr4[3][threadIdx.x]));
  __syncthreads();
  if (defl && threadIdx.x < cg.ncam) {  // z_c = (Q^T S Q)^-1 t_c with the explicit inverse
    double zc = 0.0;
    for (int m = 0; m < cg.ncam; ++m) zc += Ls[threadIdx.x * cg.ncam + m] * tcs[m];
    zcs[threadIdx.x] = zc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double extra = 0.0;  // the camera share of r.z is t_c . z_c
    if (defl)
      for (int k = 0; k < cg.ncam; ++k) extra += tcs[k] * zcs[k];
    rzn_s = tot_s + extra;
  }
  __syncthreads();
  const double rzn = rzn_s;
  const double beta = (mode == 1) ? rzn / rz_old : 0.0;
  pcg_mark(cg.trace, 4, 2, blockIdx.x);
  if (mine) {
    double z = z0 + sigma * corr;
    if (defl) {
#pragma unroll
      for (int k = 0; k < AMG_DEFL_NC; ++k)
        if (k < cg.ncam) z -= gi[k] * zcs[k];
    }
    cg.p[6 * (int64_t)i + r] = (mode == 1) ? z + beta * p0 : z;
  }
  pcg_mark(cg.trace, 4, 3, blockIdx.x);
  if (defl && blockIdx.x == 0 && threadIdx.x < cg.ncam) cg.z[o + threadIdx.x] = zcs[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < cg.ncam; ++k) {
      const double zc = defl ? zcs[k] : cg.z[o + k];
      cg.p[o + k] = (mode == 1) ? zc + beta * cg.p[o + k] : zc;
    }
    if (mode == 0) {
      cg.scal[0] = cg.scal[1] = cg.scal[2] = rzn;
      cg.scal[3] = cg.scal[4] = cg.scal[5] = 0.0;
      cg.scal[6] = rtol * sqrt(fmax(rzn, 0.0));
      cg.scal[8] = cg.scal[9] = (rzn > 0.0) ? 0.0 : 1.0;
    } else {
      cg.scal[parity ^ 1] = rzn;
      cg.scal[3] = it_old + 1.0;
      double nd = 0.0;
      if (!(pq_last > 0.0) || !(rzn == rzn)) nd = 2.0;
      else if (sqrt(fmax(rzn, 0.0)) <= target) nd = 1.0;
      cg.scal[8 + (parity ^ 1)] = nd;
    }
  }
  pcg_mark(cg.trace, 4, 4, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// host: hierarchy
// ------------------------------------------------------------------------------------------
namespace {

static inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

struct Pattern {
  int nb = 0;
  std::vector<int> rowptr, col;
};

struct Level {  // transfer from level l to level l + 1 and everything the numeric set-up of it needs
  int nb = 0, nc = 0, nA = 0, nP = 0, nAP = 0, nAc = 0;
  double omega = 0.0;
  // fine matrix (level 0: borrowed from the solver)
  const int *A_rowptr = nullptr, *A_col = nullptr;
  const double *A_val = nullptr, *Dinv = nullptr;
  // owned device arrays
  int *P_rowptr = nullptr, *P_col = nullptr, *P_row = nullptr, *P_own = nullptr, *Pa_ptr = nullptr, *Pa_idx = nullptr;
  int *AP_ptr = nullptr, *AP_la = nullptr, *AP_lb = nullptr, *Ac_ptr = nullptr, *Ac_la = nullptr, *Ac_lb = nullptr;
  int *Pt_ptr = nullptr, *Pt_blk = nullptr, *Pt_row = nullptr, *P_col_ell = nullptr;
  int2 *Pt_ell = nullptr;
  int *Ac_rowptr = nullptr, *Ac_col = nullptr, *Ac_diag = nullptr, *used_c = nullptr;
  double *Pc_val = nullptr;
  int *P_pos = nullptr;
  double *P_val = nullptr, *AP_val = nullptr, *Ac_val = nullptr, *Dinv_c = nullptr, *rc = nullptr, *y = nullptr;
};

}  // namespace

struct Amg {
  std::vector<Level> levels;
  std::vector<void *> allocs;
  double *d_tmp = nullptr, *d_tmp2 = nullptr, *d_part = nullptr, *d_norm = nullptr, *d_totals = nullptr;
  unsigned *d_flags = nullptr;
  int n_part = 0;  // per-workgroup partials of all restrictions
  char err[256] = {0};

  template <class T>
  bool dalloc(T **p, size_t n) {
    void *q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) {
      snprintf(err, sizeof err, "multilevel preconditioner: hipMalloc of %zu bytes failed", n * sizeof(T));
      return false;
    }
    allocs.push_back(q);
    *p = static_cast<T *>(q);
    return true;
  }
  template <class T>
  bool upload(T **p, const std::vector<T> &v) {
    if (!dalloc(p, v.size())) return false;
    if (!v.empty() && hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
      snprintf(err, sizeof err, "multilevel preconditioner: upload failed");
      return false;
    }
    return true;
  }
  void release() {
    for (void *q : allocs) (void)hipFree(q);
    allocs.clear();
    levels.clear();
  }
};

namespace {

// Greedy aggregation (Vanek et al.): a neighbour j of i is strong when
// |A_ij| / sqrt(|A_ii| |A_jj|) >= theta * (largest such ratio in row i), |.| = Frobenius norm.
void aggregate(const Pattern &A, const std::vector<double> &fro, double theta, std::vector<int> &agg, int &nc) {
  const int nb = A.nb;
  std::vector<double> dg(nb, 0.0);
  for (int i = 0; i < nb; ++i)
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b)
      if (A.col[b] == i) dg[i] = fro[b];
  std::vector<double> sij(A.col.size(), 0.0);
  std::vector<char> strong(A.col.size(), 0);
  for (int i = 0; i < nb; ++i) {
    double rmax = 0.0;
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b) {
      const int j = A.col[b];
      if (j == i) continue;
      const double d = dg[i] * dg[j];
      sij[b] = d > 0.0 ? fro[b] / std::sqrt(d) : 0.0;
      rmax = std::max(rmax, sij[b]);
    }
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b)
      strong[b] = (A.col[b] != i && sij[b] > 0.0 && sij[b] >= theta * rmax) ? 1 : 0;
  }
  agg.assign(nb, -1);
  nc = 0;
  for (int i = 0; i < nb; ++i) {  // roots whose strong neighbourhood is untouched
    if (agg[i] >= 0) continue;
    bool free_nb = true;
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1] && free_nb; ++b)
      if (strong[b] && agg[A.col[b]] >= 0) free_nb = false;
    if (!free_nb) continue;
    agg[i] = nc;
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b)
      if (strong[b]) agg[A.col[b]] = nc;
    ++nc;
  }
  const std::vector<int> first = agg;
  for (int i = 0; i < nb; ++i) {  // leftovers join the aggregate of their strongest aggregated neighbour
    if (agg[i] >= 0) continue;
    double best = 0.0;
    int to = -1;
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b)
      if (strong[b] && first[A.col[b]] >= 0 && sij[b] > best) {
        best = sij[b];
        to = first[A.col[b]];
      }
    if (to >= 0) agg[i] = to;
  }
  for (int i = 0; i < nb; ++i) {  // whatever is still free forms aggregates of its own
    if (agg[i] >= 0) continue;
    agg[i] = nc;
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b)
      if (strong[b] && agg[A.col[b]] < 0) agg[A.col[b]] = nc;
    ++nc;
  }
}

struct Symbolic {
  std::vector<int> P_rowptr, P_col, P_row, P_own, Pa_ptr, Pa_idx;
  std::vector<int> AP_rowptr, AP_col, AP_ptr, AP_la, AP_lb;
  std::vector<int> Ac_ptr, Ac_la, Ac_lb, Ac_diag;
  std::vector<int> Pt_ptr, Pt_blk, Pt_row, used_c, P_col_ell, P_pos;
  std::vector<int2> Pt_ell;
  Pattern Ac;
};

void symbolic(const Pattern &A, const std::vector<int> &agg, int nc, const std::vector<int> &used, Symbolic &sy) {
  const int nb = A.nb;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tt0 = now();
  // ---- P: one block per (row, aggregate seen in the row)
  sy.P_rowptr.assign(nb + 1, 0);
  sy.Pa_ptr.assign(1, 0);
  std::vector<std::pair<int, int>> tmp;
  for (int i = 0; i < nb; ++i) {
    tmp.clear();
    for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b) tmp.emplace_back(agg[A.col[b]], b);
    std::sort(tmp.begin(), tmp.end());
    for (size_t k = 0; k < tmp.size(); ++k) {
      if (k == 0 || tmp[k].first != tmp[k - 1].first) {
        if (k) sy.Pa_ptr.push_back((int)sy.Pa_idx.size());
        sy.P_col.push_back(tmp[k].first);
        sy.P_row.push_back(i);
        sy.P_own.push_back(tmp[k].first == agg[i] ? 1 : 0);
      }
      sy.Pa_idx.push_back(tmp[k].second);
    }
    if (!tmp.empty()) sy.Pa_ptr.push_back((int)sy.Pa_idx.size());
    sy.P_rowptr[i + 1] = (int)sy.P_col.size();
  }
  const int nP = (int)sy.P_col.size();
  const double tt1 = now();
  // ---- A P: row i = union of the rows of P over the neighbours of i; per block (i, c) the products (b, t) in the order
  //      they are met (b, then t, ascending).  Two passes per row with a marker per coarse column -- count, then place --
  //      instead of sorting the row's triples.
  sy.AP_rowptr.assign(nb + 1, 0);
  sy.AP_ptr.assign(1, 0);
  {
    std::vector<int> slot(nc, -1), cols, cnt, order, pos;
    for (int i = 0; i < nb; ++i) {
      cols.clear();
      cnt.clear();
      for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b) {
        const int j = A.col[b];
        for (int t = sy.P_rowptr[j]; t < sy.P_rowptr[j + 1]; ++t) {
          const int c = sy.P_col[t];
          if (slot[c] < 0) {
            slot[c] = (int)cols.size();
            cols.push_back(c);
            cnt.push_back(0);
          }
          cnt[slot[c]] += 1;
        }
      }
      const int m = (int)cols.size();
      order.resize(m);
      for (int k = 0; k < m; ++k) order[k] = k;
      std::sort(order.begin(), order.end(), [&](int x, int y) { return cols[x] < cols[y]; });
      pos.resize(m);
      int64_t base = (int64_t)sy.AP_la.size();
      for (int k = 0; k < m; ++k) {
        const int q = order[k];
        pos[q] = (int)base;
        base += cnt[q];
        sy.AP_col.push_back(cols[q]);
        sy.AP_ptr.push_back((int)base);
      }
      sy.AP_la.resize((size_t)base);
      sy.AP_lb.resize((size_t)base);
      for (int b = A.rowptr[i]; b < A.rowptr[i + 1]; ++b) {
        const int j = A.col[b];
        for (int t = sy.P_rowptr[j]; t < sy.P_rowptr[j + 1]; ++t) {
          const int q = slot[sy.P_col[t]];
          sy.AP_la[pos[q]] = b;
          sy.AP_lb[pos[q]] = t;
          pos[q] += 1;
        }
      }
      for (int k = 0; k < m; ++k) slot[cols[k]] = -1;
      sy.AP_rowptr[i + 1] = (int)sy.AP_col.size();
    }
  }
  const double tt2 = now();
  // ---- A_c = P^T (A P): the products (t, u) of block (c, d) = (P_col[t], AP_col[u]) over the fine rows, ordered by c,
  //      then d, then as generated: a counting sort by c, then count-and-place by d inside each coarse row (sorting the
  //      whole list and the rows' triples was 50 of the 65 ms this function took at 10 000 stations).
  struct Ent {
    int d, t, u;
  };
  std::vector<int64_t> start(nc + 1, 0);
  for (int i = 0; i < nb; ++i)
    for (int t = sy.P_rowptr[i]; t < sy.P_rowptr[i + 1]; ++t) start[sy.P_col[t] + 1] += sy.AP_rowptr[i + 1] - sy.AP_rowptr[i];
  for (int c = 0; c < nc; ++c) start[c + 1] += start[c];
  std::vector<Ent> ent((size_t)start[nc]);
  {
    std::vector<int64_t> pos(start.begin(), start.end() - 1);
    for (int i = 0; i < nb; ++i)
      for (int t = sy.P_rowptr[i]; t < sy.P_rowptr[i + 1]; ++t) {
        int64_t &q = pos[sy.P_col[t]];
        for (int u = sy.AP_rowptr[i]; u < sy.AP_rowptr[i + 1]; ++u) ent[(size_t)q++] = {sy.AP_col[u], t, u};
      }
  }
  sy.Ac.nb = nc;
  sy.Ac.rowptr.assign(nc + 1, 0);
  sy.Ac_ptr.assign(1, 0);
  sy.Ac_diag.assign(nc, 0);
  {
    // inside a coarse row: count per column d, then place (the entries of a column keep their order)
    std::vector<int> slot(nc, -1), cols, cnt, order, pos;
    for (int c = 0; c < nc; ++c) {
      cols.clear();
      cnt.clear();
      for (int64_t k = start[c]; k < start[c + 1]; ++k) {
        const int d = ent[(size_t)k].d;
        if (slot[d] < 0) {
          slot[d] = (int)cols.size();
          cols.push_back(d);
          cnt.push_back(0);
        }
        cnt[slot[d]] += 1;
      }
      const int m = (int)cols.size();
      order.resize(m);
      for (int k = 0; k < m; ++k) order[k] = k;
      std::sort(order.begin(), order.end(), [&](int x, int y) { return cols[x] < cols[y]; });
      pos.resize(m);
      int64_t base = (int64_t)sy.Ac_la.size();
      for (int k = 0; k < m; ++k) {
        const int q = order[k];
        if (base > 0) sy.Ac_ptr.push_back((int)base);
        pos[q] = (int)base;
        base += cnt[q];
        if (cols[q] == c) sy.Ac_diag[c] = (int)sy.Ac.col.size();
        sy.Ac.col.push_back(cols[q]);
        sy.Ac.rowptr[c + 1] += 1;
      }
      sy.Ac_la.resize((size_t)base);
      sy.Ac_lb.resize((size_t)base);
      for (int64_t k = start[c]; k < start[c + 1]; ++k) {
        const int q = slot[ent[(size_t)k].d];
        sy.Ac_la[pos[q]] = ent[(size_t)k].t;
        sy.Ac_lb[pos[q]] = ent[(size_t)k].u;
        pos[q] += 1;
      }
      for (int k = 0; k < m; ++k) slot[cols[k]] = -1;
    }
  }
  if (!ent.empty()) sy.Ac_ptr.push_back((int)sy.Ac_la.size());
  for (int c = 0; c < nc; ++c) sy.Ac.rowptr[c + 1] += sy.Ac.rowptr[c];
  const double tt3 = now();
  // ---- P by column (restriction), coarse "has observations" flags
  sy.Pt_ptr.assign(nc + 1, 0);
  for (int t = 0; t < nP; ++t) sy.Pt_ptr[sy.P_col[t] + 1] += 1;
  for (int c = 0; c < nc; ++c) sy.Pt_ptr[c + 1] += sy.Pt_ptr[c];
  sy.Pt_blk.resize(nP);
  sy.P_pos.resize(nP);
  sy.Pt_row.resize(nP);
  std::vector<int> fill(sy.Pt_ptr.begin(), sy.Pt_ptr.end() - 1);
  for (int t = 0; t < nP; ++t) {
    const int k = fill[sy.P_col[t]]++;
    sy.Pt_blk[k] = t;
    sy.P_pos[t] = k;
    sy.Pt_row[k] = sy.P_row[t];
  }
  // the padded copies the kernels of the iteration start from
  sy.Pt_ell.assign((size_t)nc * AMG_PT_ELL, make_int2(-1, 0));
  for (int c = 0; c < nc; ++c)
    for (int k = sy.Pt_ptr[c]; k < sy.Pt_ptr[c + 1] && k - sy.Pt_ptr[c] < AMG_PT_ELL; ++k)
      sy.Pt_ell[(size_t)c * AMG_PT_ELL + (k - sy.Pt_ptr[c])] = make_int2(k, sy.Pt_row[k]);  // (position in Pc_val, fine row)
  sy.P_col_ell.assign((size_t)nb * AMG_P_ELL, -1);
  for (int i = 0; i < nb; ++i)
    for (int k = sy.P_rowptr[i]; k < sy.P_rowptr[i + 1] && k - sy.P_rowptr[i] < AMG_P_ELL; ++k)
      sy.P_col_ell[(size_t)i * AMG_P_ELL + (k - sy.P_rowptr[i])] = sy.P_col[k];
  sy.used_c.assign(nc, 0);
  for (int i = 0; i < nb; ++i)
    if (used[i]) sy.used_c[agg[i]] = 1;
  if (std::getenv("BA_AMG_VERBOSE"))
    fprintf(stderr, "amg: lists of P %.1f ms, of A P %.1f (%zu products), of P^T A P %.1f (%zu products), by column %.1f\n", tt1 - tt0, tt2 - tt1,
            sy.AP_la.size(), tt3 - tt2, sy.Ac_la.size(), now() - tt3);
}

void numeric_level(hipStream_t s, const Level &L, unsigned *flags) {
  hipLaunchKernelGGL(k_amg_build_P, dim3(nblk(L.nP, 4)), dim3(256), 0, s, L.nP, L.P_row, L.P_own, L.Pa_ptr, L.Pa_idx, L.A_val,
                     L.Dinv, L.omega, L.P_val, L.P_pos, L.Pc_val);
  hipLaunchKernelGGL(k_amg_products<0>, dim3(nblk(L.nAP, 8)), dim3(256), 0, s, L.nAP, L.AP_ptr, L.AP_la, L.AP_lb, L.A_val,
                     L.P_val, L.AP_val);
  hipLaunchKernelGGL(k_amg_products<1>, dim3(nblk(L.nAc, 8)), dim3(256), 0, s, L.nAc, L.Ac_ptr, L.Ac_la, L.Ac_lb, L.P_val,
                     L.AP_val, L.Ac_val);
  launch_invert_blocks(s, L.nc, L.Ac_val, L.Ac_diag, L.used_c, L.Dinv_c, flags);
}

}  // namespace

Amg *amg_create() { return new Amg(); }
void amg_destroy(Amg *a) {
  if (!a) return;
  a->release();
  delete a;
}
const char *amg_error(const Amg *a) { return a->err; }
int amg_levels(const Amg *a) { return (int)a->levels.size() + 1; }
int amg_level_blocks(const Amg *a, int level) {
  if (level == 0) return a->levels.empty() ? 0 : a->levels[0].nb;
  return a->levels[level - 1].nc;
}

// Builds the hierarchy for the matrix (rowptr, col) whose current values are d_val (device) with
// inverse diagonal blocks d_Dinv; also runs the first numeric set-up.  Returns 0, or -1 with
// amg_error() set.  A hierarchy with no coarse level (amg_levels() == 1) is a valid outcome.
int amg_build(Amg *a, hipStream_t s, int nb0, const std::vector<int> &rowptr, const std::vector<int> &col,
              const std::vector<int> &used0, const int *d_rowptr, const int *d_col, const double *d_val, const double *d_Dinv,
              double theta) {
  a->release();
  Pattern A;
  A.nb = nb0;
  A.rowptr = rowptr;
  A.col = col;
  std::vector<int> used = used0;
  const int *dA_rowptr = d_rowptr, *dA_col = d_col;
  const double *dA_val = d_val, *dDinv = d_Dinv;
  const size_t n0 = 6 * (size_t)std::max(nb0, 1);
  size_t tmp_cap = std::max(n0, col.size());
  if (!a->dalloc(&a->d_tmp, tmp_cap) || !a->dalloc(&a->d_tmp2, n0) || !a->dalloc(&a->d_norm, 2) || !a->dalloc(&a->d_flags, 1))
    return -1;
  (void)hipMemsetAsync(a->d_flags, 0, sizeof(unsigned), s);
  const int max_levels = std::min(std::max((int)tuning_value("amg_levels", (double)AMG_MAX_LEVELS), 1), (int)AMG_MAX_LEVELS);
  const bool verbose = std::getenv("BA_AMG_VERBOSE") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int l = 0; l < max_levels - 1 && A.nb >= 2 * AMG_MIN_BLOCKS; ++l) {
    const double t_start = now();
    const int nA = (int)A.col.size();
    std::vector<double> fro(nA);
    if ((size_t)nA > tmp_cap) {
      tmp_cap = (size_t)nA;
      if (!a->dalloc(&a->d_tmp, tmp_cap)) return -1;
    }
    hipLaunchKernelGGL(k_amg_block_norms, dim3(nblk(nA, 256)), dim3(256), 0, s, nA, dA_val, a->d_tmp);
    if (hipMemcpyAsync(fro.data(), a->d_tmp, nA * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      snprintf(a->err, sizeof a->err, "multilevel preconditioner: reading the block norms failed");
      return -1;
    }
    std::vector<int> agg;
    int nc = 0;
    const double t_norms = now();
    aggregate(A, fro, theta, agg, nc);
    const double t_agg = now();
    if (nc < AMG_MIN_BLOCKS || (l >= 2 && nc < AMG_DEEP_BLOCKS) || (int64_t)nc * 10 > (int64_t)A.nb * 9) break;  // no useful coarsening left
    Symbolic sy;
    symbolic(A, agg, nc, used, sy);
    const double t_sym = now();
    Level L;
    L.nb = A.nb; L.nc = nc; L.nA = nA; L.nP = (int)sy.P_col.size(); L.nAP = (int)sy.AP_col.size(); L.nAc = (int)sy.Ac.col.size();
    L.A_rowptr = dA_rowptr; L.A_col = dA_col; L.A_val = dA_val; L.Dinv = dDinv;
    bool ok = a->upload(&L.P_rowptr, sy.P_rowptr) && a->upload(&L.P_col, sy.P_col) && a->upload(&L.P_row, sy.P_row) &&
              a->upload(&L.P_own, sy.P_own) && a->upload(&L.Pa_ptr, sy.Pa_ptr) && a->upload(&L.Pa_idx, sy.Pa_idx) &&
              a->upload(&L.AP_ptr, sy.AP_ptr) && a->upload(&L.AP_la, sy.AP_la) && a->upload(&L.AP_lb, sy.AP_lb) &&
              a->upload(&L.Ac_ptr, sy.Ac_ptr) && a->upload(&L.Ac_la, sy.Ac_la) && a->upload(&L.Ac_lb, sy.Ac_lb) &&
              a->upload(&L.Pt_ptr, sy.Pt_ptr) && a->upload(&L.Pt_blk, sy.Pt_blk) && a->upload(&L.Pt_row, sy.Pt_row) &&
              a->upload(&L.Pt_ell, sy.Pt_ell) && a->upload(&L.P_col_ell, sy.P_col_ell) &&
              a->upload(&L.Ac_rowptr, sy.Ac.rowptr) && a->upload(&L.Ac_col, sy.Ac.col) && a->upload(&L.Ac_diag, sy.Ac_diag) &&
              a->upload(&L.used_c, sy.used_c) && a->upload(&L.P_pos, sy.P_pos) && a->dalloc(&L.P_val, 36 * (size_t)L.nP) &&
              a->dalloc(&L.Pc_val, 36 * (size_t)L.nP) &&
              a->dalloc(&L.AP_val, 36 * (size_t)L.nAP) && a->dalloc(&L.Ac_val, 36 * (size_t)L.nAc) &&
              a->dalloc(&L.Dinv_c, 36 * (size_t)nc) && a->dalloc(&L.rc, 6 * (size_t)nc) && a->dalloc(&L.y, 6 * (size_t)nc);
    if (!ok) return -1;
    const double t_up = now();
    // damping factor from the spectral radius of Dinv A (power iteration, fixed start vector)
    {
      const int n = 6 * A.nb, steps = 20;
      double *x0 = a->d_tmp, *x1 = a->d_tmp2;
      hipLaunchKernelGGL(k_amg_seed, dim3(nblk(n, 256)), dim3(256), 0, s, n, x0);
      for (int k = 0; k < steps; ++k) {
        if (k == steps - 1) hipLaunchKernelGGL(k_amg_norm2, dim3(1), dim3(1024), 0, s, n, x0, a->d_norm);
        hipLaunchKernelGGL(k_amg_power_step, dim3(nblk(A.nb, 4)), dim3(256), 0, s, A.nb, dA_rowptr, dA_col, dA_val, dDinv, x0, x1);
        std::swap(x0, x1);
      }
      hipLaunchKernelGGL(k_amg_norm2, dim3(1), dim3(1024), 0, s, n, x0, a->d_norm + 1);
      double nrm[2] = {0.0, 0.0};
      if (hipMemcpyAsync(nrm, a->d_norm, sizeof nrm, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        snprintf(a->err, sizeof a->err, "multilevel preconditioner: power iteration failed");
        return -1;
      }
      double rho = (nrm[0] > 0.0 && std::isfinite(nrm[1])) ? nrm[1] / nrm[0] : 0.0;
      if (!(rho > 1.0)) rho = 1.0;  // the diagonal blocks alone give 1
      L.omega = 4.0 / (3.0 * rho);
    }
    numeric_level(s, L, a->d_flags);
    if (verbose)
      fprintf(stderr, "amg: level %d (%d -> %d blocks): block norms %.1f ms, aggregation %.1f, lists %.1f, uploads %.1f, damping factor %.1f\n", l,
              A.nb, nc, t_norms - t_start, t_agg - t_norms, t_sym - t_agg, t_up - t_sym, now() - t_up);
    a->levels.push_back(L);
    A = sy.Ac;
    used = sy.used_c;
    dA_rowptr = L.Ac_rowptr; dA_col = L.Ac_col; dA_val = L.Ac_val; dDinv = L.Dinv_c;
  }
  a->n_part = 0;
  for (const Level &L : a->levels) a->n_part += nblk(L.nc, 4);
  if (!a->dalloc(&a->d_part, (size_t)a->n_part) || !a->dalloc(&a->d_totals, 1 + AMG_DEFL_NC)) return -1;
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) {
    snprintf(a->err, sizeof a->err, "multilevel preconditioner: set-up kernels failed");
    return -1;
  }
  return 0;
}

// numeric set-up for new matrix values (same pattern); level 0 pointers may have moved
void amg_numeric(Amg *a, hipStream_t s, const double *d_val, const double *d_Dinv) {
  if (a->levels.empty()) return;
  a->levels[0].A_val = d_val;
  a->levels[0].Dinv = d_Dinv;
  for (const Level &L : a->levels) numeric_level(s, L, a->d_flags);
}

static AmgApplyArgs apply_args(const Amg *a) {
  AmgApplyArgs aa{};
  aa.n_transfers = (int)a->levels.size();
  for (int l = 0; l < aa.n_transfers; ++l) {
    const Level &L = a->levels[l];
    AmgTransfer &t = aa.t[l];
    t.nb = L.nb; t.nc = L.nc; t.P_rowptr = L.P_rowptr; t.P_col = L.P_col; t.P_val = L.P_val; t.Pc_val = L.Pc_val; t.Pt_ptr = L.Pt_ptr;
    t.Pt_blk = L.Pt_blk; t.Pt_row = L.Pt_row; t.Dinv_c = L.Dinv_c; t.rc = L.rc; t.y = L.y;
    t.Pt_ell = L.Pt_ell; t.P_col_ell = L.P_col_ell;
  }
  aa.part = a->d_part;
  aa.n_part = a->n_part;
  aa.totals = a->d_totals;
  return aa;
}

// preconditioner part of one PCG iteration: after the vector kernel has left r, z = D0^-1 r and
// the partials of r.z, finish z, the scalars and p.  mode 0 right after the start-up kernel.
// first_done: the restriction into level 1 has already run (fused into the vector kernel)
// n_part_rz: how many partials of r.z (and of G^T r) the vector kernel before this has left
static void amg_finish_iteration(hipStream_t s, const BsrCgArgs &cg, const Amg *a, int parity, int mode, double rtol, int first_done,
                                 int n_part_rz) {
  const AmgApplyArgs aa = apply_args(a);
  int off = 0;
  for (int l = 0; l < aa.n_transfers; ++l) {
    const int nb = nblk(aa.t[l].nc, 4);
    if (first_done && l == 0) {
      off += nb;
      continue;
    }
    hipLaunchKernelGGL(k_amg_restrict, dim3(nb), dim3(256), 0, s, aa.t[l], l == 0 ? cg.r : aa.t[l - 1].rc, aa.part + off, cg.trace);
    off += nb;
  }
  const int np0 = aa.n_transfers > 0 ? nblk(aa.t[0].nc, 4) : 0;
  const double *totals = nullptr;
  for (int l = aa.n_transfers - 1; l >= 1; --l) {
    AmgFoldArgs fa{};
    if (l == 1) {  // the last launch before the final kernel: one more workgroup sums what that kernel needs
      fa.part_rz = cg.part_rz; fa.part = aa.part; fa.part_gr = cg.G ? cg.part_gr : nullptr;
      fa.n_part_rz = n_part_rz; fa.n_part = aa.n_part; fa.np0 = np0; fa.ncam = cg.ncam;
      fa.sigma = AMG_COARSE_WEIGHT; fa.tau = AMG_DEEP_WEIGHT; fa.totals = aa.totals;
      totals = aa.totals;
    }
    hipLaunchKernelGGL(k_amg_prolong, dim3(nblk(aa.t[l].nb, 32) + (l == 1 ? 1 : 0)), dim3(256), 0, s, aa.t[l], aa.t[l - 1].y,
                       AMG_DEEP_WEIGHT, fa, cg.trace);
  }
  hipLaunchKernelGGL(k_amg_update2, dim3(nblk(cg.S, 32) > 0 ? nblk(cg.S, 32) : 1), dim3(256), 0, s, cg, aa.t[0],
                     aa.n_transfers > 0 ? 1 : 0, aa.part, aa.n_part, n_part_rz, parity, mode, rtol,
                     AMG_COARSE_WEIGHT, AMG_DEEP_WEIGHT, np0, totals);
}

void launch_amg_cg_init(hipStream_t s, const BsrCgArgs &cg, const Amg *a, double rtol) {
  launch_cg2_start(s, cg);
  amg_finish_iteration(s, cg, a, 0, 0, rtol, 0, cg2_blocks_vec(cg.S));
}

void launch_amg_cg_iteration(hipStream_t s, const BsrCgArgs &cg, const Amg *a, int parity) {
  const AmgApplyArgs aa = apply_args(a);
  if (aa.n_transfers == 0 || !cg.r2) {
    launch_cg2_spmv_update1(s, cg, parity);
    amg_finish_iteration(s, cg, a, parity, 1, 0.0, 0, cg2_blocks_vec(cg.S));
    return;
  }
  launch_bsr_spmv(s, cg, parity);
  launch_amg_cg_iteration_after_product(s, cg, a, parity);
}

bool amg_cg_iteration_splits(const BsrCgArgs &cg, const Amg *a) { return apply_args(a).n_transfers > 0 && cg.r2 != nullptr; }

void launch_amg_cg_iteration_after_product(hipStream_t s, const BsrCgArgs &cg, const Amg *a, int parity) {
  const AmgApplyArgs aa = apply_args(a);
  const int nbv = cg2_blocks_vec8(cg.S), nbr = nblk(aa.t[0].nc, 4);
  hipLaunchKernelGGL(k_amg_update1_restrict, dim3(nbv + nbr), dim3(256), 0, s, cg, aa.t[0], parity, cg2_blocks_spmv(cg.S), nbv, aa.part);
  amg_finish_iteration(s, cg, a, parity, 1, 0.0, 1, nbv);
}

}  // namespace ba
