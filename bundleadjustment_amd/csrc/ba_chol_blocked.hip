// ba_chol_blocked.hip -- dense fp64 Cholesky, triangular solves and inverse for reduced camera systems of order 256 and
// up, on the matrix cores (v_mfma_f64_16x16x4_f64).
//
// The reference hands its normal equations to a sparse direct Cholesky inside Ceres (src/main.cpp:4706,
// SPARSE_NORMAL_CHOLESKY); here the points are eliminated first and the dense reduced system is factored.  The
// a-posteriori covariance (src/main.cpp:5261-5279) needs the inverse of the same matrix.
//
// Right-looking, panel width CB = 128, three launches per panel:
//   1. k_chol_diag : the 128 x 128 diagonal block in LDS (one workgroup): left-looking Cholesky, then its inverse W
//                    in place (kept: the triangular solves use it too);
//   2. k_gemm_nt<TRSM> : rows below the block, X = A_panel W^T -- the triangular solve as a product;
//   3. k_gemm_nt<SYRK> : trailing update C -= X X^T over the 128 x 128 tiles of the lower triangle.
// One product kernel serves all of it: a 256-thread workgroup per 128 x 128 tile of C = beta C + alpha A B^T (both
// operands row-major with the summation index contiguous), four waves of 64 x 64 each (4 x 4 MFMA tiles, 128
// accumulator registers), K staged through LDS in chunks of 16 columns with the next chunk's global loads in flight
// while the current one is multiplied.  A staged chunk is stored k-major with a row stride of 144 doubles, so that the
// four k rows a wave reads at once fall into alternating halves of the LDS banks.
// Matrices are row-major with an EVEN leading dimension ld >= n whose padding column holds zeros: every 16-byte load
// is then aligned and in bounds, and a padded column contributes nothing to a product.
#include <hip/hip_runtime.h>

#include "ba_kernels.h"

namespace ba {

#define CB 128
#define KC 16
#define LDT (CB + 16)
typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------------
// product kernel
// ------------------------------------------------------------------------------------------------------------------
// acc += A[rows ar0 .. ar0+127][kb .. ke) * B[rows br0 .. br0+127][kb .. ke)^T ; rows >= am / bm read as zero.
// kb is even; columns up to the next even number after ke are read (padding or zeros).
__device__ __forceinline__ void gemm_tile_nt(const double *__restrict__ A, int lda, int ar0, int am, const double *__restrict__ B,
                                             int ldb, int br0, int bm, int kb, int ke, double *sA, double *sB, d4 (&acc)[4][4]) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1, lr = lane & 15, lk = lane >> 4;
  const int row = t >> 1, h = t & 1;
  const bool aon = ar0 + row < am, bon = br0 + row < bm;
  const double *ap = A + (int64_t)(ar0 + row) * lda + 8 * h, *bp = B + (int64_t)(br0 + row) * ldb + 8 * h;
  double2 ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 8 * h + 2 * j;
      ra[j] = (aon && k < ke) ? *reinterpret_cast<const double2 *>(ap + k0 + 2 * j) : make_double2(0.0, 0.0);
      rb[j] = (bon && k < ke) ? *reinterpret_cast<const double2 *>(bp + k0 + 2 * j) : make_double2(0.0, 0.0);
    }
  };
  fetch(kb);
  for (int k0 = kb; k0 < ke; k0 += KC) {
    __syncthreads();  // the previous chunk has been read
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sA[(8 * h + 2 * j) * LDT + row] = ra[j].x;
      sA[(8 * h + 2 * j + 1) * LDT + row] = ra[j].y;
      sB[(8 * h + 2 * j) * LDT + row] = rb[j].x;
      sB[(8 * h + 2 * j + 1) * LDT + row] = rb[j].y;
    }
    __syncthreads();
    if (k0 + KC < ke) fetch(k0 + KC);  // in flight during the products below
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      double fa[4], fb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        fa[a] = sA[(4 * kk + lk) * LDT + 64 * wr + 16 * a + lr];
        fb[a] = sB[(4 * kk + lk) * LDT + 64 * wc + 16 * a + lr];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[c], acc[a][c], 0, 0, 0);
    }
  }
}

enum { GEMM_TRSM = 0, GEMM_SYRK = 1, GEMM_INV_T = 2, GEMM_INV_U = 3, GEMM_INV_S = 4 };

struct GemmArgs {
  double *A;        // the matrix being factored / its factor, n x ld
  const double *W;  // inverted diagonal blocks, CB x CB each (ld CB)
  double *U;        // inverse: U = L^-T, n x ld
  double *T;        // inverse: work row of tiles, CB x ld
  int n, ld, k0, kw;  // current panel: first column, width
  int ib;             // inverse: current block row
  int part;           // trailing update: 0 all tiles, 1 the first block column only, 2 all but the first block column
};

// tile (ti, tj), tj <= ti, of a lower triangle from a linear index
__device__ __forceinline__ void tri_tile(int b, int &ti, int &tj) {
  int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while ((int64_t)i * (i + 1) / 2 > b) --i;
  while ((int64_t)(i + 1) * (i + 2) / 2 <= b) ++i;
  ti = i;
  tj = b - (int)((int64_t)i * (i + 1) / 2);
}

template <int MODE>
__global__ void __launch_bounds__(256, 2)  // two workgroups per CU: one multiplies while the other waits for its loads
k_gemm_nt(GemmArgs g) {
  __shared__ double sA[KC * LDT], sB[KC * LDT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1, lr = lane & 15, lk = lane >> 4;
  d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
  const int n = g.n, ld = g.ld;
  double *C = nullptr;
  int ldc = ld, r0 = 0, c0 = 0, rmax = n, cmax = n;
  double alpha = 1.0, beta = 0.0;
  bool lower_only = false;
  if (MODE == GEMM_TRSM) {
    // X = A[r0 .., k0 .. k0+kw) W^T, in place (a workgroup reads its rows completely before it writes them)
    r0 = g.k0 + g.kw + CB * blockIdx.x;
    gemm_tile_nt(g.A + g.k0, ld, r0, n, g.W, CB, 0, g.kw, 0, g.kw, sA, sB, acc);
    C = g.A + g.k0;
    c0 = 0;
    cmax = g.kw;
  } else if (MODE == GEMM_SYRK) {
    int ti, tj;
    if (g.part == 1) {
      ti = blockIdx.x;
      tj = 0;
    } else {
      tri_tile(blockIdx.x, ti, tj);
      if (g.part == 2) {
        ++ti;
        ++tj;
      }
    }
    const int t0 = g.k0 + g.kw;
    r0 = t0 + CB * ti;
    c0 = t0 + CB * tj;
    gemm_tile_nt(g.A + g.k0, ld, r0, n, g.A + g.k0, ld, c0, n, 0, g.kw, sA, sB, acc);
    C = g.A;
    alpha = -1.0;
    beta = 1.0;
    lower_only = true;
  } else if (MODE == GEMM_INV_T) {
    // T^T[k-block rows][i-block cols] = sum_{j = k .. i-1} U[k, j] L[i, j]^T   (blocks of CB), one tile per k < ib
    const int kblk = blockIdx.x;
    r0 = CB * kblk;
    gemm_tile_nt(g.U, ld, r0, n, g.A, ld, CB * g.ib, n, CB * kblk, CB * g.ib, sA, sB, acc);
    // T (CB x ld) holds tile k as a CB x CB block at column CB k: rows = rows of block k, columns = rows of block ib
    C = g.T;
    r0 = 0;
    c0 = CB * kblk;
    rmax = CB;
    cmax = CB * kblk + min(CB, n - CB * g.ib);
  } else if (MODE == GEMM_INV_U) {
    // U[k, ib] = - T^T[k] W_ib^T
    const int kblk = blockIdx.x;
    gemm_tile_nt(g.T + CB * kblk, ld, 0, CB, g.W + (int64_t)g.ib * CB * CB, CB, 0, CB, 0, min(CB, n - CB * g.ib), sA, sB, acc);
    C = g.U;
    r0 = CB * kblk;
    c0 = CB * g.ib;
    alpha = -1.0;
  } else {
    // S^-1[a, b] = sum_{t >= a} U[a, t] U[b, t]^T over the lower tiles (a >= b); written to A (the factor is no longer needed)
    int ti, tj;
    tri_tile(blockIdx.x, ti, tj);
    r0 = CB * ti;
    c0 = CB * tj;
    gemm_tile_nt(g.U, ld, r0, n, g.U, ld, c0, n, CB * ti, n, sA, sB, acc);
    C = g.A;
  }
  // C layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = r0 + 64 * wr + 16 * a + lk + 4 * reg;
      if (r >= rmax) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cc = c0 + 64 * wc + 16 * c + lr;
        if (cc >= cmax || (lower_only && cc > r)) continue;
        double *dst = C + (int64_t)r * ldc + cc;
        const double v = alpha * acc[a][c][reg];
        *dst = (beta != 0.0) ? beta * *dst + v : v;
        if (MODE == GEMM_INV_S && cc != r) C[(int64_t)cc * ldc + r] = v;  // the inverse is returned as a full symmetric matrix
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// diagonal block: Cholesky and inverse in LDS
// ------------------------------------------------------------------------------------------------------------------
// One workgroup, the 128 x 128 block in LDS (rows beyond the matrix are identity rows).  Both sweeps advance EIGHT
// columns at a time.  Four threads own a PAIR of rows and a quarter of the summation range each: a thread sums a
// 2 x 8 tile of dot products against the finished part (16-byte LDS reads: 10 reads per 32 multiply-adds), the quarters
// are folded by lane swaps, and the 8 x 8 diagonal piece is factored / inverted redundantly in the registers of every
// thread.  Two barriers per eight columns instead of two per column.
#define TS (CB + 2)  // even: every row starts on a 16-byte boundary
#define SB 8         // columns per step

// Cholesky of the 8 x 8 lower triangle D (row-major) in registers; il = reciprocals of the diagonal of the factor;
// bad = index of the first pivot that is not positive
__device__ __forceinline__ void chol8(const double *D, double (&L)[SB][SB], double (&il)[SB], int &bad) {
  bad = -1;
#pragma unroll
  for (int j = 0; j < SB; ++j) {
    double d = D[j * SB + j];
#pragma unroll
    for (int k = 0; k < SB; ++k)
      if (k < j) d -= L[j][k] * L[j][k];
    if (!(d > 0.0) && bad < 0) bad = j;
    // 1 / sqrt(d) from the hardware estimate and two Newton steps (nine dependent operations; the IEEE square root
    // and division of the compiler are five times as long, and this chain of 128 pivots is the kernel's critical path)
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const double e = fma(-(d * y), y, 1.0);
      y = fma(0.5 * y, e, y);
    }
    il[j] = y;
    L[j][j] = d * y;
#pragma unroll
    for (int i = 0; i < SB; ++i)
      if (i > j) {
        double v = D[i * SB + j];
#pragma unroll
        for (int k = 0; k < SB; ++k)
          if (k < j) v -= L[i][k] * L[j][k];
        L[i][j] = v * il[j];
      }
#pragma unroll
    for (int i = 0; i < SB; ++i)
      if (i < j) L[i][j] = 0.0;
  }
}

__global__ void __launch_bounds__(256)
k_chol_diag(double *__restrict__ A, int n, int ld, int k0, double *__restrict__ Wk, int *__restrict__ info) {
  extern __shared__ double Tm[];   // CB x TS
  double *Db = Tm + CB * TS;       // SB x SB: the diagonal piece of a step
  double *Cb = Db + SB * SB;       // (CB - SB) x SB: the old column block of the inversion sweep
  double *invd = Cb + (CB - SB) * SB;  // CB: reciprocals of the diagonal of the factor
  const int kw = min(CB, n - k0);
  const int t = threadIdx.x;
  {
    // half a row per thread, eight 16-byte loads in flight at a time (a loop of single guarded loads would wait for
    // every one of them in turn)
    const int r = t >> 1, cb = 64 * (t & 1);
    const double *src = A + (int64_t)(k0 + r) * ld + k0 + cb;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double2 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = cb + 16 * b + 2 * i;
        v[i] = (r < kw && c <= r) ? *reinterpret_cast<const double2 *>(src + 16 * b + 2 * i) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = cb + 16 * b + 2 * i;
        Tm[r * TS + c] = (r < kw && c <= r) ? v[i].x : (r == c ? 1.0 : 0.0);
        Tm[r * TS + c + 1] = (r < kw && c + 1 <= r) ? v[i].y : (r == c + 1 ? 1.0 : 0.0);
      }
    }
  }
  __syncthreads();
  const int rp = t >> 2, q = t & 3;   // four threads per pair of rows
  const int ra = 2 * rp, rb = ra + 1;
  // ---- factorisation, left-looking, SB columns per step
  for (int j0 = 0; j0 < CB; j0 += SB) {
    double acc[2][SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) acc[0][c] = acc[1][c] = 0.0;
    if (ra >= j0) {
      const int kq = j0 / 4, kb = q * kq, ke = kb + kq;  // j0 is a multiple of 8: even quarters
      const double *pa = Tm + ra * TS, *pb = Tm + rb * TS, *pj = Tm + j0 * TS;
      for (int k = kb; k < ke; k += 2) {
        const double2 a0 = *reinterpret_cast<const double2 *>(pa + k), a1 = *reinterpret_cast<const double2 *>(pb + k);
#pragma unroll
        for (int c = 0; c < SB; ++c) {
          const double2 b = *reinterpret_cast<const double2 *>(pj + c * TS + k);
          acc[0][c] += a0.x * b.x + a0.y * b.y;
          acc[1][c] += a1.x * b.x + a1.y * b.y;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < SB; ++c) {
        acc[i][c] += __shfl_xor(acc[i][c], 1, 64);
        acc[i][c] += __shfl_xor(acc[i][c], 2, 64);
      }
    double v[2][SB];
    if (q == 0 && ra >= j0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < SB; ++c) v[i][c] = Tm[(ra + i) * TS + j0 + c] - acc[i][c];
      if (ra < j0 + SB)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < SB; ++c) Db[(ra + i - j0) * SB + c] = v[i][c];
    }
    __syncthreads();
    double Lb[SB][SB], il[SB];
    int bad;
    chol8(Db, Lb, il, bad);
    if (t == 0) {
      if (bad >= 0 && j0 + bad < kw) atomicMax(info, k0 + j0 + bad + 1);
#pragma unroll
      for (int c = 0; c < SB; ++c) invd[j0 + c] = il[c];
    }
    if (q == 0 && ra >= j0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = ra + i;
        double x[SB];
        if (row < j0 + SB) {
#pragma unroll
          for (int c = 0; c < SB; ++c) {
            x[c] = 0.0;
#pragma unroll
            for (int m = 0; m < SB; ++m)
              if (m == row - j0) x[c] = Lb[m][c];
          }
        } else {
          // x Lb^T = v
#pragma unroll
          for (int c = 0; c < SB; ++c) {
            double sdot = v[i][c];
#pragma unroll
            for (int k = 0; k < SB; ++k)
              if (k < c) sdot -= x[k] * Lb[c][k];
            x[c] = sdot * il[c];
          }
        }
#pragma unroll
        for (int c = 0; c < SB; ++c) Tm[row * TS + j0 + c] = x[c];
      }
    }
    __syncthreads();
  }
  for (int e = t; e < CB * CB; e += 256) {
    const int rr = e / CB, c = e - rr * CB;
    if (rr < kw && c <= rr) A[(int64_t)(k0 + rr) * ld + k0 + c] = Tm[rr * TS + c];
  }
  // ---- inverse of the lower triangle in place, column blocks from the right:
  //      [L11 0; L21 L22]^-1 = [W11 0; -W22 L21 W11, W22],  W22 already in place
  for (int j0 = CB - SB; j0 >= 0; j0 -= SB) {
    // W11 = L11^-1 in registers (every thread), the old L21 to Cb
    double W11[SB][SB];
    {
      double L11[SB][SB], id[SB];
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        id[i] = invd[j0 + i];
#pragma unroll
        for (int c = 0; c < SB; ++c) L11[i][c] = (c < i) ? Tm[(j0 + i) * TS + j0 + c] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < SB; ++c)
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          if (i < c) {
            W11[i][c] = 0.0;
          } else {
            double sdot = (i == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < SB; ++k)
              if (k >= c && k < i) sdot -= L11[i][k] * W11[k][c];
            W11[i][c] = sdot * id[i];
          }
        }
    }
    const int m0 = j0 + SB;  // first row of the 22 part
    for (int e = t; e < (CB - m0) * SB; e += 256) Cb[e] = Tm[(m0 + e / SB) * TS + j0 + (e % SB)];
    __syncthreads();
    double acc[2][SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) acc[0][c] = acc[1][c] = 0.0;
    if (ra >= m0) {
      // k = m0 .. rb for both rows of the pair (the entry beyond the diagonal of the upper row is a stored zero)
      const int len = rb - m0 + 1, per = ((len / 2 + 3) / 4) * 2;
      const int kb = m0 + q * per, ke = min(kb + per, rb + 1);
      const double *pa = Tm + ra * TS, *pb = Tm + rb * TS;
      for (int k = kb; k < ke; k += 2) {
        const double2 a0 = *reinterpret_cast<const double2 *>(pa + k), a1 = *reinterpret_cast<const double2 *>(pb + k);
        const double2 *c0 = reinterpret_cast<const double2 *>(Cb + (k - m0) * SB), *c1 = c0 + SB / 2;
#pragma unroll
        for (int c = 0; c < SB / 2; ++c) {
          const double2 x0 = c0[c], x1 = c1[c];
          acc[0][2 * c] += a0.x * x0.x + a0.y * x1.x;
          acc[0][2 * c + 1] += a0.x * x0.y + a0.y * x1.y;
          acc[1][2 * c] += a1.x * x0.x + a1.y * x1.x;
          acc[1][2 * c + 1] += a1.x * x0.y + a1.y * x1.y;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < SB; ++c) {
        acc[i][c] += __shfl_xor(acc[i][c], 1, 64);
        acc[i][c] += __shfl_xor(acc[i][c], 2, 64);
      }
    if (q == 0 && ra >= j0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = ra + i;
#pragma unroll
        for (int c = 0; c < SB; ++c) {
          double x = 0.0;
          if (row >= m0) {
#pragma unroll
            for (int m = 0; m < SB; ++m)
              if (m >= c) x -= acc[i][m] * W11[m][c];
          } else {
#pragma unroll
            for (int m = 0; m < SB; ++m)
              if (m == row - j0) x = W11[m][c];
          }
          Tm[row * TS + j0 + c] = x;
        }
      }
    }
    __syncthreads();
  }
  for (int e = t; e < CB * CB; e += 256) {
    const int rr = e / CB, c = e - rr * CB;
    Wk[e] = (rr < kw && c <= rr) ? Tm[rr * TS + c] : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// triangular solves with the factor and the inverted diagonal blocks, one launch per panel
// ------------------------------------------------------------------------------------------------------------------
// forward step k: workgroup i (block row k + 1 + i) subtracts L[i, k] x_k from its part of b; the workgroup of block row
// k + 1 then finishes x_{k+1} = W_{k+1} b_{k+1}.  first != 0: only x_0 = W_0 b_0.
__global__ void __launch_bounds__(CB)
k_chol_forward(const double *__restrict__ L, int n, int ld, const double *__restrict__ W, double *__restrict__ b, int k, int first) {
  __shared__ double xs[CB];
  const int t = threadIdx.x;
  const int ib = first ? 0 : k + 1 + blockIdx.x;
  const int row = CB * ib + t;
  double v = row < n ? b[row] : 0.0;
  if (!first) {
    const int k0 = CB * k;
    xs[t] = b[k0 + t];  // block k is complete (k0 + t < n as block k is not the last one)
    __syncthreads();
    if (row < n) {
      const double2 *lr = reinterpret_cast<const double2 *>(L + (int64_t)row * ld + k0);
      double s = 0.0;
#pragma unroll 8
      for (int c = 0; c < CB / 2; ++c) {
        const double2 l = lr[c];
        s += l.x * xs[2 * c] + l.y * xs[2 * c + 1];
      }
      v -= s;
    }
    if (blockIdx.x != 0) {
      if (row < n) b[row] = v;
      return;
    }
    __syncthreads();
  }
  // x = W_ib v (lower triangular, CB x CB)
  xs[t] = v;
  __syncthreads();
  const double *wr = W + (int64_t)ib * CB * CB + (int64_t)t * CB;
  double s = 0.0;
  for (int c = 0; c <= t; ++c) s += wr[c] * xs[c];
  if (row < n) b[row] = s;
}

// backward step i (from the last block row up): workgroup k < i subtracts L[i, k]^T x_i from its part of b; the workgroup
// of block i - 1 then finishes x_{i-1} = W_{i-1}^T b_{i-1}.  first != 0: only x of the last block.
__global__ void __launch_bounds__(CB)
k_chol_backward(const double *__restrict__ L, int n, int ld, const double *__restrict__ W, double *__restrict__ b, int i, int first) {
  __shared__ double xs[CB];
  const int t = threadIdx.x;
  const int kb = first ? i : (int)blockIdx.x;
  const int col = CB * kb + t;
  double v = col < n ? b[col] : 0.0;
  if (!first) {
    const int i0 = CB * i, iw = min(CB, n - i0);
    xs[t] = t < iw ? b[i0 + t] : 0.0;
    __syncthreads();
    double s = 0.0;
    for (int r = 0; r < iw; ++r) s += L[(int64_t)(i0 + r) * ld + col] * xs[r];  // consecutive threads, consecutive columns
    v -= s;
    if (kb != i - 1) {
      b[col] = v;
      return;
    }
    __syncthreads();
  }
  xs[t] = v;
  __syncthreads();
  const double *wk = W + (int64_t)kb * CB * CB;
  double s = 0.0;
  for (int r = t; r < CB; ++r) s += wk[(int64_t)r * CB + t] * xs[r];  // W^T
  if (col < n) b[col] = s;
}

// U[k-block diagonal] = W_k^T (upper triangular), the rest of U zero
__global__ void __launch_bounds__(256)
k_inv_init_U(int n, int ld, const double *__restrict__ W, double *__restrict__ U) {
  const int kb = blockIdx.x;
  const double *wk = W + (int64_t)kb * CB * CB;
  for (int q = threadIdx.x; q < CB * CB; q += 256) {
    const int r = q / CB, c = q - r * CB;
    if (CB * kb + r < n && CB * kb + c < n) U[(int64_t)(CB * kb + r) * ld + CB * kb + c] = wk[(int64_t)c * CB + r];
  }
}

// ------------------------------------------------------------------------------------------------------------------
int chol_blocked_min_order() { return 2 * CB; }
int chol_blocked_ld(int n) { return (n + 15) & ~15; }
size_t chol_blocked_workspace(int n) { return (size_t)((n + CB - 1) / CB) * CB * CB; }

void launch_cholesky_blocked(hipStream_t s, double *A, int n, int ld, double *W, int *info) {
  (void)hipMemsetAsync(info, 0, sizeof(int), s);
  const size_t lds = (size_t)(CB * TS + SB * SB + (CB - SB) * SB + CB) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_diag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  // Look-ahead: the diagonal block of panel k + 1 (one workgroup, a chain of 128 pivots) only needs the first block
  // column of the trailing update of panel k, so the rest of that update runs on a side stream beside it.
  static thread_local hipStream_t side = nullptr;
  static thread_local hipEvent_t ev_panel = nullptr, ev_rest = nullptr;
  const int nb = (n + CB - 1) / CB;
  bool ahead = nb >= 6;
  if (ahead && !side) {
    if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_panel, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_rest, hipEventDisableTiming) != hipSuccess) {
      side = nullptr;
      (void)hipGetLastError();
    }
  }
  if (!side) ahead = false;
  GemmArgs g{};
  g.A = A; g.n = n; g.ld = ld;
  bool rest_pending = false;
  for (int k0 = 0, kb = 0; k0 < n; k0 += CB, ++kb) {
    const int kw = (n - k0 < CB) ? n - k0 : CB;
    double *Wk = W + (size_t)kb * CB * CB;
    hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), lds, s, A, n, ld, k0, Wk, info);
    const int rem = n - k0 - kw;
    if (rem <= 0) break;
    g.W = Wk; g.k0 = k0; g.kw = kw;
    const int T = (rem + CB - 1) / CB;
    g.part = 0;
    hipLaunchKernelGGL(k_gemm_nt<GEMM_TRSM>, dim3(T), dim3(256), 0, s, g);
    if (!ahead || T < 2) {
      if (rest_pending) {
        (void)hipStreamWaitEvent(s, ev_rest, 0);
        rest_pending = false;
      }
      hipLaunchKernelGGL(k_gemm_nt<GEMM_SYRK>, dim3(T * (T + 1) / 2), dim3(256), 0, s, g);
      continue;
    }
    (void)hipEventRecord(ev_panel, s);  // the panel is final: the side stream may read it
    if (rest_pending) (void)hipStreamWaitEvent(s, ev_rest, 0);  // the first block column also takes the previous panel's update
    g.part = 1;
    hipLaunchKernelGGL(k_gemm_nt<GEMM_SYRK>, dim3(T), dim3(256), 0, s, g);
    (void)hipStreamWaitEvent(side, ev_panel, 0);
    g.part = 2;
    hipLaunchKernelGGL(k_gemm_nt<GEMM_SYRK>, dim3((T - 1) * T / 2), dim3(256), 0, side, g);
    (void)hipEventRecord(ev_rest, side);
    rest_pending = true;
  }
  if (rest_pending) (void)hipStreamWaitEvent(s, ev_rest, 0);
}

void launch_chol_solve_blocked(hipStream_t s, const double *L, int n, int ld, const double *W, double *b) {
  const int nb = (n + CB - 1) / CB;
  hipLaunchKernelGGL(k_chol_forward, dim3(1), dim3(CB), 0, s, L, n, ld, W, b, 0, 1);
  for (int k = 0; k + 1 < nb; ++k) hipLaunchKernelGGL(k_chol_forward, dim3(nb - 1 - k), dim3(CB), 0, s, L, n, ld, W, b, k, 0);
  hipLaunchKernelGGL(k_chol_backward, dim3(1), dim3(CB), 0, s, L, n, ld, W, b, nb - 1, 1);
  for (int i = nb - 1; i >= 1; --i) hipLaunchKernelGGL(k_chol_backward, dim3(i), dim3(CB), 0, s, L, n, ld, W, b, i, 0);
}

// A: the factor on entry, (L L^T)^-1 as a full symmetric matrix on return.  U: n x ld work matrix, T: CB x ld work.
void launch_chol_inverse_blocked(hipStream_t s, double *A, int n, int ld, const double *W, double *U, double *T) {
  const int nb = (n + CB - 1) / CB;
  (void)hipMemsetAsync(U, 0, (size_t)n * ld * sizeof(double), s);
  (void)hipMemsetAsync(T, 0, (size_t)CB * ld * sizeof(double), s);
  hipLaunchKernelGGL(k_inv_init_U, dim3(nb), dim3(256), 0, s, n, ld, W, U);
  GemmArgs g{};
  g.A = A; g.W = W; g.U = U; g.T = T; g.n = n; g.ld = ld;
  for (int ib = 1; ib < nb; ++ib) {
    g.ib = ib;
    hipLaunchKernelGGL(k_gemm_nt<GEMM_INV_T>, dim3(ib), dim3(256), 0, s, g);
    hipLaunchKernelGGL(k_gemm_nt<GEMM_INV_U>, dim3(ib), dim3(256), 0, s, g);
  }
  hipLaunchKernelGGL(k_gemm_nt<GEMM_INV_S>, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, g);
}

}  // namespace ba
