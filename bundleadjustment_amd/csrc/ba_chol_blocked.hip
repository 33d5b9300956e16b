// ba_chol_blocked.hip -- dense fp64 Cholesky, triangular solves and inverse for reduced camera systems of order 256 and
// up, on the matrix cores (v_mfma_f64_16x16x4_f64).
//
// The reference hands its normal equations to a sparse direct Cholesky inside Ceres (src/main.cpp:4706,
// SPARSE_NORMAL_CHOLESKY); here the points are eliminated first and the dense reduced system is factored.  The
// a-posteriori covariance (src/main.cpp:5261-5279) needs the inverse of the same matrix.
//
// Right-looking, panel width CB = 128, three launches per panel:
//   1. k_chol_diag : the 128 x 128 diagonal block in LDS (one workgroup): Cholesky over block columns of 16 with the
//                    sums on the matrix cores, then its inverse W in place (kept: the triangular solves use it too);
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
  constexpr int HK = KC / 2, NJ = KC / 4;  // a thread stages half a chunk row: HK columns as NJ 16-byte pieces
  const double *ap = A + (int64_t)(ar0 + row) * lda + HK * h, *bp = B + (int64_t)(br0 + row) * ldb + HK * h;
  double2 ra[NJ], rb[NJ];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = k0 + HK * h + 2 * j;
      ra[j] = (aon && k < ke) ? *reinterpret_cast<const double2 *>(ap + k0 + 2 * j) : make_double2(0.0, 0.0);
      rb[j] = (bon && k < ke) ? *reinterpret_cast<const double2 *>(bp + k0 + 2 * j) : make_double2(0.0, 0.0);
    }
  };
  fetch(kb);
  for (int k0 = kb; k0 < ke; k0 += KC) {
    __syncthreads();  // the previous chunk has been read
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      sA[(HK * h + 2 * j) * LDT + row] = ra[j].x;
      sA[(HK * h + 2 * j + 1) * LDT + row] = ra[j].y;
      sB[(HK * h + 2 * j) * LDT + row] = rb[j].x;
      sB[(HK * h + 2 * j + 1) * LDT + row] = rb[j].y;
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
__device__ __forceinline__ void gemm_nt_block(const GemmArgs &g, const int bx, double *sA, double *sB) {
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
    r0 = g.k0 + g.kw + CB * bx;
    gemm_tile_nt(g.A + g.k0, ld, r0, n, g.W, CB, 0, g.kw, 0, g.kw, sA, sB, acc);
    C = g.A + g.k0;
    c0 = 0;
    cmax = g.kw;
  } else if (MODE == GEMM_SYRK) {
    int ti, tj;
    tri_tile(bx, ti, tj);
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
    const int kblk = bx;
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
    const int kblk = bx;
    gemm_tile_nt(g.T + CB * kblk, ld, 0, CB, g.W + (int64_t)g.ib * CB * CB, CB, 0, CB, 0, min(CB, n - CB * g.ib), sA, sB, acc);
    C = g.U;
    r0 = CB * kblk;
    c0 = CB * g.ib;
    alpha = -1.0;
  } else {
    // S^-1[a, b] = sum_{t >= a} U[a, t] U[b, t]^T over the lower tiles (a >= b); written to A (the factor is no longer needed)
    int ti, tj;
    tri_tile(bx, ti, tj);
    r0 = CB * ti;
    c0 = CB * tj;
    gemm_tile_nt(g.U, ld, r0, n, g.U, ld, c0, n, CB * ti, n, sA, sB, acc);
    C = g.A;
  }
  // C layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg.
  // Sixteen entries at a time: their old values are all requested before the first is needed (one wait per sixteen
  // instead of one per entry -- a tile of the trailing update sits on the critical path of the next panel).  Addresses
  // are clamped into the matrix so that the loads need no branch; the stores are predicated.
  constexpr bool accumulate = MODE == GEMM_SYRK;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double *dst[4][4];
    double old[4][4];
    bool ok[4][4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = r0 + 64 * wr + 16 * a + lk + 4 * reg, cc = c0 + 64 * wc + 16 * c + lr;
        ok[reg][c] = r < rmax && cc < cmax && !(lower_only && cc > r);
        dst[reg][c] = C + (int64_t)min(r, rmax - 1) * ldc + min(cc, cmax - 1);
        old[reg][c] = accumulate ? *dst[reg][c] : 0.0;
      }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!ok[reg][c]) continue;
        const double v = alpha * acc[a][c][reg];
        *dst[reg][c] = accumulate ? beta * old[reg][c] + v : v;
        if (MODE == GEMM_INV_S) {  // the inverse is returned as a full symmetric matrix
          const int r = r0 + 64 * wr + 16 * a + lk + 4 * reg, cc = c0 + 64 * wc + 16 * c + lr;
          if (cc != r) C[(int64_t)cc * ldc + r] = v;
        }
      }
  }
}

template <int MODE>
__global__ void __launch_bounds__(256, 2)  // two workgroups per CU: one multiplies while the other waits for its loads
k_gemm_nt(GemmArgs g) {
  __shared__ double sA[KC * LDT], sB[KC * LDT];
  gemm_nt_block<MODE>(g, blockIdx.x, sA, sB);
}

// ------------------------------------------------------------------------------------------------------------------
// diagonal block: Cholesky and inverse in LDS
// ------------------------------------------------------------------------------------------------------------------
// One workgroup of four waves, the 128 x 128 block row-major in LDS (rows beyond the matrix are identity rows).
//   Factorisation over block columns of 16 with one wave on the critical path.  Wave 0 holds the 16 x 16 diagonal tile
//   in the accumulator layout of the matrix instruction (four entries per lane) and eliminates its 16 pivots by rank-one
//   products on the matrix cores: per pivot one lane read (the pivot), the reciprocal square root, and two products --
//   one takes the column out of the tile, the other applies the same elimination to a unit matrix, which leaves the
//   inverse W_JJ of the tile's factor when the last pivot is done.  With W_JJ the rows below need no substitution:
//   X = P W_JJ^T is four matrix instructions per tile.  Wave 0 does that for the next block row only, subtracts X X^T
//   from the next diagonal tile and goes on with its pivots; waves 1..3 own the block rows (row I in wave 1 + I % 3),
//   multiply their tiles of the finished block column by W and bring their tiles of the next block column up to date
//   (the sums over all finished columns, on the matrix cores) while wave 0 eliminates.  Two barriers per block column.
//   Inverse: the tile inverses are there already; merged pairwise, 16 -> 32 -> 64 -> 128,
//   [L11 0; L21 L22]^-1 = [W11 0; -W22 (L21 W11), W22], two products per level over 16 x 16 tiles, in place.
#define TS (CB + 2)  // even (16-byte aligned rows) and = 2 mod 32 (conflict-free operand reads)

// acc += X[rows of xa][k] Y[rows of xb][k] over k in [kb, ke), both operands row-major in LDS (pointers at row lr, column
// lk); ke - kb is a multiple of 16: the twelve operand reads of sixteen columns are issued before their products
__device__ __forceinline__ void mfma_nt16(const double *xa, const double *xb, int kb, int ke, d4 &acc) {
  for (int k = kb; k < ke; k += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = xa[k + 4 * u];
      b[u] = xb[k + 4 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
}
// the same for X[rows][k] B[k][columns]: pb at row lk, column lr of B
__device__ __forceinline__ void mfma_nn16(const double *xa, const double *pb, int kb, int ke, d4 &acc) {
  for (int k = kb; k < ke; k += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = xa[k + 4 * u];
      b[u] = pb[(k + 4 * u) * TS];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
}

// the value lane `src` holds, in every lane (src is a constant: two lane reads into scalar registers)
__device__ __forceinline__ double lane_value(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// workgroup barrier for data that lives in LDS: stores to memory that are still on their way do not hold it up
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// what one wave writes to LDS, other lanes of the same wave read next
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void chol_diag_block(double *__restrict__ A, int n, int ld, int k0, double *__restrict__ Wk, int *__restrict__ info,
                                                double *Tm) {  // Tm: CB x TS doubles of LDS
  const int kw = min(CB, n - k0);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane & 15, lk = lane >> 4;
  const int NT = CB / 16;
  {
    // half a row per thread; the loads of the lower triangle all in flight before the first goes to LDS
    const int r = t >> 1, cb = 64 * (t & 1);
    const double *src = A + (int64_t)(k0 + r) * ld + k0 + cb;
    double2 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int c = cb + 2 * i;
      v[i] = (r < kw && c <= r) ? *reinterpret_cast<const double2 *>(src + 2 * i) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int c = cb + 2 * i;
      *reinterpret_cast<double2 *>(Tm + r * TS + c) = make_double2((r < kw && c <= r) ? v[i].x : (r == c ? 1.0 : 0.0),
                                                                   (r < kw && c + 1 <= r) ? v[i].y : (r == c + 1 ? 1.0 : 0.0));
    }
  }
  __syncthreads();
  d4 D = (d4){0.0, 0.0, 0.0, 0.0};  // wave 0: the diagonal tile, D[reg] = tile[lk + 4 reg][lr] (both triangles)
  if (w == 0) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = lk + 4 * reg;
      D[reg] = Tm[max(i, lr) * TS + min(i, lr)];
    }
  }
  int bad = -1;
  for (int J = 0; J < NT; ++J) {
    const int c0 = 16 * J, c1 = c0 + 16;
    if (w == 0) {
      // ---- the 16 pivots of tile (J, J); Wt becomes the inverse of its factor
      d4 Wt;  // rows i > j of the running inverse; a row is final when its pivot is done and is kept in wrow_done
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Wt[reg] = (lk + 4 * reg == lr) ? 1.0 : 0.0;
      double lcol[4] = {0.0, 0.0, 0.0, 0.0};       // lcol[q] = L[lr][lk + 4 q]
      double wrow_done[4] = {0.0, 0.0, 0.0, 0.0};  // wrow_done[q] = W[lk + 4 q][lr]
      // The pivot of column j + 1 is taken from the tile as it was before column j left it, minus the square of that
      // column's entry -- on the vector unit, so that the chain from pivot to pivot does not wait for the product.  The
      // matrix unit takes one product at a time: the one for the inverse is issued in the middle of the next pivot's
      // Newton steps, when the one for the tile has left the unit, so that the wave never waits to issue.
      // Only the first operand of a product is masked (lanes 16 jk + j .. 16 jk + 15: entries j .. 15 of column j): where it
      // is zero the second operand does not matter, and what the products then do to columns < j of the tile and to row j
      // of the running inverse is never read again.
      double d = lane_value(D[0], 0);
      double nl = 0.0, wrow = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jq = j >> 2, jk = j & 3;
        if (!(d > 0.0) && bad < 0) bad = c0 + j;
        // 1 / sqrt(d): hardware estimate and two Newton steps y <- y + y (1/2 - (d/2) y^2) (the IEEE square root and
        // division are five times as long, and the chain of pivots is the critical path of the whole factorisation)
        const double hd = 0.5 * d;
        double y = __builtin_amdgcn_rsq(d);
        y = fma(y, fma(-(hd * y), y, 0.5), y);
        __builtin_amdgcn_sched_barrier(0);
        if (j > 0) Wt = __builtin_amdgcn_mfma_f64_16x16x4f64(nl, wrow, Wt, 0, 0, 0);  // rows i >= j - 1 of W -= l_i (row j - 1 of W)
        __builtin_amdgcn_sched_barrier(0);
        y = fma(y, fma(-(hd * y), y, 0.5), y);
        const double dj = D[jq];
        // these 16 lanes hold row j of the tile (= column j: it is symmetric) in register jq
        const bool col = (unsigned)(lane - (16 * jk + j)) < (unsigned)(16 - j);
        const bool row = (lane >> 4) == jk;
        const double t = dj * y;  // in those lanes: column j of the factor (entry j: d y = sqrt(d)), rubbish above the diagonal
        nl = col ? -t : 0.0;
        if (j < 15) {
          const int jn = j + 1;
          const double ln = lane_value(dj, 16 * jk + jn) * y;  // entry j + 1 of the column: the lane read does not wait for y
          d = fma(-ln, ln, lane_value(D[jn >> 2], 16 * (jn & 3) + jn));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j < 15) D = __builtin_amdgcn_mfma_f64_16x16x4f64(nl, t, D, 0, 0, 0);  // tile -= l l^T
        __builtin_amdgcn_sched_barrier(0);
        wrow = Wt[jq] * y;  // in those lanes: row j of the inverse, final
        lcol[jq] = row ? t : lcol[jq];
        wrow_done[jq] = row ? wrow : wrow_done[jq];
      }
      // the inverse takes the tile's place in LDS; the factor's tile waits in a tile above the diagonal, which nothing reads
      {
        double *slot = Tm + (J < NT - 1 ? 0 : 16) * TS + (J < NT - 1 ? c1 : CB - 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) slot[lr * TS + lk + 4 * q] = lcol[q];
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Tm[(c0 + lk + 4 * reg) * TS + c0 + lr] = wrow_done[reg];
    } else if (J > 0) {
      const int first = J + 1 + ((w - 1) - (J + 1) % 3 + 3) % 3;  // this wave's first block row below J
      // ---- block column J - 1 of the rows this wave owns: X = P W^T in place
      const int cp = c0 - 16;
      for (int I = first; I < NT; I += 3) {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        mfma_nt16(Tm + (16 * I + lr) * TS + cp + lk, Tm + (cp + lr) * TS + cp + lk, 0, 16, acc);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Tm[(16 * I + lk + 4 * reg) * TS + cp + lr] = acc[reg];
      }
      wave_lds_sync();
      // ---- their tiles of block column J, and the diagonal tile of row J + 1: minus the sums over the finished columns
      for (int I = first; I < NT; I += 3) {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        mfma_nt16(Tm + (16 * I + lr) * TS + lk, Tm + (c0 + lr) * TS + lk, 0, c0, acc);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Tm[(16 * I + lk + 4 * reg) * TS + c0 + lr] -= acc[reg];
      }
      if (first == J + 1 && first < NT) {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        mfma_nt16(Tm + (c1 + lr) * TS + lk, Tm + (c1 + lr) * TS + lk, 0, c0, acc);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Tm[(c1 + lk + 4 * reg) * TS + c1 + lr] -= acc[reg];
      }
    }
    lds_barrier();
    if (w == 0 && J + 1 < NT) {
      // ---- block row J + 1 against the new tile, then its diagonal tile for the next round of pivots
      d4 X = (d4){0.0, 0.0, 0.0, 0.0};
      mfma_nt16(Tm + (c1 + lr) * TS + c0 + lk, Tm + (c0 + lr) * TS + c0 + lk, 0, 16, X);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Tm[(c1 + lk + 4 * reg) * TS + c0 + lr] = X[reg];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = lk + 4 * reg;
        D[reg] = Tm[(c1 + max(i, lr)) * TS + c1 + min(i, lr)];
      }
      wave_lds_sync();
      {
        const double *xr = Tm + (c1 + lr) * TS + c0 + lk;
        double xa[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = xr[4 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) D = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[u], xa[u], D, 0, 0, 0);
      }
    }
    lds_barrier();
  }
  if (t == 0 && bad >= 0 && bad < kw) atomicMax(info, k0 + bad + 1);
  // the factor: below the diagonal tiles from where it stands, the diagonal tiles from where wave 0 left them
#pragma unroll 8
  for (int e = t; e < CB * CB / 2; e += 256) {
    const int rr = e / (CB / 2), c = 2 * (e - rr * (CB / 2));
    if (rr < kw && c < (rr & ~15))
      *reinterpret_cast<double2 *>(A + (int64_t)(k0 + rr) * ld + k0 + c) = *reinterpret_cast<const double2 *>(Tm + rr * TS + c);
  }
#pragma unroll
  for (int e = t; e < CB * 8; e += 256) {
    const int rr = e >> 3, c = 2 * (e & 7), tile = rr >> 4;  // row of the block, column inside its diagonal tile
    if (rr < kw && c <= (rr & 15)) {
      const double2 v = *reinterpret_cast<const double2 *>(Tm + ((tile < NT - 1 ? 0 : 16) + (rr & 15)) * TS + (tile < NT - 1 ? 16 * tile + 16 : CB - 16) + c);
      double *dst = A + (int64_t)(k0 + rr) * ld + k0 + 16 * tile + c;
      if (c + 1 <= (rr & 15)) *reinterpret_cast<double2 *>(dst) = v;
      else *dst = v.x;
    }
  }
  // ---- inverse in place: the diagonal tiles hold their inverses already.  A level has 2 s x 2 s blocks of ts x ts tiles
  //      each (ts = s / 16); a wave takes ts tiles of one block: for M = L21 W11 a tile row of L21 against all tile columns
  //      of W11, for W21 = -W22 M all tile rows of W22 against a tile column of M -- so both products overwrite their own
  //      operand only, and the tiles of a wave run as independent chains on the matrix unit.
  for (int s = 16; s < CB; s *= 2) {
    const int ts = s / 16, o = 2 * s * (w / ts), r = w % ts;
    d4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (d4){0.0, 0.0, 0.0, 0.0};
    lds_barrier();
    // M = L21 W11 (W11 lower triangular: tile column m from row 16 m on)
    for (int k = 0; k < s; k += 16) {
      double xa[4], xb[4][4];
      const double *pa = Tm + (o + s + 16 * r + lr) * TS + o + k + lk;
#pragma unroll
      for (int u = 0; u < 4; ++u) xa[u] = pa[4 * u];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m < ts && 16 * m <= k) {
          const double *pb = Tm + (o + k + lk) * TS + o + 16 * m + lr;
#pragma unroll
          for (int u = 0; u < 4; ++u) xb[m][u] = pb[4 * u * TS];
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < ts && 16 * m <= k) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u], xb[m][u], acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < ts) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Tm[(o + s + 16 * r + lk + 4 * reg) * TS + o + 16 * m + lr] = acc[m][reg];
        acc[m] = (d4){0.0, 0.0, 0.0, 0.0};
      }
    lds_barrier();
    // W21 = -W22 M (W22 lower triangular: tile row m up to column 16 (m + 1))
    for (int k = 0; k < s; k += 16) {
      double xa[4][4], xb[4];
      const double *pb = Tm + (o + s + k + lk) * TS + o + 16 * r + lr;
#pragma unroll
      for (int u = 0; u < 4; ++u) xb[u] = pb[4 * u * TS];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m < ts && k < 16 * (m + 1)) {
          const double *pa = Tm + (o + s + 16 * m + lr) * TS + o + s + k + lk;
#pragma unroll
          for (int u = 0; u < 4; ++u) xa[m][u] = pa[4 * u];
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (m < ts && k < 16 * (m + 1)) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[m][u], xb[u], acc[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < ts) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Tm[(o + s + 16 * m + lk + 4 * reg) * TS + o + 16 * r + lr] = -acc[m][reg];
      }
  }
  lds_barrier();
#pragma unroll 8
  for (int e = t; e < CB * CB / 2; e += 256) {
    const int rr = e / (CB / 2), c = 2 * (e - rr * (CB / 2));
    const double2 v = *reinterpret_cast<const double2 *>(Tm + rr * TS + c);
    reinterpret_cast<double2 *>(Wk)[e] = make_double2((rr < kw && c <= rr) ? v.x : 0.0, (rr < kw && c + 1 <= rr) ? v.y : 0.0);
  }
}

__global__ void __launch_bounds__(256)
k_chol_diag(double *__restrict__ A, int n, int ld, int k0, double *__restrict__ Wk, int *__restrict__ info) {
  extern __shared__ double Tm[];  // CB x TS + CB
  chol_diag_block(A, n, ld, k0, Wk, info, Tm);
}

// ------------------------------------------------------------------------------------------------------------------
// the same three steps for many matrices at once: the fronts of one level of a sparse factorisation
// ------------------------------------------------------------------------------------------------------------------
// A front is a dense symmetric matrix of order m = nvp + nu (lower triangle, row-major, even ld) of which only the first
// nv columns are eliminated; rows nv .. nvp are padding (nvp a multiple of CB: the rows that stay start on a panel
// boundary), treated as identity rows by the diagonal step.  Panel kp of every front of the list in one launch; a front
// with fewer panels has nothing to do.
__global__ void __launch_bounds__(256)
k_front_diag(const FrontDesc *__restrict__ fd, int kp, int *__restrict__ info) {
  extern __shared__ double Tm[];
  const FrontDesc f = fd[blockIdx.x];
  const int k0 = CB * kp;
  if (k0 >= f.nvp) return;
  chol_diag_block(f.A, f.nv, f.ld, k0, f.W + (int64_t)kp * CB * CB, info, Tm);
}

// Product tile for the panel steps of the fronts: BM x BN of C from K = 128 columns, four waves of (BM/2) x (BN/2), two
// chunks of 16 columns in flight.  A 128 x 128 tile keeps one CU busy for 24 us whatever the size of the front (4.2
// MFLOP at a CU's 0.3 TFLOP/s), and a panel step of a front of a few thousand rows has only a few dozen of them for 256
// CUs: tiles of 64 rows spread the same step over four times as many CUs.
template <int BM, int BN>
__device__ __forceinline__ void front_tile_nt(const double *__restrict__ A, int lda, int ar0, int am, const double *__restrict__ B, int ldb, int br0,
                                              int bm, double *sA, double *sB, d4 (&acc)[BM / 32][BN / 32]) {
  constexpr int LA = BM + 16, LB = BN + 16;               // strides = 16 mod 32: the four k rows a wave reads fall into both bank halves
  constexpr int TA = 256 / BM, TB = 256 / BN;             // threads per staged row
  constexpr int CA = KC / TA, CBN = KC / TB;              // columns of a chunk per thread
  constexpr int NA = CA / 2, NB = CBN / 2;                // as 16-byte pieces
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1, lr = lane & 15, lk = lane >> 4;
  const int rowa = t / TA, qa = t % TA, rowb = t / TB, qb = t % TB;
  const bool aon = ar0 + rowa < am, bon = br0 + rowb < bm;
  const double *ap = A + (int64_t)(aon ? ar0 + rowa : ar0) * lda + CA * qa, *bp = B + (int64_t)(bon ? br0 + rowb : br0) * ldb + CBN * qb;
  double2 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
  auto fetch = [&](int k0, double2 (&xa)[NA], double2 (&xb)[NB]) {
#pragma unroll
    for (int j = 0; j < NA; ++j) xa[j] = aon ? *reinterpret_cast<const double2 *>(ap + k0 + 2 * j) : make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < NB; ++j) xb[j] = bon ? *reinterpret_cast<const double2 *>(bp + k0 + 2 * j) : make_double2(0.0, 0.0);
  };
  auto stage = [&](const double2 (&xa)[NA], const double2 (&xb)[NB]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      sA[(CA * qa + 2 * j) * LA + rowa] = xa[j].x;
      sA[(CA * qa + 2 * j + 1) * LA + rowa] = xa[j].y;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      sB[(CBN * qb + 2 * j) * LB + rowb] = xb[j].x;
      sB[(CBN * qb + 2 * j + 1) * LB + rowb] = xb[j].y;
    }
    __syncthreads();
  };
  auto multiply = [&]() {
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      double fa[BM / 32], fb[BN / 32];
#pragma unroll
      for (int a = 0; a < BM / 32; ++a) fa[a] = sA[(4 * kk + lk) * LA + (BM / 2) * wr + 16 * a + lr];
#pragma unroll
      for (int c = 0; c < BN / 32; ++c) fb[c] = sB[(4 * kk + lk) * LB + (BN / 2) * wc + 16 * c + lr];
#pragma unroll
      for (int a = 0; a < BM / 32; ++a)
#pragma unroll
        for (int c = 0; c < BN / 32; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[c], acc[a][c], 0, 0, 0);
    }
  };
  fetch(0, ra0, rb0);
  fetch(KC, ra1, rb1);
#pragma unroll
  for (int k0 = 0; k0 < CB; k0 += 2 * KC) {
    stage(ra0, rb0);
    if (k0 + 2 * KC < CB) fetch(k0 + 2 * KC, ra0, rb0);
    multiply();
    stage(ra1, rb1);
    if (k0 + 3 * KC < CB) fetch(k0 + 3 * KC, ra1, rb1);
    multiply();
  }
}

// rows below panel kp: X = A[rows, panel] W^T in place, RB rows (all 128 columns: a workgroup owns its rows) per workgroup.
// RB = 32 where a level has so few rows that tiles of 64 leave most of the device idle: a tile then takes half as long.
// (The same for the trailing update, tiles of 32 x 32, gains nothing: its tiles are bound by their loads and stores.)
template <int RB>
__device__ __forceinline__ void front_trsm_tile(const FrontDesc &f, int kp, double *sA, double *sB) {
  const int k0 = CB * kp;
  if (k0 >= f.nvp) return;
  const int r0 = k0 + CB + RB * blockIdx.x;
  if (r0 >= f.m) return;
  d4 acc[RB / 32][4];
#pragma unroll
  for (int a = 0; a < RB / 32; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
  front_tile_nt<RB, CB>(f.A + k0, f.ld, r0, f.m, f.W + (int64_t)kp * CB * CB, CB, 0, CB, sA, sB, acc);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1, lr = lane & 15, lk = lane >> 4;
  double *C = f.A + k0;
#pragma unroll
  for (int a = 0; a < RB / 32; ++a)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = r0 + (RB / 2) * wr + 16 * a + lk + 4 * reg;
      if (r >= f.m) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) C[(int64_t)r * f.ld + 64 * wc + 16 * c + lr] = acc[a][c][reg];
    }
}
template <int RB>
__global__ void __launch_bounds__(256, 2)
k_front_trsm(const FrontDesc *__restrict__ fd, int kp) {
  __shared__ double sA[KC * (RB + 16)], sB[KC * (CB + 16)];
  front_trsm_tile<RB>(fd[blockIdx.y], kp, sA, sB);
}
template <int RB>
__global__ void __launch_bounds__(256, 2)
k_dense_trsm(FrontDesc f, int kp) {  // one matrix, described by value
  __shared__ double sA[KC * (RB + 16)], sB[KC * (CB + 16)];
  front_trsm_tile<RB>(f, kp, sA, sB);
}

// trailing update C -= X X^T over the 64 x 64 tiles of the lower triangle
__device__ __forceinline__ void front_syrk_tile(const FrontDesc &f, int kp, double *sA, double *sB) {
  const int k0 = CB * kp;
  if (k0 >= f.nvp) return;
  const int t0 = k0 + CB, T = (f.m - t0 + 63) / 64;
  if ((int)blockIdx.x >= T * (T + 1) / 2) return;
  int ti, tj;
  tri_tile(blockIdx.x, ti, tj);
  const int r0 = t0 + 64 * ti, c0 = t0 + 64 * tj;
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
  front_tile_nt<64, 64>(f.A + k0, f.ld, r0, f.m, f.A + k0, f.ld, c0, f.m, sA, sB, acc);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1, lr = lane & 15, lk = lane >> 4;
  double *dst[2][4][2];
  double old[2][4][2];
  bool ok[2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int r = r0 + 32 * wr + 16 * a + lk + 4 * reg, cc = c0 + 32 * wc + 16 * c + lr;
        ok[a][reg][c] = r < f.m && cc <= r;
        dst[a][reg][c] = f.A + (int64_t)min(r, f.m - 1) * f.ld + min(cc, f.m - 1);
        old[a][reg][c] = *dst[a][reg][c];
      }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (ok[a][reg][c]) *dst[a][reg][c] = old[a][reg][c] - acc[a][c][reg];
}
__global__ void __launch_bounds__(256, 2)
k_front_syrk(const FrontDesc *__restrict__ fd, int kp) {
  __shared__ double sA[KC * (64 + 16)], sB[KC * (64 + 16)];
  front_syrk_tile(fd[blockIdx.y], kp, sA, sB);
}
__global__ void __launch_bounds__(256, 2)
k_dense_syrk(FrontDesc f, int kp) {
  __shared__ double sA[KC * (64 + 16)], sB[KC * (64 + 16)];
  front_syrk_tile(f, kp, sA, sB);
}

// ------------------------------------------------------------------------------------------------------------------
// triangular solves with the factor and the inverted diagonal blocks, one launch per panel
// ------------------------------------------------------------------------------------------------------------------
// Both kernels are chains of memory round trips, not of bytes: a workgroup is 2 * CB threads, two per row (column), each
// taking half of the summation index with all its loads requested at once; the halves meet in LDS.  The inverted
// diagonal blocks W hold zeros above the diagonal, so the products with them need no masks.
//
// forward step k: workgroup i (block row k + 1 + i) subtracts L[i, k] x_k from its part of b; the workgroup of block row
// k + 1 then finishes x_{k+1} = W_{k+1} b_{k+1}.  first != 0: only x_0 = W_0 b_0.
__global__ void __launch_bounds__(2 * CB)
k_chol_forward(const double *__restrict__ L, int n, int ld, const double *__restrict__ W, double *__restrict__ b, int k, int first) {
  __shared__ double xs[CB], hs[CB];
  const int t = threadIdx.x & (CB - 1), h = threadIdx.x / CB;
  const int ib = first ? 0 : k + 1 + blockIdx.x;
  const int row = CB * ib + t;
  double v = (row < n && h == 0) ? b[row] : 0.0;
  if (!first) {
    const int k0 = CB * k;
    if (h == 0) xs[t] = b[k0 + t];  // block k is complete (k0 + t < n as block k is not the last one)
    const double2 *lr = reinterpret_cast<const double2 *>(L + (int64_t)min(row, n - 1) * ld + k0 + (CB / 2) * h);
    double2 l[CB / 4];
#pragma unroll
    for (int c = 0; c < CB / 4; ++c) l[c] = lr[c];
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CB / 4; ++c) s += l[c].x * xs[(CB / 2) * h + 2 * c] + l[c].y * xs[(CB / 2) * h + 2 * c + 1];
    if (h == 1) hs[t] = s;
    __syncthreads();
    if (h == 0) v = row < n ? v - (s + hs[t]) : 0.0;
    if (blockIdx.x != 0) {
      if (row < n && h == 0) b[row] = v;
      return;
    }
    __syncthreads();
  }
  // x = W_ib v (lower triangular, CB x CB)
  const double2 *wr = reinterpret_cast<const double2 *>(W + (int64_t)ib * CB * CB + (int64_t)t * CB + (CB / 2) * h);
  double2 wv[CB / 4];
#pragma unroll
  for (int c = 0; c < CB / 4; ++c) wv[c] = wr[c];
  if (h == 0) xs[t] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < CB / 4; ++c) s += wv[c].x * xs[(CB / 2) * h + 2 * c] + wv[c].y * xs[(CB / 2) * h + 2 * c + 1];
  if (h == 1) hs[t] = s;
  __syncthreads();
  if (h == 0 && row < n) b[row] = s + hs[t];
}

// backward step i (from the last block row up): workgroup k < i subtracts L[i, k]^T x_i from its part of b; the workgroup
// of block i - 1 then finishes x_{i-1} = W_{i-1}^T b_{i-1}.  first != 0: only x of the last block.
__global__ void __launch_bounds__(2 * CB)
k_chol_backward(const double *__restrict__ L, int n, int ld, const double *__restrict__ W, double *__restrict__ b, int i, int first) {
  __shared__ double xs[CB], hs[CB];
  const int t = threadIdx.x & (CB - 1), h = threadIdx.x / CB;
  const int kb = first ? i : (int)blockIdx.x;
  const int col = CB * kb + t;
  double v = (col < n && h == 0) ? b[col] : 0.0;
  if (!first) {
    const int i0 = CB * i, iw = min(CB, n - i0);
    if (h == 0) xs[t] = t < iw ? b[i0 + t] : 0.0;
    // consecutive threads, consecutive columns; rows beyond the block: a valid row, times 0
    const double *lc = L + (int64_t)i0 * ld + col;
    double s = 0.0;
#pragma unroll
    for (int r0 = 0; r0 < CB / 2; r0 += 32) {
      double l[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) l[u] = lc[(int64_t)min((CB / 2) * h + r0 + u, iw - 1) * ld];
      if (r0 == 0) __syncthreads();
#pragma unroll
      for (int u = 0; u < 32; ++u) s += l[u] * xs[(CB / 2) * h + r0 + u];
    }
    if (h == 1) hs[t] = s;
    __syncthreads();
    if (h == 0) v -= s + hs[t];
    if (kb != i - 1) {
      if (h == 0) b[col] = v;
      return;
    }
    __syncthreads();
  }
  const double *wk = W + (int64_t)kb * CB * CB + t;
  double s = 0.0;
#pragma unroll
  for (int r0 = 0; r0 < CB / 2; r0 += 32) {
    double w[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) w[u] = wk[(int64_t)((CB / 2) * h + r0 + u) * CB];  // W^T
    if (r0 == 0) {
      if (h == 0) xs[t] = v;
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) s += w[u] * xs[(CB / 2) * h + r0 + u];
  }
  if (h == 1) hs[t] = s;
  __syncthreads();
  if (h == 0 && col < n) b[col] = s + hs[t];
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
  const size_t lds = (size_t)(CB * TS + CB) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_diag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int nb = (n + CB - 1) / CB;
  if (nb <= 64) {
    // Up to order 8192 the panel steps of the sparse factorisation's fronts, tiles of 64 (32) rows, on the one matrix: a
    // 128 x 128 tile keeps a CU busy for 24 us whatever the order, and the smaller tiles fill the device sooner (order
    // 1248: 0.57 against 0.85 ms, 3996: 2.2 against 3.0 ms, 8190: 7.1 against 9.4 ms; beyond, the larger tiles' lower
    // traffic per operation wins: 48 % of the fp64 matrix peak at order 30 000)
    FrontDesc f{};
    f.A = A; f.W = W; f.nv = n; f.nvp = CB * nb; f.m = n; f.ld = ld;
    for (int kp = 0; kp < nb; ++kp) {
      hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), lds, s, A, n, ld, CB * kp, W + (size_t)kp * CB * CB, info);
      const int T = (n - CB * kp - CB + 63) / 64;
      if (T <= 0) break;
      if (T <= 128) hipLaunchKernelGGL(k_dense_trsm<32>, dim3(2 * T), dim3(256), 0, s, f, kp);
      else hipLaunchKernelGGL(k_dense_trsm<64>, dim3(T), dim3(256), 0, s, f, kp);
      hipLaunchKernelGGL(k_dense_syrk, dim3(T * (T + 1) / 2), dim3(256), 0, s, f, kp);
    }
    return;
  }
  GemmArgs g{};
  g.A = A; g.n = n; g.ld = ld;
  for (int k0 = 0, kb = 0; k0 < n; k0 += CB, ++kb) {
    const int kw = (n - k0 < CB) ? n - k0 : CB;
    double *Wk = W + (size_t)kb * CB * CB;
    hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), lds, s, A, n, ld, k0, Wk, info);
    const int rem = n - k0 - kw;
    if (rem <= 0) break;
    g.W = Wk; g.k0 = k0; g.kw = kw;
    const int T = (rem + CB - 1) / CB;
    hipLaunchKernelGGL(k_gemm_nt<GEMM_TRSM>, dim3(T), dim3(256), 0, s, g);
    // One launch for the whole trailing update.  Running its first block column ahead of the rest (so that the next
    // diagonal block could start early, the rest on a second stream) gained nothing at order 8190 and cost 0.1 ms of
    // event hand-overs at order 1248.
    hipLaunchKernelGGL(k_gemm_nt<GEMM_SYRK>, dim3(T * (T + 1) / 2), dim3(256), 0, s, g);
  }
}

void launch_chol_solve_blocked(hipStream_t s, const double *L, int n, int ld, const double *W, double *b) {
  const int nb = (n + CB - 1) / CB;
  hipLaunchKernelGGL(k_chol_forward, dim3(1), dim3(2 * CB), 0, s, L, n, ld, W, b, 0, 1);
  for (int k = 0; k + 1 < nb; ++k) hipLaunchKernelGGL(k_chol_forward, dim3(nb - 1 - k), dim3(2 * CB), 0, s, L, n, ld, W, b, k, 0);
  hipLaunchKernelGGL(k_chol_backward, dim3(1), dim3(2 * CB), 0, s, L, n, ld, W, b, nb - 1, 1);
  for (int i = nb - 1; i >= 1; --i) hipLaunchKernelGGL(k_chol_backward, dim3(i), dim3(2 * CB), 0, s, L, n, ld, W, b, i, 0);
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

// C = (accumulate ? C : 0) + alpha A B^T for any row-major operands with the summation index contiguous: A am x k,
// B bm x k, C am x bm; one workgroup per 128 x 128 tile of C.  lda and ldb are even and the column after an odd k
// holds zero.
struct GemmNtArgs {
  const double *A, *B;
  double *C;
  int lda, ldb, ldc, am, bm, k;
  double alpha;
  int accumulate;
};

__global__ void __launch_bounds__(256, 2)
k_gemm_nt_general(GemmNtArgs g) {
  __shared__ double sA[KC * LDT], sB[KC * LDT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1, lr = lane & 15, lk = lane >> 4;
  d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
  const int r0 = CB * blockIdx.x, c0 = CB * blockIdx.y;
  gemm_tile_nt(g.A, g.lda, r0, g.am, g.B, g.ldb, c0, g.bm, 0, g.k, sA, sB, acc);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = r0 + 64 * wr + 16 * a + lk + 4 * reg, cc = c0 + 64 * wc + 16 * c + lr;
        if (r >= g.am || cc >= g.bm) continue;
        double *dst = g.C + (int64_t)r * g.ldc + cc;
        const double v = g.alpha * acc[a][c][reg];
        *dst = g.accumulate ? *dst + v : v;
      }
}

void launch_gemm_nt(hipStream_t s, const double *A, int lda, int am, const double *B, int ldb, int bm, int k, double alpha,
                    bool accumulate, double *C, int ldc) {
  if (am <= 0 || bm <= 0) return;
  GemmNtArgs g{A, B, C, lda, ldb, ldc, am, bm, k, alpha, accumulate ? 1 : 0};
  hipLaunchKernelGGL(k_gemm_nt_general, dim3((am + CB - 1) / CB, (bm + CB - 1) / CB), dim3(256), 0, s, g);
}

void launch_front_panels(hipStream_t s, const FrontDesc *fd, int n_fronts, int n_panels, int max_m, int *info) {
  if (n_fronts <= 0) return;
  const size_t lds = (size_t)(CB * TS + CB) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_front_diag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  for (int kp = 0; kp < n_panels; ++kp) {
    hipLaunchKernelGGL(k_front_diag, dim3(n_fronts), dim3(256), lds, s, fd, kp, info);
    const int T = (max_m - CB * kp - CB + 63) / 64;  // tiles of 64 rows below the panel, of the largest front
    if (T <= 0) continue;
    if (T * n_fronts <= 128) hipLaunchKernelGGL(k_front_trsm<32>, dim3(2 * T, n_fronts), dim3(256), 0, s, fd, kp);
    else hipLaunchKernelGGL(k_front_trsm<64>, dim3(T, n_fronts), dim3(256), 0, s, fd, kp);
    hipLaunchKernelGGL(k_front_syrk, dim3(T * (T + 1) / 2, n_fronts), dim3(256), 0, s, fd, kp);
  }
}
int front_panel() { return CB; }

}  // namespace ba
