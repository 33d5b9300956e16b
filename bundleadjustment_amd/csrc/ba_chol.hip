// ba_chol.hip -- dense fp64 Cholesky of the reduced camera system on the matrix cores.
//
// Right-looking blocked factorisation of a row-major symmetric matrix (lower triangle used),
// block size 32:
//   1. k_potrf_diag : the 32x32 diagonal block, one workgroup, in LDS;
//   2. k_trsm_panel : rows below it, one lane per row, L_kk staged in LDS;
//   3. k_syrk_mfma  : trailing update C -= P P^T with v_mfma_f64_16x16x4_f64, one wave per
//                     32x32 tile of the lower triangle (2x2 MFMA tiles, K = 32 in 8 steps).
// The O(n^3) work is all in step 3.  Replaces the sparse Cholesky that Ceres runs for
// SPARSE_NORMAL_CHOLESKY (src/main.cpp:4706), applied here to the reduced system.
#include "ba_kernels.h"

namespace ba {

#define NB 32
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
k_potrf_diag(double *__restrict__ A, int n, int k0, int *__restrict__ info) {
  __shared__ double T[NB][NB + 1];
  const int nb = min(NB, n - k0);
  for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) {
    const int r = t / NB, c = t % NB;
    T[r][c] = (r < nb && c < nb && c <= r) ? A[(int64_t)(k0 + r) * n + k0 + c] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    if (threadIdx.x == 0) {
      const double d = T[j][j];
      if (!(d > 0.0)) atomicMax(info, k0 + j + 1);
      T[j][j] = sqrt(d);
    }
    __syncthreads();
    const double dj = T[j][j];
    for (int r = j + 1 + threadIdx.x; r < nb; r += blockDim.x) T[r][j] /= dj;
    __syncthreads();
    // rank-1 update of the remaining lower triangle
    for (int t = threadIdx.x; t < (nb - j - 1) * (nb - j - 1); t += blockDim.x) {
      const int r = j + 1 + t / (nb - j - 1), c = j + 1 + t % (nb - j - 1);
      if (c <= r) T[r][c] -= T[r][j] * T[c][j];
    }
    __syncthreads();
  }
  for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) {
    const int r = t / NB, c = t % NB;
    if (r < nb && c < nb && c <= r) A[(int64_t)(k0 + r) * n + k0 + c] = T[r][c];
  }
}

// rows i >= k0+nb:  A[i, k0:k0+nb] <- A[i, k0:k0+nb] * L_kk^-T
__global__ void __launch_bounds__(256)
k_trsm_panel(double *__restrict__ A, int n, int k0) {
  __shared__ double T[NB][NB + 1];
  const int nb = min(NB, n - k0);
  for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) {
    const int r = t / NB, c = t % NB;
    T[r][c] = (r < nb && c <= r) ? A[(int64_t)(k0 + r) * n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int i = k0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double *row = A + (int64_t)i * n + k0;
  double x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) x[c] = (c < nb) ? row[c] : 0.0;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    if (c < nb) {
      double s = x[c];
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (k < c) s -= x[k] * T[c][k];
      x[c] = s / T[c][c];
    }
  }
#pragma unroll
  for (int c = 0; c < NB; ++c)
    if (c < nb) row[c] = x[c];
}

// trailing update, one wave per 32x32 tile (ti >= tj) of the trailing lower triangle
__global__ void __launch_bounds__(64)
k_syrk_mfma(double *__restrict__ A, int n, int k0, int t0 /* first trailing row */, int T) {
  // map linear block id -> (ti, tj) with tj <= ti
  int b = blockIdx.x, ti = 0;
  while (b >= ti + 1) {
    b -= ti + 1;
    ++ti;
  }
  const int tj = b;
  if (ti >= T) return;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  const int nb = min(NB, n - k0);
  const int i0 = t0 + 32 * ti, j0 = t0 + 32 * tj;
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < nb; kk += 4) {
    const int kc = kk + lk;
    double av[2], bv[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ri = i0 + 16 * a + lr, rj = j0 + 16 * a + lr;
      av[a] = (ri < n && kc < nb) ? A[(int64_t)ri * n + k0 + kc] : 0.0;
      bv[a] = (rj < n && kc < nb) ? A[(int64_t)rj * n + k0 + kc] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = i0 + 16 * a + lk + 4 * reg, cc = j0 + 16 * c + lr;
        if (r < n && cc < n && cc <= r) A[(int64_t)r * n + cc] -= acc[a][c][reg];
      }
}

// L L^T x = b, one workgroup; column-oriented substitution
__global__ void __launch_bounds__(1024)
k_chol_solve(const double *__restrict__ L, int n, double *__restrict__ b) {
  __shared__ double xi;
  for (int i = 0; i < n; ++i) {
    if (threadIdx.x == 0) {
      xi = b[i] / L[(int64_t)i * n + i];
      b[i] = xi;
    }
    __syncthreads();
    const double v = xi;
    for (int r = i + 1 + threadIdx.x; r < n; r += blockDim.x) b[r] -= L[(int64_t)r * n + i] * v;
    __syncthreads();
  }
  for (int i = n - 1; i >= 0; --i) {
    if (threadIdx.x == 0) {
      xi = b[i] / L[(int64_t)i * n + i];
      b[i] = xi;
    }
    __syncthreads();
    const double v = xi;
    for (int r = threadIdx.x; r < i; r += blockDim.x) b[r] -= L[(int64_t)i * n + r] * v;
    __syncthreads();
  }
}

// the same for many right-hand sides, one workgroup each: column c of B (stored as row c, n entries)
__global__ void __launch_bounds__(256)
k_chol_solve_many(const double *__restrict__ L, int n, double *__restrict__ B) {
  __shared__ double xi;
  double *b = B + (int64_t)blockIdx.x * n;
  for (int i = 0; i < n; ++i) {
    if (threadIdx.x == 0) {
      xi = b[i] / L[(int64_t)i * n + i];
      b[i] = xi;
    }
    __syncthreads();
    const double v = xi;
    if (v != 0.0)
      for (int r = i + 1 + threadIdx.x; r < n; r += blockDim.x) b[r] -= L[(int64_t)r * n + i] * v;
    __syncthreads();
  }
  for (int i = n - 1; i >= 0; --i) {
    if (threadIdx.x == 0) {
      xi = b[i] / L[(int64_t)i * n + i];
      b[i] = xi;
    }
    __syncthreads();
    const double v = xi;
    for (int r = threadIdx.x; r < i; r += blockDim.x) b[r] -= L[(int64_t)i * n + r] * v;
    __syncthreads();
  }
}

// B = identity (n x n)
__global__ void k_identity(int n, double *__restrict__ B) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < (int64_t)n * n) B[t] = (t / n == t % n) ? 1.0 : 0.0;
}

// inverse of L L^T into B (n x n, symmetric)
void launch_chol_inverse(hipStream_t s, const double *L, int n, double *B) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_identity, dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0, s, n, B);
  hipLaunchKernelGGL(k_chol_solve_many, dim3(n), dim3(256), 0, s, L, n, B);
}

void launch_cholesky(hipStream_t s, double *A, int n, int *info) {
  hipMemsetAsync(info, 0, sizeof(int), s);
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int nb = (n - k0 < NB) ? n - k0 : NB;
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), 0, s, A, n, k0, info);
    const int rem = n - k0 - nb;
    if (rem <= 0) break;
    hipLaunchKernelGGL(k_trsm_panel, dim3((rem + 255) / 256), dim3(256), 0, s, A, n, k0);
    const int T = (rem + 31) / 32;
    hipLaunchKernelGGL(k_syrk_mfma, dim3(T * (T + 1) / 2), dim3(64), 0, s, A, n, k0, k0 + nb, T);
  }
}

void launch_chol_solve(hipStream_t s, const double *L, int n, double *b) {
  if (n > 0) hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(1024), 0, s, L, n, b);
}

}  // namespace ba
