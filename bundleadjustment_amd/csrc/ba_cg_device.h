// ba_cg_device.h -- device code shared by the PCG kernels of ba_schur.hip and ba_amg.hip
#pragma once

#include "ba_device.h"
#include "ba_kernels.h"

namespace ba {

// time stamp of a phase mark (tracing only: the wait makes the stamp mean "everything before it has arrived")
__device__ __forceinline__ void pcg_mark(unsigned long long *tr, int kernel, int mark, int wg) {
  if (tr) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && wg < PCG_TRACE_MAXWG) tr[((size_t)kernel * PCG_TRACE_MAXWG + wg) * PCG_TRACE_SLOTS + mark] = wall_clock64();
  }
}

// thread per station: z_i = Minv_i r_i; returns r_i . z_i
__device__ __forceinline__ double station_precond(const double *__restrict__ Minv, const double *__restrict__ r,
                                                  double *__restrict__ z, int i) {
  const double *m = Minv + 36 * (int64_t)i, *ri = r + 6 * (int64_t)i;
  const double r0 = ri[0], r1 = ri[1], r2 = ri[2], r3 = ri[3], r4 = ri[4], r5 = ri[5];
  double dot = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double v = m[6 * k] * r0 + m[6 * k + 1] * r1 + m[6 * k + 2] * r2 + m[6 * k + 3] * r3 + m[6 * k + 4] * r4 + m[6 * k + 5] * r5;
    z[6 * (int64_t)i + k] = v;
    dot += v * ri[k];
  }
  return dot;
}

// per-workgroup partials of G^T r_s (thread i holds station i; 256 threads)
#define DEFL_NC 8
// Most free camera parameters (summed over the cameras) the kernels of the explicit block-sparse PCG hold in LDS;
// networks with more take the matrix-free PCG (ba_solve_resident) or the dense solver.
#define CG2_MAX_CAM 79
__device__ __forceinline__ void defl_partials(const BsrCgArgs &a, const double *__restrict__ r, int i, int vblock) {
  __shared__ double lds_gr[4][DEFL_NC];
  double gr[DEFL_NC];
#pragma unroll
  for (int k = 0; k < DEFL_NC; ++k) gr[k] = 0.0;
  if (i < a.S) {
    // GT is G by columns: the six entries of a station are contiguous, neighbouring threads adjoin
    const double *ri = r + 6 * (int64_t)i;
    const double r0 = ri[0], r1 = ri[1], r2 = ri[2], r3 = ri[3], r4 = ri[4], r5 = ri[5];
    const int64_t ld = 6 * (int64_t)a.S;
#pragma unroll
    for (int k = 0; k < DEFL_NC; ++k)
      if (k < a.ncam) {
        const double2 *g = reinterpret_cast<const double2 *>(a.GT + k * ld + 6 * (int64_t)i);
        const double2 g0 = g[0], g1 = g[1], g2 = g[2];
        gr[k] = g0.x * r0 + g0.y * r1 + g1.x * r2 + g1.y * r3 + g2.x * r4 + g2.y * r5;
      }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < DEFL_NC; ++k) {
    const double v = wave_sum(gr[k]);
    if (lane == 0) lds_gr[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < a.ncam)
    a.part_gr[(int64_t)vblock * a.ncam + threadIdx.x] =
        (lds_gr[0][threadIdx.x] + lds_gr[1][threadIdx.x]) + (lds_gr[2][threadIdx.x] + lds_gr[3][threadIdx.x]);
}

// Second kernel of a PCG iteration, as a function so that the multilevel variant can run the first
// restriction in the same launch.  Every workgroup folds the SpMV partials to alpha = r.z / p.Sp.
// Workgroups with vblock >= 0 (n_vblocks of them, 256 threads, one thread per station) then do
// x += alpha p, r_out = r_in - alpha q, z = D^-1 r_out and leave partials of r.z; the others return
// with alpha only.  r_in == r_out updates the residual in place.
__device__ __forceinline__ double cg2_update1_body(const BsrCgArgs &a, int parity, int nblk_spmv, const double *rin,
                                                   double *rout, int vblock, int n_vblocks) {
  __shared__ double lds[8];
  __shared__ double red[CG2_MAX_CAM + 1];   // [0] p.q over the station rows, [1 + k] camera row k of S p
  __shared__ double pcs[CG2_MAX_CAM + 1];   // camera part of p
  __shared__ double pq_s;
  __shared__ double fold4[4][8];
  __shared__ double rcs[CG2_MAX_CAM + 1];
  // Every workgroup folds the SpMV partials in the same order -> identical alpha everywhere.
  // Eight sums at a time: thread t takes sum (t & 7) and every 32nd partial from (t >> 3) on, so all
  // its loads are independent; then the 8 threads-classes are reduced across the wave and the waves.
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t o = 6 * (int64_t)a.S;
  pcg_mark(a.trace, 1, 1, blockIdx.x);
  for (int k = threadIdx.x; k < a.ncam; k += blockDim.x) pcs[k] = a.p[o + k];
  for (int q0 = 0; q0 < 1 + a.ncam; q0 += 8) {
    const int qn = q0 + (threadIdx.x & 7);
    double s = 0.0;
    if (qn < 1 + a.ncam) {
      const double *src = (qn == 0) ? a.part_pq : a.part_qc + (qn - 1);
      const int64_t st = (qn == 0) ? 1 : a.ncam;
      // at most 1024 partials (cg2_blocks_spmv): 32 independent loads per thread, summed in a fixed tree
      double v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const int m = (threadIdx.x >> 3) + 32 * j;
        v[j] = (m < nblk_spmv) ? src[st * m] : 0.0;
      }
#pragma unroll
      for (int w2 = 16; w2 >= 1; w2 >>= 1)
#pragma unroll
        for (int j = 0; j < w2; ++j) v[j] += v[j + w2];
      s = v[0];
    }
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    __syncthreads();
    if (lane < 8) fold4[w][lane] = s;
    __syncthreads();
    if (threadIdx.x < 8 && q0 + threadIdx.x < 1 + a.ncam)
      red[q0 + threadIdx.x] = (fold4[0][threadIdx.x] + fold4[1][threadIdx.x]) + (fold4[2][threadIdx.x] + fold4[3][threadIdx.x]);
  }
  __syncthreads();
  // camera rows: add the corner, S_cc p_c
  double qk = 0.0;
  if (threadIdx.x < a.ncam) {
    qk = red[1 + threadIdx.x];
    for (int m = 0; m < a.ncam; ++m) qk += a.Scc[threadIdx.x * a.ncam + m] * pcs[m];
  }
  __syncthreads();
  if (threadIdx.x < a.ncam) red[1 + threadIdx.x] = qk;
  __syncthreads();
  if (threadIdx.x == 0) {
    double pq = red[0];
    for (int k = 0; k < a.ncam; ++k) pq += red[1 + k] * pcs[k];
    pq_s = pq;
  }
  __syncthreads();
  const double pq = pq_s;
  const double rz = a.scal[parity];
  const double alpha = rz / pq;
  pcg_mark(a.trace, 1, 2, blockIdx.x);
  if (vblock < 0) return alpha;
  const int i = vblock * blockDim.x + threadIdx.x;
  double dot = 0.0;
  if (i < a.S) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      a.x[6 * (int64_t)i + k] += alpha * a.p[6 * (int64_t)i + k];
      rout[6 * (int64_t)i + k] = rin[6 * (int64_t)i + k] - alpha * a.q[6 * (int64_t)i + k];
    }
    dot = station_precond(a.Minv, rout, a.z, i);
  }
  if (a.G) defl_partials(a, rout, i, vblock);
  if (vblock == n_vblocks - 1) {
    // the last workgroup is the least loaded one: it also carries the camera entries (ncam <= 80).
    // z_c = Scc^-1 r_c with the explicit inverse, one thread per row; with the camera coarse space only
    // x and r here, the camera part of z follows in the last kernel.
    if (threadIdx.x < a.ncam) {
      a.x[o + threadIdx.x] += alpha * pcs[threadIdx.x];
      const double rk = rin[o + threadIdx.x] - alpha * red[1 + threadIdx.x];
      rout[o + threadIdx.x] = rk;
      rcs[threadIdx.x] = rk;
    }
    __syncthreads();
    if (!a.G && threadIdx.x < a.ncam) {
      double z = 0.0;
      for (int m = 0; m < a.ncam; ++m) z += a.Lcc[threadIdx.x * a.ncam + m] * rcs[m];
      a.z[o + threadIdx.x] = z;
      dot += rcs[threadIdx.x] * z;
    }
    if (threadIdx.x == 0) a.scal[4] = pq;
  }
  pcg_mark(a.trace, 1, 3, blockIdx.x);
  const double t = block_sum(dot, lds);
  if (threadIdx.x == 0) a.part_rz[vblock] = t;
  pcg_mark(a.trace, 1, 4, blockIdx.x);
  return alpha;
}


}  // namespace ba
