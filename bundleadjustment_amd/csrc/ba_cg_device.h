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

// per-workgroup partials of G^T r_s from the per-station products gr (thread i holds station i; 256 threads)
#define DEFL_NC 8
// Most free camera parameters (summed over the cameras) the kernels of the explicit block-sparse PCG hold in LDS;
// networks with more take the matrix-free PCG (ba_solve_resident) or the dense solver.
#define CG2_MAX_CAM 79
__device__ __forceinline__ void defl_partials(const BsrCgArgs &a, const double (&gr)[DEFL_NC], int vblock) {
  __shared__ double lds_gr[4][DEFL_NC];
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

// gr[k] = column k of G times the six entries ri of station i (zeros beyond the stations and the columns)
__device__ __forceinline__ void defl_products(const BsrCgArgs &a, const double (&ri)[6], int i, double (&gr)[DEFL_NC]) {
#pragma unroll
  for (int k = 0; k < DEFL_NC; ++k) gr[k] = 0.0;
  if (i < a.S) {
    const int64_t ld = 6 * (int64_t)a.S;
#pragma unroll
    for (int k = 0; k < DEFL_NC; ++k)
      if (k < a.ncam) {
        const double2 *g = reinterpret_cast<const double2 *>(a.GT + k * ld + 6 * (int64_t)i);
        const double2 g0 = g[0], g1 = g[1], g2 = g[2];
        gr[k] = g0.x * ri[0] + g0.y * ri[1] + g1.x * ri[2] + g1.y * ri[3] + g2.x * ri[4] + g2.y * ri[5];
      }
  }
}

// ---- second kernel of a PCG iteration ---------------------------------------------------------------------------
// Every workgroup folds the per-workgroup partials of the SpMV (p.q over the station rows and the camera rows of
// S p, corner included) in the same fixed order, so alpha = r.z / p.Sp is identical everywhere; the vector workgroups (one thread per
// station) then do x += alpha p, r_out = r_in - alpha q, z = D^-1 r_out and leave partials of r.z.
// The kernels are bounded by the latency of dependent loads, not by bytes, so the work is cut into an ISSUE half that
// only loads (nothing in it depends on alpha) and a FINISH half that only computes and stores: a kernel issues all
// its halves first and the loads of the fold, of the vector update and of a restriction travel together.
struct Cg2Shared {
  double lds[8];
  double red[CG2_MAX_CAM + 1];  // [0] p.q over the station rows, [1 + k] camera row k of S p
  double pcs[CG2_MAX_CAM + 1];  // camera part of p
  double rcs[CG2_MAX_CAM + 1];
  double fold4[4][8];
  double pq;
};

// Eight sums at a time (q0 .. q0 + 7 of the 1 + ncam): thread t takes sum (t & 7) and every 32nd partial from
// (t >> 3) on; at most CG2_MAX_SPMV_BLOCKS partials (cg2_blocks_spmv).
#define CG2_FOLD_LOADS 8
#define CG2_MAX_SPMV_BLOCKS (32 * CG2_FOLD_LOADS)
struct FoldLoads {
  double v[CG2_FOLD_LOADS];
  double pc;        // camera entry threadIdx.x of p
};
__device__ __forceinline__ void fold_issue(const BsrCgArgs &a, int nblk_spmv, int q0, FoldLoads &f) {
  const int qn = q0 + (threadIdx.x & 7);
  const bool on = qn < 1 + a.ncam;
  const double *src = (qn == 0) ? a.part_pq : a.part_qc + (qn - 1);
  const int64_t st = (qn == 0) ? 1 : a.ncam;
#pragma unroll
  for (int j = 0; j < CG2_FOLD_LOADS; ++j) {
    const int m = (threadIdx.x >> 3) + 32 * j;
    f.v[j] = (on && m < nblk_spmv) ? src[st * m] : 0.0;
  }
  if (q0 == 0) {
    const int t = threadIdx.x;
    f.pc = (t < a.ncam) ? a.p[6 * (int64_t)a.S + t] : 0.0;
  }
}

// alpha from the loads of fold_issue(q0 = 0) (further rounds of eight sums are loaded here); leaves red, pcs, pq in sh
__device__ __forceinline__ double cg2_alpha(const BsrCgArgs &a, double rz, int nblk_spmv, FoldLoads &f, Cg2Shared &sh) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < a.ncam) sh.pcs[threadIdx.x] = f.pc;
  for (int q0 = 0; q0 < 1 + a.ncam; q0 += 8) {
    if (q0 > 0) fold_issue(a, nblk_spmv, q0, f);
#pragma unroll
    for (int w2 = 4; w2 >= 1; w2 >>= 1)
#pragma unroll
      for (int j = 0; j < w2; ++j) f.v[j] += f.v[j + w2];
    double s = f.v[0];
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    __syncthreads();
    if (lane < 8) sh.fold4[w][lane] = s;
    __syncthreads();
    if (threadIdx.x < 8 && q0 + threadIdx.x < 1 + a.ncam)
      sh.red[q0 + threadIdx.x] = (sh.fold4[0][threadIdx.x] + sh.fold4[1][threadIdx.x]) + (sh.fold4[2][threadIdx.x] + sh.fold4[3][threadIdx.x]);
  }
  __syncthreads();
  // (the camera rows are complete: the product kernel's first workgroup has added the corner, S_cc p_c, to its partial)
  if (threadIdx.x == 0) {
    double pq = sh.red[0];
    for (int k = 0; k < a.ncam; ++k) pq += sh.red[1 + k] * sh.pcs[k];
    sh.pq = pq;
  }
  __syncthreads();
  return rz / sh.pq;
}

// vector update of station i = vblock * 256 + threadIdx.x.  u = D^-1 r_in and w = D^-1 q are formed while alpha is
// still unknown (the 36 entries of the block are dead again before the fold starts); z = u - alpha w afterwards.
struct VecLoads {
  double2 x[3], p[3];
  double r[6], q[6], u[6], w[6];
  double rc;  // camera entry threadIdx.x of r_in (last vector workgroup)
  double gr[DEFL_NC], gq[DEFL_NC];  // camera coarse space: column k of G times r_in and times q (station i)
};
__device__ __forceinline__ void vec_issue(const BsrCgArgs &a, const double *rin, int vblock, int n_vblocks, VecLoads &v) {
  const int i = vblock * blockDim.x + threadIdx.x;
  if (i < a.S) {
    const double2 *x = reinterpret_cast<const double2 *>(a.x + 6 * (int64_t)i), *p = reinterpret_cast<const double2 *>(a.p + 6 * (int64_t)i);
    const double2 *r = reinterpret_cast<const double2 *>(rin + 6 * (int64_t)i), *q = reinterpret_cast<const double2 *>(a.q + 6 * (int64_t)i);
    const double2 *m = reinterpret_cast<const double2 *>(a.Minv + 36 * (int64_t)i);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v.x[k] = x[k];
      v.p[k] = p[k];
      const double2 rr = r[k], qq = q[k];
      v.r[2 * k] = rr.x; v.r[2 * k + 1] = rr.y;
      v.q[2 * k] = qq.x; v.q[2 * k + 1] = qq.y;
    }
    // the block in two halves of three rows: 36 registers in flight instead of 72
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double2 mm[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) mm[k] = m[9 * h + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double2 m0 = mm[3 * k], m1 = mm[3 * k + 1], m2 = mm[3 * k + 2];
        v.u[3 * h + k] = m0.x * v.r[0] + m0.y * v.r[1] + m1.x * v.r[2] + m1.y * v.r[3] + m2.x * v.r[4] + m2.y * v.r[5];
        v.w[3 * h + k] = m0.x * v.q[0] + m0.y * v.q[1] + m1.x * v.q[2] + m1.y * v.q[3] + m2.x * v.q[4] + m2.y * v.q[5];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int k = 0; k < DEFL_NC; ++k) v.gr[k] = v.gq[k] = 0.0;
  if (a.G && i < a.S) {
    // G^T r_out = G^T r_in - alpha G^T q: both products before alpha is known, like u and w.  GT is G by columns:
    // the six entries of a station are contiguous, neighbouring threads adjoin.
    const int64_t ld = 6 * (int64_t)a.S;
#pragma unroll
    for (int h = 0; h < DEFL_NC; h += 4) {  // four columns in flight at a time
      double2 g[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (h + k < a.ncam) {
          const double2 *gp = reinterpret_cast<const double2 *>(a.GT + (h + k) * ld + 6 * (int64_t)i);
          g[k][0] = gp[0]; g[k][1] = gp[1]; g[k][2] = gp[2];
        }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (h + k < a.ncam) {
          v.gr[h + k] = g[k][0].x * v.r[0] + g[k][0].y * v.r[1] + g[k][1].x * v.r[2] + g[k][1].y * v.r[3] + g[k][2].x * v.r[4] + g[k][2].y * v.r[5];
          v.gq[h + k] = g[k][0].x * v.q[0] + g[k][0].y * v.q[1] + g[k][1].x * v.q[2] + g[k][1].y * v.q[3] + g[k][2].x * v.q[4] + g[k][2].y * v.q[5];
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  v.rc = (vblock == n_vblocks - 1 && threadIdx.x < a.ncam) ? rin[6 * (int64_t)a.S + threadIdx.x] : 0.0;
}
__device__ __forceinline__ void vec_finish(const BsrCgArgs &a, double alpha, double *rout, int vblock, int n_vblocks, const VecLoads &v,
                                           Cg2Shared &sh) {
  const int i = vblock * blockDim.x + threadIdx.x;
  const int64_t o = 6 * (int64_t)a.S;
  double dot = 0.0;
  double r[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (i < a.S) {
    double z[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      r[k] = v.r[k] - alpha * v.q[k];
      z[k] = v.u[k] - alpha * v.w[k];
      dot += z[k] * r[k];
    }
    double2 *xo = reinterpret_cast<double2 *>(a.x + 6 * (int64_t)i), *ro = reinterpret_cast<double2 *>(rout + 6 * (int64_t)i);
    double2 *zo = reinterpret_cast<double2 *>(a.z + 6 * (int64_t)i);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      xo[k] = make_double2(v.x[k].x + alpha * v.p[k].x, v.x[k].y + alpha * v.p[k].y);
      ro[k] = make_double2(r[2 * k], r[2 * k + 1]);
      zo[k] = make_double2(z[2 * k], z[2 * k + 1]);
    }
  }
  if (a.G) {
    double gr[DEFL_NC];
#pragma unroll
    for (int k = 0; k < DEFL_NC; ++k) gr[k] = v.gr[k] - alpha * v.gq[k];
    defl_partials(a, gr, vblock);
  }
  if (vblock == n_vblocks - 1) {
    // the last workgroup is the least loaded one: it also carries the camera entries (ncam <= CG2_MAX_CAM).
    // z_c = Scc^-1 r_c with the explicit inverse, one thread per row; with the camera coarse space only
    // x and r here, the camera part of z follows in the last kernel.
    if (threadIdx.x < a.ncam) {
      a.x[o + threadIdx.x] += alpha * sh.pcs[threadIdx.x];
      const double rk = v.rc - alpha * sh.red[1 + threadIdx.x];
      rout[o + threadIdx.x] = rk;
      sh.rcs[threadIdx.x] = rk;
    }
    __syncthreads();
    if (!a.G && threadIdx.x < a.ncam) {
      double z = 0.0;
      for (int m = 0; m < a.ncam; ++m) z += a.Lcc[threadIdx.x * a.ncam + m] * sh.rcs[m];
      a.z[o + threadIdx.x] = z;
      dot += sh.rcs[threadIdx.x] * z;
    }
    if (threadIdx.x == 0) a.scal[4] = sh.pq;
  }
  pcg_mark(a.trace, 1, 3, blockIdx.x);
  const double t = block_sum(dot, sh.lds);
  if (threadIdx.x == 0) a.part_rz[vblock] = t;
  pcg_mark(a.trace, 1, 4, blockIdx.x);
}

}  // namespace ba
