// ba_schur.hip -- the reduced camera system in explicit block-sparse form, and PCG on it.
//
// After the point blocks are eliminated, the station part of the reduced matrix is
//   S_ii'  =  [i == i'] (sum_a Js_a^T M_a Js_a + D^2)  -  sum_{points seen by i and i'} Y_a Y_b^T ,
// with Y = Js^T Jp L (6x3 per observation, V^-1 = L L^T) written by the linearisation kernel.
// The 6x6 blocks are stored in BSR form (row = station, diagonal block first, neighbours by
// ascending column).  Which observation pairs feed which block is fixed by the network and is
// prepared once on the host: the pair list is sorted by block, so every block is summed by ONE
// wave in a fixed order -- no atomics, bitwise reproducible.
// The camera (IOP/AP) columns are a dense border [S x 6 x ncam] plus an ncam x ncam corner.
// PCG then runs entirely on this explicit system: one SpMV kernel and two vector kernels per
// iteration, all multi-workgroup; the scalars (alpha, beta, stopping test) never leave the device.
#include "ba_kernels.h"

#include "ba_cg_device.h"
#include "ba_device.h"

namespace ba {

// ------------------------------------------------------------------------------------------
// assembly
// ------------------------------------------------------------------------------------------
// One wave per upper-triangular block u (lo <= hi).  Four lanes share one observation pair (each
// owns a 3x3 quadrant of Y_a Y_b^T), so sixteen pairs are in flight per wave and their row gathers
// overlap; the sixteen partial blocks are folded with wave shuffles in a fixed order.
// Rows 3h .. 3h+2 of one 6 x 3 record (18 doubles, 16-byte aligned): nine doubles that start on a 16-byte boundary
// (h = 0) or 8 bytes past one (h = 1); either way four 16-byte loads and one 8-byte load
__device__ __forceinline__ void load_half(const double *__restrict__ rec, int h, double (&v)[3][3]) {
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

__global__ void __launch_bounds__(256)
k_schur_blocks(int nu, const int *__restrict__ cstart, const int2 *__restrict__ dst, const int2 *__restrict__ pairs,
               const double *__restrict__ Y, double *__restrict__ val) {
  // Workgroups are dealt round-robin over the eight XCDs, each with an L2 of its own.  The blocks are ordered by row,
  // and the blocks of one row read the same station's records: give every XCD one contiguous eighth of the list, so
  // that a record fetched for one block of a row is still in that L2 for the next.
  __shared__ double lds_y[4][32 * 18];
  const int nwg = gridDim.x, per = (nwg + 7) >> 3;
  const int wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int u = wg * 4 + (threadIdx.x >> 6);
  if (wg >= nwg || u >= nu) return;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 2, qa = (lane >> 1) & 1, qb = lane & 1;
  const int q0 = cstart[u], q1 = cstart[u + 1];
  double acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = 0.0;
  // Sixteen pairs per step.  Their 32 records (144 B each) are fetched by the whole wave, nine lanes per record and
  // 16 bytes per lane -- seven records per load instruction, two cache lines each -- and handed to the lanes that
  // multiply them through a wave-private LDS buffer.  (A lane fetching its own half record takes five instructions of
  // which every one touches the lines of all sixteen pairs again: the kernel was bound by the address unit, not by bytes.)
  double *sy = lds_y[threadIdx.x >> 6];
  const int slot_rec = lane / 9, piece = lane - 9 * slot_rec;  // lane 63 idles
  // pair indices two steps ahead, records one step ahead: nothing in a step waits for a load issued in it
  auto pairs_of = [&](int q) { return (lane < 16 && q + lane < q1) ? pairs[q + lane] : make_int2(-1, -1); };
  int2 pr_next = pairs_of(q0), pr_next2 = pairs_of(q0 + 16);
  double2 fetched[5];
  auto fetch = [&](int2 pr) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int s = 7 * k + slot_rec;  // record slot: 0..15 the pairs' first records, 16..31 their second
      const int ia = __shfl(pr.x, s & 15, 64), ib = __shfl(pr.y, s & 15, 64);
      const int rec = s < 16 ? ia : ib;
      const bool on = lane < 63 && s < 32 && rec >= 0;
      fetched[k] = on ? *reinterpret_cast<const double2 *>(Y + (int64_t)BA_Y_STRIDE * rec + 2 * piece) : make_double2(0.0, 0.0);
    }
  };
  fetch(pr_next);
  for (int q = q0; q < q1; q += 16) {
    // hand over this step's records, request the next step's
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int s = 7 * k + slot_rec;
      if (lane < 63 && s < 32) *reinterpret_cast<double2 *>(sy + 18 * s + 2 * piece) = fetched[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (q + 16 < q1) {
      pr_next = pr_next2;
      fetch(pr_next);
      pr_next2 = pairs_of(q + 32);
    }
    // records of pairs beyond the block's last are zeros: they add nothing
    double av[3][3], bv[3][3];
    const double *pa = sy + 18 * g + 9 * qa, *pb = sy + 18 * (16 + g) + 9 * qb;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        av[i][j] = pa[3 * i + j];
        bv[i][j] = pb[3 * i + j];
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] += av[i][0] * bv[j][0] + av[i][1] * bv[j][1] + av[i][2] * bv[j][2];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double v = acc[i][j];
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[i][j] = v;
    }
  // the block and its mirror image leave as two runs of 288 contiguous bytes (16 bytes per lane) instead of 72
  // scattered 8-byte writes: assembled in the wave's LDS buffer by the four lanes that hold the quadrants
  const int2 d = dst[u];
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int r = 3 * qa + i, c = 3 * qb + j;
        sy[6 * r + c] = -acc[i][j];
        sy[36 + 6 * c + r] = -acc[i][j];  // the mirrored block (hi, lo)
      }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 36) {
    const int which = lane / 18, piece = lane - 18 * which;
    const int blk = which ? d.y : d.x;
    if (blk >= 0)
      *reinterpret_cast<double2 *>(val + 36 * (int64_t)blk + 2 * piece) = *reinterpret_cast<const double2 *>(sy + 36 * which + 2 * piece);
  }
}

// diagonal blocks: val[row i, first block] += Sdiag[i]
__global__ void k_bsr_add_diag(int S, const int *__restrict__ rowptr, const double *__restrict__ Sdiag, double *__restrict__ val) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= 36 * (int64_t)S) return;
  const int i = (int)(t / 36), k = (int)(t - 36 * (int64_t)i);
  val[36 * (int64_t)rowptr[i] + k] += Sdiag[t];
}

// dense copy of the explicit system for the Cholesky path: A is nr x nr, row-major with leading dimension ld
__global__ void k_bsr_to_dense(int S, int ncam, const int *__restrict__ rowptr, const int *__restrict__ col,
                               const double *__restrict__ val, const double *__restrict__ Ssc, const double *__restrict__ Scc,
                               double *__restrict__ A, int ld) {
  const int i = blockIdx.x;
  const int64_t nr = ld;
  if (i < S) {
    for (int b = rowptr[i]; b < rowptr[i + 1]; ++b) {
      const int j = col[b];
      for (int t = threadIdx.x; t < 36; t += blockDim.x) A[(6 * (int64_t)i + t / 6) * nr + 6 * j + t % 6] = val[36 * (int64_t)b + t];
    }
    for (int t = threadIdx.x; t < 6 * ncam; t += blockDim.x) {
      const int r = t / ncam, k = t - r * ncam;
      const double v = Ssc[(6 * (int64_t)i + r) * ncam + k];
      A[(6 * (int64_t)i + r) * nr + 6 * S + k] = v;
      A[(6 * (int64_t)S + k) * nr + 6 * i + r] = v;
    }
  } else {
    for (int t = threadIdx.x; t < ncam * ncam; t += blockDim.x)
      A[(6 * (int64_t)S + t / ncam) * nr + 6 * S + t % ncam] = Scc[t];
  }
}

// border column k from a probe result v (6S + ncam): Ssc[., k] and Scc[k, .]
__global__ void k_store_border_col(int S, int ncam, int k, const double *__restrict__ v, double *__restrict__ Ssc,
                                   double *__restrict__ Scc) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t ns = 6 * (int64_t)S;
  if (t < ns) Ssc[t * ncam + k] = v[t];
  else if (t < ns + ncam) Scc[(int64_t)k * ncam + (t - ns)] = v[t];
}

// ------------------------------------------------------------------------------------------
// explicit border for one camera with at most 8 free parameters
// ------------------------------------------------------------------------------------------
// The projected camera rows Jt = Jc - Jp V^-1 B^T (2 x nc per observation) come from the linearisation
// kernel; the corner sum Jc^T Jt from k_camera_corner.
#define NCB 8
// Station sweep over the chunks: out[chunk][0..6*NCB) = sum Js^T Jt (6 x nc)
#define NBORDER (6 * NCB)
__global__ void __launch_bounds__(256)
k_border_station(StationArgs a, const double *__restrict__ Jt) {
  __shared__ double lds[4 * NBORDER];
  const int ch = blockIdx.x;
  const int b0 = a.chunk_begin[ch], b1 = a.chunk_begin[ch + 1];
  double acc[NBORDER];
#pragma unroll
  for (int k = 0; k < NBORDER; ++k) acc[k] = 0.0;
  // the index of a thread's next observation is requested before the records of this one
  int e_next = (b0 + (int)threadIdx.x < b1) ? a.sta_list[b0 + threadIdx.x] : 0;
  for (int q = b0 + threadIdx.x; q < b1; q += blockDim.x) {
    const int e = e_next;
    if (q + (int)blockDim.x < b1) e_next = a.sta_list[q + blockDim.x];
    const double2 *rec = reinterpret_cast<const double2 *>(a.rec + (int64_t)BA_REC_STRIDE * e);
    double r0[6], r1[6];
    {
      const double2 j0 = rec[0], j1 = rec[1], j2 = rec[2], j3 = rec[3], j4 = rec[4], j5 = rec[5];
      r0[0] = j0.x; r0[1] = j0.y; r0[2] = j1.x; r0[3] = j1.y; r0[4] = j2.x; r0[5] = j2.y;
      r1[0] = j3.x; r1[1] = j3.y; r1[2] = j4.x; r1[3] = j4.y; r1[4] = j5.x; r1[5] = j5.y;
    }
    const double2 *jt = reinterpret_cast<const double2 *>(Jt + 2 * (int64_t)NCB * e);
    double t0[NCB], t1[NCB];
#pragma unroll
    for (int k = 0; k < NCB / 2; ++k) {
      const double2 x = jt[k], y = jt[NCB / 2 + k];
      t0[2 * k] = x.x; t0[2 * k + 1] = x.y;
      t1[2 * k] = y.x; t1[2 * k + 1] = y.y;
    }
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int k = 0; k < NCB; ++k) acc[p * NCB + k] += r0[p] * t0[k] + r1[p] * t1[k];
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  wave_sums_to<NBORDER>(acc, lds + w * NBORDER, lane);
  __syncthreads();
  if (threadIdx.x < NBORDER) {
    const int k = threadIdx.x;
    a.out[(int64_t)a.out_stride * ch + k] = lds[k] + lds[NBORDER + k] + lds[2 * NBORDER + k] + lds[3 * NBORDER + k];
  }
}

// fold the border chunks: Ssc[i][r][k] (row-major, ncam columns)
__global__ void __launch_bounds__(256)
k_border_fold(int S, int ncam, const int *__restrict__ sta_chunk0, const double *__restrict__ part, int stride,
              double *__restrict__ Ssc) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < 6 * (int64_t)S * ncam) {
    const int i = (int)(t / (6 * ncam));
    const int rem = (int)(t - (int64_t)i * 6 * ncam), r = rem / ncam, k = rem - r * ncam;
    double s = 0.0;
    for (int ch = sta_chunk0[i]; ch < sta_chunk0[i + 1]; ++ch) s += part[(int64_t)stride * ch + r * NCB + k];
    Ssc[t] = s;
  }
}

// Scc (ncam x ncam, symmetric, row-major) from the folded upper triangle plus prior + LM diagonal
__global__ void k_border_corner(int ncam, const double *__restrict__ corner, const double *__restrict__ camdiag,
                                double *__restrict__ Scc) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int idx = 0;
  for (int k = 0; k < NCB; ++k)
    for (int m = k; m < NCB; ++m) {
      if (k < ncam && m < ncam) {
        const double v = corner[idx] + (k == m ? camdiag[k] : 0.0);
        Scc[k * ncam + m] = v;
        Scc[m * ncam + k] = v;
      }
      ++idx;
    }
}

// ------------------------------------------------------------------------------------------
// explicit border, general case (several cameras, or more than NCB free parameters)
// ------------------------------------------------------------------------------------------
// Z_j = V_j^-1 B_j^T with B_j = sum_{e in j} Jc_e^T Jp_e over all ncam camera columns (an observation
// only touches the columns of its own camera).  One wave per point, one lane per column.
__global__ void __launch_bounds__(256)
k_border_Z(GenBorderArgs a) {
  const int jj = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (jj >= a.P) return;
  const int e0 = a.pt_start[jj], e1 = a.pt_start[jj + 1];
  const double *vi = a.Vinv + 6 * (int64_t)jj;
  const double Vi[6] = {vi[0], vi[1], vi[2], vi[3], vi[4], vi[5]};
  for (int k = lane; k < a.ncam; k += 64) {
    const int ck = a.col_cam[k], pos = a.col_pos[k];
    double b0 = 0.0, b1 = 0.0, b2 = 0.0;
    for (int e = e0; e < e1; ++e) {
      if (a.sta_cam[a.obs_sta[e]] != ck) continue;
      const double *jc = a.Jc + 2 * (int64_t)a.ncmax * e, *jp = a.Jp + 6 * (int64_t)e;
      const double c0 = jc[pos], c1 = jc[a.ncmax + pos];
      b0 += c0 * jp[0] + c1 * jp[3];
      b1 += c0 * jp[1] + c1 * jp[4];
      b2 += c0 * jp[2] + c1 * jp[5];
    }
    double *z = a.Zc + ((int64_t)jj * a.ncam + k) * 3;
    sym3_mul(Vi, b0, b1, b2, z[0], z[1], z[2]);
  }
}

// Station sweep: per chunk, column k of  sum Js^T Jt  (6 values) and of  sum Jc^T Jt  (the nc rows of the
// station's camera), Jt_e[:,k] = Jc_e[:,k] (own camera's columns only) - Jp_e Z_j[:,k].
// Thread t: column (t mod ncw) of the current tile of columns, observations t / ncw, t / ncw + nsub, ...
__global__ void __launch_bounds__(256)
k_border_general(GenBorderArgs a) {
  __shared__ double lds[256 * 7];
  const int ch = blockIdx.x, sta = a.chunk_sta[ch];
  const int b0 = a.chunk_begin[ch], b1 = a.chunk_begin[ch + 1];
  const int c = a.sta_cam[sta], nc = a.cam[c].nc, off = a.cam[c].off;
  for (int k0 = 0; k0 < a.ncam; k0 += 256) {
    const int ncw = min(256, a.ncam - k0), nsub = 256 / ncw;
    const int kk = threadIdx.x % ncw, sub = threadIdx.x / ncw, k = k0 + kk;
    const bool on = sub < nsub;
    const bool own = k >= off && k < off + nc;
    double as[6] = {0, 0, 0, 0, 0, 0}, ac[BA_NCAMPAR];
#pragma unroll
    for (int r = 0; r < BA_NCAMPAR; ++r) ac[r] = 0.0;
    if (on) {
      // observation index two steps ahead, its point index one step ahead: a step waits for its records only, not for
      // the two index loads that lead to them
      auto obs_at = [&](int q) { return q < b1 ? a.sta_list[q] : -1; };
      int e_cur = obs_at(b0 + sub), e_n1 = obs_at(b0 + sub + nsub), e_n2 = obs_at(b0 + sub + 2 * nsub);
      int jj_cur = e_cur >= 0 ? a.obs_pt[e_cur] : 0, jj_n1 = e_n1 >= 0 ? a.obs_pt[e_n1] : 0;
      for (int q = b0 + sub; q < b1; q += nsub) {
        const int e = e_cur, jj = jj_cur;
        e_cur = e_n1;
        jj_cur = jj_n1;
        e_n1 = e_n2;
        jj_n1 = e_n1 >= 0 ? a.obs_pt[e_n1] : 0;
        e_n2 = obs_at(q + 3 * nsub);
        const double *rec = a.rec + (int64_t)BA_REC_STRIDE * e, *jp = a.Jp + 6 * (int64_t)e;
        const double *jc = a.Jc + 2 * (int64_t)a.ncmax * e, *z = a.Zc + ((int64_t)jj * a.ncam + k) * 3;
        double t0 = -(jp[0] * z[0] + jp[1] * z[1] + jp[2] * z[2]), t1 = -(jp[3] * z[0] + jp[4] * z[1] + jp[5] * z[2]);
        if (own) {
          t0 += jc[k - off];
          t1 += jc[a.ncmax + k - off];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) as[r] += rec[r] * t0 + rec[6 + r] * t1;
#pragma unroll
        for (int r = 0; r < BA_NCAMPAR; ++r)
          if (r < nc) ac[r] += jc[r] * t0 + jc[a.ncmax + r] * t1;
      }
    }
    // fold the nsub partial sums of each column through LDS, 7 values at a time
    for (int pass = 0; pass < 4; ++pass) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 7; ++r) {
        const int idx = 7 * pass + r;  // 0..5 station rows, 6..24 camera rows
        double v = 0.0;
        if (idx < 6) v = as[idx];
        else if (idx - 6 < BA_NCAMPAR) v = ac[idx - 6];
        lds[threadIdx.x * 7 + r] = on ? v : 0.0;
      }
      __syncthreads();
      if (sub == 0) {
#pragma unroll
        for (int r = 0; r < 7; ++r) {
          const int idx = 7 * pass + r;
          double t = 0.0;
          for (int m = 0; m < nsub; ++m) t += lds[(m * ncw + kk) * 7 + r];
          if (idx < 6) a.out_sc[((int64_t)ch * 6 + idx) * a.ncam + k] = t;
          else if (idx - 6 < a.ncmax) a.out_cc[((int64_t)ch * a.ncmax + (idx - 6)) * a.ncam + k] = (idx - 6 < nc) ? t : 0.0;
        }
      }
    }
    __syncthreads();
  }
}

// Ssc[i][r][k] = sum over the chunks of station i (fixed order)
__global__ void k_border_general_fold_sc(int S, int ncam, const int *__restrict__ sta_chunk0, const double *__restrict__ part,
                                         double *__restrict__ Ssc) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= 6 * (int64_t)S * ncam) return;
  const int i = (int)(t / (6 * ncam));
  const int64_t rem = t - (int64_t)i * 6 * ncam;
  double s = 0.0;
  for (int ch = sta_chunk0[i]; ch < sta_chunk0[i + 1]; ++ch) s += part[(int64_t)ch * 6 * ncam + rem];
  Ssc[t] = s;
}

// raw corner row (off_c + r), column k = sum over the chunks whose station carries camera c; one workgroup per row
__global__ void __launch_bounds__(256)
k_border_general_fold_cc(int n_chunks, int ncam, int ncmax, const int *__restrict__ chunk_sta, const int *__restrict__ sta_cam,
                         const int *__restrict__ col_cam, const int *__restrict__ col_pos, const double *__restrict__ part,
                         double *__restrict__ corner) {
  const int row = blockIdx.x, c = col_cam[row], r = col_pos[row];
  for (int k = threadIdx.x; k < ncam; k += blockDim.x) {
    double s = 0.0;
    for (int ch = 0; ch < n_chunks; ++ch)
      if (sta_cam[chunk_sta[ch]] == c) s += part[((int64_t)ch * ncmax + r) * ncam + k];
    corner[(int64_t)row * ncam + k] = s;
  }
}

// Scc = symmetrised raw corner + prior and LM diagonal
__global__ void k_border_general_corner(int ncam, const double *__restrict__ corner, const double *__restrict__ camdiag,
                                        double *__restrict__ Scc) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ncam * ncam) return;
  const int r = t / ncam, k = t % ncam;
  Scc[t] = 0.5 * (corner[(int64_t)r * ncam + k] + corner[(int64_t)k * ncam + r]) + (r == k ? camdiag[r] : 0.0);
}

// ------------------------------------------------------------------------------------------
// PCG on the explicit system
// ------------------------------------------------------------------------------------------
// scal: [0],[1] r.z (double-buffered by iteration parity), [2] initial r.z, [3] iterations done,
//       [4] last p.Sp, [6] target sqrt(r.z), [8],[9] done flag (1 converged, 2 breakdown), double-buffered
//       the same way: iteration `it` reads slot it&1 and publishes the next state in the other slot,
//       so no workgroup ever sees a scalar change under it.
// q = S p for station rows (one wave per row) with per-workgroup partials of p.q and of the
// camera rows Ssc^T p.
#define SPMV_WAVES 16  // per workgroup: one partial sum per workgroup, and every workgroup of the next kernel folds them all
#define SPMV_NT 7  // triples of blocks in flight per wave: a row of up to 21 blocks is one pass (24 would spill registers)
// The kernel is a chain of dependent loads per row (row pointer -> columns -> blocks and entries of p), and a wave has
// two or three rows: the row pointers of the next two rows and the columns of the next row are requested together with
// the blocks of the current one, so a row costs one round trip after the first (two for the wave's start), and the
// camera columns of the row travel with its blocks.  The grid is sized so that every wave gets the same number of rows
// (cg2_blocks_spmv).
__global__ void __launch_bounds__(64 * SPMV_WAVES, 4)
k_bsr_spmv(BsrCgArgs a, int parity) {
  __shared__ double lds_pq[SPMV_WAVES];
  __shared__ double lds_qc[SPMV_WAVES][CG2_MAX_CAM + 1];
  pcg_mark(a.trace, 0, 0, blockIdx.x);
  const double done = a.scal[8 + parity];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row_begin = a.rows_sharded ? a.row0 : 0, row_end = a.rows_sharded ? a.row_end : a.S;
  const int nw = blockDim.x >> 6;  // 16 waves, or 8 on small systems (spmv_waves)
  const int stride = gridDim.x * nw;
  int i = row_begin + blockIdx.x * nw + w;
  int rb = 0, re = 0, rbn = 0, ren = 0;
  if (i < row_end) {
    rb = a.rowptr[i];
    re = a.rowptr[i + 1];
  }
  if (i + stride < row_end) {
    rbn = a.rowptr[i + stride];
    ren = a.rowptr[i + stride + 1];
  }
  int mycol = lane < re - rb ? a.col[rb + lane] : 0;  // the row's first 64 columns
  // camera columns 0..7 of a row: lane (r, k) = (lane >> 3, lane & 7) holds Ssc[6 i + r][k]
  const int cr = lane >> 3, ck = lane & 7;
  const bool cam_lane = cr < 6 && ck < a.ncam;
  const double *pcam = a.p + 6 * (int64_t)a.S;
  const double pk = cam_lane ? pcam[ck] : 0.0;
  // the corner of the camera rows, S_cc p_c: row threadIdx.x by the first workgroup (of the rank that holds the first
  // rows, when the rows are shared out), added to its partial of the camera rows
  double corner = 0.0;
  if (blockIdx.x == 0 && (!a.rows_sharded || a.row0 == 0) && (int)threadIdx.x < a.ncam)
    for (int m = 0; m < a.ncam; ++m) corner += a.Scc[threadIdx.x * a.ncam + m] * pcam[m];
  if (done != 0.0) return;  // waits for the flag alone
  pcg_mark(a.trace, 0, 1, blockIdx.x);
  double pq = 0.0;
  double tacc = 0.0;                // camera-row partial of column ck from block row cr (columns 0..7)
  double tcam0 = 0.0, tcam1 = 0.0;  // camera-row partials of columns lane (from 8 on) and lane + 64
  // Lanes 0..53 read a row's blocks (288 B each, contiguous) as one stream, 16 B per lane: lane l holds piece l % 18
  // of block l / 18 of each triple, i.e. always the same block row (l % 18) / 3 and column pair 2 * (l % 3).
  const int tri = lane / 18, piece = lane - 18 * tri;  // tri == 3 for the idle lanes 54..63
  const int cpair = 2 * (piece % 3);
  for (; i < row_end; i += stride) {
    const int inn = i + 2 * stride;
    const int mycoln = lane < ren - rbn ? a.col[rbn + lane] : 0;
    int rbnn = 0, renn = 0;
    if (inn < row_end) {
      rbnn = a.rowptr[inn];
      renn = a.rowptr[inn + 1];
    }
    const double cx = cam_lane ? a.Ssc[(6 * (int64_t)i + cr) * a.ncam + ck] : 0.0;
    const double pr = cr < 6 ? a.p[6 * (int64_t)i + cr] : 0.0;
    double s = 0.0;
    for (int base = rb; base < re; base += 3 * SPMV_NT) {
      const int nb = min(3 * SPMV_NT, re - base);
      int off = base - rb, cols = mycol;
      if (off + 3 * SPMV_NT > 64) {  // beyond the columns held in registers (uniform over the wave)
        cols = lane < nb ? a.col[base + lane] : 0;
        off = 0;
      }
      double2 bv[SPMV_NT], pv[SPMV_NT];
#pragma unroll
      for (int u = 0; u < SPMV_NT; ++u) {
        const int bi = 3 * u + tri;
        const int cb = __shfl(cols, off + (bi < 3 * SPMV_NT ? bi : 0), 64);
        const bool on = tri < 3 && bi < nb;
        // 32-bit byte offsets from the array bases (the block values stay below 4 GB: checked where the blocks are
        // allocated): one address register per load instead of two
        const unsigned vo = 288u * (unsigned)(base + bi) + 16u * (unsigned)piece, po = 48u * (unsigned)cb + 8u * (unsigned)cpair;
        bv[u] = on ? *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.val) + vo) : make_double2(0.0, 0.0);
        pv[u] = on ? *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.p) + po) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < SPMV_NT; ++u) s += bv[u].x * pv[u].x + bv[u].y * pv[u].y;
    }
    // block row r: pieces 3r..3r+2 of the three triples -> lanes 3r + {0,1,2} + {0,18,36}
    s += __shfl_down(s, 1, 64) + __shfl_down(s, 2, 64);   // valid where piece % 3 == 0
    s += __shfl_down(s, 18, 64) + __shfl_down(s, 36, 64);  // valid for lanes 0,3,..,15
    s = __shfl(s, 3 * (lane < 6 ? lane : 0), 64);          // lanes 0..5: rows 0..5
    // camera columns 0..7: row sums over the 8 lanes of a block row, column sums kept per lane until the end
    double cd = cx * pk;
    cd += __shfl_xor(cd, 1, 64);
    cd += __shfl_xor(cd, 2, 64);
    cd += __shfl_xor(cd, 4, 64);
    cd = __shfl(cd, 8 * (lane < 6 ? lane : 0), 64);
    const double pown = __shfl(pr, 8 * (lane < 6 ? lane : 0), 64);
    tacc += cx * pr;
    if (lane < 6) {
      s += cd;
      const double *row = a.Ssc + (6 * (int64_t)i + lane) * a.ncam;
      for (int k = 8; k < a.ncam; ++k) s += row[k] * pcam[k];
      a.q[6 * (int64_t)i + lane] = s;
      pq += s * pown;
    }
    // camera columns from 8 on: t_k += sum_r Ssc[i][r][k] p_i[r]
    if (lane >= 8 && lane < a.ncam) {
      const double *pi = a.p + 6 * (int64_t)i;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) tcam0 += a.Ssc[(6 * (int64_t)i + rr) * a.ncam + lane] * pi[rr];
    }
    if (lane + 64 < a.ncam) {
      const double *pi = a.p + 6 * (int64_t)i;
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) tcam1 += a.Ssc[(6 * (int64_t)i + rr) * a.ncam + lane + 64] * pi[rr];
    }
    rb = rbn;
    re = ren;
    rbn = rbnn;
    ren = renn;
    mycol = mycoln;
  }
  pcg_mark(a.trace, 0, 2, blockIdx.x);
  tacc += __shfl_xor(tacc, 8, 64);
  tacc += __shfl_xor(tacc, 16, 64);
  tacc += __shfl_xor(tacc, 32, 64);  // lanes 0..7: columns 0..7 (block rows 6 and 7 do not exist: zeros)
  if (lane < 8) tcam0 = tacc;
  pq = wave_sum(pq);
  if (lane == 0) lds_pq[w] = pq;
  if (lane < a.ncam) lds_qc[w][lane] = tcam0;
  if (lane + 64 < a.ncam) lds_qc[w][lane + 64] = tcam1;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int m = 0; m < nw; ++m) t += lds_pq[m];
    a.part_pq[blockIdx.x] = t;
  }
  for (int k = threadIdx.x; k < a.ncam; k += blockDim.x) {
    double t = 0.0;
    for (int m = 0; m < nw; ++m) t += lds_qc[m][k];
    a.part_qc[(int64_t)blockIdx.x * a.ncam + k] = t + corner;  // k == threadIdx.x: ncam is below the workgroup size
  }
  pcg_mark(a.trace, 0, 3, blockIdx.x);
}

// x = 0, r = b, z = M^-1 r, p = z; per-workgroup partials of r.z
__global__ void __launch_bounds__(256)
k_cg2_init(BsrCgArgs a) {
  __shared__ double lds[8];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double dot = 0.0;
  if (i < a.S) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      a.x[6 * (int64_t)i + k] = 0.0;
      a.r[6 * (int64_t)i + k] = a.b[6 * (int64_t)i + k];
    }
    dot = station_precond(a.Minv, a.r, a.z, i);
#pragma unroll
    for (int k = 0; k < 6; ++k) a.p[6 * (int64_t)i + k] = a.z[6 * (int64_t)i + k];
  }
  if (a.G) {
    double ri[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (i < a.S)
#pragma unroll
      for (int k = 0; k < 6; ++k) ri[k] = a.b[6 * (int64_t)i + k];
    double gr[DEFL_NC];
    defl_products(a, ri, i, gr);
    defl_partials(a, gr, blockIdx.x);
  }
  if (blockIdx.x == 0 && a.ncam > 0) {
    // camera entries: x = 0, r = b and (without the coarse space) z = Scc^-1 r with the explicit inverse,
    // one thread per row.  With the coarse space the camera part of z needs G^T r_s: the last kernel does it.
    const int64_t o = 6 * (int64_t)a.S;
    for (int k = threadIdx.x; k < a.ncam; k += blockDim.x) {
      a.x[o + k] = 0.0;
      a.r[o + k] = a.b[o + k];
    }
    if (!a.G)
      for (int k = threadIdx.x; k < a.ncam; k += blockDim.x) {
        double z = 0.0;
        for (int m = 0; m < a.ncam; ++m) z += a.Lcc[k * a.ncam + m] * a.b[o + m];
        a.z[o + k] = z;
        a.p[o + k] = z;
        dot += a.b[o + k] * z;
      }
  }
  const double t = block_sum(dot, lds);
  if (threadIdx.x == 0) a.part_rz[blockIdx.x] = t;
}

// sums the initial r.z partials; sets the stopping target
__global__ void __launch_bounds__(256)
k_cg2_init_finish(BsrCgArgs a, int nblk, double rtol) {
  __shared__ double lds[8];
  double s = 0.0;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) s += a.part_rz[k];
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0) {
    a.scal[0] = t;
    a.scal[1] = t;
    a.scal[2] = t;
    a.scal[3] = 0.0;
    a.scal[4] = 0.0;
    a.scal[5] = 0.0;
    a.scal[6] = rtol * sqrt(fmax(t, 0.0));
    a.scal[8] = a.scal[9] = (t > 0.0) ? 0.0 : 1.0;
  }
}

// alpha = r.z / p.Sp; x += alpha p; r -= alpha q; z = M^-1 r; partials of the new r.z
__global__ void __launch_bounds__(256)
k_cg2_update1(BsrCgArgs a, int parity, int nblk_spmv) {
  __shared__ Cg2Shared sh;
  const double done = a.scal[8 + parity], rz = a.scal[parity];
  FoldLoads f;
  VecLoads v;
  fold_issue(a, nblk_spmv, 0, f);
  if (done != 0.0) return;
  vec_issue(a, a.r, blockIdx.x, gridDim.x, v);
  const double alpha = cg2_alpha(a, rz, nblk_spmv, f, sh);
  vec_finish(a, alpha, a.r, blockIdx.x, gridDim.x, v, sh);
}

// out[0] = p^T S p from the partials a k_bsr_spmv launch has left (the fold of the PCG's second kernel on its own)
__global__ void __launch_bounds__(256)
k_quadratic_form(BsrCgArgs a, int nblk_spmv, double *__restrict__ out) {
  __shared__ Cg2Shared sh;
  FoldLoads f;
  fold_issue(a, nblk_spmv, 0, f);
  (void)cg2_alpha(a, 1.0, nblk_spmv, f, sh);
  if (threadIdx.x == 0) out[0] = sh.pq;
}

// beta = r.z(new) / r.z(old); p = z + beta p; stopping test for the next iteration
__global__ void __launch_bounds__(256)
k_cg2_update2(BsrCgArgs a, int parity, int nblk) {
  __shared__ double lds[8];
  __shared__ double rzn_s;
  const double done = a.scal[8 + parity];
  if (done != 0.0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scal[8 + (parity ^ 1)] = done;  // keep both slots set
    return;
  }
  double s = 0.0;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) s += a.part_rz[k];
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0) rzn_s = t;
  __syncthreads();
  const double rzn = rzn_s, rz = a.scal[parity];
  const double beta = rzn / rz;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.S) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a.p[6 * (int64_t)i + k] = a.z[6 * (int64_t)i + k] + beta * a.p[6 * (int64_t)i + k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t o = 6 * (int64_t)a.S;
    for (int k = 0; k < a.ncam; ++k) a.p[o + k] = a.z[o + k] + beta * a.p[o + k];
    a.scal[parity ^ 1] = rzn;
    a.scal[3] += 1.0;
    const double pq = a.scal[4];
    double nd = 0.0;
    if (!(pq > 0.0) || !(rzn == rzn)) nd = 2.0;
    else if (sqrt(fmax(rzn, 0.0)) <= a.scal[6]) nd = 1.0;
    a.scal[8 + (parity ^ 1)] = nd;
  }
}

// ------------------------------------------------------------------------------------------
// camera coarse space: set-up
// ------------------------------------------------------------------------------------------
__global__ void k_defl_get_column(int64_t n, int ncam, int k, const double *__restrict__ M, double *__restrict__ v) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) v[t] = M[t * ncam + k];
}
__global__ void k_defl_set_column(int64_t n, int ncam, int k, const double *__restrict__ v, double *__restrict__ M,
                                  double *__restrict__ MT) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) {
    M[t * ncam + k] = v[t];
    MT[k * n + t] = v[t];
  }
}

// SG = S_ss G for the ncam <= DEFL_NC columns of G at once (row-major [6S x ncam]); one wave per block row
__global__ void __launch_bounds__(256)
k_defl_spmm(BsrCgArgs a, const double *__restrict__ G, double *__restrict__ SG) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= a.S) return;
  const int g = lane >> 3, r = lane & 7;
  double acc[DEFL_NC];
#pragma unroll
  for (int k = 0; k < DEFL_NC; ++k) acc[k] = 0.0;
  if (r < 6)
    for (int b = a.rowptr[i] + g; b < a.rowptr[i + 1]; b += 8) {
      const double *B = a.val + 36 * (int64_t)b + 6 * r, *X = G + 6 * (int64_t)a.col[b] * a.ncam;
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const double bv = B[m];
#pragma unroll
        for (int k = 0; k < DEFL_NC; ++k)
          if (k < a.ncam) acc[k] += bv * X[m * a.ncam + k];
      }
    }
#pragma unroll
  for (int k = 0; k < DEFL_NC; ++k) {
    acc[k] += __shfl_xor(acc[k], 8, 64);
    acc[k] += __shfl_xor(acc[k], 16, 64);
    acc[k] += __shfl_xor(acc[k], 32, 64);
  }
  if (lane < 6)
#pragma unroll
    for (int k = 0; k < DEFL_NC; ++k)
      if (k < a.ncam) SG[(6 * (int64_t)i + lane) * a.ncam + k] = acc[k];
}

// per-workgroup partials of G^T SG - Ssc^T G - G^T Ssc (ncam x ncam, stored DEFL_NC x DEFL_NC)
__global__ void __launch_bounds__(256)
k_defl_gram(int64_t n, int ncam, const double *__restrict__ G, const double *__restrict__ SG, const double *__restrict__ Ssc,
            double *__restrict__ part) {
  __shared__ double lds[4][DEFL_NC * DEFL_NC];
  double acc[DEFL_NC * DEFL_NC];
#pragma unroll
  for (int k = 0; k < DEFL_NC * DEFL_NC; ++k) acc[k] = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    double g[DEFL_NC], sg[DEFL_NC], sc[DEFL_NC];
#pragma unroll
    for (int k = 0; k < DEFL_NC; ++k) {
      g[k] = (k < ncam) ? G[t * ncam + k] : 0.0;
      sg[k] = (k < ncam) ? SG[t * ncam + k] : 0.0;
      sc[k] = (k < ncam) ? Ssc[t * ncam + k] : 0.0;
    }
#pragma unroll
    for (int p = 0; p < DEFL_NC; ++p)
#pragma unroll
      for (int q = 0; q < DEFL_NC; ++q) acc[p * DEFL_NC + q] += g[p] * sg[q] - sc[p] * g[q] - g[p] * sc[q];
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < DEFL_NC * DEFL_NC; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) lds[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < DEFL_NC * DEFL_NC)
    part[(int64_t)blockIdx.x * (DEFL_NC * DEFL_NC) + threadIdx.x] =
        (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
}

// E = Scc + folded partials, symmetrised; copied to Lcc for the factorisation
__global__ void __launch_bounds__(64)
k_defl_corner(int ncam, int nparts, const double *__restrict__ part, const double *__restrict__ Scc, double *__restrict__ E,
              double *__restrict__ Lcc) {
  __shared__ double e[DEFL_NC * DEFL_NC];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int b = 0; b < nparts; ++b) s += part[(int64_t)b * (DEFL_NC * DEFL_NC) + t];
  e[t] = s;
  __syncthreads();
  const int p = t / DEFL_NC, q = t % DEFL_NC;
  if (p < ncam && q < ncam) {
    const double v = Scc[p * ncam + q] + 0.5 * (e[p * DEFL_NC + q] + e[q * DEFL_NC + p]);
    E[p * ncam + q] = v;
    Lcc[p * ncam + q] = v;
  }
}

// ------------------------------------------------------------------------------------------
static inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

void launch_schur_blocks(hipStream_t s, int nu, const int *cstart, const int2 *dst, const int2 *pairs, const double *Y, double *val) {
  if (nu > 0) hipLaunchKernelGGL(k_schur_blocks, dim3((nblk(nu, 4) + 7) / 8 * 8), dim3(256), 0, s, nu, cstart, dst, pairs, Y, val);
}
void launch_bsr_add_diag(hipStream_t s, int S, const int *rowptr, const double *Sdiag, double *val) {
  if (S > 0) hipLaunchKernelGGL(k_bsr_add_diag, dim3(nblk(36 * (int64_t)S, 256)), dim3(256), 0, s, S, rowptr, Sdiag, val);
}
void launch_bsr_to_dense(hipStream_t s, int S, int ncam, const int *rowptr, const int *col, const double *val, const double *Ssc,
                         const double *Scc, double *A, int ld) {
  hipLaunchKernelGGL(k_bsr_to_dense, dim3(S + 1), dim3(64), 0, s, S, ncam, rowptr, col, val, Ssc, Scc, A, ld);
}
void launch_store_border_col(hipStream_t s, int S, int ncam, int k, const double *v, double *Ssc, double *Scc) {
  hipLaunchKernelGGL(k_store_border_col, dim3(nblk(6 * (int64_t)S + ncam, 256)), dim3(256), 0, s, S, ncam, k, v, Ssc, Scc);
}
int border_ncb() { return NCB; }
int border_nacc() { return NBORDER; }
void launch_border_station(hipStream_t s, const StationArgs &sa, const double *Jt) {
  if (sa.n_chunks > 0) hipLaunchKernelGGL(k_border_station, dim3(sa.n_chunks), dim3(256), 0, s, sa, Jt);
}
void launch_border_fold(hipStream_t s, int S, int ncam, int n_chunks, const int *sta_chunk0, const double *part, int stride,
                        double *Ssc) {
  (void)n_chunks;
  const int nb = nblk(6 * (int64_t)S * ncam, 256);
  if (nb > 0) hipLaunchKernelGGL(k_border_fold, dim3(nb), dim3(256), 0, s, S, ncam, sta_chunk0, part, stride, Ssc);
}
void launch_border_general(hipStream_t s, const GenBorderArgs &a) {
  if (a.P > 0) hipLaunchKernelGGL(k_border_Z, dim3(nblk(a.P, 4)), dim3(256), 0, s, a);
  if (a.n_chunks > 0) hipLaunchKernelGGL(k_border_general, dim3(a.n_chunks), dim3(256), 0, s, a);
  const int nb = nblk(6 * (int64_t)a.S * a.ncam, 256);
  if (nb > 0) hipLaunchKernelGGL(k_border_general_fold_sc, dim3(nb), dim3(256), 0, s, a.S, a.ncam, a.sta_chunk0, a.out_sc, a.Ssc);
  hipLaunchKernelGGL(k_border_general_fold_cc, dim3(a.ncam), dim3(256), 0, s, a.n_chunks, a.ncam, a.ncmax, a.chunk_sta, a.sta_cam,
                     a.col_cam, a.col_pos, a.out_cc, a.corner);
}
void launch_border_general_corner(hipStream_t s, int ncam, const double *corner, const double *camdiag, double *Scc) {
  hipLaunchKernelGGL(k_border_general_corner, dim3(nblk((int64_t)ncam * ncam, 256)), dim3(256), 0, s, ncam, corner, camdiag, Scc);
}
void launch_border_corner(hipStream_t s, int ncam, const double *corner, const double *camdiag, double *Scc) {
  hipLaunchKernelGGL(k_border_corner, dim3(1), dim3(1), 0, s, ncam, corner, camdiag, Scc);
}
// Waves per workgroup of the product kernel: 16 (one partial sum per workgroup, and every workgroup of the next kernel
// folds them all), 8 on systems so small that workgroups of 16 would leave most CUs without one.
static int spmv_waves(int S) { return nblk(S, SPMV_WAVES / 2) <= CG2_MAX_SPMV_BLOCKS ? SPMV_WAVES / 2 : SPMV_WAVES; }
int cg2_blocks_spmv(int S) {
  // one resident round of workgroups; also bounds the partials the vector kernel folds
  const int nw = spmv_waves(S);
  const int nb = nblk(S, nw);
  if (nw < SPMV_WAVES) return nb;  // at most CG2_MAX_SPMV_BLOCKS by the choice of nw, all resident
  // as many workgroups as the device holds at once (one of sixteen waves per CU on MI355X: 256; more would wait for a
  // second round), at most the partials the fold takes
  // (looked up once per device: a process may hold problems on several)
  static int caps[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!caps[dev]) {
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bsr_spmv, 64 * SPMV_WAVES, 0) == hipSuccess && cus > 0 && per_cu > 0)
      caps[dev] = std::min((int)CG2_MAX_SPMV_BLOCKS, cus * per_cu);
    else
      caps[dev] = 256;
    (void)hipGetLastError();
  }
  const int cap = caps[dev];
  if (nb <= cap) return nb;
  // the same number of rows for every wave: the kernel ends when the waves with the most rows do
  const int rows_per_wave = nblk(S, cap * SPMV_WAVES);
  return nblk(S, rows_per_wave * SPMV_WAVES);
}
int cg2_blocks_vec(int S) { return nblk(S, 256) > 0 ? nblk(S, 256) : 1; }
int cg2_blocks_vec8(int S) { return nblk(S, 64) > 0 ? nblk(S, 64) : 1; }
void launch_cg2_init(hipStream_t s, const BsrCgArgs &a, double rtol) {
  const int nb = cg2_blocks_vec(a.S);
  hipLaunchKernelGGL(k_cg2_init, dim3(nb), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_cg2_init_finish, dim3(1), dim3(256), 0, s, a, nb, rtol);
}
// E <- (L L^T)^-1, column c by thread c (the factor is DEFL_NC x DEFL_NC at most)
__global__ void __launch_bounds__(64)
k_defl_invert(int n, const double *__restrict__ L, double *__restrict__ Einv) {
  const int c = threadIdx.x;
  if (c >= n) return;
  double z[DEFL_NC];
  for (int i = 0; i < n; ++i) {
    double s = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * z[k];
    z[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * z[k];
    z[i] = s / L[i * n + i];
  }
  for (int i = 0; i < n; ++i) Einv[i * n + c] = z[i];
}

int defl_max_cam() { return DEFL_NC; }
int cg2_max_cam() { return CG2_MAX_CAM; }
void launch_defl_get_column(hipStream_t s, int S, int ncam, int k, const double *M, double *v) {
  const int64_t n = 6 * (int64_t)S;
  if (n > 0) hipLaunchKernelGGL(k_defl_get_column, dim3(nblk(n, 256)), dim3(256), 0, s, n, ncam, k, M, v);
}
void launch_defl_set_column(hipStream_t s, int S, int ncam, int k, const double *v, double *M, double *MT) {
  const int64_t n = 6 * (int64_t)S;
  if (n > 0) hipLaunchKernelGGL(k_defl_set_column, dim3(nblk(n, 256)), dim3(256), 0, s, n, ncam, k, v, M, MT);
}
int defl_gram_blocks(int S) {
  const int b = nblk(6 * (int64_t)S, 256);
  return b < 1 ? 1 : (b > 256 ? 256 : b);
}
void launch_defl_corner(hipStream_t s, const BsrCgArgs &a, const double *G, double *SG, double *part, double *E, double *Lcc, int *info) {
  const int nb = defl_gram_blocks(a.S);
  hipLaunchKernelGGL(k_defl_spmm, dim3(nblk(a.S, 4)), dim3(256), 0, s, a, G, SG);
  hipLaunchKernelGGL(k_defl_gram, dim3(nb), dim3(256), 0, s, 6 * (int64_t)a.S, a.ncam, G, SG, a.Ssc, part);
  hipLaunchKernelGGL(k_defl_corner, dim3(1), dim3(64), 0, s, a.ncam, nb, part, a.Scc, E, Lcc);
  launch_cholesky(s, Lcc, a.ncam, info);
  hipLaunchKernelGGL(k_defl_invert, dim3(1), dim3(64), 0, s, a.ncam, Lcc, E);
}
void launch_cg2_start(hipStream_t s, const BsrCgArgs &a) {
  hipLaunchKernelGGL(k_cg2_init, dim3(cg2_blocks_vec(a.S)), dim3(256), 0, s, a);
}
void launch_cg2_spmv_update1(hipStream_t s, const BsrCgArgs &a, int parity) {
  const int nbs = cg2_blocks_spmv(a.S), nbv = cg2_blocks_vec(a.S);
  if (nbs > 0) hipLaunchKernelGGL(k_bsr_spmv, dim3(nbs), dim3(64 * spmv_waves(a.S)), 0, s, a, parity);
  hipLaunchKernelGGL(k_cg2_update1, dim3(nbv), dim3(256), 0, s, a, parity, nbs);
}
void launch_bsr_spmv(hipStream_t s, const BsrCgArgs &a, int parity) {
  const int nbs = cg2_blocks_spmv(a.S);
  if (nbs > 0) hipLaunchKernelGGL(k_bsr_spmv, dim3(nbs), dim3(64 * spmv_waves(a.S)), 0, s, a, parity);
}
// out[0] = a.p^T S a.p (a.q is overwritten; a.scal must point to zeros: the stopping flag of a finished solve would
// make the product return at once)
void launch_quadratic_form(hipStream_t s, const BsrCgArgs &a, double *out) {
  const int nbs = cg2_blocks_spmv(a.S);
  if (nbs > 0) hipLaunchKernelGGL(k_bsr_spmv, dim3(nbs), dim3(64 * spmv_waves(a.S)), 0, s, a, 0);
  hipLaunchKernelGGL(k_quadratic_form, dim3(1), dim3(256), 0, s, a, nbs, out);
}
void launch_cg2_iteration(hipStream_t s, const BsrCgArgs &a, int parity) {
  const int nbs = cg2_blocks_spmv(a.S), nbv = cg2_blocks_vec(a.S);
  if (nbs > 0) hipLaunchKernelGGL(k_bsr_spmv, dim3(nbs), dim3(64 * spmv_waves(a.S)), 0, s, a, parity);
  hipLaunchKernelGGL(k_cg2_update1, dim3(nbv), dim3(256), 0, s, a, parity, nbs);
  hipLaunchKernelGGL(k_cg2_update2, dim3(nbv), dim3(256), 0, s, a, parity, nbv);
}

}  // namespace ba
