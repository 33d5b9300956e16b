// ba_kernels.hip -- fp64 HIP kernels (gfx950) of the Levenberg-Marquardt inner loop.
//
// Two sweeps carry all per-observation work:
//   point sweep   : one lane per object point, walking that point's observations, which are
//                   contiguous because observations are stored sorted by point.  Everything
//                   that is summed per point (the 3x3 block V, its gradient, W^T p) stays in
//                   registers; no cross-lane traffic, no atomics.
//   station sweep : one 256-thread workgroup per chunk of one station's observations (taken
//                   through the station-sorted index list); per-lane partial sums are folded by
//                   wave shuffles and an LDS stage, then written once per chunk.  A second tiny
//                   kernel adds the chunks of a station in a fixed order, so results are
//                   bitwise reproducible and no global atomics are used.
// The reduced (camera-side) operator is never formed in the PCG path: S p is evaluated as
//   Jc^T (I - Jp V^-1 Jp^T) Jc p   through one point sweep and one station sweep,
// with the few camera (IOP/AP) columns kept as an explicit dense border.
#include "ba_kernels.h"

#include "ba_device.h"

namespace ba {

// ------------------------------------------------------------------------------------------
// station records
// ------------------------------------------------------------------------------------------
// rotation of src/main.cpp:1325-1335 evaluated once per station instead of once per observation
__global__ void k_station_prep(int S, const double *__restrict__ eop, const double *__restrict__ scale_sta,
                               const int *__restrict__ sta_cam, double *__restrict__ sta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const double *e = eop + 6 * (int64_t)i;
  const double so = sin(e[0]), co = cos(e[0]), sp = sin(e[1]), cp = cos(e[1]), sk = sin(e[2]), ck = cos(e[2]);
  double *o = sta + (int64_t)BA_STA_STRIDE * i;
  o[0] = cp * ck;  o[1] = co * sk + so * sp * ck;  o[2] = so * sk - co * sp * ck;
  o[3] = -cp * sk; o[4] = co * ck - so * sp * sk;  o[5] = so * ck + co * sp * sk;
  o[6] = sp;       o[7] = -so * cp;                o[8] = co * cp;
  o[9] = e[3]; o[10] = e[4]; o[11] = e[5];
  o[12] = sk; o[13] = ck;
#pragma unroll
  for (int k = 0; k < 6; ++k) o[14 + k] = scale_sta ? scale_sta[6 * (int64_t)i + k] : 1.0;  // null: unscaled columns
  o[20] = __hiloint2double(0, sta_cam[i]);
  o[21] = o[22] = o[23] = 0.0;
}

// ------------------------------------------------------------------------------------------
// linearisation: residuals, Jacobians, point elimination
// ------------------------------------------------------------------------------------------
// Two kernels.  k_point_elim (one lane per object point) walks the point's observations and only
// accumulates what the elimination needs: V, its gradient and, in the single-camera case, the
// coupling B = sum Jc^T Jp; it writes per-point records only.  k_obs_rows (one lane per
// observation, in point-sorted order) evaluates the observation again and writes every
// per-observation record exactly once; a wave's 64 records of one kind are contiguous in memory,
// so they are transposed through LDS and leave as whole cache lines.  Recomputing the projection
// is cheaper than writing the rows, reading them back and writing them a second time.
//
// CAM1 (one camera with at most LIN_NCB free parameters): k_obs_rows also writes the projected
// camera rows Jt = Jc - Jp V^-1 B^T and per-workgroup partial sums of Jc^T r (gradient) and
// Jc^T rhat (reduced right-hand side), so that no further sweep is needed for those.
#define LIN_NCB 8
#define LIN_NCORNER (LIN_NCB * (LIN_NCB + 1) / 2)
#define LIN_NPART (2 * LIN_NCB + 48)  // per-workgroup partials of k_obs_rows: Jc^T r | Jc^T rhat | upper triangle of Jc^T Jt (36 of 48)
#define ELIM_STRIDE 40  // per point: u = V^-1 g (3) | L, V^-1 = L L^T (6) | pad (7) | Z = V^-1 B^T (LIN_NCB x 3)

// 64 records of R doubles, one per lane, to contiguous global memory through a wave-private LDS
// buffer of 64 * (R + 1) doubles: each store instruction then covers 512 contiguous bytes.
// Only the first nvalid records are written.
template <int R>
__device__ __forceinline__ void tile_store(double *buf, const double (&v)[R], double *__restrict__ g, int lane, int nvalid) {
#pragma unroll
  for (int k = 0; k < R; ++k) buf[lane * (R + 1) + k] = v[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (R % 2 == 0) {
    // two doubles per lane and store: a pair never straddles records when R is even, and g is 16-byte aligned
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      const int q = 2 * (i * 64 + lane);
      const double *src = buf + (q / R) * (R + 1) + q % R;
      typedef double nt2 __attribute__((ext_vector_type(2)));
      const nt2 x = {src[0], src[1]};
      if (q < nvalid * R) __builtin_nontemporal_store(x, reinterpret_cast<nt2 *>(g + q));
    }
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int q = i * 64 + lane;
      const double x = buf[(q / R) * (R + 1) + q % R];
      if (q < nvalid * R) __builtin_nontemporal_store(x, g + q);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same for a piece of R doubles at `offset` of records of `stride` doubles (runs of R contiguous doubles per record)
template <int R>
__device__ __forceinline__ void tile_store_piece(double *buf, const double (&v)[R], double *__restrict__ g, int stride, int offset, int lane,
                                                 int nvalid) {
  static_assert(R % 2 == 0, "pairs of doubles");
#pragma unroll
  for (int k = 0; k < R; ++k) buf[lane * (R + 1) + k] = v[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int i = 0; i < R / 2; ++i) {
    const int q = 2 * (i * 64 + lane), rec = q / R, k = q % R;
    const double *src = buf + rec * (R + 1) + k;
    if (rec < nvalid) *reinterpret_cast<double2 *>(g + (int64_t)stride * rec + offset + k) = make_double2(src[0], src[1]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool CAM1, bool GEN, bool EMP>
__global__ void __launch_bounds__(256)
k_point_elim(LinArgs a) {
  constexpr int NCP = EMP ? BA_NCAMPAR : BA_NCAMPAR_BROWN;  // camera columns that can be in use
  __shared__ double lds[8];
  __shared__ double tbuf[4][64 * 21];  // per wave: the per-point records leave through it as contiguous runs
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  int bad = 0;
  // what the point leaves behind: V^-1 (6), gradient (3), column norms (3), LM diagonal (3), the elimination record
  double o_vi[6], o_g[3], o_cn[3], o_dg[3], o_el[ELIM_STRIDE];
#pragma unroll
  for (int k = 0; k < ELIM_STRIDE; ++k) o_el[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) o_vi[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) o_g[k] = o_cn[k] = o_dg[k] = 0.0;
  if (jj < a.P) {
    const int e0 = a.pt_start[jj], e1 = a.pt_start[jj + 1];
    const double X = a.xyz[3 * (int64_t)jj], Y = a.xyz[3 * (int64_t)jj + 1], Z = a.xyz[3 * (int64_t)jj + 2];
    const double s0 = a.scale_pt[3 * (int64_t)jj], s1 = a.scale_pt[3 * (int64_t)jj + 1], s2 = a.scale_pt[3 * (int64_t)jj + 2];
    double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    double B[CAM1 ? NCP : 1][3];  // all camera columns; the free ones are picked at the end
#pragma unroll
    for (int k = 0; k < (CAM1 ? NCP : 1); ++k) B[k][0] = B[k][1] = B[k][2] = 0.0;
    // the next observation's record is requested before this one is evaluated: its latency hides behind the arithmetic
    // (the station record still follows its index: one dependent load per observation instead of two)
    ObsIn o_next{};
    if (e0 < e1) o_next = load_obs(a.obs, e0);
    for (int e = e0; e < e1; ++e) {
      const ObsIn o = o_next;
      o_next = load_obs(a.obs, e + 1 < e1 ? e + 1 : e);
      const double *st = a.sta + (int64_t)BA_STA_STRIDE * o.sta;
      // one camera: its record has one address for all lanes (scalar loads, no wait for the station record first)
      const int c = CAM1 ? 0 : __double2loint(st[20]);
      const CamIn cm = load_cam(a.cam, c);
      double r0, r1, Js[12], Jp[6], dc0[NCP], dc1[NCP];
      eval_obs<true, GEN, EMP>(o, st, X, Y, Z, cm, r0, r1, Js, Jp, dc0, dc1, a.model);
      double sc;
      cost += 0.5 * huber_rho(a.huber_a, r0 * r0 + r1 * r1, sc);
      r0 *= sc;
      r1 *= sc;
      Jp[0] *= sc * s0; Jp[1] *= sc * s1; Jp[2] *= sc * s2;
      Jp[3] *= sc * s0; Jp[4] *= sc * s1; Jp[5] *= sc * s2;
      V[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3];
      V[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
      V[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
      V[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
      V[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5];
      V[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
      g[0] += Jp[0] * r0 + Jp[3] * r1;
      g[1] += Jp[1] * r0 + Jp[4] * r1;
      g[2] += Jp[2] * r0 + Jp[5] * r1;
      if (CAM1) {
        const uint32_t mask = a.cam[0].mask;
#pragma unroll
        for (int p = 0; p < NCP; ++p)
          if (mask & (1u << p)) {  // uniform: one camera
            const double c0 = dc0[p] * sc, c1 = dc1[p] * sc;
            B[p][0] += c0 * Jp[0] + c1 * Jp[3];
            B[p][1] += c0 * Jp[1] + c1 * Jp[4];
            B[p][2] += c0 * Jp[2] + c1 * Jp[5];
          }
      }
    }
    // prior rows (src/main.cpp:1064-1071, 4596-4599): r = (p - p0)/sigma, J = 1/sigma
    const double *pr = a.pt_prior + 3 * (int64_t)jj, *is = a.pt_isig + 3 * (int64_t)jj;
    const double rp0 = (X - pr[0]) * is[0], rp1 = (Y - pr[1]) * is[1], rp2 = (Z - pr[2]) * is[2];
    cost += 0.5 * (rp0 * rp0 + rp1 * rp1 + rp2 * rp2);
    const double j0 = is[0] * s0, j1 = is[1] * s1, j2 = is[2] * s2;
    V[0] += j0 * j0; V[3] += j1 * j1; V[5] += j2 * j2;
    g[0] += j0 * rp0; g[1] += j1 * rp1; g[2] += j2 * rp2;
    o_cn[0] = V[0]; o_cn[1] = V[3]; o_cn[2] = V[5];
    // Levenberg-Marquardt diagonal [Ceres 1.13 LevenbergMarquardtStrategy::ComputeStep]
    double d0, d1, d2;
    if (a.reuse_diag) {
      const double *dg = a.diag_pt + 3 * (int64_t)jj;
      d0 = dg[0]; d1 = dg[1]; d2 = dg[2];
    } else {
      d0 = fmin(fmax(V[0], a.lm_min), a.lm_max);
      d1 = fmin(fmax(V[3], a.lm_min), a.lm_max);
      d2 = fmin(fmax(V[5], a.lm_min), a.lm_max);
      o_dg[0] = d0; o_dg[1] = d1; o_dg[2] = d2;
    }
    const double l0 = sqrt(d0 / a.radius), l1 = sqrt(d1 / a.radius), l2 = sqrt(d2 / a.radius);
    V[0] += l0 * l0; V[3] += l1 * l1; V[5] += l2 * l2;
    double Vi[6];
    if (!inv_sym3(V, Vi)) bad = 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) o_vi[k] = Vi[k];
    o_g[0] = g[0]; o_g[1] = g[1]; o_g[2] = g[2];
    double (&el)[ELIM_STRIDE] = o_el;
    sym3_mul(Vi, g[0], g[1], g[2], el[0], el[1], el[2]);
    // Vinv = L L^T (lower Cholesky), for the whitened coupling blocks Y = Js^T Jp L
    const double l00 = sqrt(Vi[0]), l10 = Vi[1] / l00, l20 = Vi[2] / l00;
    const double l11 = sqrt(Vi[3] - l10 * l10), l21 = (Vi[4] - l20 * l10) / l11;
    const double l22 = sqrt(Vi[5] - l20 * l20 - l21 * l21);
    el[3] = l00; el[4] = l10; el[5] = l20; el[6] = l11; el[7] = l21; el[8] = l22;
    if (CAM1) {
      const CamRec &cr = a.cam[0];
      int pos = 0;
#pragma unroll
      for (int p = 0; p < NCP; ++p)
        if (cr.mask & (1u << p)) {
          const double cs = cr.scale[pos];
          double z0, z1, z2;
          sym3_mul(Vi, cs * B[p][0], cs * B[p][1], cs * B[p][2], z0, z1, z2);
          // (run-time position, compile-time index: the record stays in registers)
#pragma unroll
          for (int q = 0; q < LIN_NCB; ++q)
            if (q == pos) {
              el[16 + 3 * q] = z0;
              el[17 + 3 * q] = z1;
              el[18 + 3 * q] = z2;
            }
          ++pos;
        }
    }
  }
  {
    // the records of the wave's 64 points, as contiguous runs (a lane storing its own record piece by piece makes 64
    // separate 8-byte writes of every store instruction)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t jj0 = (int64_t)blockIdx.x * blockDim.x + 64 * w;
    const int nvalid = (int)(a.P - jj0 < 64 ? (a.P - jj0 > 0 ? a.P - jj0 : 0) : 64);
    double *buf = tbuf[w];
    if (nvalid > 0) {  // uniform over the wave
      tile_store<6>(buf, o_vi, a.Vinv + 6 * jj0, lane, nvalid);
      tile_store<3>(buf, o_g, a.g_pt + 3 * jj0, lane, nvalid);
      tile_store<3>(buf, o_cn, a.colnorm_pt + 3 * jj0, lane, nvalid);
      if (!a.reuse_diag) tile_store<3>(buf, o_dg, a.diag_pt + 3 * jj0, lane, nvalid);
      double half[ELIM_STRIDE / 2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < ELIM_STRIDE / 2; ++k) half[k] = o_el[h * (ELIM_STRIDE / 2) + k];
        tile_store_piece<ELIM_STRIDE / 2>(buf, half, a.elim + (int64_t)ELIM_STRIDE * jj0, ELIM_STRIDE, h * (ELIM_STRIDE / 2), lane, nvalid);
      }
    }
  }
  const double t = block_sum(cost, lds);
  if (threadIdx.x == 0) a.cost_partial[blockIdx.x] = t;
  if (bad) atomicOr(a.flags, 1);
}

#define ROWS_LDS(CAM1) (64 * ((CAM1) ? 25 : 2 * BA_NCAMPAR + 1))
#define KS_STRIDE 34  // doubles per row of the staging buffer of the per-station sums: 32 columns, rows 16-byte aligned
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// Two waves per SIMD where the register budget of that allows it without spilling (one register short otherwise): the
// collinearity projection without the empirical terms.  The other variants keep one wave with the full register file.
//
// SUMS: the kernel also forms everything a station sums over its observations -- the diagonal block sum Js^T M Js, the
// column norms, gradient and reduced right-hand side, and the camera border sum Js^T Jt -- and then writes the
// coupling blocks Y only (no row of the Jacobian leaves the chip).  A wave owns a segment of consecutive tiles.  Per
// tile every lane forms the products of its observation (39 + 6 x 8 columns), 32 columns at a time, and puts them
// into the wave's staging buffer at the row its observation has in the tile's station-sorted order (prepared once per
// problem, ba_keyed.hip).  Then the lanes change roles: lane (h, c) owns column c and rows 32 h .. 32 h + 31, adds the
// rows up in a register and, where a run of one station ends, adds the sum to that station's row of the wave's LDS
// table (ds_add_f64 -- the table is private to the wave and LDS operations of a wave execute in order, so the order
// of every sum is fixed).  Rows leave the table for partial records in global memory when the plan says so; a second
// tiny kernel adds the records of a station in ascending order.
template <bool CAM1, bool GEN, bool EMP, bool SUMS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((GEN || EMP || SUMS) ? 1 : 2)))
k_obs_rows(LinArgs a) {
  constexpr int NCP = EMP ? BA_NCAMPAR : BA_NCAMPAR_BROWN;
  constexpr int NCOLP = CAM1 ? 96 : 64;  // columns of a table row: 39 station sums (+ 48 border sums), padded to 32s
  constexpr int BUF = SUMS ? 64 * KS_STRIDE : ROWS_LDS(CAM1);
  __shared__ __attribute__((aligned(16))) double lds[4][BUF];
  __shared__ double lds_cam[4][LIN_NPART];
  __shared__ __attribute__((aligned(16))) double lds_tab[SUMS ? 4 : 1][SUMS ? BA_KS_NSLOT * NCOLP : 2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double *buf = lds[w];
  double *tab = lds_tab[SUMS ? w : 0];
  // sums of Jc^T r and Jc^T rhat over this wave's tiles: reduced across the wave tile by tile and kept in LDS
  // (sixteen accumulators per lane across the tile loop would cost the second wave per SIMD)
  if (CAM1) lds_cam[w][lane] = 0.0;  // LIN_NPART == 64
  // SUMS, one camera: which camera column each of the LIN_NCB free parameters is and its scale (zero: no such parameter),
  // once per workgroup; a tile then picks its columns with it instead of walking the camera's mask
  __shared__ int lds_selp[SUMS && CAM1 ? LIN_NCB : 1];
  __shared__ double lds_sels[SUMS && CAM1 ? LIN_NCB : 1];
  if (SUMS && CAM1) {
    if (threadIdx.x < LIN_NCB) {
      const CamRec &cr = a.cam[0];
      int pos = 0, pk = 0;
      double sk = 0.0;
      for (int p = 0; p < NCP; ++p)
        if (cr.mask & (1u << p)) {
          if (pos == (int)threadIdx.x) { pk = p; sk = cr.scale[pos]; }
          ++pos;
        }
      lds_selp[threadIdx.x] = pk;
      lds_sels[threadIdx.x] = sk;
    }
    __syncthreads();
  }
  const int64_t n_tiles = (a.N + 63) / 64;
  // SUMS: one segment of consecutive tiles per wave; otherwise the tiles are dealt round-robin over all waves
  int64_t tile_begin, tile_end, tile_step;
  int op = 0, op_end = 0;
  if (SUMS) {
    const int seg = blockIdx.x * 4 + w;
    tile_begin = (int64_t)seg * a.kp.tiles_per_seg;
    tile_end = tile_begin + a.kp.tiles_per_seg;
    if (tile_end > n_tiles) tile_end = n_tiles;
    if (seg >= a.kp.n_seg) tile_end = tile_begin;
    tile_step = 1;
    if (seg < a.kp.n_seg) {
      op = a.kp.seg_op0[seg];
      op_end = a.kp.seg_op0[seg + 1];
    }
    for (int k = lane; k < BA_KS_NSLOT * NCOLP; k += 64) tab[k] = 0.0;
  } else {
    tile_begin = (int64_t)blockIdx.x * 4 + w;
    tile_end = n_tiles;
    tile_step = (int64_t)gridDim.x * 4;
  }
  // table rows that the plan sends to their partial records before tile `trel` of the segment; the entry that is next
  // in the list is held in registers, so that a tile with nothing to write out waits for no load
  int2 op_cur = make_int2(0x7FFFFFFF, 0);
  if (SUMS && op < op_end) op_cur = a.kp.ops[op];
  auto write_out = [&](int trel) {
    while (op < op_end && (op_cur.x >> 8) <= trel) {
      double *row = tab + (op_cur.x & 255) * NCOLP;
      if (lane < NCOLP / 2) {
        const double2 v = *reinterpret_cast<const double2 *>(row + 2 * lane);
        *reinterpret_cast<double2 *>(a.kp.part + (int64_t)op_cur.y * NCOLP + 2 * lane) = v;
        *reinterpret_cast<double2 *>(row + 2 * lane) = make_double2(0.0, 0.0);
      }
      ++op;
      if (op < op_end) op_cur = a.kp.ops[op];
    }
  };
  // The observation record and point index of the NEXT tile are requested at the top of a tile: a wave's loads and
  // stores complete in order, so a load issued after a tile's stores also waits for those to drain.
  ObsIn o_next{};
  int jj_next = 0;
  if (tile_begin < tile_end) {
    const int64_t t0 = tile_begin;
    const int64_t en = (t0 * 64 + lane < a.N) ? t0 * 64 + lane : t0 * 64;
    o_next = load_obs(a.obs, en);
    jj_next = a.obs_pt[en];
  }
  // ... and the records those lead to (station, point) right before a tile's first store, for the same reason
  // (what the projection needs: rotation, position, sin/cos kappa, the point; the scales are read when they are used)
  double st_next[14], xyz_next[3];
  // SUMS (one wave per SIMD, the whole register file): everything else the tile reads per observation as well -- the
  // station's column scales, the point's scales, V^-1 and its elimination record -- so that a tile waits for no load
  // that it issued itself
  double ssc_next[SUMS ? 6 : 1], spt_next[SUMS ? 3 : 1], vi_next[SUMS ? 6 : 1], el_next[SUMS ? 10 : 1];
  auto gather_next = [&]() {
    const double2 *sp = reinterpret_cast<const double2 *>(a.sta + (int64_t)BA_STA_STRIDE * o_next.sta);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const double2 v = sp[k];
      st_next[2 * k] = v.x;
      st_next[2 * k + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) xyz_next[k] = a.xyz[3 * (int64_t)jj_next + k];
    if (SUMS) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double2 v = sp[7 + k];
        ssc_next[2 * k] = v.x;
        ssc_next[2 * k + 1] = v.y;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) spt_next[k] = a.scale_pt[3 * (int64_t)jj_next + k];
      const double2 *vp = reinterpret_cast<const double2 *>(a.Vinv + 6 * (int64_t)jj_next);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double2 v = vp[k];
        vi_next[2 * k] = v.x;
        vi_next[2 * k + 1] = v.y;
      }
      const double2 *ep = reinterpret_cast<const double2 *>(a.elim + (int64_t)ELIM_STRIDE * jj_next);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const double2 v = ep[k];
        el_next[2 * k] = v.x;
        el_next[2 * k + 1] = v.y;
      }
    }
  };
  if (tile_begin < tile_end) gather_next();
  double cam_acc[(SUMS && CAM1) ? 2 * LIN_NCB + LIN_NCORNER : 1];  // per-lane sums of the camera rows and the corner (SUMS)
#pragma unroll
  for (int k = 0; k < ((SUMS && CAM1) ? 2 * LIN_NCB + LIN_NCORNER : 1); ++k) cam_acc[k] = 0.0;
  unsigned long long tr_acc[KS_TRACE_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last = 0;
  const bool tracing = SUMS && a.kp.trace != nullptr;
  auto stamp = [&](int k) {
    if (SUMS && tracing) {
      const unsigned long long now = clock64();
      tr_acc[k] += now - tr_last;
      tr_last = now;
    }
  };
  for (int64_t tile = tile_begin; tile < tile_end; tile += tile_step) {
    if (SUMS && tracing) tr_last = clock64();
    const int64_t e0 = tile * 64;
    const int nvalid = (int)((a.N - e0 < 64) ? (a.N - e0) : 64);
    const bool on = lane < nvalid;
    // idle lanes of the last tile redo its first observation and store nothing
    const ObsIn o = o_next;
    const int jj = jj_next;
    if (tile + tile_step < tile_end) {
      const int64_t en0 = (tile + tile_step) * 64;
      const int64_t en = (en0 + lane < a.N) ? en0 + lane : en0;
      o_next = load_obs(a.obs, en);
      jj_next = a.obs_pt[en];
    }
    // SUMS: this tile's part of the plan -- the lane's row in the sorted order, the slots of the rows of its half
    int my_row = 0;
    uint32_t um = 0;
    uint4 sw0 = make_uint4(0, 0, 0, 0), sw1 = make_uint4(0, 0, 0, 0);
    if (SUMS) {
      my_row = a.kp.spos[e0 + lane];
      um = a.kp.umask[tile];
      const uint4 *sp = reinterpret_cast<const uint4 *>(a.kp.slot + e0 + 32 * (lane >> 5));
      sw0 = sp[0];
      sw1 = sp[1];
    }
    double st[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) st[k] = st_next[k];
    const double *st_rec = a.sta + (int64_t)BA_STA_STRIDE * o.sta;
    const int c = CAM1 ? 0 : __double2loint(st_rec[20]);  // one camera: scalar loads of its record
    const CamIn cm = load_cam(a.cam, c);
    const double X = xyz_next[0], Y = xyz_next[1], Z = xyz_next[2];
    double ssc[6], Vi[6], el9[9], elz[3 * LIN_NCB];
    double s0, s1, s2;
    if (SUMS) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { ssc[k] = ssc_next[k]; Vi[k] = vi_next[k]; }
#pragma unroll
      for (int k = 0; k < 9; ++k) el9[k] = el_next[k];
      if (CAM1) {
        // the camera part of the elimination record is used last in the tile: requested here, for this tile
        const double2 *zp = reinterpret_cast<const double2 *>(a.elim + (int64_t)ELIM_STRIDE * jj + 16);
#pragma unroll
        for (int k = 0; k < 3 * LIN_NCB / 2; ++k) {
          const double2 v = zp[k];
          elz[2 * k] = v.x;
          elz[2 * k + 1] = v.y;
        }
      }
      s0 = spt_next[0]; s1 = spt_next[1]; s2 = spt_next[2];
    } else {
      s0 = a.scale_pt[3 * (int64_t)jj]; s1 = a.scale_pt[3 * (int64_t)jj + 1]; s2 = a.scale_pt[3 * (int64_t)jj + 2];
    }
    double r0, r1, Js[12], Jp[6], dc0[NCP], dc1[NCP];
    eval_obs<true, GEN, EMP>(o, st, X, Y, Z, cm, r0, r1, Js, Jp, dc0, dc1, a.model);
    double sc;
    (void)huber_rho(a.huber_a, r0 * r0 + r1 * r1, sc);
    r0 *= sc;
    r1 *= sc;
    stamp(0);
    // free camera columns, compacted: into LDS at a run-time position, out as whole lines
    double c0[CAM1 ? LIN_NCB : 1], c1[CAM1 ? LIN_NCB : 1];
    if (SUMS && CAM1) {
      // all camera columns into the lane's own row of the wave's buffer, the free ones back out by their index
      double *mine = buf + lane * (2 * NCP + 1);
#pragma unroll
      for (int p = 0; p < NCP; ++p) {
        mine[p] = dc0[p];
        mine[NCP + p] = dc1[p];
      }
#pragma unroll
      for (int k = 0; k < LIN_NCB; ++k) {
        const int pk = lds_selp[k];
        const double cs = sc * lds_sels[k];
        c0[k] = mine[pk] * cs;
        c1[k] = mine[NCP + pk] * cs;
      }
      WAVE_SYNC();
    } else if (a.ncmax > 0) {
      const CamRec &cr = a.cam[c];
      const int R = 2 * a.ncmax;
      double *mine = buf + lane * (R + 1);
      for (int k = cr.nc; k < a.ncmax; ++k) {  // cameras with fewer free parameters than the widest one
        mine[k] = 0.0;
        mine[a.ncmax + k] = 0.0;
      }
      int pos = 0;
#pragma unroll
      for (int p = 0; p < NCP; ++p) {
        if (cr.mask & (1u << p)) {
          const double cs = sc * cr.scale[pos];
          mine[pos] = dc0[p] * cs;
          mine[a.ncmax + pos] = dc1[p] * cs;
          ++pos;
        }
      }
      WAVE_SYNC();
      if (!a.lean) {
        double *gj = a.Jc + (int64_t)R * e0;
        // q / R by a multiplication: exact for q < 64 * 38 with this constant (R is not known at compile time, and a
        // vector division per element would cost more than the rest of the copy)
        const unsigned magic = 0xFFFFFFFFu / (unsigned)R + 1u;
        for (int q = lane; q < nvalid * R; q += 64) {
          const int rec = (int)__umulhi((unsigned)q, magic);
          gj[q] = buf[rec * (R + 1) + (q - rec * R)];
        }
      }
      if (CAM1) {
#pragma unroll
        for (int k = 0; k < LIN_NCB; ++k) {
          c0[k] = (k < a.ncmax) ? mine[k] : 0.0;
          c1[k] = (k < a.ncmax) ? mine[a.ncmax + k] : 0.0;
        }
      }
      WAVE_SYNC();
    }
    stamp(1);
    // point data of the elimination
    // (two waves per SIMD: the elimination record is read where it is used, it would not fit the registers)
    const double *el = a.elim + (int64_t)ELIM_STRIDE * jj;
    if (!SUMS) {
      const double *vi = a.Vinv + 6 * (int64_t)jj;
#pragma unroll
      for (int k = 0; k < 6; ++k) Vi[k] = vi[k];
    }
    const double u0 = SUMS ? el9[0] : el[0], u1 = SUMS ? el9[1] : el[1], u2 = SUMS ? el9[2] : el[2];
    const double p0 = Jp[0] * (sc * s0), p1 = Jp[1] * (sc * s1), p2 = Jp[2] * (sc * s2);
    const double p3 = Jp[3] * (sc * s0), p4 = Jp[4] * (sc * s1), p5 = Jp[5] * (sc * s2);
    const double h0 = r0 - (p0 * u0 + p1 * u1 + p2 * u2), h1 = r1 - (p3 * u0 + p4 * u1 + p5 * u2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double cs = sc * (SUMS ? ssc[k] : st_rec[14 + k]);
      Js[k] *= cs;
      Js[6 + k] *= cs;
    }
    if (!SUMS) {
      double rec[16];
#pragma unroll
      for (int k = 0; k < 12; ++k) rec[k] = Js[k];
      rec[12] = r0; rec[13] = r1; rec[14] = h0; rec[15] = h1;
      if (tile + tile_step < tile_end) gather_next();  // before this tile's first store
      tile_store<16>(buf, rec, a.rec + (int64_t)BA_REC_STRIDE * e0, lane, nvalid);
    } else {
      if (tile + tile_step < tile_end) gather_next();
    }
    double mo[3];
    {
      const double jp[6] = {p0, p1, p2, p3, p4, p5};
      if (!SUMS && !a.lean) tile_store<6>(buf, jp, a.Jp + 6 * e0, lane, nvalid);
      double t0, t1, t2, w0, w1, w2;
      sym3_mul(Vi, p0, p1, p2, t0, t1, t2);
      sym3_mul(Vi, p3, p4, p5, w0, w1, w2);
      mo[0] = 1.0 - (p0 * t0 + p1 * t1 + p2 * t2);
      mo[1] = -(p0 * w0 + p1 * w1 + p2 * w2);
      mo[2] = 1.0 - (p3 * w0 + p4 * w1 + p5 * w2);
      if (!SUMS) tile_store<3>(buf, mo, a.Mo + 3 * e0, lane, nvalid);
    }
    stamp(2);
    if (a.Y) {
      // G = Jp L (2x3);  Y[r][m] = Js0[r] G0[m] + Js1[r] G1[m], row-major 6 x 3
      const double l00 = SUMS ? el9[3] : el[3], l10 = SUMS ? el9[4] : el[4], l20 = SUMS ? el9[5] : el[5];
      const double l11 = SUMS ? el9[6] : el[6], l21 = SUMS ? el9[7] : el[7], l22 = SUMS ? el9[8] : el[8];
      const double g00 = p0 * l00 + p1 * l10 + p2 * l20, g01 = p1 * l11 + p2 * l21, g02 = p2 * l22;
      const double g10 = p3 * l00 + p4 * l10 + p5 * l20, g11 = p4 * l11 + p5 * l21, g12 = p5 * l22;
      double y[BA_Y_STRIDE];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        y[3 * r] = Js[r] * g00 + Js[6 + r] * g10;
        y[3 * r + 1] = Js[r] * g01 + Js[6 + r] * g11;
        y[3 * r + 2] = Js[r] * g02 + Js[6 + r] * g12;
      }
      tile_store<BA_Y_STRIDE>(buf, y, a.Y + (int64_t)BA_Y_STRIDE * e0, lane, nvalid);
    }
    stamp(3);
    double jt[2 * LIN_NCB];  // projected camera rows (single-camera path)
    if (CAM1) {
      {
#pragma unroll
        for (int k = 0; k < LIN_NCB; ++k) {
          const double z0 = SUMS ? elz[3 * k] : el[16 + 3 * k], z1 = SUMS ? elz[3 * k + 1] : el[17 + 3 * k], z2 = SUMS ? elz[3 * k + 2] : el[18 + 3 * k];
          const bool free_k = SUMS || k < a.ncmax;  // SUMS: columns beyond the free ones are zeros, and so is their Z
          jt[k] = free_k ? c0[k] - (p0 * z0 + p1 * z1 + p2 * z2) : 0.0;
          jt[LIN_NCB + k] = free_k ? c1[k] - (p3 * z0 + p4 * z1 + p5 * z2) : 0.0;
        }
        if (!SUMS) tile_store<2 * LIN_NCB>(buf, jt, a.Jt + 2 * (int64_t)LIN_NCB * e0, lane, nvalid);
        // corner of the camera border: upper triangle of Jc^T Jt, 36 sums in three reductions of 16
        // (SUMS: one wave per SIMD has the registers to keep the 52 sums per lane over all tiles of its segment; they are
        // reduced across the wave once, after the last tile)
        const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          double cs[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            // entry idx = 16 part + q of the row-major upper triangle: (k, m), k <= m
            constexpr int tri_k[48] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3,
                                       3, 3, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            constexpr int tri_m[48] = {0, 1, 2, 3, 4, 5, 6, 7, 1, 2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5,
                                       6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const int idx = 16 * part + q;
            cs[q] = (idx < LIN_NCORNER && on) ? c0[tri_k[idx]] * jt[tri_m[idx]] + c1[tri_k[idx]] * jt[LIN_NCB + tri_m[idx]] : 0.0;
            if (SUMS && idx < LIN_NCORNER) cam_acc[2 * LIN_NCB + idx] += cs[q];
          }
          if (!SUMS) {
            const double tot = wave_sum16(cs, lane);
            if ((lane & 3) == 0) lds_cam[w][2 * LIN_NCB + 16 * part + slot] += tot;
          }
        }
      }
      {
        double gsum[2 * LIN_NCB];
#pragma unroll
        for (int k = 0; k < LIN_NCB; ++k) {
          const bool use = on && (SUMS || k < a.ncmax);
          gsum[k] = use ? c0[k] * r0 + c1[k] * r1 : 0.0;
          gsum[LIN_NCB + k] = use ? c0[k] * h0 + c1[k] * h1 : 0.0;
          if (SUMS) {
            cam_acc[k] += gsum[k];
            cam_acc[LIN_NCB + k] += gsum[LIN_NCB + k];
          }
        }
        if (!SUMS) {
          const double tot = wave_sum16(gsum, lane);
          if ((lane & 3) == 0) lds_cam[w][((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1)] += tot;
        }
      }
    }
    stamp(4);
    if (SUMS) {
      write_out((int)(tile - tile_begin));
      // factors of the products; idle lanes of the last tile contribute zeros
      double a0[6], a1[6], w0[6], w1[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        a0[k] = on ? Js[k] : 0.0;
        a1[k] = on ? Js[6 + k] : 0.0;
        w0[k] = mo[0] * a0[k] + mo[1] * a1[k];
        w1[k] = mo[1] * a0[k] + mo[2] * a1[k];
      }
      auto product = [&](int col) -> double {  // col is a constant wherever this is called
        constexpr int tp[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
        constexpr int tq[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
        if (col < 21) return w0[tp[col]] * a0[tq[col]] + w1[tp[col]] * a1[tq[col]];   // sum Js^T M Js, upper triangle
        if (col < 27) return a0[col - 21] * a0[col - 21] + a1[col - 21] * a1[col - 21];  // column norms
        if (col < 33) return a0[col - 27] * r0 + a1[col - 27] * r1;                      // gradient
        if (col < 39) return a0[col - 33] * h0 + a1[col - 33] * h1;                      // reduced right-hand side
        if (CAM1 && col < 39 + 6 * LIN_NCB)                                              // camera border, 6 x 8
          return a0[(col - 39) / LIN_NCB] * jt[(col - 39) % LIN_NCB] + a1[(col - 39) / LIN_NCB] * jt[LIN_NCB + (col - 39) % LIN_NCB];
        return 0.0;
      };
      const int h = lane >> 5, cl = lane & 31;
      double *row = buf + my_row * KS_STRIDE;
      const double *colp = buf + (32 * h) * KS_STRIDE + cl;
      const uint32_t sw[8] = {sw0.x, sw0.y, sw0.z, sw0.w, sw1.x, sw1.y, sw1.z, sw1.w};
      const uint32_t umu = __builtin_amdgcn_readfirstlane(um);
#pragma unroll
      for (int pass = 0; pass < NCOLP / 32; ++pass) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          *reinterpret_cast<double2 *>(row + 2 * k) = make_double2(product(32 * pass + 2 * k), product(32 * pass + 2 * k + 1));
        WAVE_SYNC();
        stamp(5);
        double acc = 0.0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // sixteen rows at a time keep the registers in bounds
          double x[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) x[i] = colp[(16 * half + i) * KS_STRIDE];
          if (tracing) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(6); }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            acc += x[i];
            if (umu & (1u << (16 * half + i))) {  // uniform: a run ends here in one of the halves (the other half's run is cut in two)
              const int ii = 16 * half + i;
              const int sl = (int)((sw[ii >> 2] >> (8 * (ii & 3))) & 255u);
              __hip_atomic_fetch_add(tab + sl * NCOLP + 32 * pass + cl, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              acc = 0.0;
            }
          }
        }
        WAVE_SYNC();
        stamp(7);
      }
      WAVE_SYNC();
    }
  }
  if (SUMS && tracing) tr_last = clock64();
  if (SUMS && CAM1) {
    const int slot = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
#pragma unroll
    for (int part = 0; part < 4; ++part) {  // 0: camera rows of gradient and right-hand side, 1..3: the corner
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = (16 * part + q < 2 * LIN_NCB + LIN_NCORNER) ? cam_acc[(16 * part + q < 2 * LIN_NCB + LIN_NCORNER) ? 16 * part + q : 0] : 0.0;
      const double tot = wave_sum16(v, lane);
      if ((lane & 3) == 0) lds_cam[w][16 * part + slot] = tot;
    }
  }
  if (SUMS) write_out(0x7FFFFF);
  stamp(4);
  if (SUMS && tracing && lane == 0) {
    unsigned long long *out = a.kp.trace + (int64_t)(blockIdx.x * 4 + w) * KS_TRACE_SLOTS;
#pragma unroll
    for (int k = 0; k < KS_TRACE_SLOTS; ++k) out[k] = tr_acc[k];
  }
  if (CAM1) {
    __syncthreads();
    if (threadIdx.x < LIN_NPART)
      a.cam_partial[(int64_t)blockIdx.x * LIN_NPART + threadIdx.x] =
          (lds_cam[0][threadIdx.x] + lds_cam[1][threadIdx.x]) + (lds_cam[2][threadIdx.x] + lds_cam[3][threadIdx.x]);
  }
}

// out[k] = sum over rows of part[row][k]: one workgroup per column, fixed order
__global__ void __launch_bounds__(256)
k_fold_columns(const double *__restrict__ part, int nrows, int ncols, int out_cols, double *__restrict__ out) {
  __shared__ double lds[8];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int r = threadIdx.x; r < nrows; r += blockDim.x) s += part[(int64_t)r * ncols + k];
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0 && k < out_cols) out[k] = t;
}

// ------------------------------------------------------------------------------------------
// point sweep 2: the point half of the matrix-free reduced operator
// ------------------------------------------------------------------------------------------
//   a = Js p_i (+ Jc p_c);  t = sum Jp^T a;  u = Vinv t;  slot = a - Jp u      (one lane per point)
template <bool WITHCAM>
__global__ void __launch_bounds__(256)
k_point_apply(PointArgs a) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj >= a.P) return;
  const int e0 = a.pt_start[jj], e1 = a.pt_start[jj + 1];
  double t0 = 0, t1 = 0, t2 = 0;
  for (int e = e0; e < e1; ++e) {
    double2 *rec = reinterpret_cast<double2 *>(a.rec + (int64_t)BA_REC_STRIDE * e);
    const int sta = a.obs_sta[e];
    const double2 *pv = reinterpret_cast<const double2 *>(a.vec + 6 * (int64_t)sta);
    const double2 v0 = pv[0], v1 = pv[1], v2 = pv[2];
    const double2 j0 = rec[0], j1 = rec[1], j2 = rec[2], j3 = rec[3], j4 = rec[4], j5 = rec[5];
    double a0 = j0.x * v0.x + j0.y * v0.y + j1.x * v1.x + j1.y * v1.y + j2.x * v2.x + j2.y * v2.y;
    double a1 = j3.x * v0.x + j3.y * v0.y + j4.x * v1.x + j4.y * v1.y + j5.x * v2.x + j5.y * v2.y;
    if (WITHCAM) {
      const int c = a.sta_cam[sta];
      const int nc = a.cam[c].nc, off = a.cam[c].off;
      const double *jc = a.Jc + 2 * (int64_t)a.ncmax * e;
      const double *vc = a.vec + 6 * (int64_t)a.S + off;
      for (int k = 0; k < nc; ++k) {
        a0 += jc[k] * vc[k];
        a1 += jc[a.ncmax + k] * vc[k];
      }
    }
    const double2 *jp = reinterpret_cast<const double2 *>(a.Jp + 6 * (int64_t)e);
    const double2 q0 = jp[0], q1 = jp[1], q2 = jp[2];
    t0 += q0.x * a0 + q1.y * a1;
    t1 += q0.y * a0 + q2.x * a1;
    t2 += q1.x * a0 + q2.y * a1;
    rec[7] = make_double2(a0, a1);
  }
  const double *vi = a.Vinv + 6 * (int64_t)jj;
  const double Vi[6] = {vi[0], vi[1], vi[2], vi[3], vi[4], vi[5]};
  double u0, u1, u2;
  sym3_mul(Vi, t0, t1, t2, u0, u1, u2);
  for (int e = e0; e < e1; ++e) {
    double2 *rec = reinterpret_cast<double2 *>(a.rec + (int64_t)BA_REC_STRIDE * e);
    const double2 *jp = reinterpret_cast<const double2 *>(a.Jp + 6 * (int64_t)e);
    const double2 q0 = jp[0], q1 = jp[1], q2 = jp[2];
    const double2 s = rec[7];
    rec[7] = make_double2(s.x - (q0.x * u0 + q0.y * u1 + q1.x * u2), s.y - (q1.y * u0 + q2.x * u1 + q2.y * u2));
  }
}

// ------------------------------------------------------------------------------------------
// back substitution: y_p = Vinv (g - sum Jp^T a), a = Js y_i + Jc y_c, and the partial sums of
// (J step).(r + J step / 2) for step = -y  [model cost change]
// ------------------------------------------------------------------------------------------
// Back substitution with the observation rows read the way they lie in memory.  A wave owns 64 consecutive points
// and therefore one contiguous run of observations.  It walks that run 64 observations at a time, one lane per
// observation (a = Js y_i + Jc y_c and Jp^T a from whole records), hands the three sums of each observation to the
// lane that owns its point through LDS, and after u = Vinv (g - sum Jp^T a) walks the run once more for the model
// cost change.  The products a stay in LDS when the run has at most BS_RUN observations, else in the work slot of the
// row.  Same sums in the same order as the lane-per-point walk; what changes is that every record is fetched once.
#define BS_RUN 512
template <bool WITHCAM>
__global__ void __launch_bounds__(256)
k_backsub_tiles(PointArgs a) {
  __shared__ double lds[8];
  __shared__ double lds_c[4][64 * 3];
  __shared__ double lds_u[4][64 * 3];
  __shared__ double2 lds_a[4][BS_RUN];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int jj0 = (blockIdx.x * 4 + w) * 64;
  const int jj = jj0 + lane;
  const bool has_pt = jj < a.P;
  double acc = 0.0;
  if (jj0 < a.P) {  // whole waves only
    const int jend = min(jj0 + 64, a.P);
    const int eb = a.pt_start[jj0], ee = a.pt_start[jend];
    const int e0 = has_pt ? a.pt_start[jj] : ee, e1 = has_pt ? a.pt_start[jj + 1] : ee;
    const bool fits = ee - eb <= BS_RUN;
    double t0 = 0, t1 = 0, t2 = 0;
    for (int base = eb; base < ee; base += 64) {
      const int e = base + lane;
      const bool on = e < ee;
      double c0 = 0, c1 = 0, c2 = 0;
      if (on) {
        double2 *rec = reinterpret_cast<double2 *>(a.rec + (int64_t)BA_REC_STRIDE * e);
        const int sta = a.obs_sta[e];
        const double2 *pv = reinterpret_cast<const double2 *>(a.vec + 6 * (int64_t)sta);
        const double2 v0 = pv[0], v1 = pv[1], v2 = pv[2];
        const double2 j0 = rec[0], j1 = rec[1], j2 = rec[2], j3 = rec[3], j4 = rec[4], j5 = rec[5];
        double a0 = j0.x * v0.x + j0.y * v0.y + j1.x * v1.x + j1.y * v1.y + j2.x * v2.x + j2.y * v2.y;
        double a1 = j3.x * v0.x + j3.y * v0.y + j4.x * v1.x + j4.y * v1.y + j5.x * v2.x + j5.y * v2.y;
        if (WITHCAM) {
          const int c = a.sta_cam[sta];
          const int nc = a.cam[c].nc, off = a.cam[c].off;
          const double *jc = a.Jc + 2 * (int64_t)a.ncmax * e;
          const double *vc = a.vec + 6 * (int64_t)a.S + off;
          for (int k = 0; k < nc; ++k) {
            a0 += jc[k] * vc[k];
            a1 += jc[a.ncmax + k] * vc[k];
          }
        }
        const double2 *jp = reinterpret_cast<const double2 *>(a.Jp + 6 * (int64_t)e);
        const double2 q0 = jp[0], q1 = jp[1], q2 = jp[2];
        c0 = q0.x * a0 + q1.y * a1;
        c1 = q0.y * a0 + q2.x * a1;
        c2 = q1.x * a0 + q2.y * a1;
        if (fits) lds_a[w][e - eb] = make_double2(a0, a1);
        else rec[7] = make_double2(a0, a1);
      }
      lds_c[w][3 * lane] = c0;
      lds_c[w][3 * lane + 1] = c1;
      lds_c[w][3 * lane + 2] = c2;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int q_lo = max(e0, base), q_hi = min(e1, base + 64);
      for (int q = q_lo; q < q_hi; ++q) {  // this point's observations of the chunk, in order
        t0 += lds_c[w][3 * (q - base)];
        t1 += lds_c[w][3 * (q - base) + 1];
        t2 += lds_c[w][3 * (q - base) + 2];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    double u0 = 0, u1 = 0, u2 = 0;
    if (has_pt) {
      const double *vi = a.Vinv + 6 * (int64_t)jj;
      const double Vi[6] = {vi[0], vi[1], vi[2], vi[3], vi[4], vi[5]};
      const double *g = a.g_pt + 3 * (int64_t)jj;
      sym3_mul(Vi, g[0] - t0, g[1] - t1, g[2] - t2, u0, u1, u2);
      double *yo = a.y_pt + 3 * (int64_t)jj;
      yo[0] = u0; yo[1] = u1; yo[2] = u2;
      // prior rows of this point
      const double *pr = a.pt_prior + 3 * (int64_t)jj, *is = a.pt_isig + 3 * (int64_t)jj;
      const double *x = a.xyz + 3 * (int64_t)jj, *sp = a.scale_pt + 3 * (int64_t)jj;
      const double m0 = -is[0] * sp[0] * u0, m1 = -is[1] * sp[1] * u1, m2 = -is[2] * sp[2] * u2;
      acc += m0 * ((x[0] - pr[0]) * is[0] + 0.5 * m0) + m1 * ((x[1] - pr[1]) * is[1] + 0.5 * m1) +
             m2 * ((x[2] - pr[2]) * is[2] + 0.5 * m2);
    }
    lds_u[w][3 * lane] = u0;
    lds_u[w][3 * lane + 1] = u1;
    lds_u[w][3 * lane + 2] = u2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int base = eb; base < ee; base += 64) {
      const int e = base + lane;
      if (e < ee) {
        const int pl = a.obs_pt[e] - jj0;  // the lane that owns this observation's point
        const double w0 = lds_u[w][3 * pl], w1 = lds_u[w][3 * pl + 1], w2 = lds_u[w][3 * pl + 2];
        const double2 *rec = reinterpret_cast<const double2 *>(a.rec + (int64_t)BA_REC_STRIDE * e);
        const double2 *jp = reinterpret_cast<const double2 *>(a.Jp + 6 * (int64_t)e);
        const double2 q0 = jp[0], q1 = jp[1], q2 = jp[2];
        const double2 sv = fits ? lds_a[w][e - eb] : rec[7], r = rec[6];
        const double m0 = -(sv.x + q0.x * w0 + q0.y * w1 + q1.x * w2);
        const double m1 = -(sv.y + q1.y * w0 + q2.x * w1 + q2.y * w2);
        acc += m0 * (r.x + 0.5 * m0) + m1 * (r.y + 0.5 * m1);
      }
    }
  }
  const double t = block_sum(acc, lds);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
}

// Back substitution from the whitened coupling blocks alone.  With V^-1 = L L^T and Y_e = Js_e^T Jp_e L (what the block
// assembly reads as well), V^-1 Jp_e^T Js_e y_i = L Y_e^T y_i, so
//     y_p = u - L sum_e Y_e^T y_{i(e)} - Z y_c,     u = V^-1 g,  Z = V^-1 B^T  (both in the elimination record):
// no row of the Jacobian is read.  A wave owns 64 consecutive points = one contiguous run of observations, one lane per
// observation forms Y_e^T y_i, the three sums of an observation go to the lane that owns its point through LDS.
// The model cost change follows without the rows too (launch_mcc_reduced): with H = J^T J and y_p from the exact
// elimination,  y^T H y = y_c^T S y_c - y_c^T D_c^2 y_c + g_p^T u - y_p^T D_p^2 y_p  for ANY y_c (also an inexact PCG
// solution).  This kernel leaves the point part: partial sums of  (g_p.u - y_p.D_p^2 y_p) / 2 - g_p.y_p.
__global__ void __launch_bounds__(256)
k_backsub_y(PointArgs a) {
  __shared__ double lds[8];
  __shared__ double lds_c[4][64 * 3];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int jj0 = (blockIdx.x * 4 + w) * 64;
  const int jj = jj0 + lane;
  const bool has_pt = jj < a.P;
  double acc = 0.0;
  if (jj0 < a.P) {  // whole waves only
    const int jend = min(jj0 + 64, a.P);
    const int eb = a.pt_start[jj0], ee = a.pt_start[jend];
    const int e0 = has_pt ? a.pt_start[jj] : ee, e1 = has_pt ? a.pt_start[jj + 1] : ee;
    double t0 = 0, t1 = 0, t2 = 0;
    // the point's elimination record up front: its loads travel with those of the observations (the camera columns in a
    // loop over their run-time number would be one dependent load after the other)
    double el9[9], elz[LIN_NCB][3];
    {
      const double *el = a.elim + (int64_t)ELIM_STRIDE * (has_pt ? jj : jj0);
#pragma unroll
      for (int k = 0; k < 9; ++k) el9[k] = el[k];
      const double2 *z = reinterpret_cast<const double2 *>(el + 16);
#pragma unroll
      for (int k = 0; k < 3 * LIN_NCB / 2; ++k) {
        const double2 v = (2 * k < 3 * a.ncam) ? z[k] : make_double2(0.0, 0.0);
        (&elz[0][0])[2 * k] = v.x;
        (&elz[0][0])[2 * k + 1] = v.y;
      }
    }
    // a chunk of 64 observations per pass; the station indices and Y rows of the next chunk are requested before this
    // one is worked on
    int sta_n = 0;
    double2 yn[9];
    auto request = [&](int base) {
      const int e = base + lane;
      if (e < ee) {
        sta_n = a.obs_sta[e];
        const double2 *py = reinterpret_cast<const double2 *>(a.Y + (int64_t)BA_Y_STRIDE * e);
#pragma unroll
        for (int k = 0; k < 9; ++k) yn[k] = py[k];
      }
    };
    request(eb);
    for (int base = eb; base < ee; base += 64) {
      const int e = base + lane;
      double c0 = 0, c1 = 0, c2 = 0;
      const int sta = sta_n;
      const double2 y0 = yn[0], y1 = yn[1], y2 = yn[2], y3 = yn[3], y4 = yn[4], y5 = yn[5], y6 = yn[6], y7 = yn[7], y8 = yn[8];
      const bool mine = e < ee;
      const double2 *pv = reinterpret_cast<const double2 *>(a.vec + 6 * (int64_t)(mine ? sta : 0));
      const double2 v0 = pv[0], v1 = pv[1], v2 = pv[2];
      if (base + 64 < ee) request(base + 64);
      if (mine) {
        // Y row-major 6 x 3: row r at 3 r
        c0 = y0.x * v0.x + y1.y * v0.y + y3.x * v1.x + y4.y * v1.y + y6.x * v2.x + y7.y * v2.y;
        c1 = y0.y * v0.x + y2.x * v0.y + y3.y * v1.x + y5.x * v1.y + y6.y * v2.x + y8.x * v2.y;
        c2 = y1.x * v0.x + y2.y * v0.y + y4.x * v1.x + y5.y * v1.y + y7.x * v2.x + y8.y * v2.y;
      }
      lds_c[w][3 * lane] = c0;
      lds_c[w][3 * lane + 1] = c1;
      lds_c[w][3 * lane + 2] = c2;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int q_lo = max(e0, base), q_hi = min(e1, base + 64);
      for (int q = q_lo; q < q_hi; ++q) {  // this point's observations of the chunk, in order
        t0 += lds_c[w][3 * (q - base)];
        t1 += lds_c[w][3 * (q - base) + 1];
        t2 += lds_c[w][3 * (q - base) + 2];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (has_pt) {
      const double u0 = el9[0], u1 = el9[1], u2 = el9[2];
      const double l00 = el9[3], l10 = el9[4], l20 = el9[5], l11 = el9[6], l21 = el9[7], l22 = el9[8];
      double y0 = u0 - l00 * t0, y1 = u1 - (l10 * t0 + l11 * t1), y2 = u2 - (l20 * t0 + l21 * t1 + l22 * t2);
      const double *yc = a.vec + 6 * (int64_t)a.S;
#pragma unroll
      for (int k = 0; k < LIN_NCB; ++k)
        if (k < a.ncam) {  // the record holds LIN_NCB camera columns at most
          const double c = yc[k];
          y0 -= elz[k][0] * c;
          y1 -= elz[k][1] * c;
          y2 -= elz[k][2] * c;
        }
      double *yo = a.y_pt + 3 * (int64_t)jj;
      yo[0] = y0; yo[1] = y1; yo[2] = y2;
      const double *g = a.g_pt + 3 * (int64_t)jj, *dg = a.diag_pt + 3 * (int64_t)jj;
      const double ir = 1.0 / a.radius;
      // (the sums carry the sign of the row-wise form sum m (r + m / 2), m = J step: minus the model cost change)
      acc = 0.5 * ((g[0] * u0 + g[1] * u1 + g[2] * u2) - (dg[0] * ir * y0 * y0 + dg[1] * ir * y1 * y1 + dg[2] * ir * y2 * y2)) -
            (g[0] * y0 + g[1] * y1 + g[2] * y2);
    }
  }
  const double t = block_sum(acc, lds);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
}

// camera-side part of the same sum: out[slot] += (y^T S y - sum d2 y^2) / 2 - g.y  (ytSy may be null: a second
// call for a second stretch of the vector)
__global__ void __launch_bounds__(1024)
k_mcc_reduced(int64_t nr, const double *__restrict__ g, const double *__restrict__ y, const double *__restrict__ d2,
              const double *__restrict__ ytSy, double *__restrict__ out, int slot) {
  __shared__ double lds[16];
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < nr; k += blockDim.x) s += y[k] * (g[k] + 0.5 * d2[k] * y[k]);
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0) out[slot] += (ytSy ? 0.5 * ytSy[0] : 0.0) - t;
}

// ------------------------------------------------------------------------------------------
// station sweep
// ------------------------------------------------------------------------------------------
// One workgroup per chunk (<= BA_CHUNK observations of one station).
//   SMODE_GATHER     out[chunk][0..5]            = sum Js^T v          (v = rec[voff..voff+1])
//   SMODE_GATHER_CAM out[chunk][0..5], [6..6+nc) = sum Js^T v, sum Jc^T v
//   SMODE_SETUP      out[chunk][0..20]  = upper triangle of sum Js^T M Js   (preconditioner block)
//                    out[chunk][21..26] = diag of sum Js^T Js                (column norms)
//                    out[chunk][27..32] = sum Js^T r      (gradient)
//                    out[chunk][33..38] = sum Js^T rhat   (reduced right-hand side)
//   SMODE_CAMNORM    out[chunk][0..nc)  = sum of squares of the Jc columns
//   SMODE_CAM2       out[chunk][0..nc), [19..19+nc) = sum Jc^T r, sum Jc^T rhat (camera rows of gradient and reduced rhs in one sweep)
template <int SMODE>
__global__ void __launch_bounds__(256)
k_station_pass(StationArgs a) {
  constexpr int NACC = SMODE == SMODE_GATHER ? 6 : SMODE == SMODE_GATHER_CAM ? 6 + BA_NCAMPAR
                       : SMODE == SMODE_SETUP ? 39 : SMODE == SMODE_CAM2 ? 2 * BA_NCAMPAR : BA_NCAMPAR;
  __shared__ double lds[4 * NACC];
  const int ch = blockIdx.x;
  const int sta = a.chunk_sta[ch];
  const int b0 = a.chunk_begin[ch], b1 = a.chunk_begin[ch + 1];
  double acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
  int nc = 0;
  if (SMODE == SMODE_GATHER_CAM || SMODE == SMODE_CAMNORM || SMODE == SMODE_CAM2) nc = a.cam[a.sta_cam[sta]].nc;
  // the index of a thread's next observation is requested before the records of this one
  int e_next = (b0 + (int)threadIdx.x < b1) ? a.sta_list[b0 + threadIdx.x] : 0;
  for (int q = b0 + threadIdx.x; q < b1; q += blockDim.x) {
    const int e = e_next;
    if (q + (int)blockDim.x < b1) e_next = a.sta_list[q + blockDim.x];
    const double2 *rec = reinterpret_cast<const double2 *>(a.rec + (int64_t)BA_REC_STRIDE * e);
    if (SMODE == SMODE_GATHER || SMODE == SMODE_GATHER_CAM) {
      const double2 v = rec[a.voff];
      const double2 j0 = rec[0], j1 = rec[1], j2 = rec[2], j3 = rec[3], j4 = rec[4], j5 = rec[5];
      acc[0] += j0.x * v.x + j3.x * v.y;
      acc[1] += j0.y * v.x + j3.y * v.y;
      acc[2] += j1.x * v.x + j4.x * v.y;
      acc[3] += j1.y * v.x + j4.y * v.y;
      acc[4] += j2.x * v.x + j5.x * v.y;
      acc[5] += j2.y * v.x + j5.y * v.y;
      if (SMODE == SMODE_GATHER_CAM) {
        const double *jc = a.Jc + 2 * (int64_t)a.ncmax * e;
#pragma unroll
        for (int k = 0; k < BA_NCAMPAR; ++k)
          if (k < nc) acc[6 + k] += jc[k] * v.x + jc[a.ncmax + k] * v.y;
      }
    } else if (SMODE == SMODE_CAM2) {
      const double2 rr = rec[6], rh = rec[7];
      const double *jc = a.Jc + 2 * (int64_t)a.ncmax * e;
#pragma unroll
      for (int k = 0; k < BA_NCAMPAR; ++k)
        if (k < nc) {
          const double j0 = jc[k], j1 = jc[a.ncmax + k];
          acc[k] += j0 * rr.x + j1 * rr.y;
          acc[BA_NCAMPAR + k] += j0 * rh.x + j1 * rh.y;
        }
    } else if (SMODE == SMODE_SETUP) {
      double r0[6], r1[6];
      {
        const double2 j0 = rec[0], j1 = rec[1], j2 = rec[2], j3 = rec[3], j4 = rec[4], j5 = rec[5];
        r0[0] = j0.x; r0[1] = j0.y; r0[2] = j1.x; r0[3] = j1.y; r0[4] = j2.x; r0[5] = j2.y;
        r1[0] = j3.x; r1[1] = j3.y; r1[2] = j4.x; r1[3] = j4.y; r1[4] = j5.x; r1[5] = j5.y;
      }
      const double2 rr = rec[6], rh = rec[7];
      const double *mo = a.Mo + 3 * (int64_t)e;
      const double m00 = mo[0], m01 = mo[1], m11 = mo[2];
      int idx = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const double w0 = m00 * r0[p] + m01 * r1[p], w1 = m01 * r0[p] + m11 * r1[p];
#pragma unroll
        for (int q2 = p; q2 < 6; ++q2) acc[idx++] += w0 * r0[q2] + w1 * r1[q2];
      }
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        acc[21 + p] += r0[p] * r0[p] + r1[p] * r1[p];
        acc[27 + p] += r0[p] * rr.x + r1[p] * rr.y;
        acc[33 + p] += r0[p] * rh.x + r1[p] * rh.y;
      }
    } else {
      const double *jc = a.Jc + 2 * (int64_t)a.ncmax * e;
#pragma unroll
      for (int k = 0; k < BA_NCAMPAR; ++k)
        if (k < nc) acc[k] += jc[k] * jc[k] + jc[a.ncmax + k] * jc[a.ncmax + k];
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  wave_sums_to<NACC>(acc, lds + w * NACC, lane);
  __syncthreads();
  if (threadIdx.x < NACC) {
    const int k = threadIdx.x;
    a.out[(int64_t)a.out_stride * ch + k] = lds[k] + lds[NACC + k] + lds[2 * NACC + k] + lds[3 * NACC + k];
  }
}

// Folds the chunks of each station in a fixed order: out[i][0..ncols) = sum_chunks part[ch][0..ncols)
__global__ void k_station_fold(int S, const int *__restrict__ sta_chunk0, const double *__restrict__ part, int stride,
                               int ncols, double *__restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)ncols * S) return;
  const int i = (int)(t / ncols), k = (int)(t - (int64_t)ncols * i);
  double s = 0.0;
  for (int ch = sta_chunk0[i]; ch < sta_chunk0[i + 1]; ++ch) s += part[(int64_t)stride * ch + k];
  out[t] = s;
}

// Camera rows after SMODE_GATHER_CAM / SMODE_CAMNORM: one workgroup per camera and column (blockIdx.y) sums, in
// chunk order, the partials of the chunks whose station uses that camera.
__global__ void __launch_bounds__(256)
k_camera_fold(int n_chunks, const int *__restrict__ chunk_sta, const int *__restrict__ sta_cam,
              const CamRec *__restrict__ cam, const double *__restrict__ part, int stride, int coloff,
              double *__restrict__ out_cam) {
  __shared__ double lds[8];
  const int c = blockIdx.x, k = blockIdx.y;
  const int nc = cam[c].nc, off = cam[c].off;
  if (k >= nc) return;  // uniform over the workgroup
  double s = 0.0;
  for (int ch = threadIdx.x; ch < n_chunks; ch += blockDim.x)
    if (sta_cam[chunk_sta[ch]] == c) s += part[(int64_t)stride * ch + coloff + k];
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0) out_cam[off + k] = t;
}

// v[k] += diag[k]: the diagonal term of column k of the reduced matrix after a unit-vector probe
__global__ void k_add_diag_entry(double *__restrict__ v, const double *__restrict__ diag, int64_t k) {
  if (blockIdx.x == 0 && threadIdx.x == 0) v[k] += diag[k];
}

// After SMODE_SETUP and the fold: per station, add the LM diagonal, emit the diagonal block of
// the reduced matrix, gradient, rhs and column norms.
__global__ void k_station_finish(StationFinishArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.S) return;
  double acc[39];
#pragma unroll
  for (int k = 0; k < 39; ++k) acc[k] = a.F[39 * (int64_t)i + k];
  const bool used = a.sta_used[i] != 0;
  double d2[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    a.colnorm_sta[6 * (int64_t)i + k] = acc[21 + k];
    double dg;
    if (a.reuse_diag) {
      dg = a.diag_sta[6 * (int64_t)i + k];
    } else {
      dg = fmin(fmax(acc[21 + k], a.lm_min), a.lm_max);
      a.diag_sta[6 * (int64_t)i + k] = dg;
    }
    const double l = sqrt(dg / a.radius);
    d2[k] = l * l;
    a.d2_sta[6 * (int64_t)i + k] = d2[k];
    a.g_sta[6 * (int64_t)i + k] = acc[27 + k];
    a.b_sta[6 * (int64_t)i + k] = used ? acc[33 + k] : 0.0;
  }
  // symmetric 6x6 diagonal block of the reduced matrix: sum Js^T M Js + LM diagonal
  double *sd = a.Sdiag + 36 * (int64_t)i;
  {
    int idx = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int q = p; q < 6; ++q) {
        const double v = acc[idx] + (p == q ? d2[p] : 0.0);
        sd[6 * p + q] = v;
        sd[6 * q + p] = v;
        ++idx;
      }
  }
}

// Inverse of the 6x6 diagonal blocks (block-Jacobi preconditioner).  Block i is read at
// src + 36 * (idx ? idx[i] : i).  Stations without observations get a zero block.
__global__ void k_invert_blocks(int S, const double *__restrict__ src, const int *__restrict__ idx,
                                const int *__restrict__ sta_used, double *__restrict__ Minv, unsigned *flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const double *sp = src + 36 * (int64_t)(idx ? idx[i] : i);
  double A[6][6];
#pragma unroll
  for (int p = 0; p < 6; ++p)
#pragma unroll
    for (int q = 0; q < 6; ++q) A[p][q] = sp[6 * p + q];
  const bool used = sta_used[i] != 0;
  double L[6][6];
  bool ok = used;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (k < j) d -= L[j][k] * L[j][k];
    if (!(d > 0.0)) ok = false;
    const double ld = sqrt(d);
    L[j][j] = ld;
    const double il = 1.0 / ld;
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (r > j) {
        double s = A[r][j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (k < j) s -= L[r][k] * L[j][k];
        L[r][j] = s * il;
      }
  }
  double *mi = Minv + 36 * (int64_t)i;
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 36; ++k) mi[k] = 0.0;
    if (used) atomicOr(flags, 2);
    return;
  }
  // inverse column by column: solve L L^T x = e_c
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double y[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (k < r) s -= L[r][k] * y[k];
      y[r] = s / L[r][r];
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
      double s = y[r];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (k > r) s -= L[k][r] * y[k];
      y[r] = s / L[r][r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) mi[6 * r + c] = y[r];
  }
}

// ------------------------------------------------------------------------------------------
// cost at a candidate point (residuals only), src/main.cpp:1322-1388 + Huber
// ------------------------------------------------------------------------------------------
template <bool GEN, bool EMP, bool ONECAM>
__global__ void __launch_bounds__(256)
k_cost_obs(CostArgs a) {
  __shared__ double lds[8];
  double cost = 0.0;
  // the record and point index of a thread's next observation are requested before this one is evaluated
  const int64_t first = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  ObsIn o_next{};
  int jj_next = 0;
  if (first < a.N) {
    o_next = load_obs(a.obs, first);
    jj_next = a.obs_pt[first];
  }
  for (int64_t e = first; e < a.N; e += step) {
    const ObsIn o = o_next;
    const int jj = jj_next;
    if (e + step < a.N) {
      o_next = load_obs(a.obs, e + step);
      jj_next = a.obs_pt[e + step];
    }
    const double *st = a.sta + (int64_t)BA_STA_STRIDE * o.sta;
    const CamIn cm = load_cam(a.cam, ONECAM ? 0 : __double2loint(st[20]));
    const double *x = a.xyz + 3 * (int64_t)jj;
    double r0, r1;
    eval_obs<false, GEN, EMP>(o, st, x[0], x[1], x[2], cm, r0, r1, nullptr, nullptr, nullptr, nullptr, a.model);
    double sc;
    const double rho = huber_rho(a.huber_a, r0 * r0 + r1 * r1, sc);
    cost += 0.5 * rho;
    if (a.res_out) {
      const double f = a.apply_loss ? sc : 1.0;
      a.res_out[2 * (int64_t)o.row] = f * r0;
      a.res_out[2 * (int64_t)o.row + 1] = f * r1;
    }
  }
  const double t = block_sum(cost, lds);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = t;
}

__global__ void __launch_bounds__(256)
k_cost_prior(int P, const double *__restrict__ xyz, const double *__restrict__ prior, const double *__restrict__ isig,
             const int *__restrict__ pt_file, double *__restrict__ res_out /* file order, may be null */,
             double *__restrict__ partial) {
  __shared__ double lds[8];
  double cost = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < 3 * (int64_t)P; t += (int64_t)gridDim.x * blockDim.x) {
    const double r = (xyz[t] - prior[t]) * isig[t];
    cost += 0.5 * r * r;
    if (res_out) {
      const int64_t jj = t / 3;
      res_out[3 * (int64_t)pt_file[jj] + (t - 3 * jj)] = r;
    }
  }
  const double s = block_sum(cost, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[slot] = sum partial[0..n)
__global__ void __launch_bounds__(1024)
k_sum_partials(const double *__restrict__ partial, int n, double *__restrict__ out, int slot) {
  __shared__ double lds[16];
  double s = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) s += partial[k];
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0) out[slot] = t;
}

// IOP prior blocks (src/main.cpp:4626-4629): out[slot] += cost of the variable blocks,
// out[fixed_slot] = cost of the constant ones; residuals (file order) optionally written.
__global__ void k_camera_prior_cost(int C, const CamRec *__restrict__ cam, double *__restrict__ out, int slot,
                                    int fixed_slot, double *__restrict__ res_out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double v = 0.0, fx = 0.0;
  for (int c = 0; c < C; ++c) {
    double q = 0.0;
    for (int k = 0; k < 3; ++k) {
      const double r = (cam[c].iop[k] - cam[c].prior[k]) * cam[c].isig[k];
      q += 0.5 * r * r;
      if (res_out) res_out[3 * c + k] = r;
    }
    if (cam[c].in_program & 1) v += q; else fx += q;
  }
  out[slot] += v;
  if (fixed_slot >= 0) out[fixed_slot] = fx;
}

__global__ void __launch_bounds__(1024)
k_max_partials(const double *__restrict__ partial, int n, double *__restrict__ out, int slot) {
  __shared__ double lds[16];
  double s = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) s = fmax(s, partial[k]);
  const double t = block_max(s, lds);
  if (threadIdx.x == 0) out[slot] = t;
}

// max |g_scaled / scale| over a vector
__global__ void __launch_bounds__(256)
k_absmax_ratio(int64_t n, const double *__restrict__ g, const double *__restrict__ scale, const double *__restrict__ x,
               double *__restrict__ partial) {
  // max |x - Plus(x, -gradient)|, gradient = g / scale  [Ceres 1.13 TrustRegionMinimizer::EvaluateGradientAndJacobian]:
  // the gradient as it survives the addition to x, not the gradient itself
  __shared__ double lds[8];
  double m = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const double xv = x[t], moved = xv + (-(g[t] / scale[t]));
    m = fmax(m, fabs(xv - moved));
  }
  const double r = block_max(m, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// the same over the camera blocks, where Plus also projects xp yp c onto their box.  One thread: a few values.
__global__ void k_camera_gradient_step(int C, const CamRec *__restrict__ cam, const double *__restrict__ g_cam,
                                       double *__restrict__ out, int slot) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double m = 0.0;
  for (int c = 0; c < C; ++c) {
    const CamRec &r = cam[c];
    double step[BA_NCAMPAR];
    int pos = 0;
    for (int p = 0; p < BA_NCAMPAR; ++p) {
      step[p] = 0.0;
      if (r.mask & (1u << p)) {
        step[p] = -(g_cam[r.off + pos] / r.scale[pos]);
        ++pos;
      }
    }
    if (r.in_program & 1)
      for (int k = 0; k < 3; ++k) {
        const double v = fmin(fmax(r.iop[k] + step[k], r.lo[k]), r.hi[k]);
        m = fmax(m, fabs(r.iop[k] - v));
      }
    if (r.in_program & 2)
      for (int k = 0; k < 16; ++k) m = fmax(m, fabs(r.ap[k] - (r.ap[k] + step[3 + k])));
  }
  out[slot] = m;
}

__global__ void __launch_bounds__(256)
k_dot(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ partial) {
  __shared__ double lds[8];
  double acc = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) acc += a[t] * b[t];
  const double r = block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void __launch_bounds__(256)
k_absmax_prod(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ partial) {
  __shared__ double lds[8];
  double m = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) m = fmax(m, fabs(a[t] * b[t]));
  const double r = block_max(m, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// scale = 1 / (1 + sqrt(colnorm))   [Ceres 1.13 TrustRegionMinimizer::EvaluateGradientAndJacobian]
__global__ void k_make_scale(int64_t n, const double *__restrict__ colnorm, double *__restrict__ scale) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) scale[t] = 1.0 / (1.0 + sqrt(colnorm[t]));
}

__global__ void k_fill(int64_t n, double *__restrict__ v, double val) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) v[t] = val;
}

// ------------------------------------------------------------------------------------------
// candidate point  Plus(x, alpha * delta),  delta = -y * scale  on the free coordinates; xp yp c are projected
// onto their box  [Ceres 1.13 ParameterBlock::Plus]
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_candidate(CandArgs a) {
  __shared__ double lds[8];
  // sums over the points (local to a shard) and over stations + cameras (replicated) are kept apart
  double step2 = 0.0, norm2 = 0.0, pstep2 = 0.0, pnorm2 = 0.0;
  const int64_t n_s = 6 * (int64_t)a.S, n_p = 3 * (int64_t)a.P;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_s + n_p; t += (int64_t)gridDim.x * blockDim.x) {
    if (t < n_s) {
      const int i = (int)(t / 6);
      const bool used = a.sta_used[i] != 0;
      const double d = used ? a.alpha * (-a.y_cam[t] * a.scale_sta[t]) : 0.0;
      const double v = a.eop[t] + d;
      a.eop_c[t] = v;
      // the step as Ceres measures it: x - candidate, not delta  [TrustRegionMinimizer::ParameterToleranceReached]
      const double m = a.eop[t] - v;
      if (used) { step2 += m * m; norm2 += v * v; }
    } else {
      const int64_t u = t - n_s;
      const double d = a.alpha * (-a.y_pt[u] * a.scale_pt[u]);
      const double v = a.xyz[u] + d;
      a.xyz_c[u] = v;
      const double m = a.xyz[u] - v;
      pstep2 += m * m;
      pnorm2 += v * v;
    }
  }
  // cameras: a handful of values, done by one thread
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int c = 0; c < a.C; ++c) {
      CamRec r = a.cam[c];
      int pos = 0;
      for (int p = 0; p < BA_NCAMPAR; ++p) {
        if (r.mask & (1u << p)) {
          const double d = a.alpha * (-a.y_cam[n_s + r.off + pos] * r.scale[pos]);
          if (p < 3) r.iop[p] += d;
          else {
            r.ap[p - 3] += d;
            const double m = a.cam[c].ap[p - 3] - r.ap[p - 3];
            step2 += m * m;
          }
          ++pos;
        }
      }
      for (int k = 0; k < 3; ++k) {
        if (r.in_program & 1) r.iop[k] = fmin(fmax(r.iop[k], r.lo[k]), r.hi[k]);
        const double moved = r.iop[k] - a.cam[c].iop[k];  // the step as taken, after the projection
        step2 += moved * moved;
      }
      if (r.in_program & 1) for (int k = 0; k < 3; ++k) norm2 += r.iop[k] * r.iop[k];
      if (r.in_program & 2) for (int k = 0; k < 16; ++k) norm2 += r.ap[k] * r.ap[k];
      a.cam_c[c] = r;
    }
  }
  const double s = block_sum(step2, lds);
  const double n = block_sum(norm2, lds);
  const double ps = block_sum(pstep2, lds);
  const double pn = block_sum(pnorm2, lds);
  if (threadIdx.x == 0) {
    a.partial[blockIdx.x] = s;
    a.partial[gridDim.x + blockIdx.x] = n;
    a.partial[2 * gridDim.x + blockIdx.x] = ps;
    a.partial[3 * gridDim.x + blockIdx.x] = pn;
  }
}

// |x|^2 over the ambient coordinates of the blocks in the reduced program
__global__ void __launch_bounds__(256)
k_xnorm(CandArgs a) {
  __shared__ double lds[8];
  double norm2 = 0.0, pnorm2 = 0.0;
  const int64_t n_s = 6 * (int64_t)a.S, n_p = 3 * (int64_t)a.P;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_s + n_p; t += (int64_t)gridDim.x * blockDim.x) {
    if (t < n_s) {
      const int i = (int)(t / 6);
      if (a.sta_used[i]) norm2 += a.eop[t] * a.eop[t];
    } else {
      pnorm2 += a.xyz[t - n_s] * a.xyz[t - n_s];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int c = 0; c < a.C; ++c) {
      const CamRec &r = a.cam[c];
      if (r.in_program & 1) for (int k = 0; k < 3; ++k) norm2 += r.iop[k] * r.iop[k];
      if (r.in_program & 2) for (int k = 0; k < 16; ++k) norm2 += r.ap[k] * r.ap[k];
    }
  }
  const double n = block_sum(norm2, lds);
  const double pn = block_sum(pnorm2, lds);
  if (threadIdx.x == 0) {
    a.partial[blockIdx.x] = n;
    a.partial[gridDim.x + blockIdx.x] = pn;
  }
}

// camera prior rows: column norms, gradient, rhs, LM diagonal.  One thread; a few values.
__global__ void k_camera_finish(CamFinishArgs a) {
  // a thread per camera and parameter (a single thread walking them is a chain of dependent memory accesses: 0.17 ms for
  // four cameras of twelve free parameters)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.C * BA_NCAMPAR) return;
  {
    const int c = t / BA_NCAMPAR, p = t - c * BA_NCAMPAR;
    const CamRec &r = a.cam[c];
    {
      if (!(r.mask & (1u << p))) return;
      const int pos = __popc(r.mask & ((1u << p) - 1u));  // free parameters before this one
      const int col = r.off + pos;
      double jc = 0.0, rc = 0.0;
      if (p < 3) {
        jc = r.isig[p] * r.scale[pos];
        rc = (r.iop[p] - r.prior[p]) * r.isig[p];
      }
      a.jc2[col] = jc * jc;
      const double cn = (a.colnorm_in ? a.colnorm_in[col] : 0.0) + jc * jc;
      a.colnorm_cam[col] = cn;
      double dg;
      if (a.reuse_diag) {
        dg = a.diag_cam[col];
      } else {
        dg = a.need_norms ? fmin(fmax(cn, a.lm_min), a.lm_max) : a.lm_min;
        a.diag_cam[col] = dg;
      }
      const double l = sqrt(dg / a.radius);
      a.d2_cam[col] = l * l;
      a.camdiag[col] = jc * jc + l * l;
      a.g_cam[col] += jc * rc;  // g_cam / b_cam hold the image-row sums on entry
      a.b_cam[col] += jc * rc;
    }
  }
}

// model cost change contribution of the camera prior rows, added to out[slot]
__global__ void k_camera_mcc(int C, const CamRec *__restrict__ cam, const double *__restrict__ y_cam,
                             double *__restrict__ out, int slot) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double acc = 0.0;
  for (int c = 0; c < C; ++c) {
    const CamRec &r = cam[c];
    int pos = 0;
    for (int p = 0; p < BA_NCAMPAR; ++p) {
      if (!(r.mask & (1u << p))) continue;
      if (p < 3) {
        const double m = -r.isig[p] * r.scale[pos] * y_cam[r.off + pos];
        acc += m * ((r.iop[p] - r.prior[p]) * r.isig[p] + 0.5 * m);
      }
      ++pos;
    }
  }
  out[slot] += acc;
}

// ------------------------------------------------------------------------------------------
// PCG vector phase: one 1024-thread workgroup does everything between two operator
// applications (the vectors have 6S + ncam entries: a few hundred KB at most).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void precond(const CgArgs &a, const double *__restrict__ r, double *__restrict__ z) {
  for (int i = threadIdx.x; i < a.S; i += blockDim.x) {
    const double *m = a.Minv + 36 * (int64_t)i;
    const double *ri = r + 6 * (int64_t)i;
    const double r0 = ri[0], r1 = ri[1], r2 = ri[2], r3 = ri[3], r4 = ri[4], r5 = ri[5];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      z[6 * (int64_t)i + k] = m[6 * k] * r0 + m[6 * k + 1] * r1 + m[6 * k + 2] * r2 + m[6 * k + 3] * r3 +
                              m[6 * k + 4] * r4 + m[6 * k + 5] * r5;
  }
  if (a.ncam > 0 && threadIdx.x == 0) {
    // dense Cholesky factor L (row-major, lower) of the camera block
    const int n = a.ncam;
    const double *L = a.Lcc;
    double *zc = z + 6 * (int64_t)a.S;
    const double *rc = r + 6 * (int64_t)a.S;
    for (int i = 0; i < n; ++i) {
      double s = rc[i];
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * zc[k];
      zc[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = zc[i];
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * zc[k];
      zc[i] = s / L[i * n + i];
    }
  }
}

__device__ __forceinline__ double block_dot(const double *__restrict__ x, const double *__restrict__ y, int n, double *lds,
                                            double *bcast) {
  double s = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) s += x[k] * y[k];
  const double t = block_sum(s, lds);
  if (threadIdx.x == 0) *bcast = t;
  __syncthreads();
  const double v = *bcast;
  __syncthreads();
  return v;
}

// x = 0, r = b, z = M^-1 r, p = z, rz = r.z
__global__ void __launch_bounds__(1024)
k_cg_init(CgArgs a) {
  __shared__ double lds[16];
  __shared__ double bc;
  const int n = a.n;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    a.x[k] = 0.0;
    a.r[k] = a.b[k];
  }
  __syncthreads();
  precond(a, a.r, a.z);
  __syncthreads();
  const double rz = block_dot(a.r, a.z, n, lds, &bc);
  for (int k = threadIdx.x; k < n; k += blockDim.x) a.p[k] = a.z[k];
  if (threadIdx.x == 0) {
    a.scal[0] = rz;   // current r.z
    a.scal[1] = rz;   // initial r.z
    a.scal[2] = 0.0;  // iterations done
  }
}

// q holds the implicit station part of S p; add the explicit camera border, then one CG update.
__global__ void __launch_bounds__(1024)
k_cg_step(CgArgs a) {
  __shared__ double lds[16];
  __shared__ double bc;
  const int n = a.n, ns = 6 * a.S, nc = a.ncam;
  // LM diagonal of the station columns (the implicit operator carries only Js^T M Js)
  for (int t = threadIdx.x; t < ns; t += blockDim.x) a.q[t] += a.d2_sta[t] * a.p[t];
  __syncthreads();
  // border:  q_s += Ssc p_c ;  q_c = Ssc^T p_s + Scc p_c      (Ssc row-major [ns x nc])
  if (nc > 0) {
    const double *pc = a.p + ns;
    for (int k = 0; k < nc; ++k) {
      double s = 0.0;
      for (int t = threadIdx.x; t < ns; t += blockDim.x) s += a.Ssc[(int64_t)t * nc + k] * a.p[t];
      const double d = block_sum(s, lds);
      if (threadIdx.x == 0) {
        double v = d;
        for (int m = 0; m < nc; ++m) v += a.Scc[k * nc + m] * pc[m];
        a.q[ns + k] = v;
      }
    }
    for (int t = threadIdx.x; t < ns; t += blockDim.x) {
      double s = a.q[t];
      for (int k = 0; k < nc; ++k) s += a.Ssc[(int64_t)t * nc + k] * pc[k];
      a.q[t] = s;
    }
    __syncthreads();
  }
  const double rz = a.scal[0];
  const double pq = block_dot(a.p, a.q, n, lds, &bc);
  const double alpha = rz / pq;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    a.x[k] += alpha * a.p[k];
    a.r[k] -= alpha * a.q[k];
  }
  __syncthreads();
  precond(a, a.r, a.z);
  __syncthreads();
  const double rzn = block_dot(a.r, a.z, n, lds, &bc);
  const double beta = rzn / rz;
  for (int k = threadIdx.x; k < n; k += blockDim.x) a.p[k] = a.z[k] + beta * a.p[k];
  if (threadIdx.x == 0) {
    a.scal[0] = rzn;
    a.scal[2] += 1.0;
    a.scal[3] = pq;
  }
}

// new constant corrections (file order) into the observation records (point-sorted order)
__global__ void k_set_corrections(int64_t N, double *__restrict__ obs, const double *__restrict__ cx, const double *__restrict__ cy) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= N) return;
  double *o = obs + e * BA_OBS_STRIDE;
  const int row = __double2hiint(o[6]);
  o[4] = cx[row];
  o[5] = cy[row];
}

// copy helpers ------------------------------------------------------------------------------
__global__ void k_copy(int64_t n, const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) dst[t] = src[t];
}
__global__ void k_unit(int64_t n, double *__restrict__ v, int64_t k) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < n) v[t] = (t == k) ? 1.0 : 0.0;
}

// Per-observation residuals and dense 2x28 Jacobians in file order (no loss, no scaling):
// the counterpart of AutoDiffCostFunction<collinearity,2,6,3,3,16>::Evaluate, src/main.cpp:3845-3847.
template <bool GEN>
__global__ void __launch_bounds__(256)
k_resjac_dense(CostArgs a, double *__restrict__ r_out, double *__restrict__ J_out) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= a.N) return;
  const ObsIn o = load_obs(a.obs, e);
  const double *st = a.sta + (int64_t)BA_STA_STRIDE * o.sta;
  const CamIn cm = load_cam(a.cam, __double2loint(st[20]));
  const double *x = a.xyz + 3 * (int64_t)a.obs_pt[e];
  double r0, r1, Js[12], Jp[6], dc0[BA_NCAMPAR], dc1[BA_NCAMPAR];
  eval_obs<true, GEN>(o, st, x[0], x[1], x[2], cm, r0, r1, Js, Jp, dc0, dc1, a.model);
  if (r_out) {
    r_out[2 * (int64_t)o.row] = r0;
    r_out[2 * (int64_t)o.row + 1] = r1;
  }
  if (J_out) {
    double *J = J_out + 56 * (int64_t)o.row;
#pragma unroll
    for (int k = 0; k < 6; ++k) { J[k] = Js[k]; J[28 + k] = Js[6 + k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { J[6 + k] = Jp[k]; J[28 + 6 + k] = Jp[3 + k]; }
#pragma unroll
    for (int k = 0; k < BA_NCAMPAR; ++k) { J[9 + k] = dc0[k]; J[28 + 9 + k] = dc1[k]; }
  }
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// covariance of the parameters (src/main.cpp:5261-5279, 5443-5565)
// ------------------------------------------------------------------------------------------
// With the points eliminated, Cx of the camera-side parameters is S^-1 and that of point j is
//   V^-1 + V^-1 W^T S^-1 W V^-1,   W V^-1 = sum_a [Js_a | Jc_a]^T (Jp_a V^-1),
// everything in the Jacobi-scaled variables; the scales are undone at the end.
__global__ void k_cov_camside(CovArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n) return;
  const double v = a.Sinv[(int64_t)t * a.ld + t];
  const int ns = 6 * a.S;
  double sc;
  if (t < ns) sc = a.sta_used[t / 6] ? a.scale_sta[t] : 0.0;
  else sc = a.scale_cam[t - ns];
  a.var_cs[t] = v * sc * sc;
}

__global__ void __launch_bounds__(64)
k_cov_points(CovArgs a) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj >= a.P) return;
  if (a.pt_home && a.pt_home[jj] != a.home) return;  // another local matrix holds this point's blocks
  const int e0 = a.pt_start[jj], e1 = a.pt_start[jj + 1];
  const int ns = 6 * a.S, ncam = a.n - ns;
  const double *vi = a.Vinv + 6 * (int64_t)jj;
  const double Vi[6] = {vi[0], vi[1], vi[2], vi[3], vi[4], vi[5]};
  double *Tc = a.Tc + (int64_t)jj * ncam * 3;
  for (int k = 0; k < 3 * ncam; ++k) Tc[k] = 0.0;
  double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  // T_a = Js_a^T Q_a (6 x 3), Q_a = Jp_a V^-1 (2 x 3)
  auto station_part = [&](int e, double T[6][3], double Q[2][3]) {
    const double *rec = a.rec + (int64_t)BA_REC_STRIDE * e, *jp = a.Jp + 6 * (int64_t)e;
    sym3_mul(Vi, jp[0], jp[1], jp[2], Q[0][0], Q[0][1], Q[0][2]);
    sym3_mul(Vi, jp[3], jp[4], jp[5], Q[1][0], Q[1][1], Q[1][2]);
    for (int r = 0; r < 6; ++r)
      for (int m = 0; m < 3; ++m) T[r][m] = rec[r] * Q[0][m] + rec[6 + r] * Q[1][m];
  };
  for (int ea = e0; ea < e1; ++ea) {
    double Ta[6][3], Qa[2][3];
    station_part(ea, Ta, Qa);
    const int sa = a.obs_sta[ea], ia = a.sta_loc ? a.sta_loc[sa] : sa;
    const int c = a.sta_cam[sa], nc = a.cam[c].nc, off = a.cam[c].off;
    const double *jc = a.Jc + 2 * (int64_t)a.ncmax * ea;
    for (int k = 0; k < nc; ++k)
      for (int m = 0; m < 3; ++m) Tc[3 * (off + k) + m] += jc[k] * Qa[0][m] + jc[a.ncmax + k] * Qa[1][m];
    for (int eb = e0; eb < e1; ++eb) {
      double Tb[6][3], Qb[2][3];
      station_part(eb, Tb, Qb);
      const int sb = a.obs_sta[eb], ib = a.sta_loc ? a.sta_loc[sb] : sb;
      for (int r = 0; r < 6; ++r) {
        const double *row = a.Sinv + (int64_t)(6 * ia + r) * a.ld + 6 * ib;
        double u[3] = {0, 0, 0};  // row r of Sinv[ia, ib] times T_b
        for (int q = 0; q < 6; ++q)
          for (int m = 0; m < 3; ++m) u[m] += row[q] * Tb[q][m];
        for (int l = 0; l < 3; ++l)
          for (int m = 0; m < 3; ++m) acc[l][m] += Ta[r][l] * u[m];
      }
    }
  }
  // camera columns: 2 T_s^T Sinv[s, c] T_c + T_c^T Sinv[c, c] T_c
  for (int ea = e0; ea < e1; ++ea) {
    double Ta[6][3], Qa[2][3];
    station_part(ea, Ta, Qa);
    const int sa = a.obs_sta[ea], ia = a.sta_loc ? a.sta_loc[sa] : sa;
    for (int r = 0; r < 6; ++r) {
      const double *row = a.Sinv + (int64_t)(6 * ia + r) * a.ld + ns;
      double u[3] = {0, 0, 0};
      for (int k = 0; k < ncam; ++k)
        for (int m = 0; m < 3; ++m) u[m] += row[k] * Tc[3 * k + m];
      for (int l = 0; l < 3; ++l)
        for (int m = 0; m < 3; ++m) acc[l][m] += Ta[r][l] * u[m] + u[l] * Ta[r][m];
    }
  }
  for (int k = 0; k < ncam; ++k) {
    const double *row = a.Sinv + (int64_t)(ns + k) * a.ld + ns;
    double u[3] = {0, 0, 0};
    for (int q = 0; q < ncam; ++q)
      for (int m = 0; m < 3; ++m) u[m] += row[q] * Tc[3 * q + m];
    for (int l = 0; l < 3; ++l)
      for (int m = 0; m < 3; ++m) acc[l][m] += Tc[3 * k + l] * u[m];
  }
  const double *sp = a.scale_pt + 3 * (int64_t)jj;
  double *out = a.var_xyz + 3 * (int64_t)a.pt_file[jj];
  out[0] = (Vi[0] + acc[0][0]) * sp[0] * sp[0];
  out[1] = (Vi[3] + acc[1][1]) * sp[1] * sp[1];
  out[2] = (Vi[5] + acc[2][2]) * sp[2] * sp[2];
}

// stations without observations are not part of the program: unit diagonal keeps the factorisation defined
__global__ void k_dense_fix_unused(int S, int ld, const int *__restrict__ sta_used, double *__restrict__ A) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 6 * S && !sta_used[t / 6]) A[(int64_t)t * ld + t] = 1.0;
}

static inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

void launch_station_prep(hipStream_t s, int S, const double *eop, const double *scale_sta, const int *sta_cam, double *sta) {
  if (S > 0) hipLaunchKernelGGL(k_station_prep, dim3(nblk(S, 128)), dim3(128), 0, s, S, eop, scale_sta, sta_cam, sta);
}
void launch_cov(hipStream_t s, const CovArgs &a) {
  if (a.n > 0) hipLaunchKernelGGL(k_cov_camside, dim3(nblk(a.n, 256)), dim3(256), 0, s, a);
  if (a.P > 0) hipLaunchKernelGGL(k_cov_points, dim3(nblk(a.P, 64)), dim3(64), 0, s, a);
}
void launch_cov_points(hipStream_t s, const CovArgs &a) {
  if (a.P > 0) hipLaunchKernelGGL(k_cov_points, dim3(nblk(a.P, 64)), dim3(64), 0, s, a);
}
void launch_dense_fix_unused(hipStream_t s, int S, int ld, const int *sta_used, double *A) {
  if (S > 0) hipLaunchKernelGGL(k_dense_fix_unused, dim3(nblk(6 * (int64_t)S, 256)), dim3(256), 0, s, S, ld, sta_used, A);
}
int linearize_blocks(int P) { return nblk(P, 256); }
int obs_rows_blocks(int64_t N) {
  const int b = nblk((N + 63) / 64, 4);
  return b < 1 ? 1 : (b > 4096 ? 4096 : b);
}
// part: 1 = per-point kernel, 2 = per-observation kernel, 3 = both (a linearisation)
void launch_linearize(hipStream_t s, const LinArgs &a, int part) {
  if (a.P <= 0) return;
  const dim3 gp(nblk(a.P, 256)), go(obs_rows_blocks(a.N)), b(256);
  const bool gen = a.model != 0;
  // kernel variants: [single-camera fast path][general projection][empirical terms in use]
  static void (*const elim_tab[2][2][2])(LinArgs) = {
      {{k_point_elim<false, false, false>, k_point_elim<false, false, true>}, {k_point_elim<false, true, false>, k_point_elim<false, true, true>}},
      {{k_point_elim<true, false, false>, k_point_elim<true, false, true>}, {k_point_elim<true, true, false>, k_point_elim<true, true, true>}}};
  static void (*const rows_tab[2][2][2])(LinArgs) = {
      {{k_obs_rows<false, false, false, false>, k_obs_rows<false, false, true, false>}, {k_obs_rows<false, true, false, false>, k_obs_rows<false, true, true, false>}},
      {{k_obs_rows<true, false, false, false>, k_obs_rows<true, false, true, false>}, {k_obs_rows<true, true, false, false>, k_obs_rows<true, true, true, false>}}};
  static void (*const sums_tab[2][2][2])(LinArgs) = {
      {{k_obs_rows<false, false, false, true>, k_obs_rows<false, false, true, true>}, {k_obs_rows<false, true, false, true>, k_obs_rows<false, true, true, true>}},
      {{k_obs_rows<true, false, false, true>, k_obs_rows<true, false, true, true>}, {k_obs_rows<true, true, false, true>, k_obs_rows<true, true, true, true>}}};
  void (*elim)(LinArgs) = elim_tab[a.Jt ? 1 : 0][gen ? 1 : 0][a.emp ? 1 : 0];
  void (*rows)(LinArgs) = (a.fused ? sums_tab : rows_tab)[a.Jt ? 1 : 0][gen ? 1 : 0][a.emp ? 1 : 0];
  if (part & 1) hipLaunchKernelGGL(elim, gp, b, 0, s, a);
  if (part & 2) hipLaunchKernelGGL(rows, a.fused ? dim3(keyed_blocks(a.kp.n_seg)) : go, b, 0, s, a);
}
int keyed_ncolp(bool cam1) { return cam1 ? 96 : 64; }
int keyed_blocks(int n_seg) { return n_seg < 1 ? 1 : (n_seg + 3) / 4; }
int elim_stride() { return ELIM_STRIDE; }
int lin_ncb() { return LIN_NCB; }
// camera rows of the gradient (out_g) and of the reduced right-hand side (out_b) from the partials
// that k_linearize<true> left; out arrays have ncam entries
void launch_camera_rows_fold(hipStream_t s, const double *cam_partial, int nblocks, int ncam, double *out_g, double *out_b) {
  hipLaunchKernelGGL(k_fold_columns, dim3(LIN_NCB), dim3(256), 0, s, cam_partial, nblocks, LIN_NPART, ncam, out_g);
  hipLaunchKernelGGL(k_fold_columns, dim3(LIN_NCB), dim3(256), 0, s, cam_partial + LIN_NCB, nblocks, LIN_NPART, ncam, out_b);
}
int lin_npart() { return LIN_NPART; }
// corner[0 .. NCB(NCB+1)/2) = upper triangle of sum Jc^T Jt, from the partials that k_obs_rows<true> left
void launch_camera_corner_fold(hipStream_t s, const double *cam_partial, int nblocks, double *corner) {
  hipLaunchKernelGGL(k_fold_columns, dim3(LIN_NCORNER), dim3(256), 0, s, cam_partial + 2 * LIN_NCB, nblocks, LIN_NPART, LIN_NCORNER, corner);
}
void launch_point_pass(hipStream_t s, const PointArgs &a, int mode, bool withcam) {
  if (a.P <= 0) return;
  const dim3 g(nblk(a.P, 256)), b(256);
  if (mode == MODE_APPLY) {
    if (withcam) hipLaunchKernelGGL(k_point_apply<true>, g, b, 0, s, a);
    else hipLaunchKernelGGL(k_point_apply<false>, g, b, 0, s, a);
  } else {
    if (withcam) hipLaunchKernelGGL(k_backsub_tiles<true>, g, b, 0, s, a);
    else hipLaunchKernelGGL(k_backsub_tiles<false>, g, b, 0, s, a);
  }
}
void launch_backsub_y(hipStream_t s, const PointArgs &a) {
  if (a.P > 0) hipLaunchKernelGGL(k_backsub_y, dim3(nblk(a.P, 256)), dim3(256), 0, s, a);
}
void launch_mcc_reduced(hipStream_t s, int64_t nr, const double *g, const double *y, const double *d2, const double *ytSy, double *out, int slot) {
  hipLaunchKernelGGL(k_mcc_reduced, dim3(1), dim3(1024), 0, s, nr, g, y, d2, ytSy, out, slot);
}
void launch_station_pass(hipStream_t s, const StationArgs &a, int smode) {
  if (a.n_chunks <= 0) return;
  const dim3 g(a.n_chunks), b(256);
  switch (smode) {
    case SMODE_GATHER: hipLaunchKernelGGL((k_station_pass<SMODE_GATHER>), g, b, 0, s, a); break;
    case SMODE_GATHER_CAM: hipLaunchKernelGGL((k_station_pass<SMODE_GATHER_CAM>), g, b, 0, s, a); break;
    case SMODE_SETUP: hipLaunchKernelGGL((k_station_pass<SMODE_SETUP>), g, b, 0, s, a); break;
    case SMODE_CAM2: hipLaunchKernelGGL((k_station_pass<SMODE_CAM2>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((k_station_pass<SMODE_CAMNORM>), g, b, 0, s, a); break;
  }
}
void launch_station_fold(hipStream_t s, int S, const int *sta_chunk0, const double *part, int stride, int ncols, double *out) {
  if (S > 0) hipLaunchKernelGGL(k_station_fold, dim3(nblk((int64_t)ncols * S, 256)), dim3(256), 0, s, S, sta_chunk0, part, stride, ncols, out);
}
void launch_camera_fold(hipStream_t s, int C, int n_chunks, const int *chunk_sta, const int *sta_cam, const CamRec *cam,
                        const double *part, int stride, int coloff, double *out_cam) {
  if (C > 0) hipLaunchKernelGGL(k_camera_fold, dim3(C, BA_NCAMPAR), dim3(256), 0, s, n_chunks, chunk_sta, sta_cam, cam, part, stride, coloff, out_cam);
}
void launch_add_diag_entry(hipStream_t s, double *v, const double *diag, int64_t k) {
  hipLaunchKernelGGL(k_add_diag_entry, dim3(1), dim3(1), 0, s, v, diag, k);
}
void launch_camera_prior_cost(hipStream_t s, int C, const CamRec *cam, double *out, int slot, int fixed_slot, double *res_out) {
  hipLaunchKernelGGL(k_camera_prior_cost, dim3(1), dim3(1), 0, s, C, cam, out, slot, fixed_slot, res_out);
}
void launch_station_finish(hipStream_t s, const StationFinishArgs &a) {
  if (a.S > 0) hipLaunchKernelGGL(k_station_finish, dim3(nblk(a.S, 64)), dim3(64), 0, s, a);
}
void launch_invert_blocks(hipStream_t s, int S, const double *src, const int *idx, const int *sta_used, double *Minv, unsigned *flags) {
  if (S > 0) hipLaunchKernelGGL(k_invert_blocks, dim3(nblk(S, 64)), dim3(64), 0, s, S, src, idx, sta_used, Minv, flags);
}
int cost_blocks(int64_t n) {
  const int b = nblk(n, 256);
  return b < 1 ? 1 : (b > 4096 ? 4096 : b);
}
void launch_cost_obs(hipStream_t s, const CostArgs &a, int blocks) {
  static void (*const tab[2][2][2])(CostArgs) = {
      {{k_cost_obs<false, false, false>, k_cost_obs<false, false, true>}, {k_cost_obs<false, true, false>, k_cost_obs<false, true, true>}},
      {{k_cost_obs<true, false, false>, k_cost_obs<true, false, true>}, {k_cost_obs<true, true, false>, k_cost_obs<true, true, true>}}};
  hipLaunchKernelGGL(tab[a.model ? 1 : 0][a.emp ? 1 : 0][a.one_cam ? 1 : 0], dim3(blocks), dim3(256), 0, s, a);
}
void launch_cost_prior(hipStream_t s, int P, const double *xyz, const double *prior, const double *isig, const int *pt_file,
                       double *res_out, double *partial, int blocks) {
  hipLaunchKernelGGL(k_cost_prior, dim3(blocks), dim3(256), 0, s, P, xyz, prior, isig, pt_file, res_out, partial);
}
// Small arrays gathered behind one large one (or scattered back), so that one reduction over the ranks covers them all
__global__ void k_pack_segments(PackList l, double *__restrict__ packed, int gather) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int off = 0;
#pragma unroll
  for (int k = 0; k < BA_PACK_MAX; ++k) {
    const int n = l.n[k];
    if (t >= off && t < off + n) {
      if (gather) packed[t] = l.ptr[k][t - off];
      else l.ptr[k][t - off] = packed[t];
    }
    off += n;
  }
}
void launch_pack_segments(hipStream_t s, const PackList &l, double *packed, int gather) {
  int total = 0;
  for (int k = 0; k < BA_PACK_MAX; ++k) total += l.n[k];
  if (total > 0) hipLaunchKernelGGL(k_pack_segments, dim3(nblk(total, 256)), dim3(256), 0, s, l, packed, gather);
}
// out[slot] = flags[0] as a number (the flag travels with the maximum over the ranks)
__global__ void k_flag_to_scal(const unsigned *__restrict__ flags, double *__restrict__ out, int slot) {
  if (blockIdx.x == 0 && threadIdx.x == 0) out[slot] = (double)flags[0];
}
void launch_flag_to_scal(hipStream_t s, const unsigned *flags, double *out, int slot) {
  hipLaunchKernelGGL(k_flag_to_scal, dim3(1), dim3(64), 0, s, flags, out, slot);
}
void launch_sum_partials(hipStream_t s, const double *partial, int n, double *out, int slot) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, s, partial, n, out, slot);
}
void launch_max_partials(hipStream_t s, const double *partial, int n, double *out, int slot) {
  hipLaunchKernelGGL(k_max_partials, dim3(1), dim3(1024), 0, s, partial, n, out, slot);
}
void launch_absmax_ratio(hipStream_t s, int64_t n, const double *g, const double *scale, const double *x, double *partial, int blocks) {
  hipLaunchKernelGGL(k_absmax_ratio, dim3(blocks), dim3(256), 0, s, n, g, scale, x, partial);
}
void launch_camera_gradient_step(hipStream_t s, int C, const CamRec *cam, const double *g_cam, double *out, int slot) {
  hipLaunchKernelGGL(k_camera_gradient_step, dim3(1), dim3(1), 0, s, C, cam, g_cam, out, slot);
}
void launch_dot(hipStream_t s, int64_t n, const double *a, const double *b, double *partial, int blocks) {
  hipLaunchKernelGGL(k_dot, dim3(blocks), dim3(256), 0, s, n, a, b, partial);
}
void launch_absmax_prod(hipStream_t s, int64_t n, const double *a, const double *b, double *partial, int blocks) {
  hipLaunchKernelGGL(k_absmax_prod, dim3(blocks), dim3(256), 0, s, n, a, b, partial);
}
void launch_make_scale(hipStream_t s, int64_t n, const double *colnorm, double *scale) {
  if (n > 0) hipLaunchKernelGGL(k_make_scale, dim3(nblk(n, 256)), dim3(256), 0, s, n, colnorm, scale);
}
void launch_fill(hipStream_t s, int64_t n, double *v, double val) {
  if (n > 0) hipLaunchKernelGGL(k_fill, dim3(nblk(n, 256)), dim3(256), 0, s, n, v, val);
}
void launch_candidate(hipStream_t s, const CandArgs &a, int blocks) {
  hipLaunchKernelGGL(k_candidate, dim3(blocks), dim3(256), 0, s, a);
}
void launch_xnorm(hipStream_t s, const CandArgs &a, int blocks) {
  hipLaunchKernelGGL(k_xnorm, dim3(blocks), dim3(256), 0, s, a);
}
void launch_camera_finish(hipStream_t s, const CamFinishArgs &a) {
  if (a.C > 0) hipLaunchKernelGGL(k_camera_finish, dim3(nblk((int64_t)a.C * BA_NCAMPAR, 64)), dim3(64), 0, s, a);
}
void launch_camera_mcc(hipStream_t s, int C, const CamRec *cam, const double *y_cam, double *out, int slot) {
  hipLaunchKernelGGL(k_camera_mcc, dim3(1), dim3(1), 0, s, C, cam, y_cam, out, slot);
}
void launch_cg_init(hipStream_t s, const CgArgs &a) { hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(1024), 0, s, a); }
void launch_cg_step(hipStream_t s, const CgArgs &a) { hipLaunchKernelGGL(k_cg_step, dim3(1), dim3(1024), 0, s, a); }
void launch_set_corrections(hipStream_t s, int64_t N, double *obs, const double *cx, const double *cy) {
  if (N > 0) hipLaunchKernelGGL(k_set_corrections, dim3(nblk(N, 256)), dim3(256), 0, s, N, obs, cx, cy);
}
void launch_copy(hipStream_t s, int64_t n, const double *src, double *dst) {
  if (n > 0) hipLaunchKernelGGL(k_copy, dim3(nblk(n, 256)), dim3(256), 0, s, n, src, dst);
}
void launch_unit(hipStream_t s, int64_t n, double *v, int64_t k) {
  if (n > 0) hipLaunchKernelGGL(k_unit, dim3(nblk(n, 256)), dim3(256), 0, s, n, v, k);
}
void launch_resjac_dense(hipStream_t s, const CostArgs &a, double *r_out, double *J_out) {
  void (*k)(CostArgs, double *, double *) = a.model ? k_resjac_dense<true> : k_resjac_dense<false>;
  if (a.N > 0) hipLaunchKernelGGL(k, dim3(nblk(a.N, 256)), dim3(256), 0, s, a, r_out, J_out);
}

}  // namespace ba
