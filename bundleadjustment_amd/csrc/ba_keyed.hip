// ba_keyed.hip -- per-station sums formed inside the observation kernel: the plan and the fold.
//
// The observation kernel (k_obs_rows<..., SUMS> in ba_kernels.hip) walks the observations in point order, 64 at a time,
// one wave per segment of consecutive tiles.  What a station needs from its observations -- sum Js^T M Js, sum Js^T Js,
// sum Js^T r, sum Js^T rhat and the camera border sum Js^T Jt -- is a reduction by station over that order.  The order is
// fixed by the network, so everything that depends on it is prepared here, once per problem:
//   * per tile, the position of every observation in the tile's station-sorted order (spos) -- the lanes put their
//     products into a wave-private LDS buffer at that row, so that rows of one station are adjacent and a lane that owns
//     a column can add them up in a register;
//   * per tile, where the runs of equal station end (umask) and which slot of the wave's LDS table a row belongs to
//     (slot): the table holds one row of sums per station the segment is working on;
//   * per segment, the list of table rows to write out (ops): a row that must make room for another station, and
//     every occupied row at the end of the segment, goes to a partial record in global memory;
//   * per station, the list of its partial records in ascending order: k_keyed_fold adds them in that order.
// Nothing here depends on the numbers, and every sum has one fixed order: results are bitwise reproducible.
#include "ba_kernels.h"

#include <algorithm>
#include <cstring>
#include <thread>

namespace ba {

int keyed_nslot() { return BA_KS_NSLOT; }

namespace {

struct SegOut {
  std::vector<int2> ops;     // x: (tile_rel << 8) | slot, y: record index inside the segment
  std::vector<int> rec_sta;  // station of each record of the segment
  bool ok = true;
};

// One segment: tiles [t0, t1).  Slots are kept by least-recent use.
void plan_segment(int64_t N, const int *e_sta, int64_t t0, int64_t t1, uint8_t *spos, uint8_t *slot, uint32_t *umask, SegOut &out) {
  const int NS = BA_KS_NSLOT;
  int slot_sta[BA_KS_NSLOT];
  int64_t last_use[BA_KS_NSLOT];
  for (int k = 0; k < NS; ++k) { slot_sta[k] = -1; last_use[k] = -1; }
  std::pair<int, int> key[64];
  for (int64_t t = t0; t < t1; ++t) {
    const int64_t e0 = t * 64;
    const int nv = (int)std::min<int64_t>(64, N - e0);
    for (int l = 0; l < nv; ++l) key[l] = {e_sta[e0 + l], l};
    std::sort(key, key + nv);
    // slots of the tile's stations: those already in the table first, so that none of them is pushed out below
    int pos_slot[64];
    int n_distinct = 0;
    for (int i = 0; i < nv; ++i)
      if (i == 0 || key[i].first != key[i - 1].first) ++n_distinct;
    if (n_distinct > NS) { out.ok = false; return; }
    for (int i = 0; i < nv; ++i) {
      if (i > 0 && key[i].first == key[i - 1].first) { pos_slot[i] = pos_slot[i - 1]; continue; }
      pos_slot[i] = -1;
      for (int k = 0; k < NS; ++k)
        if (slot_sta[k] == key[i].first) { pos_slot[i] = k; last_use[k] = t; break; }
    }
    for (int i = 0; i < nv; ++i) {
      if (pos_slot[i] >= 0) continue;
      if (i > 0 && key[i].first == key[i - 1].first) { pos_slot[i] = pos_slot[i - 1]; continue; }
      int pick = -1;
      for (int k = 0; k < NS; ++k)
        if (slot_sta[k] < 0) { pick = k; break; }
      if (pick < 0) {
        for (int k = 0; k < NS; ++k)
          if (last_use[k] < t && (pick < 0 || last_use[k] < last_use[pick])) pick = k;
        // n_distinct <= NS: a row not used by this tile exists
        out.ops.push_back(make_int2((int)((t - t0) << 8) | pick, (int)out.rec_sta.size()));
        out.rec_sta.push_back(slot_sta[pick]);
      }
      slot_sta[pick] = key[i].first;
      last_use[pick] = t;
      pos_slot[i] = pick;
    }
    uint32_t um = 0x80000000u;
    for (int i = 0; i < 64; ++i) {
      const int ps = i < nv ? pos_slot[i] : pos_slot[nv - 1];  // idle lanes of the last tile add zeros to its last run
      slot[e0 + i] = (uint8_t)ps;
      const int lane = i < nv ? key[i].second : i;
      spos[e0 + lane] = (uint8_t)i;
      const bool last_of_run = i + 1 < nv && key[i + 1].first != key[i].first;
      if (last_of_run) um |= 1u << (i & 31);
    }
    umask[t] = um;
  }
  for (int k = 0; k < NS; ++k)
    if (slot_sta[k] >= 0) {
      out.ops.push_back(make_int2((int)(0x7FFFFFu << 8) | k, (int)out.rec_sta.size()));
      out.rec_sta.push_back(slot_sta[k]);
    }
}

}  // namespace

bool build_keyed_plan(int64_t N, int S, const int *e_sta, int tiles_per_seg, int threads, KeyedPlanHost &out) {
  const int64_t n_tiles = (N + 63) / 64;
  const int n_seg = (int)((n_tiles + tiles_per_seg - 1) / tiles_per_seg);
  out.tiles_per_seg = tiles_per_seg;
  out.n_seg = n_seg;
  out.spos.assign((size_t)n_tiles * 64, 0);
  out.slot.assign((size_t)n_tiles * 64, 0);
  out.umask.assign((size_t)n_tiles, 0);
  std::vector<SegOut> seg((size_t)n_seg);
  auto work = [&](int s0, int s1) {
    for (int s = s0; s < s1; ++s) {
      const int64_t t0 = (int64_t)s * tiles_per_seg, t1 = std::min<int64_t>(n_tiles, t0 + tiles_per_seg);
      plan_segment(N, e_sta, t0, t1, out.spos.data(), out.slot.data(), out.umask.data(), seg[s]);
    }
  };
  const int T = (n_seg < 64) ? 1 : std::max(1, threads);
  if (T == 1) {
    work(0, n_seg);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, (int)((int64_t)n_seg * t / T), (int)((int64_t)n_seg * (t + 1) / T));
    for (auto &x : th) x.join();
  }
  for (const SegOut &so : seg)
    if (!so.ok) return false;
  // records and ops of the segments one after another
  out.seg_op0.assign((size_t)n_seg + 1, 0);
  int64_t n_rec = 0, n_ops = 0;
  for (int s = 0; s < n_seg; ++s) {
    out.seg_op0[s] = (int)n_ops;
    n_ops += (int64_t)seg[s].ops.size();
    n_rec += (int64_t)seg[s].rec_sta.size();
  }
  out.seg_op0[n_seg] = (int)n_ops;
  if (n_rec > 0x7FFFFFFF || n_ops > 0x7FFFFFFF) return false;
  out.ops.resize((size_t)n_ops);
  out.n_rec = (int)n_rec;
  std::vector<int> rec_sta((size_t)n_rec);
  {
    int64_t r0 = 0, o0 = 0;
    for (int s = 0; s < n_seg; ++s) {
      for (size_t k = 0; k < seg[s].ops.size(); ++k) out.ops[(size_t)o0 + k] = make_int2(seg[s].ops[k].x, seg[s].ops[k].y + (int)r0);
      std::memcpy(rec_sta.data() + r0, seg[s].rec_sta.data(), seg[s].rec_sta.size() * sizeof(int));
      o0 += (int64_t)seg[s].ops.size();
      r0 += (int64_t)seg[s].rec_sta.size();
    }
  }
  // the records of each station, ascending
  out.sta_rec0.assign((size_t)S + 1, 0);
  for (int64_t r = 0; r < n_rec; ++r) out.sta_rec0[rec_sta[r] + 1]++;
  for (int i = 0; i < S; ++i) out.sta_rec0[i + 1] += out.sta_rec0[i];
  out.sta_recs.resize((size_t)n_rec);
  {
    std::vector<int> fill(out.sta_rec0.begin(), out.sta_rec0.end() - 1);
    for (int64_t r = 0; r < n_rec; ++r) out.sta_recs[fill[rec_sta[r]]++] = (int)r;
  }
  return true;
}

// F[i][0..39) and the camera border Ssc[i] (6 x ncam, row-major) of every station from its partial records, in
// record order.  One thread per station and column; the threads of a station read a record as one contiguous run.
__global__ void __launch_bounds__(256)
k_keyed_fold(int S, int ncolp, int ncam, const int *__restrict__ sta_rec0, const int *__restrict__ sta_recs,
             const double *__restrict__ part, double *__restrict__ F, double *__restrict__ Ssc) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)S * ncolp) return;
  const int i = (int)(t / ncolp), c = (int)(t - (int64_t)i * ncolp);
  if (c >= 39 + 6 * BA_KS_NCB) return;
  double s = 0.0;
  const int q0 = sta_rec0[i], q1 = sta_rec0[i + 1];
  // the record index of the next step is requested before this step's value
  int r_next = q0 < q1 ? sta_recs[q0] : 0;
  for (int q = q0; q < q1; ++q) {
    const int r = r_next;
    if (q + 1 < q1) r_next = sta_recs[q + 1];
    s += part[(int64_t)r * ncolp + c];
  }
  if (c < 39) {
    F[39 * (int64_t)i + c] = s;
  } else {
    const int row = (c - 39) / BA_KS_NCB, k = (c - 39) - row * BA_KS_NCB;
    if (k < ncam) Ssc[(6 * (int64_t)i + row) * ncam + k] = s;
  }
}

void launch_keyed_fold(hipStream_t s, int S, int ncolp, int ncam, const int *sta_rec0, const int *sta_recs, const double *part,
                       double *F, double *Ssc) {
  if (S <= 0) return;
  const int64_t n = (int64_t)S * ncolp;
  hipLaunchKernelGGL(k_keyed_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S, ncolp, ncam, sta_rec0, sta_recs, part, F, Ssc);
}

}  // namespace ba
