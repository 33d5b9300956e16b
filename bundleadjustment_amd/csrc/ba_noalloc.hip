// ba_noalloc.hip -- ba_evaluate_noalloc: cost and residual vector from arrays that already live in device
// memory, for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
// forbid hipMalloc).  Same arithmetic as ba_evaluate (the station records come from the same kernel, the image
// rows from the same eval_obs); what differs is where the work memory comes from: ba_evaluate keeps its station
// records, camera records and block partials in buffers that ba_create allocated, this entry point keeps them in
// one file-scope device array of fixed size, so it calls no hipMalloc, hipFree or any other allocating routine,
// does not synchronise and copies nothing to the host.
//
// The array exists once per GPU (every device that loads the code object holds its own copy at its own address).
// All calls on one GPU share it and it keeps the last call's contents: every word a call reads it has written
// itself before (each workgroup writes its own partial, the finishing kernel reads exactly the partials of this
// call), so nothing has to start at zero.  Calls that overlap in time on one GPU (two streams, two host threads)
// race on it; calls on one stream are ordered and safe.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>

#include "ba_device.h"
#include "ba_gpu.h"
#include "ba_kernels.h"

namespace ba {
int fail_msg(int code, const char *fmt, ...);  // ba_api.hip: sets the text of ba_last_error()
}

namespace {

enum { NA_MAX_STA = BA_NOALLOC_MAX_STATIONS, NA_MAX_CAM = BA_NOALLOC_MAX_CAMERAS, NA_MAX_PART = 2048, NA_MAX_DEV = 64 };

// layout of the work memory
struct NaScratch {
  double sta[(size_t)NA_MAX_STA * BA_STA_STRIDE];  // station records (rotation, position, sin/cos kappa, unit column scales)
  CamRec cam[NA_MAX_CAM];                           // camera records
  double part_cost[NA_MAX_PART];                    // per-workgroup partial sums of the cost
  double part_bad[NA_MAX_PART];                     // per-workgroup counts of rejected rows
};

// The work memory itself.  In host code this name is only a handle for hipGetSymbolAddress: the device address
// is looked up once per device (scratch_for_current_device) and that pointer is what the kernels receive.
static __device__ __align__(256) unsigned char scratch[sizeof(NaScratch)];

std::mutex g_mu;
NaScratch *g_ptr[NA_MAX_DEV];  // device address of `scratch` on each device, null until looked up

int scratch_for_current_device(NaScratch **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return ba::fail_msg(BA_ERR_DEVICE, "ba_evaluate_noalloc: hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= NA_MAX_DEV) return ba::fail_msg(BA_ERR_UNSUPPORTED, "ba_evaluate_noalloc: device ordinal %d (at most %d devices)", dev, NA_MAX_DEV);
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_ptr[dev]) {
    void *p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // the current device's copy
    if (e != hipSuccess || !p)
      return ba::fail_msg(BA_ERR_DEVICE, "ba_evaluate_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
    g_ptr[dev] = static_cast<NaScratch *>(p);
  }
  *out = g_ptr[dev];
  return 0;
}

struct NaArgs {
  int N, S, P, C;
  const int *obs_pt, *obs_sta, *sta_cam;
  const double *obs_x, *obs_y, *obs_sx, *obs_sy, *obs_cx, *obs_cy;
  const double *pt_prior, *pt_sigma, *iop_prior, *iop_sigma;
  const double *xyz, *iop, *ap;
  double huber_a;
  int model, apply_loss;
  NaScratch *w;
  double *res;   // [2N + 3P + 3C] or null
  double *cost;  // [2]
};

// camera records from the flat iop / ap arrays (the fields the evaluation reads)
__global__ void k_na_cameras(NaArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  CamRec &r = a.w->cam[c];
  for (int k = 0; k < 3; ++k) {
    r.iop[k] = a.iop[3 * c + k];
    r.prior[k] = a.iop_prior[3 * c + k];
    r.isig[k] = 1.0 / a.iop_sigma[3 * c + k];
  }
  for (int k = 0; k < 16; ++k) r.ap[k] = a.ap[16 * c + k];
}

// image rows in file order (src/main.cpp:2033-2120 + HuberLoss), partial cost per workgroup
template <bool GEN>
__global__ void __launch_bounds__(256)
k_na_cost_obs(NaArgs a) {
  __shared__ double lds[8];
  double cost = 0.0, bad = 0.0;
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < a.N; n += (int64_t)gridDim.x * blockDim.x) {
    ObsIn o;
    o.x = a.obs_x[n]; o.y = a.obs_y[n]; o.sx = a.obs_sx[n]; o.sy = a.obs_sy[n]; o.cx = a.obs_cx[n]; o.cy = a.obs_cy[n];
    o.sta = a.obs_sta[n];
    o.row = (int)n;
    const int jj = a.obs_pt[n];
    double r0 = 0.0, r1 = 0.0, f = 1.0;
    bool ok = o.sta >= 0 && o.sta < a.S && jj >= 0 && jj < a.P && o.sx > 0.0 && o.sy > 0.0;
    int c = 0;
    if (ok) {
      c = a.sta_cam[o.sta];
      ok = c >= 0 && c < a.C;
    }
    if (ok) {
      const double *st = a.w->sta + (int64_t)BA_STA_STRIDE * o.sta;
      const CamIn cm = load_cam(a.w->cam, c);
      const double *x = a.xyz + 3 * (int64_t)jj;
      eval_obs<false, GEN, true>(o, st, x[0], x[1], x[2], cm, r0, r1, nullptr, nullptr, nullptr, nullptr, a.model);
      double sc;
      cost += 0.5 * huber_rho(a.huber_a, r0 * r0 + r1 * r1, sc);
      if (a.apply_loss) f = sc;
    } else {
      bad += 1.0;  // a row that ba_create would have refused: it counts as rejected and contributes nothing
    }
    if (a.res) {
      a.res[2 * n] = f * r0;
      a.res[2 * n + 1] = f * r1;
    }
  }
  const double t = block_sum(cost, lds);
  const double b = block_sum(bad, lds);
  if (threadIdx.x == 0) {
    a.w->part_cost[blockIdx.x] = t;
    a.w->part_bad[blockIdx.x] = b;
  }
}

// XYZ prior rows (src/main.cpp:1064-1071, 4596-4599) in file order; partials behind those of the image rows
__global__ void __launch_bounds__(256)
k_na_cost_prior(NaArgs a, int slot0) {
  __shared__ double lds[8];
  double cost = 0.0, bad = 0.0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < 3 * (int64_t)a.P; t += (int64_t)gridDim.x * blockDim.x) {
    const double sg = a.pt_sigma[t];
    double r = 0.0;
    if (sg > 0.0) {
      r = (a.xyz[t] - a.pt_prior[t]) * (1.0 / sg);
      cost += 0.5 * r * r;
    } else {
      bad += 1.0;
    }
    if (a.res) a.res[2 * (int64_t)a.N + t] = r;
  }
  const double s = block_sum(cost, lds);
  const double b = block_sum(bad, lds);
  if (threadIdx.x == 0) {
    a.w->part_cost[slot0 + blockIdx.x] = s;
    a.w->part_bad[slot0 + blockIdx.x] = b;
  }
}

// folds the partials of this call in a fixed order, adds the IOP prior rows (src/main.cpp:4626-4629)
__global__ void __launch_bounds__(1024)
k_na_finish(NaArgs a, int nparts) {
  __shared__ double lds[16];
  double s = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nparts; k += blockDim.x) {
    s += a.w->part_cost[k];
    b += a.w->part_bad[k];
  }
  const double ts = block_sum(s, lds);
  const double tb = block_sum(b, lds);
  if (threadIdx.x == 0) {
    double cost = ts, bad = tb;
    for (int c = 0; c < a.C; ++c) {
      const CamRec &r = a.w->cam[c];
      for (int k = 0; k < 3; ++k) {
        double v = 0.0;
        if (a.iop_sigma[3 * c + k] > 0.0) {
          v = (r.iop[k] - r.prior[k]) * r.isig[k];
          cost += 0.5 * v * v;
        } else {
          bad += 1.0;
        }
        if (a.res) a.res[2 * (int64_t)a.N + 3 * (int64_t)a.P + 3 * c + k] = v;
      }
    }
    a.cost[0] = cost;
    a.cost[1] = bad;
  }
}

}  // namespace

extern "C" int ba_evaluate_noalloc(const ba_dims *dims, const ba_inputs *in, const double *eop, const double *xyz,
                                   const double *iop, const double *ap, int apply_loss, double *cost, double *residuals,
                                   void *hip_stream) {
  if (!dims || !in || !cost) return ba::fail_msg(BA_ERR_INVALID, "ba_evaluate_noalloc: null argument");
  const int N = dims->n_obs, S = dims->n_sta, P = dims->n_pts, C = dims->n_cam;
  if (N < 0 || S < 0 || P < 0 || C < 0) return ba::fail_msg(BA_ERR_INVALID, "ba_evaluate_noalloc: negative dimension");
  if (S > NA_MAX_STA)
    return ba::fail_msg(BA_ERR_UNSUPPORTED, "ba_evaluate_noalloc: %d stations, the fixed work memory holds at most %d (use ba_evaluate)", S, (int)NA_MAX_STA);
  if (C > NA_MAX_CAM)
    return ba::fail_msg(BA_ERR_UNSUPPORTED, "ba_evaluate_noalloc: %d cameras, the fixed work memory holds at most %d (use ba_evaluate)", C, (int)NA_MAX_CAM);
  if (in->model < 0 || in->model > 7) return ba::fail_msg(BA_ERR_UNSUPPORTED, "ba_evaluate_noalloc: projection model %d", in->model);
  if (N > 0 && (!in->obs_pt || !in->obs_sta || !in->obs_x || !in->obs_y || !in->obs_sx || !in->obs_sy || !in->obs_cx || !in->obs_cy))
    return ba::fail_msg(BA_ERR_INVALID, "ba_evaluate_noalloc: null observation array");
  if ((S > 0 && (!in->sta_cam || !eop)) || (P > 0 && (!in->pt_prior || !in->pt_sigma || !xyz)) ||
      (C > 0 && (!in->iop_prior || !in->iop_sigma || !iop || !ap)))
    return ba::fail_msg(BA_ERR_INVALID, "ba_evaluate_noalloc: null station / point / camera array");
  NaScratch *w = nullptr;
  const int rc = scratch_for_current_device(&w);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  // an error that an earlier, unrelated call left behind is not this call's: only a change of the state is reported
  const hipError_t before = hipPeekAtLastError();
  NaArgs a{};
  a.N = N; a.S = S; a.P = P; a.C = C;
  a.obs_pt = in->obs_pt; a.obs_sta = in->obs_sta; a.sta_cam = in->sta_cam;
  a.obs_x = in->obs_x; a.obs_y = in->obs_y; a.obs_sx = in->obs_sx; a.obs_sy = in->obs_sy; a.obs_cx = in->obs_cx; a.obs_cy = in->obs_cy;
  a.pt_prior = in->pt_prior; a.pt_sigma = in->pt_sigma; a.iop_prior = in->iop_prior; a.iop_sigma = in->iop_sigma;
  a.xyz = xyz; a.iop = iop; a.ap = ap;
  a.huber_a = in->huber_a; a.model = in->model; a.apply_loss = apply_loss;
  a.w = w; a.res = residuals; a.cost = cost;
  // station records by the kernel ba_evaluate uses (unit column scales), camera records, rows, fold
  ba::launch_station_prep(s, S, eop, nullptr, in->sta_cam, w->sta);
  if (C > 0) hipLaunchKernelGGL(k_na_cameras, dim3((C + 63) / 64), dim3(64), 0, s, a);
  int nb1 = 0, nb2 = 0;
  if (N > 0) {
    nb1 = (int)std::min<int64_t>(((int64_t)N + 255) / 256, NA_MAX_PART / 2);
    if (in->model) hipLaunchKernelGGL(k_na_cost_obs<true>, dim3(nb1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_na_cost_obs<false>, dim3(nb1), dim3(256), 0, s, a);
  }
  if (P > 0) {
    nb2 = (int)std::min<int64_t>((3 * (int64_t)P + 255) / 256, NA_MAX_PART / 2);
    hipLaunchKernelGGL(k_na_cost_prior, dim3(nb2), dim3(256), 0, s, a, nb1);
  }
  hipLaunchKernelGGL(k_na_finish, dim3(1), dim3(1024), 0, s, a, nb1 + nb2);
  const hipError_t e = hipPeekAtLastError();
  if (e != hipSuccess && e != before) return ba::fail_msg(BA_ERR_DEVICE, "ba_evaluate_noalloc: kernel launch: %s", hipGetErrorString(e));
  return 0;
}
