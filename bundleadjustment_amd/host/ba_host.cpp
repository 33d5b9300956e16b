// ba_host.cpp -- file readers/writers, index maps, priors, fixed/free selection, sigma0 and
// the synthetic network generator.  Interface and reference line citations: include/ba_host.h.
#include "ba_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <numeric>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

const double kPi = 3.141592653589793238462643383279502884197169399;  // src/main.cpp:39
const double kApScale = 1000.0;                                       // src/main.cpp:55

}  // namespace

struct ba_network {
  int32_t model = 0;  // projection functor, 0..7
  // .pho (src/main.cpp:3058-3059)
  std::vector<int32_t> imageTarget, imageStation;
  std::vector<double> imageX, imageY, imageXStdDev, imageYStdDev, imageXCorr, imageYCorr;
  // .eop (:3151-3153)
  std::vector<int32_t> eopStation, eopCamera;
  std::vector<double> EOP;  // 6 per station: omega,phi,kappa (rad), Xo,Yo,Zo
  // .iop (:3541-3544)
  std::vector<int32_t> iopCamera, iopAxis;
  std::vector<double> iopXMin, iopYMin, iopXMax, iopYMax;
  std::vector<double> IOP;  // 3 per camera
  std::vector<double> AP;   // 16 per camera
  // .xyz (:3655-3657)
  std::vector<int32_t> xyzTarget;
  std::vector<double> XYZ, xyzPrior, xyzStdDev;  // 3 per point
  // derived
  std::vector<int32_t> indexPoint, indexPose, staSensor;
  std::vector<double> iopPrior, iopStdDev;
  std::vector<uint16_t> apFree;
  std::vector<uint8_t> iopFree;
  std::vector<uint8_t> camInUse;  // a shard only (ba_net_shard): cameras that the whole network observes
  // truth (synthetic only)
  std::vector<double> tEOP, tXYZ, tIOP, tAP;
  // variances of the parameters (empty until ba_net_set_variances; src/main.cpp:5251-5256 defaults otherwise)
  std::vector<double> vEOP, vXYZ, vIOP, vAP;
};

namespace {

// Token reader with the end-of-file behaviour of the reference's extraction loops: `hit_eof`
// becomes true as soon as an extraction touches the end of the stream, which is what
// `inp.eof()` reports at src/main.cpp:3082, 3185, 3608, 3679.
struct Reader {
  std::ifstream in;
  bool bad = false;
  explicit Reader(const char *path) : in(path) {}
  bool ok() const { return in.is_open(); }
  template <class T>
  void get(T &v) {
    if (bad) return;
    in >> v;
    if (in.fail() && !in.eof()) bad = true;  // malformed token: the reference would spin forever
  }
  bool eof() const { return in.eof(); }
};

int read_pho(const char *path, ba_network &n) {
  Reader r(path);
  if (!r.ok()) return fail(BA_ERR_INVALID, "cannot open %s", path);
  while (true) {  // src/main.cpp:3061-3084
    int c1 = 0, c2 = 0;
    double c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0, c8 = 0;
    r.get(c1); r.get(c2); r.get(c3); r.get(c4); r.get(c5); r.get(c6); r.get(c7); r.get(c8);
    if (r.bad) return fail(BA_ERR_INVALID, "%s: malformed token near record %zu", path, n.imageTarget.size());
    n.imageTarget.push_back(c1); n.imageStation.push_back(c2);
    n.imageX.push_back(c3); n.imageY.push_back(c4);
    n.imageXStdDev.push_back(c5); n.imageYStdDev.push_back(c6);
    n.imageXCorr.push_back(c7); n.imageYCorr.push_back(c8);
    if (r.eof()) break;
  }
  n.imageTarget.pop_back(); n.imageStation.pop_back();  // :3086-3094
  n.imageX.pop_back(); n.imageY.pop_back(); n.imageXStdDev.pop_back(); n.imageYStdDev.pop_back();
  n.imageXCorr.pop_back(); n.imageYCorr.pop_back();
  return 0;
}

int read_eop(const char *path, ba_network &n) {
  Reader r(path);
  if (!r.ok()) return fail(BA_ERR_INVALID, "cannot open %s", path);
  while (true) {  // :3154-3187
    int c1 = 0, c2 = 0;
    double c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0, c8 = 0;
    r.get(c1); r.get(c2); r.get(c3); r.get(c4); r.get(c5); r.get(c6); r.get(c7); r.get(c8);
    if (r.bad) return fail(BA_ERR_INVALID, "%s: malformed token near record %zu", path, n.eopStation.size());
    c6 *= kPi / 180.0;  // :3161-3163
    c7 *= kPi / 180.0;
    c8 *= kPi / 180.0;
    n.eopStation.push_back(c1); n.eopCamera.push_back(c2);
    const double e[6] = {c6, c7, c8, c3, c4, c5};  // :3177-3182
    n.EOP.insert(n.EOP.end(), e, e + 6);
    if (r.eof()) break;
  }
  n.eopStation.pop_back(); n.eopCamera.pop_back();
  n.EOP.resize(n.EOP.size() - 6);
  return 0;
}

int read_iop(const char *path, ba_network &n) {
  Reader r(path);
  if (!r.ok()) return fail(BA_ERR_INVALID, "cannot open %s", path);
  while (true) {  // :3545-3610
    int c1 = 0, c2 = 0;
    double c[26] = {0};
    r.get(c1); r.get(c2);
    for (int k = 3; k <= 25; ++k) r.get(c[k]);
    if (r.bad) return fail(BA_ERR_INVALID, "%s: malformed token near record %zu", path, n.iopCamera.size());
    n.iopCamera.push_back(c1); n.iopAxis.push_back(c2);
    n.iopXMin.push_back(c[3]); n.iopYMin.push_back(c[4]); n.iopXMax.push_back(c[5]); n.iopYMax.push_back(c[6]);
    n.IOP.insert(n.IOP.end(), c + 7, c + 10);   // xp yp c  :3581-3583
    n.AP.insert(n.AP.end(), c + 10, c + 26);    // a1 a2 k1 k2 k3 p1 p2 ep1..ep9  :3588-3604
    if (r.eof()) break;
  }
  n.iopCamera.pop_back(); n.iopAxis.pop_back();
  n.iopXMin.pop_back(); n.iopYMin.pop_back(); n.iopXMax.pop_back(); n.iopYMax.pop_back();
  n.IOP.resize(n.IOP.size() - 3);
  n.AP.resize(n.AP.size() - 16);
  return 0;
}

int read_xyz(const char *path, ba_network &n) {
  Reader r(path);
  if (!r.ok()) return fail(BA_ERR_INVALID, "cannot open %s", path);
  while (true) {  // :3658-3681
    int c1 = 0;
    double c[8] = {0};
    r.get(c1);
    for (int k = 2; k <= 7; ++k) r.get(c[k]);
    if (r.bad) return fail(BA_ERR_INVALID, "%s: malformed token near record %zu", path, n.xyzTarget.size());
    n.xyzTarget.push_back(c1);
    n.XYZ.insert(n.XYZ.end(), c + 2, c + 5);
    n.xyzStdDev.insert(n.xyzStdDev.end(), c + 5, c + 8);
    if (r.eof()) break;
  }
  n.xyzTarget.pop_back();
  n.XYZ.resize(n.XYZ.size() - 3);
  n.xyzStdDev.resize(n.xyzStdDev.size() - 3);
  return 0;
}

int first_match(const int32_t *ids, int32_t n_ids, const int32_t *keys, int32_t n_keys, int32_t *out) {
  std::unordered_map<int32_t, int32_t> first;
  first.reserve(static_cast<size_t>(n_ids) * 2 + 1);
  for (int32_t i = 0; i < n_ids; ++i) first.emplace(ids[i], i);  // emplace keeps the first occurrence
  int missing = 0;
  for (int32_t n = 0; n < n_keys; ++n) {
    auto it = first.find(keys[n]);
    if (it == first.end()) {
      out[n] = n_ids;
      ++missing;
    } else {
      out[n] = it->second;
    }
  }
  return missing;
}

// index maps, priors and default fixed/free selection once the four tables are filled
int finish(ba_network &n) {
  const int32_t N = static_cast<int32_t>(n.imageTarget.size());
  const int32_t S = static_cast<int32_t>(n.eopStation.size());
  const int32_t P = static_cast<int32_t>(n.xyzTarget.size());
  const int32_t C = static_cast<int32_t>(n.iopCamera.size());
  n.indexPoint.resize(N);
  n.indexPose.resize(N);
  n.staSensor.resize(S);
  // src/main.cpp:3821-3831
  if (first_match(n.xyzTarget.data(), P, n.imageTarget.data(), N, n.indexPoint.data()))
    return fail(BA_ERR_INVALID, "an image observation names a target id that is not in the .xyz file");
  if (first_match(n.eopStation.data(), S, n.imageStation.data(), N, n.indexPose.data()))
    return fail(BA_ERR_INVALID, "an image observation names a station id that is not in the .eop file");
  if (first_match(n.iopCamera.data(), C, n.eopCamera.data(), S, n.staSensor.data()))
    return fail(BA_ERR_INVALID, "a station names a camera id that is not in the .iop file");
  n.xyzPrior = n.XYZ;  // the .xyz coordinates are both the start value (:3674) and the prior mean (:4598)
  n.iopPrior = n.IOP;  // :4628
  n.iopStdDev.resize(3 * static_cast<size_t>(C));
  for (int32_t c = 0; c < C; ++c) {  // :4612-4623
    n.iopStdDev[3 * c] = 10.0;
    n.iopStdDev[3 * c + 1] = 10.0;
    n.iopStdDev[3 * c + 2] = 1.0E3;
  }
  if (n.apFree.size() != static_cast<size_t>(C)) n.apFree.assign(C, uint16_t(1u << 2));  // only k1 free, :4487-4504
  if (n.iopFree.size() != static_cast<size_t>(C)) n.iopFree.assign(C, uint8_t(7));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// synthetic generator
// ---------------------------------------------------------------------------------------------
struct Rng {
  std::mt19937_64 g;
  explicit Rng(uint64_t s) : g(s) {}
  double uni() { return (g() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * uni(); }
  double normal() {
    double u1 = uni(), u2 = uni();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * kPi * u2);
  }
};

void rotation(const double *e, double R[9]) {  // src/main.cpp:1325-1335
  const double so = std::sin(e[0]), co = std::cos(e[0]), sp = std::sin(e[1]), cp = std::cos(e[1]);
  const double sk = std::sin(e[2]), ck = std::cos(e[2]);
  R[0] = cp * ck; R[1] = co * sk + so * sp * ck; R[2] = so * sk - co * sp * ck;
  R[3] = -cp * sk; R[4] = co * ck - so * sp * sk; R[5] = so * ck + co * sp * sk;
  R[6] = sp; R[7] = -so * cp; R[8] = co * cp;
}

// distortion at normalised image coordinates (src/main.cpp:1371-1376)
void distortion(const double *ap, double xb, double yb, double &dx, double &dy) {
  const double rr = xb * xb + yb * yb;
  double pw[14];
  pw[0] = 1.0;
  for (int k = 1; k < 14; ++k) pw[k] = pw[k - 1] * rr;
  const double G = ap[2] * pw[1] + ap[3] * pw[2] + ap[4] * pw[3] + ap[7] * pw[4] + ap[8] * pw[5] + ap[9] * pw[6] +
                   ap[10] * (pw[7] + pw[8]) + ap[11] * pw[9] + ap[12] * pw[10] + ap[13] * pw[11] + ap[14] * pw[12] +
                   ap[15] * pw[13];
  dx = xb * G + ap[5] * (rr + 2.0 * xb * xb) + 2.0 * ap[6] * xb * yb + ap[0] * xb + ap[1] * yb;
  dy = yb * G + ap[6] * (rr + 2.0 * yb * yb) + 2.0 * ap[5] * xb * yb;
}

// image coordinates of a point at (Xc, Yc, Zc) in the camera frame, before principal point and distortion:
// src/main.cpp:1349-1350, :2161-2166, :2255-2256, :2341-2342, :2427-2428, :2513-2514, :1441-1442, :2737-2738
void project(int model, double c, double Xc, double Yc, double Zc, double &x, double &y) {
  double h;
  if (model == 0) {
    h = -1.0 / Zc;
  } else if (model == 1) {
    h = 2.0 / (std::sqrt(Xc * Xc + Yc * Yc + Zc * Zc) - Zc);
  } else if (model == 6) {
    h = 1.0;
  } else {
    const double rho = std::sqrt(Xc * Xc + Yc * Yc), th = std::atan2(rho, -Zc);
    const double f = (model == 2 || model == 7) ? th : model == 3 ? 2.0 * std::sin(0.5 * th) : model == 4 ? std::sin(th) : 2.0 * std::tan(0.5 * th);
    h = f / rho;
  }
  x = c * h * Xc;
  y = c * h * Yc;
  if (model == 7) {
    x = std::fabs(x);
    y = std::fabs(y);
  }
}

}  // namespace

extern "C" {

const char *ba_host_last_error(void) { return g_err.c_str(); }

int ba_index_first_match(const int32_t *ids, int32_t n_ids, const int32_t *keys, int32_t n_keys, int32_t *out) {
  if (n_ids < 0 || n_keys < 0 || (!ids && n_ids) || (!keys && n_keys) || (!out && n_keys))
    return fail(BA_ERR_INVALID, "ba_index_first_match: bad arguments");
  const int missing = first_match(ids, n_ids, keys, n_keys, out);
  if (missing) return fail(BA_ERR_INVALID, "%d key(s) not found", missing);
  return 0;
}

int ba_net_read(const char *pho, const char *eop, const char *iop, const char *xyz, ba_network **out) {
  if (!pho || !eop || !iop || !xyz || !out) return fail(BA_ERR_INVALID, "ba_net_read: null argument");
  ba_network *n = new ba_network();
  int rc = read_pho(pho, *n);
  if (!rc) rc = read_eop(eop, *n);
  if (!rc) rc = read_iop(iop, *n);
  if (!rc) rc = read_xyz(xyz, *n);
  if (!rc) rc = finish(*n);
  if (rc) {
    delete n;
    return rc;
  }
  *out = n;
  return 0;
}

void ba_synth_default(ba_synth_config *c) {
  std::memset(c, 0, sizeof *c);
  c->n_cam = 1;
  c->n_sta = 24;
  c->n_pts = 500;
  c->rays_per_point = 20;
  c->seed = 0xBA5EBA11ull;
  c->ap_free_mask = 0x7F;  // a1 a2 k1 k2 k3 p1 p2
  c->iop_free_mask = 7;
  c->empirical_terms = 0;
  c->noise_sigma = 0.5;
  c->outlier_fraction = 0.01;
  c->control_sigma = 10.0;
  c->free_sigma = 1.0e4;
  c->control_every = 64;
  c->perfect_start = 0;
  c->model = 0;
}

int ba_net_synth(const ba_synth_config *cfg, ba_network **out) {
  if (!cfg || !out) return fail(BA_ERR_INVALID, "ba_net_synth: null argument");
  const int32_t C = cfg->n_cam, S = cfg->n_sta, P = cfg->n_pts;
  if (C < 1 || S < 1 || P < 1 || cfg->rays_per_point < 1) return fail(BA_ERR_INVALID, "ba_net_synth: sizes must be positive");
  if (cfg->model < 0 || cfg->model > 7) return fail(BA_ERR_INVALID, "ba_net_synth: projection model %d out of range 0..7", cfg->model);
  const int32_t K = std::min(cfg->rays_per_point, S);
  if (static_cast<int64_t>(P) * K > 2000000000LL) return fail(BA_ERR_INVALID, "ba_net_synth: too many observations");
  ba_network *n = new ba_network();
  n->model = cfg->model;
  Rng rng(cfg->seed);

  // cameras: image units are pixels, sensor about +-2000 px
  n->tIOP.resize(3 * static_cast<size_t>(C));
  n->tAP.assign(16 * static_cast<size_t>(C), 0.0);
  for (int32_t c = 0; c < C; ++c) {
    n->iopCamera.push_back(1 + c);
    n->iopAxis.push_back(1);
    n->iopXMin.push_back(-2000.0); n->iopYMin.push_back(-2000.0);
    n->iopXMax.push_back(2000.0); n->iopYMax.push_back(2000.0);
    n->tIOP[3 * c] = rng.uni(-5, 5);
    n->tIOP[3 * c + 1] = rng.uni(-5, 5);
    n->tIOP[3 * c + 2] = cfg->model == 6 ? 0.3 : 3000.0;  // orthographic: image = c x object coordinates
    double *a = &n->tAP[16 * static_cast<size_t>(c)];
    a[0] = 1e-2; a[1] = 1e-2; a[2] = 0.5; a[3] = -0.02; a[4] = 5e-4; a[5] = 0.05; a[6] = 0.05;
    if (cfg->empirical_terms) { a[7] = 1e-5; a[10] = 1e-8; }
    // parameters that stay fixed keep their true value out of the model: zero them so that the
    // network is consistent with what will be estimated
    for (int k = 0; k < 16; ++k)
      if (!(cfg->ap_free_mask & (1u << k))) a[k] = 0.0;
  }

  // stations: a gx x gy grid of cells of size D at height H, each looking obliquely at its cell centre
  const double H = 10000.0;
  const double D = 0.3 * H / std::sqrt(K / kPi);
  const int32_t gx = static_cast<int32_t>(std::ceil(std::sqrt(static_cast<double>(S))));
  n->tEOP.resize(6 * static_cast<size_t>(S));
  for (int32_t s = 0; s < S; ++s) {
    const int32_t ix = s % gx, iy = s / gx;
    const double cx = (ix + 0.5) * D, cy = (iy + 0.5) * D;
    const double om = rng.uni(-15, 15) * kPi / 180.0, ph = rng.uni(-15, 15) * kPi / 180.0;
    const double ka = rng.uni(-kPi, kPi);
    const double h = H * rng.uni(0.9, 1.1);
    const double t = h / (std::cos(om) * std::cos(ph));
    double *e = &n->tEOP[6 * static_cast<size_t>(s)];
    e[0] = om; e[1] = ph; e[2] = ka;
    e[3] = cx + t * std::sin(ph) + rng.uni(-0.1, 0.1) * D;
    e[4] = cy - t * std::sin(om) * std::cos(ph) + rng.uni(-0.1, 0.1) * D;
    e[5] = h;
    n->eopStation.push_back(101 + s);
    n->eopCamera.push_back(1 + s % C);
  }

  // points and their observing stations: the K stations whose cell centres are nearest
  n->tXYZ.resize(3 * static_cast<size_t>(P));
  std::vector<int32_t> obsSta(static_cast<size_t>(P) * K);
  {
    std::vector<std::pair<double, int32_t>> cand;
    const int32_t gy = (S + gx - 1) / gx;
    for (int32_t j = 0; j < P; ++j) {
      const int32_t home = static_cast<int32_t>(rng.uni() * S) % S;
      const int32_t hx = home % gx, hy = home / gx;
      const double X = (hx + rng.uni()) * D, Y = (hy + rng.uni()) * D, Z = rng.uni(-0.1, 0.1) * H;
      n->tXYZ[3 * static_cast<size_t>(j)] = X;
      n->tXYZ[3 * static_cast<size_t>(j) + 1] = Y;
      n->tXYZ[3 * static_cast<size_t>(j) + 2] = Z;
      int32_t w = static_cast<int32_t>(std::ceil(std::sqrt(static_cast<double>(K)) / 2.0)) + 1;
      for (;;) {
        cand.clear();
        for (int32_t iy = std::max(0, hy - w); iy <= std::min(gy - 1, hy + w); ++iy)
          for (int32_t ix = std::max(0, hx - w); ix <= std::min(gx - 1, hx + w); ++ix) {
            const int32_t s = iy * gx + ix;
            if (s >= S) continue;
            const double dx = (ix + 0.5) * D - X, dy = (iy + 0.5) * D - Y;
            cand.emplace_back(dx * dx + dy * dy, s);
          }
        if (static_cast<int32_t>(cand.size()) >= K || (w >= gx && w >= gy)) break;
        ++w;
      }
      std::sort(cand.begin(), cand.end());
      for (int32_t k = 0; k < K; ++k) obsSta[static_cast<size_t>(j) * K + k] = cand[k].second;
      n->xyzTarget.push_back(1001 + j);
    }
  }

  // .pho rows in station-major order (image by image), points ascending inside an image
  const int64_t N = static_cast<int64_t>(P) * K;
  std::vector<int64_t> start(S + 1, 0);
  for (int64_t q = 0; q < N; ++q) start[obsSta[q] + 1]++;
  for (int32_t s = 0; s < S; ++s) start[s + 1] += start[s];
  n->imageTarget.resize(N); n->imageStation.resize(N);
  n->imageX.resize(N); n->imageY.resize(N);
  n->imageXStdDev.assign(N, cfg->noise_sigma); n->imageYStdDev.assign(N, cfg->noise_sigma);
  n->imageXCorr.assign(N, 0.0); n->imageYCorr.assign(N, 0.0);
  {
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    std::vector<double> R(9 * static_cast<size_t>(S));
    for (int32_t s = 0; s < S; ++s) rotation(&n->tEOP[6 * static_cast<size_t>(s)], &R[9 * static_cast<size_t>(s)]);
    for (int32_t j = 0; j < P; ++j)
      for (int32_t k = 0; k < K; ++k) {
        const int32_t s = obsSta[static_cast<size_t>(j) * K + k];
        const int64_t row = fill[s]++;
        const double *e = &n->tEOP[6 * static_cast<size_t>(s)], *r = &R[9 * static_cast<size_t>(s)];
        const double *p = &n->tXYZ[3 * static_cast<size_t>(j)];
        const int32_t c = s % C;
        const double *io = &n->tIOP[3 * static_cast<size_t>(c)], *a = &n->tAP[16 * static_cast<size_t>(c)];
        const double dX = p[0] - e[3], dY = p[1] - e[4], dZ = p[2] - e[5];
        const double Xc = r[0] * dX + r[1] * dY + r[2] * dZ, Yc = r[3] * dX + r[4] * dY + r[5] * dZ;
        const double Zc = r[6] * dX + r[7] * dY + r[8] * dZ;
        double x, y;
        project(cfg->model, io[2], Xc, Yc, Zc, x, y);
        // the correction is a function of the observed coordinate: solve x_obs = x + xp + dx(x_obs)
        double xo = x + io[0], yo = y + io[1];
        for (int it = 0; it < 5; ++it) {
          double dx, dy;
          distortion(a, (xo - io[0]) / kApScale, (yo - io[1]) / kApScale, dx, dy);
          xo = x + io[0] + dx;
          yo = y + io[1] + dy;
        }
        n->imageTarget[row] = n->xyzTarget[j];
        n->imageStation[row] = n->eopStation[s];
        n->imageX[row] = xo;
        n->imageY[row] = yo;
      }
    // noise and gross errors in file order, so that they do not depend on the point loop above
    for (int64_t row = 0; row < N; ++row) {
      double nx = cfg->noise_sigma * rng.normal(), ny = cfg->noise_sigma * rng.normal();
      if (rng.uni() < cfg->outlier_fraction) {
        nx += (rng.uni() < 0.5 ? -1.0 : 1.0) * rng.uni(5, 20) * cfg->noise_sigma;
        ny += (rng.uni() < 0.5 ? -1.0 : 1.0) * rng.uni(5, 20) * cfg->noise_sigma;
      }
      n->imageX[row] += nx;
      n->imageY[row] += ny;
    }
  }

  // start values and priors
  n->EOP = n->tEOP; n->XYZ = n->tXYZ; n->IOP = n->tIOP;
  n->AP.assign(16 * static_cast<size_t>(C), 0.0);
  if (cfg->perfect_start) n->AP = n->tAP;
  if (!cfg->perfect_start) {
    for (int32_t s = 0; s < S; ++s) {
      double *e = &n->EOP[6 * static_cast<size_t>(s)];
      for (int k = 0; k < 3; ++k) e[k] += 0.05 * kPi / 180.0 * rng.normal();
      for (int k = 3; k < 6; ++k) e[k] += 5.0 * rng.normal();
    }
    for (size_t q = 0; q < n->XYZ.size(); ++q) n->XYZ[q] += 5.0 * rng.normal();
    for (int32_t c = 0; c < C; ++c)
      for (int k = 0; k < 3; ++k)
        if (cfg->iop_free_mask & (1u << k))
          n->IOP[3 * static_cast<size_t>(c) + k] += (k == 2 && cfg->model == 6 ? 1e-4 : 1.0) * rng.normal();
  }
  n->xyzStdDev.resize(3 * static_cast<size_t>(P));
  const int32_t every = cfg->control_every > 0 ? cfg->control_every : 64;
  for (int32_t j = 0; j < P; ++j) {
    const double sg = (j % every == 0) ? cfg->control_sigma : cfg->free_sigma;
    for (int k = 0; k < 3; ++k) n->xyzStdDev[3 * static_cast<size_t>(j) + k] = sg;
  }
  n->apFree.assign(C, cfg->ap_free_mask);
  n->iopFree.assign(C, cfg->iop_free_mask);
  const int rc = finish(*n);
  if (rc) {
    delete n;
    return rc;
  }
  *out = n;
  return 0;
}

void ba_net_free(ba_network *net) { delete net; }

int ba_net_set_model(ba_network *n, int32_t model) {
  if (!n || model < 0 || model > 7) return fail(BA_ERR_INVALID, "ba_net_set_model: model must be 0..7");
  n->model = model;
  return 0;
}

int ba_net_iop_bounds(const ba_network *n, double *lower, double *upper) {
  if (!n || !lower || !upper) return fail(BA_ERR_INVALID, "ba_net_iop_bounds: null argument");
  for (size_t c = 0; c < n->iopCamera.size(); ++c) {
    lower[3 * c + 0] = n->iopXMin[c];  upper[3 * c + 0] = n->iopXMax[c];
    lower[3 * c + 1] = -n->iopYMax[c]; upper[3 * c + 1] = n->iopYMin[c];  // "Y is flipped"
    lower[3 * c + 2] = 0.0;            upper[3 * c + 2] = DBL_MAX;        // principal distance positive
  }
  return 0;
}

int ba_net_dims(const ba_network *n, ba_dims *d) {
  if (!n || !d) return fail(BA_ERR_INVALID, "ba_net_dims: null argument");
  d->n_obs = static_cast<int32_t>(n->imageTarget.size());
  d->n_sta = static_cast<int32_t>(n->eopStation.size());
  d->n_pts = static_cast<int32_t>(n->xyzTarget.size());
  d->n_cam = static_cast<int32_t>(n->iopCamera.size());
  return 0;
}

int ba_net_inputs(ba_network *n, ba_inputs *in) {
  if (!n || !in) return fail(BA_ERR_INVALID, "ba_net_inputs: null argument");
  in->obs_pt = n->indexPoint.data();
  in->obs_sta = n->indexPose.data();
  in->sta_cam = n->staSensor.data();
  in->obs_x = n->imageX.data(); in->obs_y = n->imageY.data();
  in->obs_sx = n->imageXStdDev.data(); in->obs_sy = n->imageYStdDev.data();
  in->obs_cx = n->imageXCorr.data(); in->obs_cy = n->imageYCorr.data();
  in->pt_prior = n->xyzPrior.data(); in->pt_sigma = n->xyzStdDev.data();
  in->iop_prior = n->iopPrior.data(); in->iop_sigma = n->iopStdDev.data();
  in->ap_free_mask = n->apFree.data();
  in->iop_free_mask = n->iopFree.data();
  in->huber_a = 1.0;  // src/main.cpp:3810
  in->model = n->model;
  in->cam_in_use = n->camInUse.empty() ? nullptr : n->camInUse.data();
  return 0;
}

double *ba_net_eop(ba_network *n) { return n->EOP.data(); }
double *ba_net_xyz(ba_network *n) { return n->XYZ.data(); }
double *ba_net_iop(ba_network *n) { return n->IOP.data(); }
double *ba_net_ap(ba_network *n) { return n->AP.data(); }
const double *ba_net_truth_eop(const ba_network *n) { return n->tEOP.empty() ? nullptr : n->tEOP.data(); }
const double *ba_net_truth_xyz(const ba_network *n) { return n->tXYZ.empty() ? nullptr : n->tXYZ.data(); }
const double *ba_net_truth_iop(const ba_network *n) { return n->tIOP.empty() ? nullptr : n->tIOP.data(); }
const double *ba_net_truth_ap(const ba_network *n) { return n->tAP.empty() ? nullptr : n->tAP.data(); }

// Point boundaries of the ranks: contiguous ranges with about the same sum of (observations + 1) per rank.  The "+ 1"
// stands for a point's own prior row and keeps ranges of unobserved points from collapsing.
int ba_net_point_ranges(const ba_network *n, int32_t world, int64_t *bounds) {
  if (!n || !bounds || world < 1) return fail(BA_ERR_INVALID, "ba_net_point_ranges: bad argument");
  const int64_t P = static_cast<int64_t>(n->xyzTarget.size());
  std::vector<int64_t> w(static_cast<size_t>(P), 1);
  for (int32_t p : n->indexPoint) w[p]++;
  for (int64_t p = 1; p < P; ++p) w[p] += w[p - 1];
  const double total = P ? static_cast<double>(w[P - 1]) : 0.0;
  bounds[0] = 0;
  for (int32_t r = 1; r < world; ++r) {
    const double target = total * r / world;
    const int64_t b = std::lower_bound(w.begin(), w.end(), target,
                                       [](int64_t a, double t) { return static_cast<double>(a) < t; }) - w.begin();
    bounds[r] = std::max(b, bounds[r - 1]);
  }
  bounds[world] = P;
  return 0;
}

// The sub-network of one rank: its point range with every observation of those points (file order kept), all stations
// and cameras.  The station and camera unknowns are replicated; only the reduced camera system is summed over ranks.
int ba_net_shard(const ba_network *n, int32_t rank, int32_t world, ba_network **out, int64_t *pt_lo, int64_t *pt_hi) {
  if (!n || !out || world < 1 || rank < 0 || rank >= world) return fail(BA_ERR_INVALID, "ba_net_shard: bad argument");
  std::vector<int64_t> b(static_cast<size_t>(world) + 1);
  if (int rc = ba_net_point_ranges(n, world, b.data())) return rc;
  const int64_t lo = b[rank], hi = b[rank + 1];
  ba_network *s = new ba_network();
  s->model = n->model;
  s->eopStation = n->eopStation; s->eopCamera = n->eopCamera; s->EOP = n->EOP;
  s->iopCamera = n->iopCamera; s->iopAxis = n->iopAxis;
  s->iopXMin = n->iopXMin; s->iopYMin = n->iopYMin; s->iopXMax = n->iopXMax; s->iopYMax = n->iopYMax;
  s->IOP = n->IOP; s->AP = n->AP;
  s->staSensor = n->staSensor; s->iopPrior = n->iopPrior; s->iopStdDev = n->iopStdDev;
  s->apFree = n->apFree; s->iopFree = n->iopFree;
  // the cameras that the whole network observes: a rank without rows of one of them carries its columns all the same
  s->camInUse.assign(n->iopCamera.size(), 0);
  for (size_t i = 0; i < n->indexPose.size(); ++i) s->camInUse[n->staSensor[n->indexPose[i]]] = 1;
  s->tEOP = n->tEOP; s->tIOP = n->tIOP; s->tAP = n->tAP;
  s->vEOP = n->vEOP; s->vIOP = n->vIOP; s->vAP = n->vAP;
  auto cut = [&](const std::vector<double> &v, std::vector<double> &o, int per) {
    if (!v.empty()) o.assign(v.begin() + per * lo, v.begin() + per * hi);
  };
  s->xyzTarget.assign(n->xyzTarget.begin() + lo, n->xyzTarget.begin() + hi);
  cut(n->XYZ, s->XYZ, 3); cut(n->xyzPrior, s->xyzPrior, 3); cut(n->xyzStdDev, s->xyzStdDev, 3);
  cut(n->tXYZ, s->tXYZ, 3); cut(n->vXYZ, s->vXYZ, 3);
  const size_t N = n->indexPoint.size();
  size_t cnt = 0;
  for (size_t i = 0; i < N; ++i) cnt += (n->indexPoint[i] >= lo && n->indexPoint[i] < hi);
  for (auto *v : {&s->imageTarget, &s->imageStation, &s->indexPoint, &s->indexPose}) v->reserve(cnt);
  for (auto *v : {&s->imageX, &s->imageY, &s->imageXStdDev, &s->imageYStdDev, &s->imageXCorr, &s->imageYCorr}) v->reserve(cnt);
  for (size_t i = 0; i < N; ++i) {
    const int64_t p = n->indexPoint[i];
    if (p < lo || p >= hi) continue;
    s->imageTarget.push_back(n->imageTarget[i]); s->imageStation.push_back(n->imageStation[i]);
    s->indexPoint.push_back(static_cast<int32_t>(p - lo)); s->indexPose.push_back(n->indexPose[i]);
    s->imageX.push_back(n->imageX[i]); s->imageY.push_back(n->imageY[i]);
    s->imageXStdDev.push_back(n->imageXStdDev[i]); s->imageYStdDev.push_back(n->imageYStdDev[i]);
    s->imageXCorr.push_back(n->imageXCorr[i]); s->imageYCorr.push_back(n->imageYCorr[i]);
  }
  if (pt_lo) *pt_lo = lo;
  if (pt_hi) *pt_hi = hi;
  *out = s;
  return 0;
}

const int32_t *ba_net_ids(const ba_network *n, int which) {
  switch (which) {
    case 0: return n->imageTarget.data();
    case 1: return n->imageStation.data();
    case 2: return n->eopStation.data();
    case 3: return n->eopCamera.data();
    case 4: return n->xyzTarget.data();
    case 5: return n->iopCamera.data();
  }
  return nullptr;
}

int ba_net_set_free(ba_network *n, uint16_t ap_free_mask, uint8_t iop_free_mask) {
  if (!n) return fail(BA_ERR_INVALID, "ba_net_set_free: null argument");
  n->apFree.assign(n->iopCamera.size(), ap_free_mask);
  n->iopFree.assign(n->iopCamera.size(), static_cast<uint8_t>(iop_free_mask & 7u));
  return 0;
}

int32_t ba_net_num_ap_correction(const ba_network *n) {
  // src/main.cpp:4509: fixAP.size() of the last camera processed, not multiplied by the camera count
  if (!n || n->apFree.empty()) return 0;
  int fixed = 0;
  for (int k = 0; k < 16; ++k)
    if (!(n->apFree.back() & (1u << k))) ++fixed;
  return fixed;
}

double ba_sigma0(const ba_summary *s, int32_t n_pts, int32_t num_ap_correction, double *redundancy) {
  // src/main.cpp:4907-4910
  const double red = static_cast<double>(s->num_residuals_reduced) - s->num_parameters_reduced - 3.0 * n_pts + 7 +
                     num_ap_correction;
  if (redundancy) *redundancy = red;
  const double aposterioriVariance = 2 * s->final_cost / red;
  return std::sqrt(aposterioriVariance);
}

int ba_net_write_inputs(const ba_network *n, const char *dir, const char *stem) {
  if (!n || !dir || !stem) return fail(BA_ERR_INVALID, "ba_net_write_inputs: null argument");
  const std::string base = std::string(dir) + "/" + stem;
  FILE *f = fopen((base + ".pho").c_str(), "w");
  if (!f) return fail(BA_ERR_INVALID, "cannot write %s.pho", base.c_str());
  for (size_t i = 0; i < n->imageTarget.size(); ++i)
    fprintf(f, "%i %i %.17g %.17g %.17g %.17g %.17g %.17g\n", n->imageTarget[i], n->imageStation[i], n->imageX[i],
            n->imageY[i], n->imageXStdDev[i], n->imageYStdDev[i], n->imageXCorr[i], n->imageYCorr[i]);
  fclose(f);
  f = fopen((base + ".eop").c_str(), "w");
  if (!f) return fail(BA_ERR_INVALID, "cannot write %s.eop", base.c_str());
  for (size_t i = 0; i < n->eopStation.size(); ++i) {
    const double *e = &n->EOP[6 * i];
    fprintf(f, "%i %i %.17g %.17g %.17g %.17g %.17g %.17g\n", n->eopStation[i], n->eopCamera[i], e[3], e[4], e[5],
            e[0] * 180.0 / kPi, e[1] * 180.0 / kPi, e[2] * 180.0 / kPi);
  }
  fclose(f);
  f = fopen((base + ".iop").c_str(), "w");
  if (!f) return fail(BA_ERR_INVALID, "cannot write %s.iop", base.c_str());
  for (size_t i = 0; i < n->iopCamera.size(); ++i) {
    fprintf(f, "%i %i %.17g %.17g %.17g %.17g", n->iopCamera[i], n->iopAxis[i], n->iopXMin[i], n->iopYMin[i],
            n->iopXMax[i], n->iopYMax[i]);
    for (int k = 0; k < 3; ++k) fprintf(f, " %.17g", n->IOP[3 * i + k]);
    for (int k = 0; k < 16; ++k) fprintf(f, " %.17g", n->AP[16 * i + k]);
    fprintf(f, "\n");
  }
  fclose(f);
  f = fopen((base + ".xyz").c_str(), "w");
  if (!f) return fail(BA_ERR_INVALID, "cannot write %s.xyz", base.c_str());
  for (size_t i = 0; i < n->xyzTarget.size(); ++i)
    fprintf(f, "%i %.17g %.17g %.17g %.17g %.17g %.17g\n", n->xyzTarget[i], n->xyzPrior[3 * i], n->xyzPrior[3 * i + 1],
            n->xyzPrior[3 * i + 2], n->xyzStdDev[3 * i], n->xyzStdDev[3 * i + 1], n->xyzStdDev[3 * i + 2]);
  fclose(f);
  return 0;
}

int ba_net_set_variances(ba_network *n, const double *var_eop, const double *var_xyz, const double *var_iop, const double *var_ap) {
  if (!n) return fail(BA_ERR_INVALID, "ba_net_set_variances: null network");
  auto set = [](std::vector<double> &dst, const double *src, size_t cnt) {
    if (src) dst.assign(src, src + cnt);
    else dst.clear();
  };
  set(n->vEOP, var_eop, n->EOP.size());
  set(n->vXYZ, var_xyz, n->XYZ.size());
  set(n->vIOP, var_iop, n->IOP.size());
  set(n->vAP, var_ap, n->AP.size());
  return 0;
}

int ba_net_write_results(const ba_network *n, const char *dir, const double *residuals) {
  if (!n || !dir) return fail(BA_ERR_INVALID, "ba_net_write_results: null argument");
  const std::string d = std::string(dir) + "/";
  const double PI = kPi;
  // sqrt(variance) as at src/main.cpp:7085-7118; variances default to 1E6 when no covariance was computed (:5251-5256)
  auto sd = [](const std::vector<double> &v, size_t k) { return std::sqrt(v.empty() ? 1E6 : v[k]); };
  FILE *fout = fopen((d + "temp.pho").c_str(), "w");
  if (!fout) return fail(BA_ERR_INVALID, "cannot write into %s", dir);
  for (size_t i = 0; i < n->imageTarget.size(); ++i)  // src/main.cpp:3130
    fprintf(fout, "%i %i %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf\n", n->imageTarget[i], n->imageStation[i], n->imageX[i],
            n->imageY[i], n->imageXStdDev[i], n->imageYStdDev[i], n->imageXCorr[i], n->imageYCorr[i]);
  fclose(fout);

  fout = fopen((d + "EOP.jck").c_str(), "w");
  for (size_t i = 0; i < n->eopStation.size(); ++i) {  // src/main.cpp:7096
    const double *E = &n->EOP[6 * i];
    fprintf(fout, "%i %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf\n", n->eopStation[i],
            E[0] * 180.0 / PI, E[1] * 180.0 / PI, E[2] * 180.0 / PI, E[3], E[4], E[5], sd(n->vEOP, 6 * i) * 180.0 / PI,
            sd(n->vEOP, 6 * i + 1) * 180.0 / PI, sd(n->vEOP, 6 * i + 2) * 180.0 / PI, sd(n->vEOP, 6 * i + 3),
            sd(n->vEOP, 6 * i + 4), sd(n->vEOP, 6 * i + 5));
  }
  fclose(fout);

  fout = fopen((d + "temp.eop").c_str(), "w");
  for (size_t i = 0; i < n->eopStation.size(); ++i) {  // src/main.cpp:4804
    const double *E = &n->EOP[6 * i];
    fprintf(fout, "%i %i %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf\n", n->eopStation[i], n->eopCamera[i], E[3], E[4], E[5],
            E[0] * 180.0 / PI, E[1] * 180.0 / PI, E[2] * 180.0 / PI);
  }
  fclose(fout);

  fout = fopen((d + "iop.jck").c_str(), "w");
  for (size_t i = 0; i < n->iopCamera.size(); ++i)  // src/main.cpp:7107
    fprintf(fout, "%i %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf\n", n->iopCamera[i], n->IOP[3 * i], n->IOP[3 * i + 1],
            n->IOP[3 * i + 2], sd(n->vIOP, 3 * i), sd(n->vIOP, 3 * i + 1), sd(n->vIOP, 3 * i + 2));
  fclose(fout);

  fout = fopen((d + "temp.iop").c_str(), "w");
  for (size_t i = 0; i < n->iopCamera.size(); ++i) {  // src/main.cpp:4843 (6 decimals, as the reference)
    fprintf(fout, "%i %i %.6lf %.6lf %.6lf %.6lf", n->iopCamera[i], n->iopAxis[i], n->iopXMin[i], n->iopYMin[i],
            n->iopXMax[i], n->iopYMax[i]);
    for (int k = 0; k < 3; ++k) fprintf(fout, " %.6lf", n->IOP[3 * i + k]);
    for (int k = 0; k < 16; ++k) fprintf(fout, " %.6lf", n->AP[16 * i + k]);
    fprintf(fout, "\n");
  }
  fclose(fout);

  fout = fopen((d + "ap.jck").c_str(), "w");
  for (size_t i = 0; i < n->iopCamera.size(); ++i) {  // src/main.cpp:7118
    fprintf(fout, "%i", n->iopCamera[i]);
    for (int k = 0; k < 16; ++k) fprintf(fout, " %.6lf", n->AP[16 * i + k]);
    for (int k = 0; k < 16; ++k) fprintf(fout, " %.6lf", sd(n->vAP, 16 * i + k));
    fprintf(fout, "\n");
  }
  fclose(fout);

  fout = fopen((d + "XYZ.jck").c_str(), "w");
  for (size_t i = 0; i < n->xyzTarget.size(); ++i)  // src/main.cpp:7085
    fprintf(fout, "%i %.6lf %.6lf %.6lf %.6lf %.6lf %.6lf\n", n->xyzTarget[i], n->XYZ[3 * i], n->XYZ[3 * i + 1],
            n->XYZ[3 * i + 2], sd(n->vXYZ, 3 * i), sd(n->vXYZ, 3 * i + 1), sd(n->vXYZ, 3 * i + 2));
  fclose(fout);

  if (residuals) {
    fout = fopen((d + "image.jck").c_str(), "w");
    for (size_t i = 0; i < n->imageTarget.size(); ++i) {  // src/main.cpp:6955-6965, 6996
      const double vx = residuals[2 * i] * n->imageXStdDev[i], vy = residuals[2 * i + 1] * n->imageYStdDev[i];
      const int32_t pose = n->indexPose[i];
      // reference ids as the collinearity block fills them (src/main.cpp:3838-3840)
      const int pointReferenceID = n->xyzTarget[n->indexPoint[i]], frameReferenceID = n->eopStation[pose];
      const int sensorReferenceID = n->iopCamera[n->staSensor[pose]];
      // redundancy numbers default to 1E6 (:6670); CvDiag carries the observation variance (:6673-6676)
      fprintf(fout, "%i %i %i %.6lf %.6lf %.6lf %.6lf %.2lf %.2lf %.6lf %.6lf\n", pointReferenceID, frameReferenceID,
              sensorReferenceID, n->imageX[i], n->imageY[i], vx, vy, 1E6, 1E6,
              std::sqrt(n->imageXStdDev[i] * n->imageXStdDev[i]), std::sqrt(n->imageYStdDev[i] * n->imageYStdDev[i]));
    }
    fclose(fout);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// reporting: statistics helpers, angles.jck, the console report after the solve, QC against a truth file
// ---------------------------------------------------------------------------------------------
double ba_stat_median(const double *v, int64_t n) {  // calcMedian, src/main.cpp:791-808
  if (!v || n <= 0) return 0.0;
  std::vector<double> s(v, v + n);
  std::sort(s.begin(), s.end());
  const size_t size = s.size();
  return (size % 2 == 0) ? (s[size / 2 - 1] + s[size / 2]) / 2 : s[size / 2];
}

double ba_stat_stddev(const double *v, int64_t n) {  // calcStdDev, src/main.cpp:811-823 (divides by n)
  if (!v || n <= 0) return 0.0;
  double sum = 0.0;
  for (int64_t i = 0; i < n; ++i) sum += v[i];
  const double mean = sum / n;
  double sq = 0.0;
  for (int64_t i = 0; i < n; ++i) sq += (v[i] - mean) * (v[i] - mean);
  return std::sqrt(sq / n);
}

void ba_stat_all(const double *v, int64_t n, ba_stats *out) {  // calcStatistics, src/main.cpp:884-920
  if (!out) return;
  std::memset(out, 0, sizeof *out);
  if (!v || n <= 0) return;
  out->median = ba_stat_median(v, n);
  double sum = 0.0;
  for (int64_t i = 0; i < n; ++i) sum += v[i];
  out->mean = sum / n;
  out->stdev = ba_stat_stddev(v, n);
  out->max = *std::max_element(v, v + n);
  out->min = *std::min_element(v, v + n);
  double r = 0.0;
  for (int64_t i = 0; i < n; ++i) r += v[i] * v[i];
  out->rmse = std::sqrt(r / double(n));
}

double ba_aic(double num_obs, double mse, double num_unk) { return num_obs * std::log(mse) + 2.0 * num_unk; }  // :1008-1012
double ba_bic(double num_obs, double mse, double num_unk) { return num_obs * std::log(mse) + num_unk * std::log(num_obs); }  // :1015-1019

double ba_incidence_angle(const double *eop, const double *xyz) {  // src/main.cpp:923-945
  double R[9];
  rotation(eop, R);
  const double dX = xyz[0] - eop[3], dY = xyz[1] - eop[4], dZ = xyz[2] - eop[5];
  const double Xs = R[0] * dX + R[1] * dY + R[2] * dZ, Ys = R[3] * dX + R[4] * dY + R[5] * dZ;
  const double Zs = R[6] * dX + R[7] * dY + R[8] * dZ;
  return std::atan2(std::sqrt(Xs * Xs + Ys * Ys), -Zs);
}

static void corrected_image_point(double x, double y, const double *iop, const double *ap, double &xc, double &yc) {
  double dx, dy;  // src/main.cpp:950-963
  distortion(ap, (x - iop[0]) / kApScale, (y - iop[1]) / kApScale, dx, dy);
  xc = x - iop[0] - dx;
  yc = y - iop[1] - dy;
}

double ba_refraction_angle(double x, double y, const double *iop, const double *ap) {  // src/main.cpp:947-966
  double xc, yc;
  corrected_image_point(x, y, iop, ap, xc, yc);
  return std::atan2(std::sqrt(xc * xc + yc * yc), iop[2]);
}

double ba_radial_distance(double x, double y, const double *iop, const double *ap) {  // src/main.cpp:968-987
  double xc, yc;
  corrected_image_point(x, y, iop, ap, xc, yc);
  return std::sqrt(xc * xc + yc * yc);
}

int ba_net_write_angles(const ba_network *n, const char *dir) {  // src/main.cpp:4756-4782
  if (!n || !dir) return fail(BA_ERR_INVALID, "ba_net_write_angles: null argument");
  FILE *fout = fopen((std::string(dir) + "/angles.jck").c_str(), "w");
  if (!fout) return fail(BA_ERR_INVALID, "cannot write into %s", dir);
  for (size_t i = 0; i < n->imageTarget.size(); ++i) {
    const int32_t pose = n->indexPose[i], cam = n->staSensor[pose];
    const double alpha = ba_incidence_angle(&n->EOP[6 * (size_t)pose], &n->XYZ[3 * (size_t)n->indexPoint[i]]);
    const double beta = ba_refraction_angle(n->imageX[i], n->imageY[i], &n->IOP[3 * (size_t)cam], &n->AP[16 * (size_t)cam]);
    fprintf(fout, "%i %i %i %.6lf %.6lf\n", n->imageTarget[i], n->imageStation[i], n->eopCamera[pose], beta * 180.0 / kPi,
            alpha * 180.0 / kPi);
  }
  fclose(fout);
  return 0;
}

int ba_net_report(const ba_network *n, const ba_summary *sm, const double *residuals, int64_t n_residuals, int32_t bins,
                  ba_report *out, ba_bin_stats *bin_stats) {
  if (!n || !sm || !residuals || !out) return fail(BA_ERR_INVALID, "ba_net_report: null argument");
  const size_t N = n->imageX.size(), P = n->xyzTarget.size();
  if (n_residuals < (int64_t)(2 * N)) return fail(BA_ERR_INVALID, "ba_net_report: %lld residuals for %zu image points", (long long)n_residuals, N);
  if (bins < 0 || (bins > 0 && !bin_stats)) return fail(BA_ERR_INVALID, "ba_net_report: bins without an output array");
  std::memset(out, 0, sizeof *out);
  const int32_t nap = ba_net_num_ap_correction(n);
  // ---- src/main.cpp:4905-4915
  out->quadratic_error = 2 * sm->final_cost;
  out->redundancy = sm->num_residuals_reduced - sm->num_parameters_reduced - 3 * (double)P + 7 + nap;
  out->aposteriori_variance = 2 * sm->final_cost / out->redundancy;
  out->aposteriori_stddev = std::sqrt(out->aposteriori_variance);
  out->ic_n = (double)(sm->num_residuals_reduced - 3 * (int64_t)P);
  out->ic_k = (double)(sm->num_parameters_reduced - nap);
  out->aic = ba_aic(out->ic_n, 2 * sm->final_cost, out->ic_k);
  out->bic = ba_bic(out->ic_n, 2 * sm->final_cost, out->ic_k);
  // ---- :4919-4942; frames and targets are the distinct ids of the .pho file (:3097-3107)
  std::vector<int32_t> frames(n->imageStation), targets(n->imageTarget);
  std::sort(frames.begin(), frames.end());
  frames.erase(std::unique(frames.begin(), frames.end()), frames.end());
  std::sort(targets.begin(), targets.end());
  targets.erase(std::unique(targets.begin(), targets.end()), targets.end());
  out->n_frames = (int32_t)frames.size();
  out->n_targets = (int32_t)targets.size();
  double vtpv = 0.0;
  for (int64_t i = 0; i < n_residuals; ++i) vtpv += residuals[i] * residuals[i];
  out->vtpv = vtpv;
  out->approx_dof = (double)(2 * (int64_t)N - 6 * (int64_t)frames.size() - 3 * (int64_t)targets.size() + 7);
  double vi = 0.0;
  for (size_t i = 0; i < 2 * N; ++i) vi += residuals[i] * residuals[i];
  out->vtpv_image = vi;
  // ---- :6952-6984 image residuals in image units, reprojection errors
  double sx = 0.0, sy = 0.0, mx = 0.0, my = 0.0;
  out->res_x_min = out->res_y_min = N ? 1e300 : 0.0;
  out->res_x_max = out->res_y_max = N ? -1e300 : 0.0;
  std::vector<double> vx(N), vy(N);
  for (size_t i = 0; i < N; ++i) {
    vx[i] = residuals[2 * i] * n->imageXStdDev[i];
    vy[i] = residuals[2 * i + 1] * n->imageYStdDev[i];
    sx += vx[i] * vx[i];
    sy += vy[i] * vy[i];
    mx += vx[i];
    my += vy[i];
    out->res_x_min = std::min(out->res_x_min, vx[i]); out->res_x_max = std::max(out->res_x_max, vx[i]);
    out->res_y_min = std::min(out->res_y_min, vy[i]); out->res_y_max = std::max(out->res_y_max, vy[i]);
  }
  if (N) {
    out->res_x_mean = mx / N;
    out->res_y_mean = my / N;
    out->rmse_x = std::sqrt(sx / N);
    out->rmse_y = std::sqrt(sy / N);
    out->rmse_xy = std::sqrt((sx + sy) / (2 * N));
  }
  // ---- :4960-5090 quantile bins over the corrected radial distance (stable order of equal distances, :990-1005)
  out->bins = bins;
  if (bins > 0) {
    std::vector<double> dist(N);
    for (size_t i = 0; i < N; ++i) {
      const int32_t cam = n->staSensor[n->indexPose[i]];
      dist[i] = ba_radial_distance(n->imageX[i], n->imageY[i], &n->IOP[3 * (size_t)cam], &n->AP[16 * (size_t)cam]);
    }
    std::vector<int> idx(N);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&dist](size_t a, size_t b) { return dist[a] < dist[b]; });
    const int64_t width = (int64_t)std::ceil(double(N) / double(bins));
    out->bin_width = width;
    size_t l = 0;
    std::vector<double> tImg, tRes, tDist;
    for (int b = 0; b < bins; ++b) {
      tImg.clear(); tRes.clear(); tDist.clear();
      for (int64_t m = 0; m < width && l < N; ++m, ++l) {
        tImg.push_back(n->imageX[idx[l]]);
        tImg.push_back(n->imageY[idx[l]]);
        tRes.push_back(vx[idx[l]]);
        tRes.push_back(vy[idx[l]]);
        tDist.push_back(dist[idx[l]]);
      }
      bin_stats[b].count = (int64_t)tDist.size();
      ba_stat_all(tImg.data(), (int64_t)tImg.size(), &bin_stats[b].obs);
      ba_stat_all(tRes.data(), (int64_t)tRes.size(), &bin_stats[b].res);
      ba_stat_all(tDist.data(), (int64_t)tDist.size(), &bin_stats[b].dist);
    }
  }
  return 0;
}

int ba_net_print_report(const ba_network *n, const ba_report *r, const ba_bin_stats *bs, const char *path) {
  if (!n || !r) return fail(BA_ERR_INVALID, "ba_net_print_report: null argument");
  std::ostringstream o;  // the reference prints with std::cout in its default format (6 significant digits)
  const double dof = r->approx_dof;
  o << "  Estimating A Posterior Variance Factor..." << std::endl;
  o << "     *CERES GLOBAL A Posteriori Variance (using the robust weights): " << std::endl;  // src/main.cpp:4905-4915
  o << "        *Ceres Quadratic Error: " << r->quadratic_error << std::endl;
  o << "        *Ceres Redundancy: " << r->redundancy << std::endl;
  o << "        *Ceres A Posteriori Variance: " << r->aposteriori_variance << std::endl;
  o << "        *Ceres A Posteriori Std Dev: " << r->aposteriori_stddev << std::endl;
  o << "        *AIC n: " << r->aic << ". n = " << (int64_t)r->ic_n << ", k = " << (int64_t)r->ic_k << std::endl;
  o << "        *BIC n: " << r->bic << ". n = " << (int64_t)r->ic_n << ", k = " << (int64_t)r->ic_k << std::endl;
  o << "     MANUAL GLOBAL A Posteriori Variance (assumes normal weights): " << std::endl;  // :4925-4931
  o << "        vTPv: " << r->vtpv << std::endl;
  o << "        Approximate dof (2*img - 6*EOP - 3*XYZ + 7datum): " << (int64_t)dof << std::endl;
  o << "        vTPv/dof: " << r->vtpv / dof << std::endl;
  o << "        sqrt(vTPv/dof): " << std::sqrt(r->vtpv / dof) << std::endl;
  o << "        vTPv/ceresRedundancy: " << r->vtpv / r->redundancy << std::endl;
  o << "        sqrt(vTPv/ceresRedundancy): " << std::sqrt(r->vtpv / r->redundancy) << std::endl;
  o << "     Image A Posteriori Variance: " << std::endl;  // :4935-4940
  o << "        vTPv: " << r->vtpv_image << std::endl;
  o << "        Approx dof: " << (int64_t)dof << std::endl;
  o << "        Approx a posteriori variance image (used for scaling Cx = apostVar*Qx): " << r->vtpv_image / dof << std::endl;
  o << "        Approx a posteriori std dev image: " << std::sqrt(r->vtpv_image / dof) << std::endl;
  if (r->bins > 0 && bs) {  // :4962-5137
    o << "  Computing quantile residual statistics by dividing into " << r->bins << " bins..." << std::endl;
    o << "    binWidth: " << r->bin_width << " points" << std::endl;
    o << "    Min distanc: ";
    for (int b = 0; b < r->bins; ++b) o << bs[b].dist.min << "\t";
    o << std::endl;
    o << "    Median dist: ";
    for (int b = 0; b < r->bins; ++b) o << bs[b].dist.median << "\t";
    o << std::endl;
    o << "    Max distanc: ";
    for (int b = 0; b < r->bins; ++b) o << bs[b].dist.max << "\t";
    o << std::endl;
    std::ios state(nullptr);
    state.copyfmt(o);
    o << std::setprecision(2) << std::fixed;
    o << "    Mean residuals  : ";
    for (int b = 0; b < r->bins; ++b) o << bs[b].res.mean << "\t";
    o << std::endl;
    o << "    StdDev residuals: ";
    for (int b = 0; b < r->bins; ++b) o << bs[b].res.stdev << "\t";
    o << std::endl;
    o << "    RMSE residuals  : ";
    for (int b = 0; b < r->bins; ++b) o << bs[b].res.rmse << "\t";
    o << std::endl;
    o.copyfmt(state);
  }
  // :6982-6984
  o << "  Residual x mean (range): " << r->res_x_mean << " (" << r->res_x_min << " to " << r->res_x_max << ")" << std::endl;
  o << "  Residual y mean (range): " << r->res_y_mean << " (" << r->res_y_min << " to " << r->res_y_max << ")" << std::endl;
  o << "  Reprojection errors (RMSE in x, y, and average): " << r->rmse_x << ", " << r->rmse_y << " --> " << r->rmse_xy << std::endl;
  const std::string text = o.str();
  if (!path) {
    fputs(text.c_str(), stdout);
    return 0;
  }
  FILE *f = fopen(path, "w");
  if (!f) return fail(BA_ERR_INVALID, "cannot write %s", path);
  fputs(text.c_str(), f);
  fclose(f);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// correlation analysis
// ---------------------------------------------------------------------------------------------
namespace {
// |correlation| of one camera parameter against one coordinate of every station (or point), NaN left out
// (extractAPCorrelation, src/main.cpp:837-848), then calcStatistics (:851-881) into slots max, median, mean, min
void correlation_column(const double *cov_row, const double *var, int64_t count, int stride, int coord, double var_cam,
                        bool have_row, double *t_max, double *t_median, double *t_mean, double *t_min) {
  std::vector<double> v;
  if (have_row && cov_row && var) {
    v.reserve((size_t)count);
    for (int64_t i = 0; i < count; ++i) {
      const double c = cov_row[stride * i + coord] / (std::sqrt(var[stride * i + coord]) * std::sqrt(var_cam));
      if (!std::isnan(c)) v.push_back(std::fabs(c));
    }
  }
  ba_stats st;
  ba_stat_all(v.data(), (int64_t)v.size(), &st);
  if (v.empty()) st.max = st.median = st.mean = NAN;  // :5833-5834; min stays 0
  *t_max = st.max; *t_median = st.median; *t_mean = st.mean; *t_min = st.min;
}

// one row of an Eigen matrix on a stream: coefficients right-aligned to the widest of the row, one blank between
void print_row(std::ostringstream &o, const double *v, int n) {
  std::vector<std::string> t((size_t)n);
  size_t w = 0;
  for (int i = 0; i < n; ++i) {
    std::ostringstream c;
    c.copyfmt(o);
    c << v[i];
    t[i] = c.str();
    w = std::max(w, t[i].size());
  }
  for (int i = 0; i < n; ++i) {
    if (i) o << " ";
    o << std::string(w - t[i].size(), ' ') << t[i];
  }
}

// "Mean(fabs)" and "Median(fabs)" lines over the entries of a table that are not NaN (:5885-5896)
void print_table_stats(std::ostringstream &o, const double *v, int n, bool off_diagonal_of_16 = false) {
  std::vector<double> s;
  for (int i = 0; i < n; ++i) {
    if (off_diagonal_of_16 && i / 16 == i % 16) continue;
    if (!std::isnan(v[i])) s.push_back(std::fabs(v[i]));
  }
  ba_stats st;
  ba_stat_all(s.data(), (int64_t)s.size(), &st);
  o << "   Mean(fabs): " << st.mean << " +/- " << st.stdev << std::endl;
  o << "   Median(fabs): " << st.median << " (" << st.min << " to " << st.max << ")" << std::endl;
}
}  // namespace

int ba_correlations_compute(int32_t n_sta, int64_t n_pts, int32_t camera, int32_t n_rows, const int32_t *cam_param,
                            const double *var_eop, const double *var_xyz, const double *var_iop, const double *var_ap,
                            const double *cov_eop, const double *cov_xyz, const double *cov_cam, ba_correlations *out) {
  if (!out || n_sta < 0 || n_pts < 0 || camera < 0 || n_rows < 0 || (n_rows > 0 && !cam_param) || !var_iop || !var_ap)
    return fail(BA_ERR_INVALID, "ba_correlations_compute: bad argument");
  int row_of[19];
  for (int q = 0; q < 19; ++q) row_of[q] = -1;
  for (int k = 0; k < n_rows; ++k)
    if (cam_param[k] / 19 == camera) row_of[cam_param[k] % 19] = k;
  auto var_cam = [&](int q) { return q < 3 ? var_iop[3 * (size_t)camera + q] : var_ap[16 * (size_t)camera + q - 3]; };
  for (int q = 0; q < 19; ++q) {
    const int k = row_of[q];
    const double vc = var_cam(q);
    for (int m = 0; m < 6; ++m) {
      double *t = q < 3 ? &out->eop_iop[0][m][q] : &out->eop_ap[0][m][q - 3];
      const size_t step = q < 3 ? 6 * 3 : 6 * 16;
      correlation_column(k >= 0 && cov_eop ? cov_eop + (size_t)k * 6 * n_sta : nullptr, var_eop, n_sta, 6, m, vc, k >= 0, t, t + step,
                         t + 2 * step, t + 3 * step);
    }
    for (int m = 0; m < 3; ++m) {
      double *t = q < 3 ? &out->xyz_iop[0][m][q] : &out->xyz_ap[0][m][q - 3];
      const size_t step = q < 3 ? 3 * 3 : 3 * 16;
      correlation_column(k >= 0 && cov_xyz ? cov_xyz + (size_t)k * 3 * n_pts : nullptr, var_xyz, n_pts, 3, m, vc, k >= 0, t, t + step,
                         t + 2 * step, t + 3 * step);
    }
  }
  // camera corner: a constant parameter has zero covariance and zero variance, 0 / 0 = NaN as in the reference
  auto cov_cc = [&](int qa, int qb) {
    const int ka = row_of[qa], kb = row_of[qb];
    return (ka >= 0 && kb >= 0 && cov_cam) ? cov_cam[(size_t)ka * n_rows + kb] : 0.0;
  };
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 16; ++b) out->iop_ap[a][b] = cov_cc(a, 3 + b) / (std::sqrt(var_cam(a)) * std::sqrt(var_cam(3 + b)));
  for (int a = 0; a < 16; ++a)
    for (int b = 0; b < 16; ++b) out->ap_ap[a][b] = cov_cc(3 + a, 3 + b) / (std::sqrt(var_cam(3 + a)) * std::sqrt(var_cam(3 + b)));
  return 0;
}

int ba_print_correlations(const ba_correlations *c, const char *path) {
  if (!c) return fail(BA_ERR_INVALID, "ba_print_correlations: null argument");
  std::ostringstream o;
  o << std::setprecision(2) << std::fixed;  // :5618-5619
  static const char *kind[4] = {"Max", "Median", "Mean", "Min"};
  static const char *eop_name[6] = {"omega: ", "phi  : ", "kappa: ", "Xo   : ", "Yo   : ", "Zo   : "};
  static const char *xyz_name[3] = {"X: ", "Y: ", "Z: "};
  static const char *iop_head = "       xp\typ\tc", *ap_head = "       a1\ta2\tk1\tk2\tk3\tp1\tp2\tk4\tk5...";
  o << "Correlation Matrix AP-AP..." << std::endl;  // :5621-5636
  print_table_stats(o, &c->ap_ap[0][0], 256, true);
  for (int a = 0; a < 16; ++a) {
    print_row(o, c->ap_ap[a], 16);
    o << std::endl;
  }
  o << "============================= Correlation EOP-IOP =============================" << std::endl;  // :5882-5967
  for (int t = 0; t < 4; ++t) {
    o << "Correlation EOP-IOP " << kind[t] << "(fabs)" << std::endl;
    print_table_stats(o, &c->eop_iop[t][0][0], 18);
    o << iop_head << std::endl;
    for (int m = 0; m < 6; ++m) {
      o << eop_name[m];
      print_row(o, c->eop_iop[t][m], 3);
      o << std::endl;
    }
  }
  for (int t = 0; t < 4; ++t) {  // :6052-6139
    o << "Correlation EOP-AP " << kind[t] << "(fabs)" << std::endl;
    print_table_stats(o, &c->eop_ap[t][0][0], 96);
    o << ap_head << std::endl;
    for (int m = 0; m < 6; ++m) {
      o << eop_name[m];
      print_row(o, c->eop_ap[t][m], 16);
      o << std::endl;
    }
  }
  for (int t = 0; t < 4; ++t) {  // :6263-6338
    o << "Correlation XYZ-IOP " << kind[t] << "(fabs)" << std::endl;
    print_table_stats(o, &c->xyz_iop[t][0][0], 9);
    o << iop_head << std::endl;
    for (int m = 0; m < 3; ++m) {
      o << xyz_name[m];
      print_row(o, c->xyz_iop[t][m], 3);
      o << std::endl;
    }
  }
  o << "Correlation XYZ-AP Max(fabs)" << std::endl;  // :6391-6408: the maxima only
  print_table_stats(o, &c->xyz_ap[0][0][0], 48);
  o << ap_head << std::endl;
  for (int m = 0; m < 3; ++m) {
    o << xyz_name[m];
    print_row(o, c->xyz_ap[0][m], 16);
    o << std::endl;
  }
  o << "Correlation IOP-AP" << std::endl;  // :6469-6492
  print_table_stats(o, &c->iop_ap[0][0], 48);
  o << ap_head << std::endl;
  static const char *iop_name[3] = {"xp: ", "yp: ", "c : "};
  for (int a = 0; a < 3; ++a) {
    o << iop_name[a];
    print_row(o, c->iop_ap[a], 16);
    o << std::endl;
  }
  const std::string text = o.str();
  if (!path) {
    fputs(text.c_str(), stdout);
    return 0;
  }
  FILE *f = fopen(path, "w");
  if (!f) return fail(BA_ERR_INVALID, "cannot write %s", path);
  fputs(text.c_str(), f);
  fclose(f);
  return 0;
}

int ba_net_qc(const ba_network *n, const char *truth_path, ba_qc *out) {  // src/main.cpp:7242-7346
  if (!n || !truth_path || !out) return fail(BA_ERR_INVALID, "ba_net_qc: null argument");
  std::memset(out, 0, sizeof *out);
  std::ifstream inp(truth_path);
  if (!inp.is_open()) return fail(BA_ERR_INVALID, "cannot open %s", truth_path);
  std::vector<double> id;  // the reference keeps the ids as doubles (:7249)
  std::vector<double> xyz;
  while (true) {  // seven columns per row; the row read at end of file is dropped again (:7251-7271)
    int c0 = 0;
    double c[6] = {0, 0, 0, 0, 0, 0};
    inp >> c0 >> c[0] >> c[1] >> c[2] >> c[3] >> c[4] >> c[5];
    id.push_back(c0);
    xyz.insert(xyz.end(), c, c + 3);
    if (inp.eof()) break;
    if (inp.fail()) return fail(BA_ERR_INVALID, "%s: malformed token near record %zu", truth_path, id.size());
  }
  id.pop_back();
  xyz.resize(xyz.size() - 3);
  out->n_truth = (int32_t)id.size();
  out->n_estimated = (int32_t)n->xyzTarget.size();
  double mean[3] = {0, 0, 0}, rmse[3] = {0, 0, 0}, mn[3] = {1E10, 1E10, 1E10}, mx[3] = {-1E10, -1E10, -1E10};
  std::unordered_map<int32_t, int32_t> first;  // first estimated point with an id, as the inner loop's break finds it
  for (int32_t j = (int32_t)n->xyzTarget.size() - 1; j >= 0; --j) first[n->xyzTarget[j]] = j;
  int matches = 0;
  for (size_t i = 0; i < id.size(); ++i) {
    const auto it = first.find((int32_t)id[i]);
    if (it == first.end()) continue;
    for (int k = 0; k < 3; ++k) {
      const double diff = n->XYZ[3 * (size_t)it->second + k] - xyz[3 * i + k];
      mean[k] += diff;
      rmse[k] += std::pow(diff, 2.0);
      if (diff < mn[k]) mn[k] = diff;
      if (diff > mx[k]) mx[k] = diff;
    }
    ++matches;
  }
  out->n_matches = matches;
  for (int k = 0; k < 3; ++k) {
    out->mean[k] = mean[k] / matches;
    out->min[k] = mn[k];
    out->max[k] = mx[k];
    out->rmse[k] = std::sqrt(rmse[k] / matches);
  }
  out->rmse_average = std::sqrt((out->rmse[0] * out->rmse[0] + out->rmse[1] * out->rmse[1] + out->rmse[2] * out->rmse[2]) / 3.0);
  return 0;
}

// ---- hand-over to and from the regression step of the outer loop (src/main.cpp:3040-3056, :7544-7613)
int ba_net_set_corrections(ba_network *n, const double *cx, const double *cy) {
  if (!n || (!n->imageX.empty() && (!cx || !cy))) return fail(BA_ERR_INVALID, "ba_net_set_corrections: null argument");
  std::copy(cx, cx + n->imageX.size(), n->imageXCorr.begin());  // columns 7-8 of temp.pho (:3130)
  std::copy(cy, cy + n->imageY.size(), n->imageYCorr.begin());
  return 0;
}

int ba_read_ml_cost(const char *path, double *cost, double *redundancy, double *hyper) {  // :7562-7590
  if (!path || !cost || !redundancy || !hyper) return fail(BA_ERR_INVALID, "ba_read_ml_cost: null argument");
  std::ifstream inp(path);
  if (!inp.is_open()) return fail(BA_ERR_INVALID, "cannot open %s", path);
  std::vector<double> rows;
  while (true) {  // five columns per row; the row read at end of file is dropped again
    double c[5] = {0, 0, 0, 0, 0};
    inp >> c[0] >> c[1] >> c[2] >> c[3] >> c[4];
    rows.insert(rows.end(), c, c + 5);
    if (inp.eof()) break;
    if (inp.fail()) return fail(BA_ERR_INVALID, "%s: malformed token", path);
  }
  rows.resize(rows.size() - 5);
  if (rows.empty()) return fail(BA_ERR_INVALID, "%s: no complete row (the reference would read past the end, :7592)", path);
  *cost = rows[0];
  *redundancy = rows[1];
  hyper[0] = rows[2]; hyper[1] = rows[3]; hyper[2] = rows[4];
  return 0;
}

int ba_append_costs(const char *dir, int32_t iteration, double lsq_cost, double lsq_redundancy, double ml_cost,
                    double ml_redundancy, double reprojection_2d, double reconstruction_3d, const double *hyper) {
  if (!dir) return fail(BA_ERR_INVALID, "ba_append_costs: null argument");
  const std::string d = std::string(dir) + "/";
  const char *mode = iteration == 0 ? "w" : "a";  // :7594-7597, :7606-7609
  if (hyper) {
    FILE *f = fopen((d + "hyperparameters.jck").c_str(), mode);
    if (!f) return fail(BA_ERR_INVALID, "cannot write into %s", dir);
    fprintf(f, "%.6lf %.6lf %.6lf\n", hyper[0], hyper[1], hyper[2]);  // :7599
    fclose(f);
  }
  FILE *f = fopen((d + "costs.jck").c_str(), mode);
  if (!f) return fail(BA_ERR_INVALID, "cannot write into %s", dir);
  fprintf(f, "%.6lf %.6lf %.6lf %.6lf %.6lf %.6lf\n", 2.0 * lsq_cost, lsq_redundancy, ml_cost, ml_redundancy, reprojection_2d,
          reconstruction_3d);  // :7611
  fclose(f);
  return 0;
}

int ba_net_print_qc(const ba_qc *q, const char *truth_path, const char *path) {  // src/main.cpp:7246, 7274-7344
  if (!q) return fail(BA_ERR_INVALID, "ba_net_print_qc: null argument");
  std::ostringstream o;
  o << "  Ground truth filename: " << (truth_path ? truth_path : "") << std::endl;
  o << "  Number of XYZ Ground Truth Read: " << q->n_truth << std::endl;
  o << "  Number of XYZ estimated        : " << q->n_estimated << std::endl;
  o << "  Number of matching points used : " << q->n_matches << std::endl;
  const char *axis[3] = {"X", "Y", "Z"};
  for (int k = 0; k < 3; ++k)
    o << "    Mean " << axis[k] << " (range): " << q->mean[k] << " (" << q->min[k] << " to " << q->max[k] << ")" << std::endl;
  o << "    Direct method - RMSE X, Y, Z, Average: " << q->rmse[0] << ", " << q->rmse[1] << ", " << q->rmse[2] << " --> "
    << q->rmse_average << std::endl;
  const std::string text = o.str();
  FILE *f = path ? fopen(path, "w") : stdout;
  if (!f) return fail(BA_ERR_INVALID, "cannot write %s", path);
  fputs(text.c_str(), f);
  if (path) fclose(f);
  return 0;
}

}  // extern "C"
