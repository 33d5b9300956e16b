// Host arithmetic of the projected line search that a bounds-constrained solve runs on every trust-region step:
// the step-size contraction of [Ceres 1.13] line_search.cc (ArmijoLineSearch, CUBIC interpolation) with the
// polynomial routines of polynomial.cc.  No device code; the trial points are evaluated by ba_api.hip.
#include <algorithm>
#include <array>
#include <cmath>
#include <vector>

#include "ba_gpu.h"

namespace {

using Poly = std::vector<double>;  // coefficients, highest power first

double horner(const Poly &p, double x) {
  double v = 0.0;
  for (double c : p) v = v * x + c;
  return v;
}

Poly derivative(const Poly &p) {
  Poly d;
  const int deg = (int)p.size() - 1;
  for (int i = 0; i < deg; ++i) d.push_back((deg - i) * p[i]);
  return d;
}

// polynomial of degree (#constraints - 1) through the valid values and gradients [FindInterpolatingPolynomial]
Poly interpolate(const std::vector<ba_ls_sample> &samples) {
  int m = 0;
  for (const auto &s : samples) m += (s.value_is_valid ? 1 : 0) + (s.gradient_is_valid ? 1 : 0);
  const int deg = m - 1;
  std::vector<std::vector<double>> a(m, std::vector<double>(m + 1, 0.0));
  int row = 0;
  for (const auto &s : samples) {
    if (s.value_is_valid) {
      for (int j = 0; j <= deg; ++j) a[row][j] = std::pow(s.x, deg - j);
      a[row++][m] = s.value;
    }
    if (s.gradient_is_valid) {
      for (int j = 0; j < deg; ++j) a[row][j] = (deg - j) * std::pow(s.x, deg - j - 1);
      a[row++][m] = s.gradient;
    }
  }
  // Gauss-Jordan with complete pivoting (the reference solver uses Eigen's fullPivLu)
  std::vector<int> col_of(m);
  for (int j = 0; j < m; ++j) col_of[j] = j;
  for (int k = 0; k < m; ++k) {
    int pr = k, pc = k;
    for (int r = k; r < m; ++r)
      for (int c = k; c < m; ++c)
        if (std::fabs(a[r][c]) > std::fabs(a[pr][pc])) { pr = r; pc = c; }
    if (a[pr][pc] == 0.0) break;
    std::swap(a[k], a[pr]);
    for (auto &r : a) std::swap(r[k], r[pc]);
    std::swap(col_of[k], col_of[pc]);
    for (int r = 0; r < m; ++r) {
      if (r == k) continue;
      const double f = a[r][k] / a[k][k];
      if (f != 0.0)
        for (int c = k; c <= m; ++c) a[r][c] -= f * a[k][c];
    }
  }
  Poly p(m, 0.0);
  for (int k = 0; k < m; ++k)
    if (a[k][k] != 0.0) p[col_of[k]] = a[k][m] / a[k][k];
  return p;
}

// real roots inside [lo, hi].  Between neighbouring stationary points a polynomial is monotone, so each sign change
// there holds one root; bisection runs until the bracket is one ulp wide.
std::vector<double> roots_within(Poly p, double lo, double hi) {
  while (!p.empty() && p.front() == 0.0) p.erase(p.begin());
  std::vector<double> out;
  if (p.size() < 2) return out;
  if (p.size() == 2) {
    const double r = -p[1] / p[0];
    if (r >= lo && r <= hi) out.push_back(r);
    return out;
  }
  std::vector<double> knots = roots_within(derivative(p), lo, hi);
  std::sort(knots.begin(), knots.end());
  knots.insert(knots.begin(), lo);
  knots.push_back(hi);
  auto add = [&out](double r) { if (out.empty() || out.back() != r) out.push_back(r); };
  for (size_t i = 0; i + 1 < knots.size(); ++i) {
    double l = knots[i], h = knots[i + 1], fl = horner(p, l);
    const double fh = horner(p, h);
    if (fl == 0.0) { add(l); continue; }
    if (fh == 0.0 || (fl < 0.0) == (fh < 0.0)) continue;
    for (;;) {
      const double mid = 0.5 * (l + h);
      if (!(mid > l && mid < h)) break;
      const double fm = horner(p, mid);
      if (fm == 0.0) { l = h = mid; break; }
      if ((fm < 0.0) == (fl < 0.0)) { l = mid; fl = fm; } else h = mid;
    }
    add(0.5 * (l + h));
  }
  if (horner(p, hi) == 0.0) add(hi);
  return out;
}

}  // namespace

extern "C" double ba_line_search_contract(const ba_ls_sample *start, const ba_ls_sample *previous, const ba_ls_sample *current,
                                          double min_step, double max_step) {
  if (!current->value_is_valid) return std::min(std::max(current->x * 0.5, min_step), max_step);
  std::vector<ba_ls_sample> samples{*start, *current};
  if (previous && previous->value_is_valid) samples.push_back(*previous);
  const Poly p = interpolate(samples);
  // [MinimizePolynomial] the middle of the interval, its ends, the stationary points inside; strict improvement only
  double best_x = (min_step + max_step) / 2.0, best = horner(p, best_x);
  auto offer = [&](double x) {
    const double v = horner(p, x);
    if (v < best) { best = v; best_x = x; }
  };
  offer(min_step);
  offer(max_step);
  // (Ceres also offers the real parts of complex root pairs of the derivative; such a point is not stationary, so its value
  //  cannot lie below the smallest of the ends and the real stationary points inside -- offering it changes nothing)
  if (p.size() > 2)
    for (double r : roots_within(derivative(p), min_step, max_step)) offer(r);
  // [MinimizeInterpolatingPolynomial] and the samples that lie inside
  for (const auto &s : samples)
    if (s.x >= min_step && s.x <= max_step) offer(s.x);
  return best_x;
}
