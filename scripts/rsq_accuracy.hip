// Accuracy of the hardware reciprocal square root (v_rsq_f64) and of one and two Newton steps after it, over 35 decades:
// hipcc --offload-arch=gfx950 -O2 scripts/rsq_accuracy.hip -o /tmp/rsq && /tmp/rsq
// MI355X: 5.2e-08 (2^-24.2) raw, 37.6 ulp after one step, 1.24 ulp after two -- why the pivot chain of k_chol_diag takes two.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *y0, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y0[i] = __builtin_amdgcn_rsq(x[i]);
}
int main() {
  const int n = 1 << 22;
  double *hx = new double[n], *hy = new double[n];
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (s >> 11) * (1.0 / 9007199254740992.0);
    hx[i] = std::exp((u - 0.5) * 80.0);  // 1e-17 .. 1e17
  }
  double *dx, *dy;
  if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dy, n * 8) != hipSuccess) return 1;
  if (hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
  k<<<n / 256, 256>>>(dx, dy, n);
  if (hipMemcpy(hy, dy, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  double worst = 0, worst1 = 0, worst2 = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / sqrtl((long double)hx[i]);
    double e = std::fabs((double)((hy[i] - t) / t));
    if (e > worst) worst = e;
    double y = hy[i], d = hx[i];
    double r = std::fma(-(d * y), y, 1.0); y = std::fma(0.5 * y, r, y);
    e = std::fabs((double)((y - t) / t)); if (e > worst1) worst1 = e;
    r = std::fma(-(d * y), y, 1.0); y = std::fma(0.5 * y, r, y);
    e = std::fabs((double)((y - t) / t)); if (e > worst2) worst2 = e;
  }
  printf("v_rsq_f64 worst relative error %.3e (2^%.1f); after one Newton step %.3e (%.2f ulp); after two %.3e (%.2f ulp)\n", worst, std::log2(worst), worst1, worst1 / 1.11e-16, worst2, worst2 / 1.11e-16);
  return 0;
}
