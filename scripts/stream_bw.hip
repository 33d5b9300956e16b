// Achievable HBM bandwidth of this device for the access shapes the solver's kernels use: 16-byte loads and stores per lane,
// whole arrays of 1-4 GB, grid-stride.  Prints GB/s for read-only, write-only, copy and triad.
//   hipcc --offload-arch=gfx950 -O3 scripts/stream_bw.hip -o scripts/stream_bw && scripts/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_read(const double2 *__restrict__ a, size_t n, double *__restrict__ out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 12345.678) out[0] = s;  // never true: keeps the loads
}
__global__ void __launch_bounds__(256) k_write(double2 *__restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_double2(1.0, 2.0);
}
__global__ void __launch_bounds__(256) k_write_nt(double2 *__restrict__ a, size_t n) {
  typedef double nt2 __attribute__((ext_vector_type(2)));
  nt2 v = {1.0, 2.0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, reinterpret_cast<nt2 *>(a) + i);
}
__global__ void __launch_bounds__(256) k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_triad(const double2 *__restrict__ a, const double2 *__restrict__ b, double2 *__restrict__ c, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 x = a[i], y = b[i];
    c[i] = make_double2(x.x + 3.0 * y.x, x.y + 3.0 * y.y);
  }
}

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30), n = bytes / 16;
  double2 *a, *b, *c;
  double *out;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes)); CHECK(hipMalloc(&out, 8));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grids[] = {2048, 8192, 32768};
  for (int g : grids) {
    for (int kind = 0; kind < 5; ++kind) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        switch (kind) {
          case 0: hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, out); break;
          case 1: hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, a, n); break;
          case 2: hipLaunchKernelGGL(k_write_nt, dim3(g), dim3(256), 0, 0, a, n); break;
          case 3: hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); break;
          default: hipLaunchKernelGGL(k_triad, dim3(g), dim3(256), 0, 0, a, b, c, n); break;
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      const double moved = bytes * (kind == 3 ? 2.0 : kind == 4 ? 3.0 : 1.0);
      const char *name[] = {"read", "write", "write (nontemporal)", "copy", "triad"};
      printf("grid %6d  %-20s %8.3f ms  %7.0f GB/s\n", g, name[kind], best, moved / best / 1e6);
    }
  }
  return 0;
}
