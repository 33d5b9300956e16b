// Issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on one SIMD: a single wave, long runs of independent
// instructions, clocked with s_memtime.
// hipcc --offload-arch=gfx950 -O2 scripts/mfma_f64_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma(double *out, long long *cycles, int reps) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  const long long t1 = clock64();
  out[threadIdx.x & 63] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ void __launch_bounds__(64) k_mfma_dep(double *out, long long *cycles, int reps) {
  d4 a0 = {0, 0, 0, 0};
  const double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  const long long t0 = clock64();
  for (int r = 0; r < 4 * reps; ++r) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
  const long long t1 = clock64();
  out[threadIdx.x] = a0[0] + a0[1] + a0[2] + a0[3];
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_fma(double *out, long long *cycles, int reps) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  const double x = 1.0 + threadIdx.x * 1e-9, y = threadIdx.x * 1e-9;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = fma(a[i], x, y);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x & 63] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

__global__ void __launch_bounds__(64) k_fma_dep(double *out, long long *cycles, int reps) {
  double a = threadIdx.x;
  const double x = 1.0 + threadIdx.x * 1e-9, y = threadIdx.x * 1e-9;
  const long long t0 = clock64();
  for (int r = 0; r < 8 * reps; ++r) a = fma(a, x, y);
  const long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
  double *out;
  long long *cyc, h = 0;
  if (hipMalloc(&out, 64 * 8) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
  const int reps = 4096;
  for (int pass = 0; pass < 2; ++pass) {
    k_mfma<<<1, 64>>>(out, cyc, reps);
    if (hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (pass) printf("independent v_mfma_f64_16x16x4_f64: %.1f shader clocks each (2048 flop)\n", (double)h / (4.0 * reps));
    k_mfma_dep<<<1, 64>>>(out, cyc, reps);
    if (hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (pass) printf("dependent   v_mfma_f64_16x16x4_f64: %.1f shader clocks each\n", (double)h / (4.0 * reps));
    k_fma<<<1, 64>>>(out, cyc, reps);
    if (hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (pass) printf("independent v_fma_f64: %.2f shader clocks each (128 flop)\n", (double)h / (8.0 * reps));
    k_fma_dep<<<1, 64>>>(out, cyc, reps);
    if (hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (pass) printf("dependent   v_fma_f64: %.2f shader clocks each\n", (double)h / (8.0 * reps));
  }
  // the clock of the counter against wall time, and the rates in wall time
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
  const int big = 1 << 18;
  float ms = 0;
  (void)hipEventRecord(e0, 0);
  k_mfma<<<1, 64>>>(out, cyc, big);
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1;
  if (hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  printf("counter: %.0f MHz; one v_mfma_f64_16x16x4_f64 per %.1f ns on a SIMD = %.1f GFLOP/s per SIMD, x 1024 SIMDs = %.1f TFLOP/s\n",
         (double)h / (ms * 1e3), ms * 1e6 / (4.0 * big), 2048.0 * 4.0 * big / (ms * 1e6), 2048.0 * 4.0 * big / (ms * 1e6) * 1024 * 1e-3);
  (void)hipEventRecord(e0, 0);
  k_fma<<<1, 64>>>(out, cyc, 8 * big);
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1;
  printf("one v_fma_f64 per %.2f ns on a SIMD = %.1f GFLOP/s per SIMD, x 1024 SIMDs = %.1f TFLOP/s\n", ms * 1e6 / (64.0 * big),
         128.0 * 64.0 * big / (ms * 1e6), 128.0 * 64.0 * big / (ms * 1e6) * 1024 * 1e-3);
  // all SIMDs busy: 1024 waves of the same
  (void)hipEventRecord(e0, 0);
  k_mfma<<<1024, 64>>>(out, cyc, big);
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1;
  printf("1024 waves of v_mfma_f64_16x16x4_f64: %.1f TFLOP/s\n", 1024.0 * 2048.0 * 4.0 * big / (ms * 1e9));
  for (int waves = 2048; waves <= 8192; waves *= 2) {
    (void)hipEventRecord(e0, 0);
    k_mfma<<<waves / 4, 256>>>(out, cyc, big / 4);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1;
    printf("%d waves (workgroups of 4) of v_mfma_f64_16x16x4_f64: %.1f TFLOP/s\n", waves, (double)waves * 2048.0 * big / (ms * 1e9));
    (void)hipEventRecord(e0, 0);
    k_fma<<<waves / 4, 256>>>(out, cyc, 2 * big);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1;
    printf("%d waves (workgroups of 4) of v_fma_f64: %.1f TFLOP/s\n", waves, (double)waves * 128.0 * 16.0 * big / (ms * 1e9));
  }
  return 0;
}
