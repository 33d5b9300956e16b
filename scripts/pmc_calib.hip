// pmc_calib.hip -- read patterns of known size for calibrating the FETCH_SIZE counter of rocprofv3 on this device.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/pmc_calib scripts/pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_calib -- scripts/pmc_calib
//
// Every kernel reads BYTES bytes of a 2 GiB array exactly once (far more than the 256 MiB Infinity Cache holds), in one
// of the access shapes the solver's kernels use, and adds what it read into one number per workgroup so that nothing is
// optimised away.  The counter's value over BYTES is the factor to apply to that shape.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__device__ __forceinline__ void finish(double s, double *out) {
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x % 1024, s);
}

// (a) contiguous, 16 bytes per lane and load
__global__ void calib_stream16(const double2 *__restrict__ a, size_t n2, double *out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  finish(s, out);
}
// (b) contiguous, 8 bytes per lane and load
__global__ void calib_stream8(const double *__restrict__ a, size_t n, double *out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  finish(s, out);
}
// (c) contiguous, 4 bytes per lane and load
__global__ void calib_stream4(const float *__restrict__ a, size_t n, double *out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  finish(s, out);
}
// (d) records of R * 16 bytes, one record per lane (consecutive lanes take consecutive records), R loads of 16 bytes:
//     R = 4 is the 64-byte observation record, R = 8 the 128-byte row record, R = 9 the 144-byte coupling block
template <int R>
__global__ void calib_records(const double2 *__restrict__ a, size_t nrec, double *out) {
  double s = 0.0;
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
    const double2 *p = a + r * R;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const double2 v = p[k];
      s += v.x + v.y;
    }
  }
  finish(s, out);
}
// (e) the same records taken in a scattered order (a fixed permutation by multiplication with an odd constant): the
//     gathers of the station sweeps
template <int R>
__global__ void calib_gather(const double2 *__restrict__ a, size_t nrec_pow2, double *out) {
  double s = 0.0;
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < nrec_pow2; r += (size_t)gridDim.x * blockDim.x) {
    const size_t q = (r * 2654435761ull) & (nrec_pow2 - 1);
    const double2 *p = a + q * R;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const double2 v = p[k];
      s += v.x + v.y;
    }
  }
  finish(s, out);
}
// (f) 144-byte records fetched by the whole wave, nine lanes per record (seven records per load)
__global__ void calib_records9(const double2 *__restrict__ a, size_t nrec, double *out) {
  double s = 0.0;
  const int lane = threadIdx.x & 63, slot = lane / 9, piece = lane - 9 * slot;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r0 = wave * 7; r0 < nrec; r0 += nwaves * 7) {
    const size_t r = r0 + slot;
    if (lane < 63 && r < nrec) {
      const double2 v = a[r * 9 + piece];
      s += v.x + v.y;
    }
  }
  finish(s, out);
}

int main() {
  const size_t BYTES = (size_t)2 << 30;
  double *buf = nullptr, *out = nullptr;
  CHECK(hipMalloc((void **)&buf, BYTES));
  CHECK(hipMalloc((void **)&out, 1024 * sizeof(double)));
  CHECK(hipMemset(buf, 0, BYTES));
  CHECK(hipMemset(out, 0, 1024 * sizeof(double)));
  const dim3 g(8192), b(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_stream16, g, b, 0, 0, (const double2 *)buf, BYTES / 16, out);
    hipLaunchKernelGGL(calib_stream8, g, b, 0, 0, (const double *)buf, BYTES / 8, out);
    hipLaunchKernelGGL(calib_stream4, g, b, 0, 0, (const float *)buf, BYTES / 4, out);
    hipLaunchKernelGGL(calib_records<4>, g, b, 0, 0, (const double2 *)buf, BYTES / 64, out);
    hipLaunchKernelGGL(calib_records<8>, g, b, 0, 0, (const double2 *)buf, BYTES / 128, out);
    hipLaunchKernelGGL(calib_records<9>, g, b, 0, 0, (const double2 *)buf, BYTES / 144, out);
    hipLaunchKernelGGL(calib_gather<4>, g, b, 0, 0, (const double2 *)buf, BYTES / 64, out);
    hipLaunchKernelGGL(calib_gather<8>, g, b, 0, 0, (const double2 *)buf, BYTES / 128, out);
    hipLaunchKernelGGL(calib_records9, g, b, 0, 0, (const double2 *)buf, BYTES / 144, out);
    CHECK(hipDeviceSynchronize());
  }
  printf("bytes per kernel: stream16 %zu stream8 %zu stream4 %zu records<4> %zu records<8> %zu records<9> %zu gather<4> %zu gather<8> %zu records9 %zu\n",
         BYTES, BYTES, BYTES, BYTES / 64 * 64, BYTES / 128 * 128, BYTES / 144 * 144, BYTES / 64 * 64, BYTES / 128 * 128, BYTES / 144 * 144);
  CHECK(hipFree(buf));
  CHECK(hipFree(out));
  return 0;
}
