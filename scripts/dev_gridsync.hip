// Micro-benchmark: cost of a grid-wide barrier on one device, three designs:
//  0 one arrival counter; 1 per-workgroup flags polled by every workgroup's first wave; 2 two-level counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define LIMIT (1u << 16)
template <int KIND, int FENCE>
__global__ void k_sync(unsigned *w, int iters, double *buf) {
  unsigned epoch = 0;
  for (int it = 0; it < iters; ++it) {
    if (buf) buf[(size_t)blockIdx.x * blockDim.x + threadIdx.x] += 1.0;
    __syncthreads();
    epoch += 1;
    if (KIND == 0) {
      if (threadIdx.x == 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * gridDim.x;
        unsigned spins = 0;
        while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > LIMIT) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    } else if (KIND == 1) {
      if (threadIdx.x < 64) {
        if (threadIdx.x == 0) {
          if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_store(w + 64 + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        for (;;) {
          unsigned m = 0xffffffffu;
          for (unsigned k = threadIdx.x; k < gridDim.x; k += 64) {
            const unsigned v = __hip_atomic_load(w + 64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = v < m ? v : m;
          }
          const bool ok = m >= epoch;
          if (__all(ok)) break;
          if (++spins > LIMIT) break;
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    } else {
      if (threadIdx.x == 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned grp = blockIdx.x >> 4, ngrp = (gridDim.x + 15) >> 4;
        const unsigned gsize = min(16u, gridDim.x - grp * 16u);
        const unsigned old = __hip_atomic_fetch_add(w + 64 + 32 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * gsize) __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * ngrp;
        unsigned spins = 0;
        while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > LIMIT) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
  }
}
template <int KIND, int FENCE>
void run(unsigned *w, double *buf, int nb, int th, int wb) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipMemset(w, 0, 1 << 16);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_sync<KIND, FENCE>), dim3(nb), dim3(th), 0, 0, w, iters, wb ? buf : nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("kind=%d fence=%d blocks=%4d threads=%4d dirty=%d : %.2f us per barrier\n", KIND, FENCE, nb, th, wb, 1e3 * ms / iters);
  fflush(stdout);
}
int main() {
  unsigned *w; double *buf;
  (void)hipMalloc(&w, 1 << 16); (void)hipMalloc(&buf, 8u << 20); (void)hipMemset(buf, 0, 8u << 20);
  for (int nb : {128, 256})
    for (int th : {512, 1024})
      for (int wb = 0; wb < 2; ++wb) {
        run<0, 0>(w, buf, nb, th, wb); run<0, 1>(w, buf, nb, th, wb);
        run<1, 0>(w, buf, nb, th, wb); run<1, 1>(w, buf, nb, th, wb);
        run<2, 0>(w, buf, nb, th, wb); run<2, 1>(w, buf, nb, th, wb);
      }
  return 0;
}
