// Dev micro-benchmark: what does a grid-wide barrier cost inside ONE persistent launch (256 workgroups x 1024 threads, one per
// CU) on gfx950, against the ~5 us a dependent kernel launch costs the decode step today?
//   variant 0: release/acquire atomics at agent scope (the compiler adds the L2 write-back / invalidate)
//   variant 1: relaxed atomics only (data exchanged through coherent sc0 sc1 accesses, no cache maintenance)
// Each barrier is followed by a small coherent exchange (every workgroup writes 1 KB, then reads its neighbour's) so the
// measured figure includes making data visible.   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbar tools/grid_barrier_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int VARIANT>
__device__ inline bool grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (VARIANT == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { ok = false; break; }
      }
    } else {
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { ok = false; break; }
      }
    }
  }
  __syncthreads();
  return ok;
}

template <int VARIANT>
__global__ __launch_bounds__(1024) void bar_kernel(unsigned* counter, unsigned* data, int n_barriers, unsigned* bad) {
  const int nb = gridDim.x;
  unsigned wrong = 0;
  for (int b = 0; b < n_barriers; ++b) {
    // every workgroup publishes 1 KB
    if (threadIdx.x < 256) {
      unsigned* p = data + (size_t)(b & 1) * nb * 256 + blockIdx.x * 256 + threadIdx.x;
      if (VARIANT == 0) *p = (unsigned)(b * 1000 + blockIdx.x);
      else *(volatile unsigned*)p = (unsigned)(b * 1000 + blockIdx.x);
    }
    if (VARIANT == 1) __builtin_amdgcn_s_waitcnt(0);
    if (!grid_barrier<VARIANT>(counter, (unsigned)(b + 1) * nb)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000u); return; }
    if (threadIdx.x < 256) {
      const int nbr = (blockIdx.x + 37) % nb;
      const unsigned* p = data + (size_t)(b & 1) * nb * 256 + nbr * 256 + threadIdx.x;
      const unsigned v = VARIANT == 0 ? *p : *(const volatile unsigned*)p;
      wrong += v != (unsigned)(b * 1000 + nbr);
    }
  }
  if (wrong) atomicAdd(bad, wrong);
}

int main() {
  const int nb = 256, n_bar = 400;
  unsigned *counter, *data, *bad;
  hipMalloc(&counter, 4);
  hipMalloc(&bad, 4);
  hipMalloc(&data, (size_t)2 * nb * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int variant = 0; variant < 2; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(counter, 0, 4);
      hipMemset(bad, 0, 4);
      hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(nb), dim3(1024), 0, 0, counter, data, n_bar, bad);
      else hipLaunchKernelGGL(bar_kernel<1>, dim3(nb), dim3(1024), 0, 0, counter, data, n_bar, bad);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned hb = 0;
      hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("variant %d (%s): %d barriers in %.1f us = %.2f us per barrier + exchange; wrong reads %u\n", variant,
             variant == 0 ? "release/acquire, agent scope" : "relaxed atomics + sc0 sc1 data", n_bar, ms * 1e3, ms * 1e3 / n_bar, hb);
    }
  }
  return 0;
}
