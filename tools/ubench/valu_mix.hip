// valu_mix.hip -- why does a v_exp_f32 : v_fma_f32 mix issue slower than the two rates predict?  Variants of the
// scan inner loop's instruction mix: dependent vs independent fma chains, exps interleaved vs batched.
//   hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix && ./valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
#define C 8

// A: per chain: exp, then 6 DEPENDENT fmas (the r01 "exp+6fma" test)
__global__ void k_dep(float* out, float a, float b) {
  float x[C], e[C];
  for (int i = 0; i < C; ++i) { x[i] = threadIdx.x * 1e-3f + i; e[i] = -x[i]; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < C; ++i) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                   : "+v"(x[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0; for (int i = 0; i < C; ++i) s += x[i] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// B: 8 exps back to back, then 6 rounds of 8 INDEPENDENT fmas
__global__ void k_batched(float* out, float a, float b) {
  float x[C], e[C];
  for (int i = 0; i < C; ++i) { x[i] = threadIdx.x * 1e-3f + i; e[i] = -x[i]; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < C; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < C; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  float s = 0; for (int i = 0; i < C; ++i) s += x[i] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// C: interleaved: exp(i) followed by 6 fmas on 6 DIFFERENT chains
__global__ void k_inter(float* out, float a, float b) {
  float x[C], e[C];
  for (int i = 0; i < C; ++i) { x[i] = threadIdx.x * 1e-3f + i; e[i] = -x[i]; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < C; ++i) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
#pragma unroll
      for (int r = 0; r < 6; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i + r) % C]) : "v"(a), "v"(b));
    }
  }
  float s = 0; for (int i = 0; i < C; ++i) s += x[i] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// D: 48 dependent-in-pairs fmas only (2 chains alternate) -- dependent-issue cost without exp
__global__ void k_dep2(float* out, float a, float b) {
  float x[C];
  for (int i = 0; i < C; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < C; ++i)
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                   : "+v"(x[i]) : "v"(a), "v"(b));
  }
  float s = 0; for (int i = 0; i < C; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// E: exp result consumed by the next fma (exp -> fma dependency, as a = exp2(..); h = fma(a, h, b))
__global__ void k_expdep(float* out, float a, float b) {
  float x[C], e[C];
  for (int i = 0; i < C; ++i) { x[i] = threadIdx.x * 1e-3f + i; e[i] = -x[i] * 1e-3f; }
  for (int it = 0; it < ITERS; ++it) {
    float t[C];
#pragma unroll
    for (int i = 0; i < C; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(t[i]) : "v"(e[i]));
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < C; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(t[i]), "v"(b));
  }
  float s = 0; for (int i = 0; i < C; ++i) s += x[i] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// F: mul+exp pairs (exp input produced by the instruction right before it), then fmas
__global__ void k_mulexp(float* out, float a, float b) {
  float x[C], e[C];
  for (int i = 0; i < C; ++i) { x[i] = threadIdx.x * 1e-3f + i; e[i] = -x[i] * 1e-3f; }
  for (int it = 0; it < ITERS; ++it) {
    float t[C];
#pragma unroll
    for (int i = 0; i < C; ++i) asm volatile("v_mul_f32 %0, %1, %2\n v_exp_f32 %0, %0" : "=&v"(t[i]) : "v"(e[i]), "v"(a));
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int i = 0; i < C; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(t[i]), "v"(b));
  }
  float s = 0; for (int i = 0; i < C; ++i) s += x[i] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K> static void run(K k, const char* name, int w) {
  int blocks = 256 * w, threads = 256;
  float* out; hipMalloc(&out, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double wave_insts = (double)ITERS * C * 7 * blocks * threads / 64;
  double per_simd = wave_insts / (ms * 1e-3) / 1024.0;
  printf("%-22s waves/SIMD=%d  %.3f ms  %.2f cycles/wave-inst @2.4GHz\n", name, w, ms, 2.4e9 / per_simd);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 3, 4, 8}) {
    run(k_dep, "exp+6 dep fma", w);
    run(k_batched, "8exp then 48 indep", w);
    run(k_inter, "exp+6 indep (interl.)", w);
    run(k_dep2, "7 dep fma", w);
    run(k_expdep, "8exp -> 48 fma(use)", w);
    run(k_mulexp, "8(mul,exp) -> 40 fma", w);
  }
  return 0;
}
