// valu_rates.hip -- measures the VALU issue rates the scan kernel is planned against on gfx950:
// v_fma_f32, v_pk_fma_f32, v_exp_f32, v_mul_f32 with a DPP operand.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float float2_ __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define CHAINS 8

__global__ void k_fma(float* out, float a, float b) {
  float x[CHAINS];
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkfma(float* out, float a, float b) {
  float2_ x[CHAINS]; float2_ av = {a, a}, bv = {b, b};
  for (int i = 0; i < CHAINS; ++i) x[i] = float2_{threadIdx.x * 1e-3f + i, 1.0f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
  }
  float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_exp(float* out, float a, float b) {
  float x[CHAINS];
  for (int i = 0; i < CHAINS; ++i) x[i] = -(threadIdx.x * 1e-3f + i);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
  }
  float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dppmul(float* out, float a, float b) {
  float x[CHAINS];
  for (int i = 0; i < CHAINS; ++i) x[i] = 1.0f + threadIdx.x * 1e-6f + i * 1e-7f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) asm volatile("v_mul_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(a));
  }
  float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 1 exp : 6 fma mix, as in the scan inner loop
__global__ void k_mix(float* out, float a, float b) {
  float x[CHAINS], e[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { x[i] = threadIdx.x * 1e-3f + i; e[i] = -x[i]; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                   : "+v"(x[i]) : "v"(a), "v"(b));
    }
  }
  float s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K> static double run(K k, const char* name, double lane_ops_per_iter, int waves_per_simd) {
  int blocks = 256 * waves_per_simd, threads = 256;  // 4 waves per block -> waves_per_simd per SIMD
  float* out; hipMalloc(&out, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double insts_per_wave = (double)ITERS * CHAINS * lane_ops_per_iter;
  double waves = (double)blocks * threads / 64;
  double wave_insts = insts_per_wave * waves;
  double per_simd_per_s = wave_insts / (ms * 1e-3) / 1024.0;   // wave-instructions / s / SIMD
  printf("%-10s waves/SIMD=%d  %.3f ms  %.2f G wave-inst/s/SIMD  => %.2f cycles/wave-inst @2.4GHz (%.2f @2.0)\n",
         name, waves_per_simd, ms, per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s, 2.0e9 / per_simd_per_s);
  hipFree(out);
  return ms;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run(k_fma, "fma", 1, w);
    run(k_pkfma, "pk_fma", 1, w);
    run(k_exp, "exp", 1, w);
    run(k_dppmul, "mul_dpp", 1, w);
    run(k_mix, "exp+6fma", 7, w);
  }
  return 0;
}
