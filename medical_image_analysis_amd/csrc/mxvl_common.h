// mxvl_common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels of libmxvl.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "mxvl.h"

namespace mxvl {

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;

// ---- io element <-> fp32 ---------------------------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

template <typename T> struct Io;
template <> struct Io<float> {
  static constexpr int dtype = MXVL_F32;
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float x) { *p = x; }
};
template <> struct Io<bf16_t> {
  static constexpr int dtype = MXVL_BF16;
  __device__ static inline float ld(const bf16_t* p) {
    return __builtin_bit_cast(float, (uint32_t)p->v << 16);
  }
  __device__ static inline void st(bf16_t* p, float x) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    if ((u & 0x7fffffffu) > 0x7f800000u) { p->v = (uint16_t)((u >> 16) | 0x40); return; }
    u += 0x7fffu + ((u >> 16) & 1u);
    p->v = (uint16_t)(u >> 16);
  }
};
// two fp32 -> packed bf16x2 (lo = a, hi = b), round-to-nearest-even, in ONE VALU op (gfx950 v_cvt_pk_bf16_f32;
// no compiler builtin exists for it)
__device__ inline uint32_t cvt_pk_bf16(float a, float b) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <> struct Io<f16_t> {
  static constexpr int dtype = MXVL_F16;
  __device__ static inline float ld(const f16_t* p) { return (float)p->v; }
  __device__ static inline void st(f16_t* p, float x) { p->v = (_Float16)x; }
};

// ---- math ------------------------------------------------------------------------------------
__device__ inline float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }       // v_exp_f32
__device__ inline float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ inline float fast_log2(float x) { return __builtin_amdgcn_logf(x); }         // v_log_f32
__device__ inline float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// F.softplus(x), beta=1, threshold=20.  log1p(w) = log(1+w) * w / ((1+w)-1) keeps full relative
// accuracy for tiny w = e^x (dt_min = 1e-3 lives there) at the cost of one v_rcp.
__device__ inline float softplus(float x) {
  const float w = fast_exp(x);
  const float s = 1.0f + w;
  const float den = s - 1.0f;
  const float l = fast_log2(s) * 0.6931471805599453f;
  const float r = (den == 0.0f) ? w : l * w * fast_rcp(den);
  return x > 20.0f ? x : r;
}
__device__ inline float sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ inline float silu(float x) { return x * sigmoid(x); }

// ---- DPP cross-lane moves (gfx9 encodings) -------------------------------------------------------
// Lanes whose source is outside the row / disabled by row_mask keep `old` (bound_ctrl = 0).
constexpr int DPP_ROW_SHR(int n) { return 0x110 + n; }
constexpr int DPP_ROW_SHL(int n) { return 0x100 + n; }
constexpr int DPP_WAVE_SHR1 = 0x138;
constexpr int DPP_WAVE_SHL1 = 0x130;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK = 0xf>
__device__ inline float dpp(float old, float src) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                         ROW_MASK, 0xf, false));
}

}  // namespace mxvl
