// decode_elt.h -- the 16-bit element type of the decoder-step kernels (decode.hip, decode_gemm.h) as a trait, gfx950 only.
//
// The reference loads its report LLM with torch_dtype=torch.float16 (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:
// 72, 85, 92) and the round-1..4 decode kernels were bf16-only, so the reference's own checkpoint dtype could not decode on them.
// Every kernel of the step is now a template over one of the two traits below; what differs between them is exactly
//   f(v)      the element's value as fp32,
//   r(x)      fp32 -> element bits, round-to-nearest-even (fp16: overflow to inf, as torch's .half()),
//   rr(x)     x rounded through the element type (the rounding points of the modules' 16-bit tensor ops),
//   dot2      v_dot2_f32_bf16 / v_dot2_f32_f16,
//   mfma32    v_mfma_f32_16x16x32_bf16 / _f16   (8 elements per lane),
//   mfma16    v_mfma_f32_16x16x16_bf16 / _f16   (4 elements per lane),
// the memory layout, the LDS-DMA rings, the transpose reads (ds_read_b64_tr_b16 moves 16-bit lanes whatever they mean) and every
// wait count are the same instruction streams.
#pragma once
#include "mxvl_common.h"

namespace mxvl {

typedef float elt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int elt_u32x4 __attribute__((ext_vector_type(4)));
typedef short elt_s16x4 __attribute__((ext_vector_type(4)));

struct EltBf16 {
  static constexpr int dtype = MXVL_BF16;
  __device__ static inline float f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
  __device__ static inline uint16_t r(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  __device__ static inline float rr(float x) { return f(r(x)); }
  // the two elements of a packed word as fp32
  __device__ static inline float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
  __device__ static inline float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
  __device__ static inline float dot2(uint32_t a, uint32_t b, float c) {
    typedef short bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
  }
  __device__ static inline elt_f32x4 mfma32(elt_u32x4 a, elt_u32x4 b, elt_f32x4 c) {
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
  }
  __device__ static inline elt_f32x4 mfma16(elt_s16x4 a, elt_s16x4 b, elt_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
};

struct EltF16 {
  static constexpr int dtype = MXVL_F16;
  __device__ static inline float f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
  __device__ static inline uint16_t r(float x) { return __builtin_bit_cast(uint16_t, (_Float16)x); }
  __device__ static inline float rr(float x) { return (float)(_Float16)x; }
  __device__ static inline float lo(uint32_t w) { return f((uint16_t)w); }
  __device__ static inline float hi(uint32_t w) { return f((uint16_t)(w >> 16)); }
  __device__ static inline float dot2(uint32_t a, uint32_t b, float c) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
  }
  __device__ static inline elt_f32x4 mfma32(elt_u32x4 a, elt_u32x4 b, elt_f32x4 c) {
    typedef _Float16 v8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
  }
  __device__ static inline elt_f32x4 mfma16(elt_s16x4 a, elt_s16x4 b, elt_f32x4 c) {
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(v4, a), __builtin_bit_cast(v4, b), c, 0, 0, 0);
  }
};

// eight elements (one 16-byte load) as fp32
template <typename E> __device__ inline void elt_unpack8(const uint4 v, float* out) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    out[2 * j] = E::lo(w[j]);
    out[2 * j + 1] = E::hi(w[j]);
  }
}

// mxvl_dtype of a decode descriptor: 0 (what ABI <= 7 callers leave in the reserved field) means bf16
inline int decode_dtype(int32_t v) { return v == 0 ? MXVL_BF16 : v; }
// anything but bf16 / fp16 behind that default is refused (MXVL_ERR_DTYPE), not decoded as bf16
inline bool decode_dtype_ok(int32_t v) { return v == 0 || v == MXVL_BF16 || v == MXVL_F16; }

}  // namespace mxvl
