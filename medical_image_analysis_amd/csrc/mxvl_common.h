// mxvl_common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels of libmxvl.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "mxvl.h"

// Measurement-only ablation switches (tools/*_bench.py).  They exist ONLY in builds made with -DMXVL_ABLATE
// (`python -m medical_image_analysis_amd.build --ablate` -> build/libmxvl_ablate.so); in the product library every
// MXVL_ABL(...) condition is the constant false, no environment variable is read and no ablation code is emitted.
#ifdef MXVL_ABLATE
#include <stdlib.h>
#define MXVL_ABL(cond) (cond)
#define MXVL_ABL_ENV(name) (getenv(name) ? atoi(getenv(name)) : 0)
#else
#define MXVL_ABL(cond) (false)
#define MXVL_ABL_ENV(name) (0)
#endif

// A/B experiment bits (build.py --exp N -> build/libmxvl_exp<N>.so): 0 in the product library.  An `#if MXVL_EXP & bit` block
// lives only while its experiment is open; the winner becomes the code, the loser is deleted.
#ifndef MXVL_EXP
#define MXVL_EXP 0
#endif

namespace mxvl {

constexpr int kWave = 64;
typedef float v2f __attribute__((ext_vector_type(2)));   // an aligned VGPR pair: operands of v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32
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

// F.softplus(x), beta=1, threshold=20.  log1p(w), w = e^x, with full relative accuracy for tiny w (dt_min = 1e-3 lives there):
//   log(1 + w) = log(s) + (w - (s - 1)) / s + O(e^2),  s = fl(1 + w),  s - 1 exact (Sterbenz)
// and 1 / s only has to be right to a factor where the correction matters at all: max(1 - (s - 1), 0) -- within 2 x of 1 / s for s < 2,
// zero beyond, where the rounding error of s is below an ulp of log(s).  One transcendental less than the round-5 form
// log(s) * w / (s - 1) (a v_rcp per element: 8 per 128-step chunk and lane) and no special case for s == 1; against float64 over
// x in [-30, 20]: max relative error 2.1e-7 (2.8e-7 before).
__device__ inline float softplus(float x) {
  const float w = fast_exp(x);
  const float s = 1.0f + w;
  const float den = s - 1.0f;
  const float l = fast_log2(s) * 0.6931471805599453f;
  const float r = fmaf(w - den, fmaxf(1.0f - den, 0.0f), l);
  return x > 20.0f ? x : r;
}
__device__ inline float sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ inline float silu(float x) { return x * sigmoid(x); }

// ---- 4-element row accesses in the io dtype (16 bytes fp32, 8 bytes bf16/fp16) -------------------------------------
template <typename io_t> __device__ inline float4 ld4(const io_t* p);
template <> __device__ inline float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ inline float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 r = *(const uint2*)p;
  return make_float4(__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                     __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u));
}
template <> __device__ inline float4 ld4<f16_t>(const f16_t* p) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 r = *(const h4*)p;
  return make_float4((float)r.x, (float)r.y, (float)r.z, (float)r.w);
}
template <typename io_t> __device__ inline void st4(io_t* p, float4 v);
template <> __device__ inline void st4<float>(float* p, float4 v) { *(float4*)p = v; }
template <> __device__ inline void st4<bf16_t>(bf16_t* p, float4 v) {
  *(uint2*)p = make_uint2(cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w));
}
template <> __device__ inline void st4<f16_t>(f16_t* p, float4 v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  *(h4*)p = h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
}

// `out` (forward) / `dout` (backward) of the scan: io dtype, or fp32 for any io dtype when the caller set MXVL_SCAN_OUT_F32
// (the reference's "oflex" i16o32 mode, cusoflex/selective_scan_oflex.cpp:150,207).  `off` is an element offset; the flag is
// uniform, so the branch costs one scalar compare per store site.
template <typename io_t> __device__ inline void st4_out(void* base, int64_t off, float4 v, bool f32) {
  if (f32) *(float4*)((float*)base + off) = v; else st4<io_t>((io_t*)base + off, v);
}
template <typename io_t> __device__ inline void st_out(void* base, int64_t off, float v, bool f32) {
  if (f32) ((float*)base)[off] = v; else Io<io_t>::st((io_t*)base + off, v);
}
template <typename io_t> __device__ inline float4 ld4_out(const void* base, int64_t off, bool f32) {
  return f32 ? *(const float4*)((const float*)base + off) : ld4<io_t>((const io_t*)base + off);
}
template <typename io_t> __device__ inline float ld_out(const void* base, int64_t off, bool f32) {
  return f32 ? ((const float*)base)[off] : Io<io_t>::ld((const io_t*)base + off);
}

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

// Row of `delta` / entry of `delta_bias` that channel d reads when delta carries dim / ratio channels (the vendored oflex
// kernels' dim_deltagroups_ratio, cusoflex/selective_scan_fwd_kernel_oflex.cuh:91-107).  ratio is a kernel argument: the
// select is on a scalar condition and the division is one v_mul_hi by the host-computed reciprocal (delta_magic).
__device__ inline int delta_row(int d, int ratio, uint32_t magic) { return ratio <= 1 ? d : (int)__umulhi((uint32_t)d, magic); }
// magic = floor(2^32 / ratio) + 1: umulhi(d, magic) == d / ratio for every d with d * ratio < 2^32 (checked by the callers: d < dim)
inline uint32_t delta_magic(int ratio) { return ratio <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)ratio + 1ull); }

// ---- batch folded into the sequence (MXVL_SCAN_FOLD_BATCH; scan_fwd_stream.h / scan_bwd.hip, FOLD) -------------------------------
// tv / seqlen by one multiply-high: exact while tv * seqlen < 2^32 (mxvl_scan_fold_ok checks batch * seqlen^2)
inline uint32_t scan_fold_magic(int seqlen) { return (uint32_t)((1ull << 32) / (uint64_t)seqlen + 1); }
// element offset of (segment sb, step sl) for an array with batch stride bs: ONE v_mad_u64_u32 -- the launchers only fold when every
// batch stride fits 32 bits (a 64 x 32-bit multiply per array and chunk was a tenth of the folded forward)
__device__ inline int64_t seg_off(int sb, int64_t bs, int sl) { return (int64_t)((uint64_t)(uint32_t)sb * (uint32_t)bs + (uint32_t)sl); }
// batch elements per workgroup sequence, the SAME function for the forward (which lays the checkpoints out by it) and the
// backward.  The backward runs ONE 32-row workgroup per CU, so the number of parts is chosen to make (32-row tiles x parts) fill
// whole rounds of the 256 CUs (128 tiles x 3 parts = 1.5 rounds cost the 197-token encoder a quarter of the kernel), with at
// least 512 sixteen-row workgroups for the forward where possible, sequences of >= 1024 steps; ties go to fewer parts.
inline int scan_fold_bpp(int batch, int seqlen, int dim, int n_groups) {
  const int dpg = dim / n_groups;
  const long tiles32 = (long)n_groups * ((dpg + 31) / 32), tiles16 = (long)n_groups * ((dpg + 15) / 16);
  int best_parts = 1;
  double best = -1.0;
  for (int parts = 1; parts <= batch && parts <= 32; ++parts) {
    const int bpp = (batch + parts - 1) / parts;
    const int real_parts = (batch + bpp - 1) / bpp;
    if (real_parts != parts) continue;
    const long wgs = tiles32 * parts, rounds = (wgs + 255) / 256;
    double score = (double)wgs / (double)(rounds * 256);          // fill of the backward's rounds
    if (tiles16 * parts < 512) score *= 0.5 + 0.5 * (double)(tiles16 * parts) / 512.0;   // the forward wants >= 2 workgroups per CU
    if ((long)bpp * seqlen < 1024 && parts > 1) continue;         // a sequence shorter than 8 chunks folds little padding away
    if (score > best + 1e-9) { best = score; best_parts = parts; }
  }
  return (batch + best_parts - 1) / best_parts;
}

}  // namespace mxvl
