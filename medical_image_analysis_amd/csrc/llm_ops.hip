// llm_ops.hip -- the element-wise work of a (frozen) Llama / Qwen2 decoder layer in the TRAINING step of the report-generation
// stages, gfx950: rotary position embedding of q and k (forward + adjoint) and RMSNorm (forward + input gradient).
//
// The stage-3 / R2GenCSR step runs an fp16-loaded 7B decoder under bf16 autocast, forward and activation-gradient backward
// (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:195-241, R2GenCSR/models/R2GenCSR.py:309-474).  As torch expressions
// (EMRRG/models/hybrid_decoder_layer.py:185-199 Qwen2RMSNorm, :290-323 rotate_half / apply_rotary_pos_emb; HF's Llama classes are the
// same text) a layer's RoPE is 12 element-wise launches forward and ~18 backward over (B, H, T, D) -- slices, negations, a cat, two
// multiplies by cos / sin that PROMOTE bf16 x fp16 to fp32, an add, the cast back, and in backward the zero-padded slice gradients --
// and an RMSNorm 5 + 8 with three fp32 round trips of (B, T, hidden): 46 ms + 24 ms of a 357 ms R2GenCSR step
// (profiles/r06_step_eager_r2gencsr.txt).  Both are one pass over their operands here, with the reference's rounding points:
//
//   RoPE     y1 = x1 c1 - x2 s1,  y2 = x2 c2 + x1 s2   (x1 | x2 the halves of a head, c / s the cos / sin rows of the token)
//            every product and the sum rounded to R = promote(dtype(x), dtype(cos)) (fp32 unless the two are the same 16-bit type),
//            the result to dtype(x) -- bit-identical to apply_rotary_pos_emb followed by `.to(v.dtype)`;
//   adjoint  dx1 = io(io(R(g1 c1)) + io(R(g2 s2))),  dx2 = io(io(R(g2 c2)) - io(R(g1 s1))): autograd's own sequence (the gradient of a
//            promoted product is cast to the operand's dtype, the slice gradients are added in that dtype);
//   RMSNorm  y = out(P(w * dt(x * rsqrt(mean(x^2) + eps)))),  statistics in fp32, dt = dtype(x), P = promote(dtype(w), dt), out = the
//            dtype the consumer reads (the autocast dtype under autocast: the nn.Linear behind the norm casts to it anyway);
//   dx       = dt(rstd * (gh - xhat * mean(gh * xhat))),  gh = dt(P(g * w)),  xhat = x * rstd in fp32.  No weight gradient: callers with a
//            trainable norm weight keep the torch expression (Qwen2RMSNorm.forward decides).
#include "mxvl_common.h"
#include <type_traits>

// Every product and sum below is its own rounding step of the torch expression it restates: this file is compiled with
// -ffp-contract=off (build.py FILE_FLAGS; the pragma alone does not stop the backend's fusion under -ffp-contract=fast -- it produced
// v_fma_f16 / v_fma_mixlo_f16 here); fmaf where a fused step is meant.
#pragma clang fp contract(off)

namespace mxvl {

template <typename T>
__device__ __forceinline__ float rnd_to(float v) {          // the value after a store in T
  if constexpr (sizeof(T) == 4) return v;
  // v is an fp32 RESULT (torch's promoted product) that is then cast: keep the two roundings apart -- left to itself the compiler
  // folds fptrunc(fmul) into v_fma_mixlo_f16, ONE rounding of the exact product (5 of 75 776 gradients differed by an fp16 ulp)
  asm volatile("" : "+v"(v));
  T t;
  Io<T>::st(&t, v);
  return Io<T>::ld(&t);
}

struct RopeArgs {
  int B, T, Hq, Hk, D, backward;
  int64_t q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, qo_bs, qo_ts, qo_hs, ko_bs, ko_ts, ko_hs, cs_bs, cs_ts;
  const void *q, *k, *cos, *sin;
  void *qo, *ko;
};

// V consecutive cos / sin values as vector loads (8, 16 or 2 x 16 bytes; the launcher checks the alignment and passes CSV = false otherwise)
template <typename cs_t, int V, bool CSV>
__device__ __forceinline__ void rope_ld_cs(const cs_t* q, float (&v)[V]) {
  if constexpr (!CSV) {
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = Io<cs_t>::ld(q + e);
  } else {
    cs_t t[V];
    constexpr int BYTES = V * (int)sizeof(cs_t);
    if constexpr (BYTES == 8) *(uint2*)t = *(const uint2*)q;
    else if constexpr (BYTES == 16) *(uint4*)t = *(const uint4*)q;
    else { *(uint4*)t = *(const uint4*)q; *((uint4*)t + 1) = *((const uint4*)q + 1); }
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = Io<cs_t>::ld(t + e);
  }
}

// one thread: V elements of the first half of a head and the V matching elements of the second half
template <typename io_t, typename cs_t, bool CSV>
__global__ __launch_bounds__(256) void rope_kernel(const RopeArgs p) {
  constexpr int V = 16 / (int)sizeof(io_t);                 // 8 (16-bit) / 4 (fp32) elements = 16 bytes
  constexpr bool R16 = std::is_same<io_t, cs_t>::value && sizeof(io_t) == 2;   // no promotion: intermediates live in the io dtype
  const int half = p.D / 2, vph = half / V, H = p.Hq + p.Hk;
  const int64_t total = (int64_t)p.B * p.T * H * vph;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int j = (int)(idx % vph);
    int64_t r = idx / vph;
    const int h = (int)(r % H);
    r /= H;
    const int t = (int)(r % p.T), b = (int)(r / p.T);
    const bool isq = h < p.Hq;
    const int hh = isq ? h : h - p.Hq;
    const io_t* x = isq ? (const io_t*)p.q + b * p.q_bs + t * p.q_ts + hh * p.q_hs : (const io_t*)p.k + b * p.k_bs + t * p.k_ts + hh * p.k_hs;
    io_t* y = isq ? (io_t*)p.qo + b * p.qo_bs + t * p.qo_ts + hh * p.qo_hs : (io_t*)p.ko + b * p.ko_bs + t * p.ko_ts + hh * p.ko_hs;
    const cs_t* c = (const cs_t*)p.cos + b * p.cs_bs + t * p.cs_ts;
    const cs_t* s = (const cs_t*)p.sin + b * p.cs_bs + t * p.cs_ts;
    const int i0 = j * V;
    io_t x1[V], x2[V], o1[V], o2[V];
    *(uint4*)x1 = *(const uint4*)(x + i0);
    *(uint4*)x2 = *(const uint4*)(x + half + i0);
    auto rr = [](float v) { return R16 ? rnd_to<io_t>(v) : v; };
    float cv1[V], cv2[V], sv1[V], sv2[V];
    rope_ld_cs<cs_t, V, CSV>(c + i0, cv1);
    rope_ld_cs<cs_t, V, CSV>(c + half + i0, cv2);
    rope_ld_cs<cs_t, V, CSV>(s + i0, sv1);
    rope_ld_cs<cs_t, V, CSV>(s + half + i0, sv2);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float a1 = Io<io_t>::ld(x1 + e), a2 = Io<io_t>::ld(x2 + e);
      const float c1 = cv1[e], c2 = cv2[e], s1 = sv1[e], s2 = sv2[e];
      if (!p.backward) {
        Io<io_t>::st(o1 + e, rr(rr(a1 * c1) + rr(-a2 * s1)));
        Io<io_t>::st(o2 + e, rr(rr(a2 * c2) + rr(a1 * s2)));
      } else {
        const float gA1 = rnd_to<io_t>(rr(a1 * c1)), gA2 = rnd_to<io_t>(rr(a2 * c2));
        const float gB1 = rnd_to<io_t>(rr(a1 * s1)), gB2 = rnd_to<io_t>(rr(a2 * s2));
        Io<io_t>::st(o1 + e, gA1 + gB2);
        Io<io_t>::st(o2 + e, gA2 - gB1);
      }
    }
    *(uint4*)(y + i0) = *(const uint4*)o1;
    *(uint4*)(y + half + i0) = *(const uint4*)o2;
  }
}

struct RmsArgs {
  int rows, cols, x_dtype, w_dtype, y_dtype;
  float eps;
  const void *x, *w, *g;
  void *y;          // forward: y; backward: dx
  float* rstd;
};

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]) {
  if constexpr (sizeof(T) == 4) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    T t[8];
    *(uint4*)t = *(const uint4*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = Io<T>::ld(t + i);
  }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]) {
  if constexpr (sizeof(T) == 4) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    T t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) Io<T>::st(t + i, v[i]);
    *(uint4*)p = *(const uint4*)t;
  }
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// a wave per row, 8 elements per lane and pass; the row is read twice (the second time from L1 / L2)
template <typename x_t, typename w_t, typename y_t>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const RmsArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int C = p.cols;
  const x_t* x = (const x_t*)p.x + (int64_t)row * C;
  float ss = 0.0f;
  for (int i = lane * 8; i < C; i += 512) {
    float v[8];
    ld8<x_t>(x + i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)C + p.eps);
  if (lane == 0 && p.rstd) p.rstd[row] = rstd;
  constexpr bool PF32 = !(std::is_same<x_t, w_t>::value && sizeof(x_t) == 2);      // promote(w, x) is fp32 unless both are the same 16-bit type
  y_t* y = (y_t*)p.y + (int64_t)row * C;
  const w_t* w = (const w_t*)p.w;
  for (int i = lane * 8; i < C; i += 512) {
    float v[8], g[8], o[8];
    ld8<x_t>(x + i, v);
    ld8<w_t>(w + i, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float h = rnd_to<x_t>(v[e] * rstd);
      const float m = g[e] * h;
      o[e] = PF32 ? m : rnd_to<x_t>(m);
    }
    st8<y_t>(y + i, o);
  }
}

template <typename x_t, typename w_t, typename y_t>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const RmsArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int C = p.cols;
  constexpr bool PF32 = !(std::is_same<x_t, w_t>::value && sizeof(x_t) == 2);
  const x_t* x = (const x_t*)p.x + (int64_t)row * C;
  const y_t* g = (const y_t*)p.g + (int64_t)row * C;
  const w_t* w = (const w_t*)p.w;
  const float rstd = p.rstd[row];
  float dot = 0.0f;
  for (int i = lane * 8; i < C; i += 512) {
    float v[8], gv[8], wv[8];
    ld8<x_t>(x + i, v);
    ld8<y_t>(g + i, gv);
    ld8<w_t>(w + i, wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float m = gv[e] * wv[e];
      const float gh = rnd_to<x_t>(PF32 ? m : rnd_to<x_t>(m));
      dot = fmaf(gh, v[e] * rstd, dot);
    }
  }
  dot = wave_sum(dot) / (float)C;
  x_t* dx = (x_t*)p.y + (int64_t)row * C;
  for (int i = lane * 8; i < C; i += 512) {
    float v[8], gv[8], wv[8], o[8];
    ld8<x_t>(x + i, v);
    ld8<y_t>(g + i, gv);
    ld8<w_t>(w + i, wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float m = gv[e] * wv[e];
      const float gh = rnd_to<x_t>(PF32 ? m : rnd_to<x_t>(m));
      o[e] = rstd * (gh - v[e] * rstd * dot);
    }
    st8<x_t>(dx + i, o);
  }
}

// silu(a) * b of the gated MLP (Qwen2MLP / LlamaMLP: act_fn(gate_proj(x)) * up_proj(x)) with the two roundings of the two torch kernels;
// backward: autograd's sequence -- db = io(dy * s), ds = io(dy * b), da = io(ds * sig (1 + a (1 - sig))) -- from a, b, dy alone (s = io(silu(a))
// is recomputed: the same arithmetic, the same bits)
template <typename io_t>
__global__ __launch_bounds__(256) void silu_mul_kernel(const io_t* __restrict__ a, const io_t* __restrict__ b, const io_t* __restrict__ dy,
                                                        io_t* __restrict__ y, io_t* __restrict__ db, int64_t n8) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n8; v += (int64_t)gridDim.x * 256) {
    float av[8], bv[8], o[8];
    ld8<io_t>(a + v * 8, av);
    ld8<io_t>(b + v * 8, bv);
    if (dy == nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float s = rnd_to<io_t>(av[e] / (1.0f + expf(-av[e])));
        o[e] = s * bv[e];
      }
      st8<io_t>(y + v * 8, o);
    } else {
      float gv[8], ob[8];
      ld8<io_t>(dy + v * 8, gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sig = 1.0f / (1.0f + expf(-av[e]));
        const float s = rnd_to<io_t>(av[e] / (1.0f + expf(-av[e])));
        ob[e] = gv[e] * s;
        const float ds = rnd_to<io_t>(gv[e] * bv[e]);
        o[e] = ds * sig * (1.0f + av[e] * (1.0f - sig));
      }
      st8<io_t>(y + v * 8, o);         // da
      st8<io_t>(db + v * 8, ob);
    }
  }
}

template <typename io_t>
static int rope_launch_cs(const RopeArgs& a, int cs_dtype, hipStream_t s) {
  const int V = 16 / (int)sizeof(io_t);
  const int64_t total = (int64_t)a.B * a.T * (a.Hq + a.Hk) * (a.D / 2 / V);
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 * 16 ? (total + 255) / 256 : 65536 * 16);
  // cos / sin rows as vectors when every (batch, token) row and both half-head offsets are aligned to the vector
  const int csz = cs_dtype == MXVL_F32 ? 4 : 2, bytes = V * csz, al = bytes > 16 ? 16 : bytes;
  const bool csv = ((uintptr_t)a.cos) % al == 0 && ((uintptr_t)a.sin) % al == 0 && (a.cs_bs * csz) % al == 0 && (a.cs_ts * csz) % al == 0 &&
                   ((int64_t)(a.D / 2) * csz) % al == 0;
#define MXVL_ROPE_GO(CS)                                                                              \
  do {                                                                                                \
    if (csv) hipLaunchKernelGGL((rope_kernel<io_t, CS, true>), dim3(grid), dim3(256), 0, s, a);       \
    else hipLaunchKernelGGL((rope_kernel<io_t, CS, false>), dim3(grid), dim3(256), 0, s, a);          \
  } while (0)
  switch (cs_dtype) {
    case MXVL_F32: MXVL_ROPE_GO(float); break;
    case MXVL_BF16: MXVL_ROPE_GO(bf16_t); break;
    default: MXVL_ROPE_GO(f16_t); break;
  }
#undef MXVL_ROPE_GO
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

template <typename x_t, typename w_t>
static int rms_launch_y(const RmsArgs& a, bool bwd, hipStream_t s) {
  const dim3 grid((a.rows + 3) / 4), block(256);
#define MXVL_RMS_GO(Y)                                                                     \
  do {                                                                                     \
    if (bwd) hipLaunchKernelGGL((rmsnorm_bwd_kernel<x_t, w_t, Y>), grid, block, 0, s, a);  \
    else hipLaunchKernelGGL((rmsnorm_fwd_kernel<x_t, w_t, Y>), grid, block, 0, s, a);      \
  } while (0)
  switch (a.y_dtype) {
    case MXVL_F32: MXVL_RMS_GO(float); break;
    case MXVL_BF16: MXVL_RMS_GO(bf16_t); break;
    default: MXVL_RMS_GO(f16_t); break;
  }
#undef MXVL_RMS_GO
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
template <typename x_t>
static int rms_launch_w(const RmsArgs& a, bool bwd, hipStream_t s) {
  switch (a.w_dtype) {
    case MXVL_F32: return rms_launch_y<x_t, float>(a, bwd, s);
    case MXVL_BF16: return rms_launch_y<x_t, bf16_t>(a, bwd, s);
    default: return rms_launch_y<x_t, f16_t>(a, bwd, s);
  }
}

static bool dtype_ok(int d) { return d == MXVL_F32 || d == MXVL_BF16 || d == MXVL_F16; }

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_rope(const mxvl_rope_desc* d, void* hip_stream) {
  if (!d || !d->q || !d->k || !d->cos || !d->sin || !d->q_out || !d->k_out) return MXVL_ERR_NULL;
  if (d->batch <= 0 || d->seqlen <= 0 || d->n_q_heads <= 0 || d->n_k_heads < 0 || d->head_dim <= 0) return MXVL_ERR_SHAPE;
  if (!dtype_ok(d->io_dtype) || !dtype_ok(d->cs_dtype)) return MXVL_ERR_DTYPE;
  const int esz = d->io_dtype == MXVL_F32 ? 4 : 2, V = 16 / esz;
  if (d->head_dim % (2 * V) != 0) return MXVL_ERR_UNSUPPORTED;          // whole 16-byte vectors in each half of a head
  const int64_t st[] = {d->q_bs, d->q_ts, d->q_hs, d->k_bs, d->k_ts, d->k_hs, d->qo_bs, d->qo_ts, d->qo_hs, d->ko_bs, d->ko_ts, d->ko_hs};
  for (int64_t s : st)
    if (s % V != 0) return MXVL_ERR_UNSUPPORTED;
  for (const void* q : {d->q, d->k, (const void*)d->q_out, (const void*)d->k_out})
    if (((uintptr_t)q) % 16 != 0) return MXVL_ERR_UNSUPPORTED;
  RopeArgs a;
  a.B = d->batch; a.T = d->seqlen; a.Hq = d->n_q_heads; a.Hk = d->n_k_heads; a.D = d->head_dim; a.backward = d->backward ? 1 : 0;
  a.q_bs = d->q_bs; a.q_ts = d->q_ts; a.q_hs = d->q_hs; a.k_bs = d->k_bs; a.k_ts = d->k_ts; a.k_hs = d->k_hs;
  a.qo_bs = d->qo_bs; a.qo_ts = d->qo_ts; a.qo_hs = d->qo_hs; a.ko_bs = d->ko_bs; a.ko_ts = d->ko_ts; a.ko_hs = d->ko_hs;
  a.cs_bs = d->cs_bs; a.cs_ts = d->cs_ts;
  a.q = d->q; a.k = d->k; a.cos = d->cos; a.sin = d->sin; a.qo = d->q_out; a.ko = d->k_out;
  hipStream_t s = (hipStream_t)hip_stream;
  switch (d->io_dtype) {
    case MXVL_F32: return rope_launch_cs<float>(a, d->cs_dtype, s);
    case MXVL_BF16: return rope_launch_cs<bf16_t>(a, d->cs_dtype, s);
    default: return rope_launch_cs<f16_t>(a, d->cs_dtype, s);
  }
}

static int rms_common(const mxvl_rms_train_desc* d, bool bwd, void* hip_stream) {
  if (!d || !d->x || !d->weight || !d->y || (bwd && (!d->grad || !d->rstd))) return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->cols <= 0) return MXVL_ERR_SHAPE;
  if (!dtype_ok(d->x_dtype) || !dtype_ok(d->w_dtype) || !dtype_ok(d->y_dtype)) return MXVL_ERR_DTYPE;
  if (d->cols % 8 != 0) return MXVL_ERR_UNSUPPORTED;
  for (const void* q : {d->x, d->weight, (const void*)d->y, d->grad})
    if (q && ((uintptr_t)q) % 16 != 0) return MXVL_ERR_UNSUPPORTED;
  RmsArgs a;
  a.rows = d->rows; a.cols = d->cols; a.x_dtype = d->x_dtype; a.w_dtype = d->w_dtype; a.y_dtype = d->y_dtype; a.eps = d->eps;
  a.x = d->x; a.w = d->weight; a.g = d->grad; a.y = d->y; a.rstd = (float*)d->rstd;
  hipStream_t s = (hipStream_t)hip_stream;
  switch (d->x_dtype) {
    case MXVL_F32: return rms_launch_w<float>(a, bwd, s);
    case MXVL_BF16: return rms_launch_w<bf16_t>(a, bwd, s);
    default: return rms_launch_w<f16_t>(a, bwd, s);
  }
}
extern "C" int mxvl_rmsnorm_train_fwd(const mxvl_rms_train_desc* d, void* hip_stream) { return rms_common(d, false, hip_stream); }
extern "C" int mxvl_rmsnorm_train_bwd(const mxvl_rms_train_desc* d, void* hip_stream) { return rms_common(d, true, hip_stream); }

extern "C" int mxvl_silu_mul(const void* a, const void* b, const void* dy, void* y, void* db, int64_t n, int io_dtype, void* hip_stream) {
  if (!a || !b || !y || (dy && !db)) return MXVL_ERR_NULL;
  if (n <= 0) return MXVL_ERR_SHAPE;
  if (!dtype_ok(io_dtype)) return MXVL_ERR_DTYPE;
  if (n % 8 != 0) return MXVL_ERR_UNSUPPORTED;
  for (const void* q : {a, b, dy, (const void*)y, (const void*)db})
    if (q && ((uintptr_t)q) % 16 != 0) return MXVL_ERR_UNSUPPORTED;
  const int64_t n8 = n / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 < 65536 * 8 ? (n8 + 255) / 256 : 65536 * 8);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (io_dtype) {
    case MXVL_F32: hipLaunchKernelGGL(silu_mul_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, (const float*)dy, (float*)y, (float*)db, n8); break;
    case MXVL_BF16: hipLaunchKernelGGL(silu_mul_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)dy, (bf16_t*)y, (bf16_t*)db, n8); break;
    default: hipLaunchKernelGGL(silu_mul_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)a, (const f16_t*)b, (const f16_t*)dy, (f16_t*)y, (f16_t*)db, n8); break;
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
