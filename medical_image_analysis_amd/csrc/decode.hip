// decode.hip -- single-token decoder step of the report generator (Llama-2 / Qwen2 layer) for gfx950.
//
// Replaces what the reference leaves to HF transformers + cuBLAS per generated token
// (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:292-301 -> LlamaForCausalLM.generate; layer arithmetic
// restated in EMRRG/models/hybrid_decoder_layer.py:185-199 RMSNorm, :266-322 RoPE, :326-337 MLP, :392-457 attention).
// At batch*beams <= 8 rows every projection is a GEMV that streams its bf16 weight matrix exactly once: the step is
// HBM-bound (13.5 GB per token for Llama-2-7B -> 1.7 ms at 8 TB/s), and in eager PyTorch it is launch-bound (~1400
// small kernels per token).  Two kernels cover a layer:
//   gemv_bf16_kernel     y[m][n] = epi( sum_k W[n][k] * xhat[m][k] ), xhat = x or RMSNorm(x)*g (fused prologue);
//                        epilogues: +bias, +residual, SwiGLU over two weight matrices, fp32 output (logits).
//                        One persistent workgroup per CU (16 waves): the normalised activations are staged once in
//                        LDS (bf16), every wave streams whole weight rows with 16-byte loads (8 in flight per lane),
//                        products on v_dot2_f32_bf16, one DPP/shuffle reduction per row.
//   decode_attn_kernel   RoPE on q/k, append k/v to the cache, one-query attention over the cached positions with a
//                        per-beam slot table (beam re-ordering moves 4-byte slot ids, never the cache itself).
#include <algorithm>

#include "mxvl_common.h"

namespace mxvl {

constexpr int kMaxRows = 8;

struct GemvArgs {
  int rows, K, N, swiglu, out_f32;
  float eps;
  const uint16_t *x, *g, *W, *W2, *bias, *res;
  void* y;
};

__device__ inline float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ inline uint16_t f2bf(float x) {
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
typedef short bf16x2_t __attribute__((ext_vector_type(2)));
__device__ inline float dot2(uint32_t a, uint32_t b, float c) {
#if __has_builtin(__builtin_amdgcn_fdot2_f32_bf16)
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
#else
  c = fmaf(__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, b << 16), c);
  return fmaf(__builtin_bit_cast(float, a & 0xffff0000u), __builtin_bit_cast(float, b & 0xffff0000u), c);
#endif
}
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int M>
__global__ __launch_bounds__(1024) void gemv_bf16_kernel(const GemvArgs p) {
  constexpr int NW = 16;  // waves per workgroup (one workgroup per CU)
  extern __shared__ __attribute__((aligned(16))) uint16_t sx[];  // [M][K] (normalised) activations, bf16
  __shared__ float s_rstd[kMaxRows];
  __shared__ float s_part[kMaxRows][NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K, N = p.N;

  if (p.g) {  // RMSNorm statistics in fp32, as Qwen2RMSNorm (hybrid_decoder_layer.py:193-198)
    for (int m = 0; m < M; ++m) {
      float s = 0.0f;
      for (int k = tid; k < K; k += 1024) { const float v = bf2f(p.x[(size_t)m * K + k]); s = fmaf(v, v, s); }
      s = wave_sum(s);
      if (lane == 0) s_part[m][wave] = s;
    }
    __syncthreads();
    if (tid < M) {
      float s = 0.0f;
      for (int w = 0; w < NW; ++w) s += s_part[tid][w];
      s_rstd[tid] = rsqrtf(s / (float)K + p.eps);
    }
    __syncthreads();
  }
  for (int i = tid; i < M * (K / 2); i += 1024) {  // stage x (or bf16(bf16(x*rstd) * g)) once per workgroup
    const int m = i / (K / 2), kk = (i - m * (K / 2)) * 2;
    const uint32_t raw = *(const uint32_t*)(p.x + (size_t)m * K + kk);
    uint32_t packed = raw;
    if (p.g) {
      const uint32_t gg = *(const uint32_t*)(p.g + kk);
      const float a = bf2f(f2bf(bf2f((uint16_t)raw) * s_rstd[m])) * bf2f((uint16_t)gg);
      const float b = bf2f(f2bf(bf2f((uint16_t)(raw >> 16)) * s_rstd[m])) * bf2f((uint16_t)(gg >> 16));
      packed = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
    }
    *(uint32_t*)(sx + (size_t)m * K + kk) = packed;
  }
  __syncthreads();

  const int per_wg = (N + gridDim.x - 1) / gridDim.x;
  const int n_lo = blockIdx.x * per_wg, n_hi = min(N, n_lo + per_wg);
  for (int n0 = n_lo + wave * 2; n0 < n_hi; n0 += NW * 2) {   // two weight rows in flight per wave
    float acc[2][M], acc2[2][M];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) { acc[r][m] = 0.0f; acc2[r][m] = 0.0f; }
    const int n1 = (n0 + 1 < n_hi) ? n0 + 1 : n0;
    const uint16_t* w0 = p.W + (size_t)n0 * K;
    const uint16_t* w1 = p.W + (size_t)n1 * K;
    const uint16_t* v0 = p.swiglu ? p.W2 + (size_t)n0 * K : nullptr;
    const uint16_t* v1 = p.swiglu ? p.W2 + (size_t)n1 * K : nullptr;
#pragma unroll 4
    for (int kk = lane * 8; kk < K; kk += 512) {
      const uint4 a0 = *(const uint4*)(w0 + kk);
      const uint4 a1 = *(const uint4*)(w1 + kk);
      uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0;
      if (p.swiglu) { b0 = *(const uint4*)(v0 + kk); b1 = *(const uint4*)(v1 + kk); }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const uint4 xv = *(const uint4*)(sx + (size_t)m * K + kk);
        acc[0][m] = dot2(a0.x, xv.x, dot2(a0.y, xv.y, dot2(a0.z, xv.z, dot2(a0.w, xv.w, acc[0][m]))));
        acc[1][m] = dot2(a1.x, xv.x, dot2(a1.y, xv.y, dot2(a1.z, xv.z, dot2(a1.w, xv.w, acc[1][m]))));
        if (p.swiglu) {
          acc2[0][m] = dot2(b0.x, xv.x, dot2(b0.y, xv.y, dot2(b0.z, xv.z, dot2(b0.w, xv.w, acc2[0][m]))));
          acc2[1][m] = dot2(b1.x, xv.x, dot2(b1.y, xv.y, dot2(b1.z, xv.z, dot2(b1.w, xv.w, acc2[1][m]))));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int n = n0 + r;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float v = wave_sum(acc[r][m]);
        const float v2 = p.swiglu ? wave_sum(acc2[r][m]) : 0.0f;
        if (lane == 0 && n < n_hi) {
          const size_t o = (size_t)m * N + n;
          if (p.swiglu) {  // bf16(bf16(silu(gate)) * up), gate/up rounded to bf16 first (what the torch modules do)
            const float gte = bf2f(f2bf(v)), up = bf2f(f2bf(v2));
            ((uint16_t*)p.y)[o] = f2bf(bf2f(f2bf(gte * sigmoid(gte))) * up);
          } else {
            if (p.bias) v += bf2f(p.bias[n]);
            if (p.res) v = bf2f(f2bf(v)) + bf2f(p.res[o]);  // the linear output rounds to bf16 before the residual add
            if (p.out_f32) ((float*)p.y)[o] = v; else ((uint16_t*)p.y)[o] = f2bf(v);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct AttnArgs {
  int rows, H, Hkv, D, max_len;
  float scale;
  const uint16_t* qkv;       // (rows, (H + 2*Hkv) * D)
  const float *cosv, *sinv;  // (rows, D)
  uint16_t *kc, *vc;         // (rows, Hkv, max_len, D)
  const int* slot;           // (rows, max_len)
  const int64_t* pos;        // device scalar
  const int64_t* mask;       // (rows, max_len), nonzero = attend
  uint16_t* out;             // (rows, H * D)
};

__global__ __launch_bounds__(256) void decode_attn_kernel(const AttnArgs p) {
  extern __shared__ float sm[];
  const int D = p.D, T = p.max_len;
  float* sq = sm;            // [D] rotated query
  float* sk = sq + D;        // [D] rotated new key
  float* sv = sk + D;        // [D] new value
  float* sc = sv + D;        // [T] scores / probabilities
  float* red = sc + T;       // [8] reductions + [2][D] partial outputs
  float* so = red + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, m = blockIdx.y;
  const int group = p.H / p.Hkv, hk = h / group;
  const int pos = (int)*p.pos;
  const size_t row = (size_t)m * (p.H + 2 * p.Hkv) * D;
  const uint16_t* q = p.qkv + row + (size_t)h * D;
  const uint16_t* kn = p.qkv + row + (size_t)(p.H + hk) * D;
  const uint16_t* vn = p.qkv + row + (size_t)(p.H + p.Hkv + hk) * D;
  // RoPE (hybrid_decoder_layer.py:284-322): x*cos + rotate_half(x)*sin, computed in the activation dtype (bf16)
  if (tid < D) {
    const int d = tid, half = D / 2;
    const float c = bf2f(f2bf(p.cosv[(size_t)m * D + d])), s = bf2f(f2bf(p.sinv[(size_t)m * D + d]));
    const float qd = bf2f(q[d]), qo = d < half ? -bf2f(q[d + half]) : bf2f(q[d - half]);
    const float kd = bf2f(kn[d]), ko = d < half ? -bf2f(kn[d + half]) : bf2f(kn[d - half]);
    const float qr = bf2f(f2bf(bf2f(f2bf(qd * c)) + bf2f(f2bf(qo * s))));
    const float kr = bf2f(f2bf(bf2f(f2bf(kd * c)) + bf2f(f2bf(ko * s))));
    sq[d] = qr;
    sk[d] = kr;
    sv[d] = bf2f(vn[d]);
    if (h % group == 0) {  // one head of the group appends to the cache (slot m owns position pos of beam m)
      const size_t o = (((size_t)m * p.Hkv + hk) * T + pos) * D + d;
      p.kc[o] = f2bf(kr);
      p.vc[o] = vn[d];
    }
  }
  __syncthreads();
  // scores over the allowed cached positions (t <= pos); position pos uses the fresh key
  float lmax = -INFINITY;
  for (int t = tid; t <= pos; t += 256) {
    float s = -INFINITY;
    if (p.mask[(size_t)m * T + t] != 0) {
      float acc = 0.0f;
      if (t == pos) {
        for (int d = 0; d < D; ++d) acc = fmaf(sq[d], sk[d], acc);
      } else {
        const int sl = p.slot[(size_t)m * T + t];
        const uint4* kr = (const uint4*)(p.kc + (((size_t)sl * p.Hkv + hk) * T + t) * D);
        for (int d8 = 0; d8 < D / 8; ++d8) {
          const uint4 kv = kr[d8];
          const uint32_t w[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc = fmaf(sq[d8 * 8 + 2 * j], __builtin_bit_cast(float, w[j] << 16), acc);
            acc = fmaf(sq[d8 * 8 + 2 * j + 1], __builtin_bit_cast(float, w[j] & 0xffff0000u), acc);
          }
        }
      }
      s = acc * p.scale;
    }
    sc[t] = s;
    lmax = fmaxf(lmax, s);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
  if (lane == 0) red[wave] = lmax;
  __syncthreads();
  const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float lsum = 0.0f;
  for (int t = tid; t <= pos; t += 256) {
    const float e = (sc[t] == -INFINITY) ? 0.0f : fast_exp(sc[t] - gmax);
    sc[t] = e;
    lsum += e;
  }
  lsum = wave_sum(lsum);
  __syncthreads();
  if (lane == 0) red[4 + wave] = lsum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  // out[d] = sum_t p_t V[t][d]: thread = (d, half of the positions)
  const int d = tid % D, part = tid / D, parts = 256 / D;
  float o = 0.0f;
  if (part < parts) {
    for (int t = part; t <= pos; t += parts) {
      const float pt = sc[t];
      if (pt != 0.0f) {
        float v;
        if (t == pos) v = sv[d];
        else {
          const int sl = p.slot[(size_t)m * T + t];
          v = bf2f(p.vc[(((size_t)sl * p.Hkv + hk) * T + t) * D + d]);
        }
        o = fmaf(pt, v, o);
      }
    }
    so[part * D + d] = o;
  }
  __syncthreads();
  if (tid < D) {
    float tot = 0.0f;
    for (int q2 = 0; q2 < parts; ++q2) tot += so[q2 * D + tid];
    p.out[(size_t)m * p.H * D + (size_t)h * D + tid] = f2bf(tot * inv);
  }
}

static int dec_check() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

template <int M>
static int launch_gemv(const GemvArgs& a, hipStream_t s) {
  const int grid = std::max(1, std::min(256, (a.N + 31) / 32));
  const size_t lds = (size_t)M * a.K * sizeof(uint16_t);
  if (lds > 150 * 1024) return MXVL_ERR_UNSUPPORTED;  // rows * K bf16 must fit one CU's LDS
  if (lds > 64 * 1024) {
    static bool raised = false;  // per template instance
    if (!raised) {
      if (hipFuncSetAttribute((const void*)gemv_bf16_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
        return MXVL_ERR_LAUNCH;
      raised = true;
    }
  }
  hipLaunchKernelGGL(gemv_bf16_kernel<M>, dim3(grid), dim3(1024), lds, s, a);
  return MXVL_OK;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" {

int mxvl_decode_gemv(const mxvl_gemv_desc* d, void* hip_stream) {
  if (!d || !d->x || !d->W || !d->y) return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->rows > kMaxRows || d->K <= 0 || d->N <= 0) return MXVL_ERR_SHAPE;
  if (d->K % 8 != 0) return MXVL_ERR_UNSUPPORTED;  // 16-byte weight loads
  if (d->swiglu && (!d->W2 || d->out_f32)) return MXVL_ERR_UNSUPPORTED;
  GemvArgs a;
  a.rows = d->rows; a.K = d->K; a.N = d->N; a.swiglu = d->swiglu; a.out_f32 = d->out_f32; a.eps = d->eps;
  a.x = (const uint16_t*)d->x; a.g = (const uint16_t*)d->norm_weight; a.W = (const uint16_t*)d->W;
  a.W2 = (const uint16_t*)d->W2; a.bias = (const uint16_t*)d->bias; a.res = (const uint16_t*)d->residual; a.y = d->y;
  hipStream_t s = (hipStream_t)hip_stream;
  int rc;
  switch (d->rows) {
    case 1: rc = launch_gemv<1>(a, s); break;
    case 2: rc = launch_gemv<2>(a, s); break;
    case 3: rc = launch_gemv<3>(a, s); break;
    case 4: rc = launch_gemv<4>(a, s); break;
    case 5: rc = launch_gemv<5>(a, s); break;
    case 6: rc = launch_gemv<6>(a, s); break;
    case 7: rc = launch_gemv<7>(a, s); break;
    default: rc = launch_gemv<8>(a, s); break;
  }
  return rc != MXVL_OK ? rc : dec_check();
}

int mxvl_decode_attn(const mxvl_decode_attn_desc* d, void* hip_stream) {
  if (!d || !d->qkv || !d->cos || !d->sin || !d->k_cache || !d->v_cache || !d->slot_table || !d->pos || !d->mask || !d->out)
    return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->n_heads <= 0 || d->n_kv_heads <= 0 || d->n_heads % d->n_kv_heads != 0) return MXVL_ERR_SHAPE;
  if (d->head_dim % 8 != 0 || d->head_dim > 256 || 256 % d->head_dim != 0) return MXVL_ERR_UNSUPPORTED;
  AttnArgs a;
  a.rows = d->rows; a.H = d->n_heads; a.Hkv = d->n_kv_heads; a.D = d->head_dim; a.max_len = d->max_len;
  a.scale = d->scale; a.qkv = (const uint16_t*)d->qkv; a.cosv = (const float*)d->cos; a.sinv = (const float*)d->sin;
  a.kc = (uint16_t*)d->k_cache; a.vc = (uint16_t*)d->v_cache; a.slot = (const int*)d->slot_table;
  a.pos = (const int64_t*)d->pos; a.mask = (const int64_t*)d->mask; a.out = (uint16_t*)d->out;
  const size_t lds = sizeof(float) * ((size_t)3 * a.D + a.max_len + 8 + (256 / a.D) * a.D);
  if (lds > 64 * 1024) return MXVL_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(decode_attn_kernel, dim3(a.H, a.rows), dim3(256), lds, (hipStream_t)hip_stream, a);
  return dec_check();
}

}  // extern "C"
