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
#include <cstdlib>

#include "decode_bodies.h"

namespace mxvl {


// Work unit = one output column n: weight row n of W (with SwiGLU: gate row n of W, then up row n of W2).  Units are
// dealt round-robin to the 16 x gridDim.x waves of the launch, a whole weight row per wave at a time, so every wave
// streams 2*K contiguous bytes with PF 16-byte loads in flight per lane.  The next row's first PF loads are issued
// before the current row's cross-lane reduction, and the very first row's before the RMSNorm prologue, so the HBM
// stream never waits for the prologue or an epilogue.
template <int M>
__global__ __launch_bounds__(1024) void gemv_bf16_kernel(const GemvArgs p) {
  constexpr int NW = 16;  // waves per workgroup (one workgroup per CU)
  constexpr int PF = 8;   // 16-byte loads in flight per lane
  extern __shared__ __attribute__((aligned(16))) uint16_t sx[];  // [M][K] (normalised) activations, bf16
  __shared__ float s_part[kMaxRows][NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K, N = p.N;
  if (MXVL_ABL(p.ablate == 1)) return;
  const int S = p.swiglu ? 2 : 1;
  const int TW = gridDim.x * NW;                      // waves in the launch
  const int n_first = blockIdx.x * NW + wave;
  const int n_items = n_first < N ? ((N - 1 - n_first) / TW + 1) * S : 0;   // weight rows this wave streams
  auto row_ptr = [&](int item) -> const uint16_t* {
    const int n = n_first + (p.swiglu ? item >> 1 : item) * TW;
    return ((p.swiglu && (item & 1)) ? p.W2 : p.W) + (size_t)n * K;
  };
  uint4 pre[PF];
  if (n_items > 0) {
    const uint16_t* w = row_ptr(0);
#pragma unroll
    for (int j = 0; j < PF; ++j) pre[j] = ldw(w, lane * 8 + j * 512, K);
  }

  // Stage x (or RMSNorm(x)*g, Qwen2RMSNorm hybrid_decoder_layer.py:193-198: bf16(bf16(x*rstd) * g)) in LDS.  One
  // 16-byte global load per thread per (row, 8192-column block), ALL issued before the first use: the prologue costs
  // one L2 round trip, not one per element.
  if (MXVL_ABL(p.ablate == 2)) return;
  if (!MXVL_ABL(p.ablate == 3))
  for (int c0 = 0; c0 < K; c0 += 8192) {
    const int kk = c0 + tid * 8;
    const bool on = kk < K;
    uint4 xr[M];
#pragma unroll
    for (int m = 0; m < M; ++m) xr[m] = on ? *(const uint4*)(p.x + (size_t)m * K + kk) : make_uint4(0, 0, 0, 0);
    if (!p.g) {
      if (on) {
#pragma unroll
        for (int m = 0; m < M; ++m) *(uint4*)(sx + (size_t)m * K + kk) = xr[m];
      }
    } else {
      // K <= 8192 on this path (checked by the launcher): the whole row is in the workgroup's registers
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const uint32_t w[4] = {xr[m].x, xr[m].y, xr[m].z, xr[m].w};
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf2f((uint16_t)w[j]), b = bf2f((uint16_t)(w[j] >> 16));
          s = fmaf(a, a, fmaf(b, b, s));
        }
        s = wave_sum(s);
        if (lane == 0) s_part[m][wave] = s;
      }
      __syncthreads();
      const uint4 gv = on ? *(const uint4*)(p.g + kk) : make_uint4(0, 0, 0, 0);
      const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float tot = 0.0f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) tot += s_part[m][w2];
        const float rstd = rsqrtf(tot / (float)K + p.eps);
        const uint32_t w[4] = {xr[m].x, xr[m].y, xr[m].z, xr[m].w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bf2f(f2bf(bf2f((uint16_t)w[j]) * rstd)) * bf2f((uint16_t)gw[j]);
          const float b = bf2f(f2bf(bf2f((uint16_t)(w[j] >> 16)) * rstd)) * bf2f((uint16_t)(gw[j] >> 16));
          o[j] = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
        }
        if (on) *(uint4*)(sx + (size_t)m * K + kk) = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  __syncthreads();

  if (MXVL_ABL(p.ablate == 4)) return;
  float gate[M];
#pragma unroll
  for (int m = 0; m < M; ++m) gate[m] = 0.0f;
  for (int item = 0; item < n_items; ++item) {
    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.0f;
    const uint16_t* w = row_ptr(item);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int kk = lane * 8 + j * 512;
      if (kk < K) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const uint4 xv = *(const uint4*)(sx + (size_t)m * K + kk);
          acc[m] = dot2(pre[j].x, xv.x, dot2(pre[j].y, xv.y, dot2(pre[j].z, xv.z, dot2(pre[j].w, xv.w, acc[m]))));
        }
      }
    }
#pragma unroll 4
    for (int kk = lane * 8 + PF * 512; kk < K; kk += 512) {
      const uint4 a0 = ldnt(w + kk);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const uint4 xv = *(const uint4*)(sx + (size_t)m * K + kk);
        acc[m] = dot2(a0.x, xv.x, dot2(a0.y, xv.y, dot2(a0.z, xv.z, dot2(a0.w, xv.w, acc[m]))));
      }
    }
    if (item + 1 < n_items) {  // next row's head goes in flight before this row's reduction
      const uint16_t* wn = row_ptr(item + 1);
#pragma unroll
      for (int j = 0; j < PF; ++j) pre[j] = ldw(wn, lane * 8 + j * 512, K);
    }
    const int n = n_first + (p.swiglu ? item >> 1 : item) * TW;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float v = wave_sum(acc[m]);
      const size_t o = (size_t)m * N + n;
      if (p.swiglu) {
        if (!(item & 1)) gate[m] = v;
        else if (lane == 0) {  // bf16(bf16(silu(gate)) * up), gate/up rounded to bf16 first (what the torch modules do)
          const float gte = bf2f(f2bf(gate[m])), up = bf2f(f2bf(v));
          ((uint16_t*)p.y)[o] = f2bf(bf2f(f2bf(gte * sigmoid(gte))) * up);
        }
      } else if (lane == 0) {
        if (p.bias) v += bf2f(p.bias[n]);
        if (p.res) v = bf2f(f2bf(v)) + bf2f(p.res[o]);  // the linear output rounds to bf16 before the residual add
        if (p.out_f32) ((float*)p.y)[o] = v; else ((uint16_t*)p.y)[o] = f2bf(v);
      }
    }
  }
}


template <int D>
__global__ __launch_bounds__(kAttnWaves * 64) void decode_attn_kernel(const AttnArgs p) {
  extern __shared__ float sm[];
  decode_attn_body<D, kAttnWaves>(p, blockIdx.x, blockIdx.y, sm, PlainIO());
}
template <int D>
__global__ __launch_bounds__(kAttnWaves * 64) void decode_cross_attn_kernel(const CrossAttnArgs p) {
  extern __shared__ float sm[];
  decode_cross_attn_body<D, kAttnWaves>(p, blockIdx.x, blockIdx.y, sm, PlainIO());
}

static int dec_check() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

template <int M>
static int launch_gemv(const GemvArgs& a, hipStream_t s) {
  const int grid = std::max(1, std::min(256, (a.N + 15) / 16));
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
  if (d->norm_weight && d->K > 8192) return MXVL_ERR_UNSUPPORTED;  // fused RMSNorm keeps a whole row in registers
  if (d->swiglu && (!d->W2 || d->out_f32)) return MXVL_ERR_UNSUPPORTED;
  GemvArgs a;
  a.ablate = MXVL_ABL_ENV("MXVL_GEMV_ABLATE");
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
  if (d->head_dim != 64 && d->head_dim != 128 && d->head_dim != 256) return MXVL_ERR_UNSUPPORTED;
  AttnArgs a;
  a.rows = d->rows; a.H = d->n_heads; a.Hkv = d->n_kv_heads; a.D = d->head_dim; a.max_len = d->max_len;
  a.scale = d->scale; a.qkv = (const uint16_t*)d->qkv; a.cosv = (const float*)d->cos; a.sinv = (const float*)d->sin;
  a.kc = (uint16_t*)d->k_cache; a.vc = (uint16_t*)d->v_cache; a.slot = (const int*)d->slot_table;
  a.pos = (const int64_t*)d->pos; a.mask = (const int64_t*)d->mask; a.out = (uint16_t*)d->out;
  a.q_rope = (uint16_t*)d->q_rope;
  const int NG = kAttnWaves * 64 / (a.D / 8);
  const size_t lds = sizeof(float) * ((size_t)3 * a.D + 2 * NG + (size_t)NG * a.D + a.max_len);
  if (lds > 64 * 1024) return MXVL_ERR_UNSUPPORTED;
  const dim3 grid(a.H, a.rows), block(kAttnWaves * 64);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (a.D) {
    case 64: hipLaunchKernelGGL(decode_attn_kernel<64>, grid, block, lds, s, a); break;
    case 128: hipLaunchKernelGGL(decode_attn_kernel<128>, grid, block, lds, s, a); break;
    case 256: hipLaunchKernelGGL(decode_attn_kernel<256>, grid, block, lds, s, a); break;
    default: return MXVL_ERR_UNSUPPORTED;
  }
  return dec_check();
}

int mxvl_decode_cross_attn(const mxvl_decode_cross_attn_desc* d, void* hip_stream) {
  if (!d || !d->q_rope || !d->k || !d->v || !d->text_state || !d->gate_weight || !d->gate_bias || !d->out) return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->n_heads <= 0 || d->n_kv_heads <= 0 || d->n_heads % d->n_kv_heads != 0 || d->n_keys <= 0) return MXVL_ERR_SHAPE;
  if (d->kv_rows_div <= 0 || d->rows % d->kv_rows_div != 0) return MXVL_ERR_SHAPE;
  if (d->head_dim != 64 && d->head_dim != 128 && d->head_dim != 256) return MXVL_ERR_UNSUPPORTED;
  if (d->out == d->text_state) return MXVL_ERR_UNSUPPORTED;   // every head's workgroup reads the whole row for the gate
  CrossAttnArgs a;
  a.rows = d->rows; a.H = d->n_heads; a.Hkv = d->n_kv_heads; a.D = d->head_dim; a.n_keys = d->n_keys;
  a.kv_rows_div = d->kv_rows_div; a.gate_flags = d->gate_flags; a.scale = d->scale;
  a.q_rope = (const uint16_t*)d->q_rope; a.k = (const uint16_t*)d->k; a.v = (const uint16_t*)d->v;
  a.key_mask = (const uint8_t*)d->key_mask; a.row_on = (const uint8_t*)d->row_on; a.text_state = (const uint16_t*)d->text_state;
  a.gate_w = (const uint16_t*)d->gate_weight; a.gate_b = (const uint16_t*)d->gate_bias; a.warm = (const uint16_t*)d->warm_up_gate;
  a.out = (uint16_t*)d->out;
  const int NG = kAttnWaves * 64 / (a.D / 8);
  const size_t lds = sizeof(float) * ((size_t)a.D + 2 * NG + (size_t)NG * a.D + kAttnWaves);
  if (lds > 64 * 1024) return MXVL_ERR_UNSUPPORTED;
  const dim3 grid(a.H, a.rows), block(kAttnWaves * 64);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (a.D) {
    case 64: hipLaunchKernelGGL(decode_cross_attn_kernel<64>, grid, block, lds, s, a); break;
    case 128: hipLaunchKernelGGL(decode_cross_attn_kernel<128>, grid, block, lds, s, a); break;
    case 256: hipLaunchKernelGGL(decode_cross_attn_kernel<256>, grid, block, lds, s, a); break;
    default: return MXVL_ERR_UNSUPPORTED;
  }
  return dec_check();
}

}  // extern "C"
