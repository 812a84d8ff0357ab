// decode_stack.hip -- ONE persistent launch for the whole decoder stack of a report-generation step on gfx950.
//
// Same arithmetic, launch by launch, as the per-kernel step (decode.hip: fused RMSNorm+QKV GEMV, RoPE / cache append /
// one-query attention, [gated image cross-attention of a conditioned hybrid layer], o_proj GEMV + residual, fused RMSNorm +
// gate/up GEMV + SwiGLU, down GEMV + residual, final RMSNorm + lm_head) -- the reference leaves these to HF generate
// (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:292-301; layer arithmetic EMRRG/models/hybrid_decoder_layer.py:
// 185-199, 266-337, 392-457, 653-697).
//
// Why: a Llama-2-7B token streams 13.5 GB of weights through 129 GEMVs; as separate launches every one of them pays a
// dependent-launch gap, a first-byte latency and a drain -- measured 5 us per launch against 6-36 us of streaming
// (profiles/r03_decode_timeline.txt), 46 % of HBM peak for the token.  Weights do not depend on activations: here one
// workgroup per CU stays resident for the whole stack and every wave issues the loads of its first weight row of the NEXT
// phase before it finishes the current one, so the HBM stream keeps running while the (tiny) activations change hands.
//
// How the activations change hands: NOT through a grid barrier.  A barrier (arrival counter + poll) measured 6 us idle and
// 13 us under the weight stream -- four serialised memory round trips (store ack, atomic, poll, reload) -- and made the step
// SLOWER than separate launches (6.0 vs 3.6 ms per token, tools/decode_ab.py).  Instead every inter-phase vector lives in
// 8-byte slots {2 x bf16, epoch} (LLIO, decode_bodies.h): the producer's store carries its own "ready" flag, the consumer
// polls the data it needs anyway.  One store + one load between two phases; dependencies, not barriers, keep a slot array
// from being overwritten before its readers are done (every reader of an exchange produces into the next one, and every
// consumer waits for all producers of the vector it reads).  Polls are bounded: a launch that is not co-resident raises the
// error word and finishes with garbage instead of hanging the GPU.
#include <algorithm>

#include "decode_bodies.h"

namespace mxvl {

struct StackLayer {   // = mxvl_decode_layer (include/mxvl.h)
  const uint16_t *ln1, *wqkv, *bqkv, *wo, *ln2, *wgate, *wup, *wdown;
  uint16_t *kc, *vc;
  const uint16_t *img_k, *img_v;
  const uint8_t *img_key_mask, *img_row_on;
  const uint16_t *img_gate_w, *img_gate_b, *img_warm;
  float eps1, eps2;
  int img_n_keys, img_kv_rows_div, img_gate_flags, reserved;
};
static_assert(sizeof(StackLayer) == sizeof(mxvl_decode_layer), "layer table layout");

struct StackArgs {
  int rows, hidden, inter, H, Hkv, D, max_len, vocab, n_layers;
  float scale, eps_final;
  const StackLayer* layers;
  const uint16_t* x_in;          // (rows, hidden) bf16 token embeddings
  uint16_t* ws;                  // slot arrays (LLIO, decode_bodies.h): x | x2 | qkv | att | att2 | q_rope | act
  const float *cosv, *sinv;
  const int* slot;
  const int64_t *pos, *mask;
  const uint16_t *norm_w, *lm_head;
  float* logits;
  unsigned* sync;                // [0] launch counter (epoch base), [1] error word
};

constexpr int kSW = 16;        // waves per workgroup
constexpr int kPF = 8;         // 16-byte loads in flight per lane = one 4096-column weight row per wave
constexpr int kMaxItems = 4;   // output columns per wave of a phase whose outputs other workgroups read (N <= 4 x 16 x gridDim.x)

// exchanges of a layer, numbered so that every slot array sees strictly growing epochs: epoch = base + 8 * layer + id
enum { EX_X = 0, EX_QKV = 1, EX_ATT = 2, EX_ATT2 = 3, EX_X2 = 4, EX_ACT = 5 };

// One GEMV phase = the arithmetic of gemv_bf16_kernel (decode.hip) between two slot arrays.
struct Phase {
  int K, N, swiglu;
  float eps;
  const uint16_t* x;      // slots (rows, K)
  uint32_t e_in;
  const uint16_t *g, *W, *W2, *bias;
  const uint16_t* res;    // slots (rows, N) of an earlier exchange, optional
  uint16_t* y;            // slots (rows, N); the lm_head phase writes plain fp32 logits instead
  uint32_t e_out;
};
struct Prime { const uint16_t* W; int K, N; };

// first weight row of this wave for the phase `n`: issued, not waited for
__device__ inline void gemv_prime(const Prime& n, uint4 (&pre)[kPF]) {
  const int lane = threadIdx.x & 63, n_first = blockIdx.x * kSW + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (n_first < n.N) {                                   // wave-uniform
    ld_batch<kPF>(n.W + (size_t)n_first * n.K, 0, n.K, lane, pre);
  } else {
#pragma unroll
    for (int j = 0; j < kPF; ++j) pre[j] = make_uint4(0, 0, 0, 0);
  }
}

// pre[] holds this wave's first row on entry.  LOGITS = false: outputs are gathered in LDS and leave as slot pairs {2 x bf16,
// e_out} from wave 0, and every wave puts its first row of phase `next` in flight before it finishes (there is nothing to
// wait for: data and flag are one store).  LOGITS = true (lm_head): plain fp32 stores, nothing follows.
template <int M, bool LOGITS>
__device__ inline void gemv_phase(const Phase& p, int rows, uint4 (&pre)[kPF], const Prime& next, const LLIO& io, uint16_t* sx,
                                  float (*s_part)[kSW], float* s_out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K, N = p.N;
  const int S = p.swiglu ? 2 : 1;
  const int TW = gridDim.x * kSW;
  const int n_first = blockIdx.x * kSW + wave;
  const int n_items = n_first < N ? ((N - 1 - n_first) / TW + 1) * S : 0;
  auto row_ptr = [&](int item) -> const uint16_t* {
    const int n = n_first + (p.swiglu ? item >> 1 : item) * TW;
    return ((p.swiglu && (item & 1)) ? p.W2 : p.W) + (size_t)n * K;
  };
  // a workgroup without a row in this phase has nothing to read either (and must not: it is not part of the dependency
  // chain that keeps a slot array from being overwritten while it is still being read)
#if MXVL_EXP & 2
  const bool wg_active = false;      // timing only: no activation reads at all
#else
  const bool wg_active = blockIdx.x * kSW < N;
#endif
  if (wg_active) {
    for (int c0 = 0; c0 < K; c0 += 8192) {
      const int kk = c0 + tid * 8;
      const bool on = kk < K;
      // poll the 4 slots (8 elements) of every row together: one round trip when the producer is done
      u32x4_t lo[M], hi[M];
      int spins = 0;
      while (true) {
        bool ok = true;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          if (on && m < rows) {
            lo[m] = LLIO::slot2(p.x, (size_t)m * K + kk);
            hi[m] = LLIO::slot2(p.x, (size_t)m * K + kk + 4);
            ok = ok && lo[m].y == io.e_in && lo[m].w == io.e_in && hi[m].y == io.e_in && hi[m].w == io.e_in;
          } else {
            lo[m] = u32x4_t{0, 0, 0, 0};
            hi[m] = lo[m];
          }
        }
#if MXVL_EXP & 1
        break;              // timing only: one load, no waiting
#endif
        if (ok || io.give_up(spins)) break;
      }
      uint4 xr[M];
#pragma unroll
      for (int m = 0; m < M; ++m) xr[m] = make_uint4(lo[m].x, lo[m].z, hi[m].x, hi[m].z);
      if (!p.g) {
        if (on) {
#pragma unroll
          for (int m = 0; m < M; ++m) *(uint4*)(sx + (size_t)m * K + kk) = xr[m];
        }
      } else {   // K <= 8192 with a norm (checked by the launcher)
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
        const uint4 gv = on ? ldg16(p.g + kk) : make_uint4(0, 0, 0, 0);
        const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int m = 0; m < M; ++m) {
          float tot = 0.0f;
#pragma unroll
          for (int w2 = 0; w2 < kSW; ++w2) tot += s_part[m][w2];
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
  }
  __syncthreads();

  float gate[M];
#pragma unroll
  for (int m = 0; m < M; ++m) gate[m] = 0.0f;
  for (int item = 0; item < n_items; ++item) {
    float acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = 0.0f;
    const uint16_t* w = row_ptr(item);
#pragma unroll
    for (int j = 0; j < kPF; ++j) {
      const int kk = lane * 8 + j * 512;
      if (kk < K) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const uint4 xv = *(const uint4*)(sx + (size_t)m * K + kk);
          acc[m] = dot2(pre[j].x, xv.x, dot2(pre[j].y, xv.y, dot2(pre[j].z, xv.z, dot2(pre[j].w, xv.w, acc[m]))));
        }
      }
    }
    // the rest of a row longer than 4096 columns, kPF loads at a time (a single load per trip leaves the wave one round trip
    // per 1 KB: the down projection, K = 11008, ran at 3.9 TB/s that way)
    for (int k0 = kPF * 512; k0 < K; k0 += kPF * 512) {
      ld_batch<kPF>(w, k0, K, lane, pre);
#pragma unroll
      for (int j = 0; j < kPF; ++j) {
        const int kk = k0 + lane * 8 + j * 512;
        if (kk < K) {
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const uint4 xv = *(const uint4*)(sx + (size_t)m * K + kk);
            acc[m] = dot2(pre[j].x, xv.x, dot2(pre[j].y, xv.y, dot2(pre[j].z, xv.z, dot2(pre[j].w, xv.w, acc[m]))));
          }
        }
      }
    }
    if (item + 1 < n_items) {
      ld_batch<kPF>(row_ptr(item + 1), 0, K, lane, pre);
    } else if (!LOGITS) {
      gemv_prime(next, pre);
    }
    const int col = p.swiglu ? item >> 1 : item;
    const int n = n_first + col * TW;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float v = wave_sum(acc[m]);
      const size_t o = (size_t)m * N + n;
      bool have = false;
      if (p.swiglu) {
        if (!(item & 1)) gate[m] = v;
        else {
          const float gte = bf2f(f2bf(gate[m])), up = bf2f(f2bf(v));
          v = bf2f(f2bf(bf2f(f2bf(gte * sigmoid(gte))) * up));
          have = true;
        }
      } else {
        if (p.bias) v += bf2f(p.bias[n]);
        if (p.res && m < rows) v = bf2f(f2bf(v)) + bf2f(io.peek(p.res, o));
        have = true;
      }
      if (have && lane == 0 && m < rows) {
        if constexpr (LOGITS) ((float*)p.y)[o] = v;
        else s_out[(wave * kMaxItems + col) * M + m] = v;
      }
    }
  }
  if (!LOGITS && n_items == 0) gemv_prime(next, pre);   // no row in this phase: the wave still prefetches for the next one
  if constexpr (!LOGITS) {
    __syncthreads();
    if (wave == 0 && wg_active) {
      const int cols = (N - 1 - blockIdx.x * kSW) / TW + 1;     // columns of the workgroup's first wave (the most any wave has)
      const int total = (kSW / 2) * cols * M;
      for (int i = lane; i < total; i += 64) {
        const int m = i % M, w2 = (i / M) % (kSW / 2), col = i / (M * (kSW / 2));
        const int n = blockIdx.x * kSW + 2 * w2 + col * TW;      // even; N is even: n + 1 < N whenever n < N
        if (n < N && m < rows) {
          const uint32_t a = f2bf(s_out[((2 * w2) * kMaxItems + col) * M + m]), b = f2bf(s_out[((2 * w2 + 1) * kMaxItems + col) * M + m]);
          const u32x2_t v = {a | (b << 16), p.e_out};
          *(__attribute__((address_space(1))) volatile u32x2_t*)((char*)p.y + (((size_t)m * N + n) >> 1) * 8) = v;
        }
      }
    }
  }
}

struct StackWs {     // slot arrays inside the workspace, in elements (= 4-byte units)
  uint32_t x, x2, qkv, att, att2, q_rope, act, total;
};
__host__ __device__ inline StackWs stack_ws(int rows, int hidden, int inter, int H, int Hkv, int D) {
  StackWs w;
  uint32_t o = 0;
  auto take = [&](size_t n) { const uint32_t at = o; o += ((uint32_t)n + 3u) & ~3u; return at; };
  w.x = take((size_t)rows * hidden); w.x2 = take((size_t)rows * hidden); w.qkv = take((size_t)rows * (H + 2 * Hkv) * D);
  w.att = take((size_t)rows * H * D); w.att2 = take((size_t)rows * H * D); w.q_rope = take((size_t)rows * H * D);
  w.act = take((size_t)rows * inter);
  w.total = o;
  return w;
}
__device__ inline uint16_t* ws_at(const StackArgs& a, uint32_t elem_off) { return (uint16_t*)((char*)a.ws + (size_t)elem_off * 4); }

// kind: 0 RMSNorm + QKV, 2 o_proj + residual, 3 RMSNorm + gate/up + SwiGLU, 4 down + residual, 5 final RMSNorm + lm_head
__device__ inline Phase stack_phase(const StackArgs& a, const StackWs& w, uint32_t base, int l, int kind) {
  Phase p;
  p.swiglu = 0; p.eps = 0.0f; p.g = nullptr; p.W2 = nullptr; p.bias = nullptr; p.res = nullptr;
  const uint32_t e = base + 8u * (uint32_t)l;
  if (kind == 5) {
    p.x = ws_at(a, w.x); p.e_in = e + EX_X; p.g = a.norm_w; p.eps = a.eps_final; p.W = a.lm_head; p.y = (uint16_t*)a.logits; p.e_out = 0;
    p.K = a.hidden; p.N = a.vocab;
    return p;
  }
  const StackLayer& L = a.layers[l];
  switch (kind) {
    case 0:
      p.x = ws_at(a, w.x); p.e_in = e + EX_X; p.g = L.ln1; p.eps = L.eps1; p.W = L.wqkv; p.bias = L.bqkv;
      p.y = ws_at(a, w.qkv); p.e_out = e + EX_QKV; p.K = a.hidden; p.N = (a.H + 2 * a.Hkv) * a.D;
      break;
    case 2:
      p.x = ws_at(a, L.img_k ? w.att2 : w.att); p.e_in = e + (L.img_k ? EX_ATT2 : EX_ATT); p.W = L.wo; p.res = ws_at(a, w.x);
      p.y = ws_at(a, w.x2); p.e_out = e + EX_X2; p.K = a.H * a.D; p.N = a.hidden;
      break;
    case 3:
      p.x = ws_at(a, w.x2); p.e_in = e + EX_X2; p.g = L.ln2; p.eps = L.eps2; p.W = L.wgate; p.W2 = L.wup; p.swiglu = 1;
      p.y = ws_at(a, w.act); p.e_out = e + EX_ACT; p.K = a.hidden; p.N = a.inter;
      break;
    default:
      p.x = ws_at(a, w.act); p.e_in = e + EX_ACT; p.W = L.wdown; p.res = ws_at(a, w.x2);
      p.y = ws_at(a, w.x); p.e_out = e + 8u + EX_X; p.K = a.inter; p.N = a.hidden;      // the next layer's input
      break;
  }
  return p;
}
__device__ inline Prime prime_of(const Phase& p) { return Prime{p.W, p.K, p.N}; }

template <int D>
__device__ inline void stack_attention(const StackArgs& a, const StackWs& w, const StackLayer& L, uint32_t e, float* smf, LLIO io) {
  const bool cond = L.img_k != nullptr;
  AttnArgs at;
  at.rows = a.rows; at.H = a.H; at.Hkv = a.Hkv; at.D = a.D; at.max_len = a.max_len; at.scale = a.scale;
  at.qkv = ws_at(a, w.qkv); at.cosv = a.cosv; at.sinv = a.sinv; at.slot = a.slot; at.pos = a.pos; at.mask = a.mask;
  at.out = ws_at(a, w.att);
  at.kc = L.kc; at.vc = L.vc; at.q_rope = cond ? ws_at(a, w.q_rope) : nullptr;
  io.e_in = e + EX_QKV;
  for (int item = blockIdx.x; item < a.H * a.rows; item += gridDim.x) {
    decode_attn_body<D, kSW>(at, item % a.H, item / a.H, smf, io, e + EX_ATT);
    __syncthreads();
  }
}
template <int D>
__device__ inline void stack_cross_attention(const StackArgs& a, const StackWs& w, const StackLayer& L, uint32_t e, float* smf, LLIO io) {
  CrossAttnArgs ca;
  ca.rows = a.rows; ca.H = a.H; ca.Hkv = a.Hkv; ca.D = a.D; ca.n_keys = L.img_n_keys; ca.kv_rows_div = L.img_kv_rows_div;
  ca.gate_flags = L.img_gate_flags; ca.scale = a.scale; ca.q_rope = ws_at(a, w.q_rope); ca.k = L.img_k; ca.v = L.img_v;
  ca.key_mask = L.img_key_mask; ca.row_on = L.img_row_on; ca.text_state = ws_at(a, w.att); ca.gate_w = L.img_gate_w;
  ca.gate_b = L.img_gate_b; ca.warm = L.img_warm; ca.out = ws_at(a, w.att2);
  io.e_in = e + EX_ATT;
  for (int item = blockIdx.x; item < a.H * a.rows; item += gridDim.x) {
    decode_cross_attn_body<D, kSW>(ca, item % a.H, item / a.H, smf, io, e + EX_ATT2);
    __syncthreads();
  }
}

// Runs on the stream right before the stack kernel: a fresh epoch base, and the argument block copied to device memory (the
// stack kernel reads its arguments from there field by field where it needs them: as by-value kernel arguments all 33 dwords
// stay live in scalar registers for the whole launch and push the weight-stream loops into scratch spills).
__global__ void stack_setup_kernel(StackArgs a, StackArgs* dst) {
  *dst = a;
  a.sync[0] += 1;
}

// A layer is written out phase by phase (not looped over a phase index) so that the prefetch registers have short, explicit
// live ranges: defined at the end of one phase, consumed at the start of the next, dead across the attention.  The loop is
// rotated so that its back edge is crossed where no prefetch is live: an iteration = attention(l), o_proj(l), gate/up(l),
// down(l), then layer l + 1's QKV -- or, behind the last layer, the final RMSNorm + lm_head.
template <int M, int D>
__global__ __launch_bounds__(1024) void decode_stack_kernel(const StackArgs* __restrict__ ap) {
  const StackArgs& a = *ap;
  extern __shared__ __attribute__((aligned(16))) uint16_t dyn[];   // [M][K] activations of a GEMV phase | attention scratch
  __shared__ float s_part[kMaxRows][kSW];
  __shared__ float s_out[kSW * kMaxItems * kMaxRows];
  __shared__ int s_dead;
  float* smf = (float*)dyn;
  uint4 pre[kPF];
  const int rows = a.rows;
  const StackWs w = stack_ws(rows, a.hidden, a.inter, a.H, a.Hkv, a.D);
  // epochs of this launch: the launch counter (bumped on the stream before the launch) keeps them apart from whatever an
  // earlier token / generation left in the slots; never 0 (fresh workspace)
  const uint32_t base = ((a.sync[0] % 4194303u) + 1u) << 10;
  if (threadIdx.x == 0) s_dead = 0;
  __syncthreads();
  LLIO io;
  io.e_in = 0; io.dead = &s_dead; io.err = a.sync + 1;

  {   // token embeddings -> slots of exchange X of layer 0 (every workgroup a slice), then layer 0's QKV
    const Phase p0 = stack_phase(a, w, base, 0, 0);
    gemv_prime(prime_of(p0), pre);
    const int pairs = rows * a.hidden / 2;
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < pairs; i += gridDim.x * 1024) {
      const u32x2_t v = {*(const uint32_t*)(a.x_in + 2 * (size_t)i), base + EX_X};
      *(__attribute__((address_space(1))) volatile u32x2_t*)((char*)ws_at(a, w.x) + (size_t)i * 8) = v;
    }
    io.e_in = p0.e_in;
    gemv_phase<M, false>(p0, rows, pre, Prime{p0.W, 0, 0}, io, dyn, s_part, s_out);   // nothing prefetched: the attention follows
  }
  for (int l = 0; l < a.n_layers; ++l) {
    const StackLayer& L = a.layers[l];
    const bool last = l + 1 == a.n_layers;
    const uint32_t e = base + 8u * (uint32_t)l;
    stack_attention<D>(a, w, L, e, smf, io);
    if (L.img_k != nullptr) stack_cross_attention<D>(a, w, L, e, smf, io);
    {
      const Phase po = stack_phase(a, w, base, l, 2), pg = stack_phase(a, w, base, l, 3);
      gemv_prime(prime_of(po), pre);
      io.e_in = po.e_in;
      gemv_phase<M, false>(po, rows, pre, prime_of(pg), io, dyn, s_part, s_out);
    }
    {
      const Phase pg = stack_phase(a, w, base, l, 3), pd = stack_phase(a, w, base, l, 4);
      io.e_in = pg.e_in;
      gemv_phase<M, false>(pg, rows, pre, prime_of(pd), io, dyn, s_part, s_out);
    }
    {
      const Phase pd = stack_phase(a, w, base, l, 4), pn = stack_phase(a, w, base, l + 1, last ? 5 : 0);
      io.e_in = pd.e_in;
      gemv_phase<M, false>(pd, rows, pre, prime_of(pn), io, dyn, s_part, s_out);
    }
    {
      const Phase pn = stack_phase(a, w, base, l + 1, last ? 5 : 0);
      io.e_in = pn.e_in;
      if (last) gemv_phase<M, true>(pn, rows, pre, prime_of(pn), io, dyn, s_part, s_out);
      else gemv_phase<M, false>(pn, rows, pre, Prime{pn.W, 0, 0}, io, dyn, s_part, s_out);   // the attention follows
    }
  }
}

static size_t stack_lds_bytes(const StackArgs& a, int M) {
  const size_t gemv = (size_t)M * std::max(std::max(a.hidden, a.inter), a.H * a.D) * sizeof(uint16_t);
  const size_t attn = sizeof(float) * (size_t)decode_attn_lds_floats(a.D, kSW) + sizeof(int) * (size_t)a.max_len;
  const size_t cross = sizeof(float) * (size_t)decode_cross_attn_lds_floats(a.D, kSW);
  return std::max(gemv, std::max(attn, cross));
}

static int stack_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  }
  return cus;
}

template <int M, int D>
static int launch_stack_d(const StackArgs& a, hipStream_t s) {
  const int grid = stack_cus();
  if (grid <= 0) return MXVL_ERR_LAUNCH;
  const size_t lds = stack_lds_bytes(a, M);
  if (lds > 150 * 1024) return MXVL_ERR_UNSUPPORTED;
  // every phase whose outputs other workgroups read keeps <= kMaxItems output columns per wave
  const int widest = std::max(std::max((a.H + 2 * a.Hkv) * a.D, a.inter), a.hidden);
  if (widest > kMaxItems * kSW * grid) return MXVL_ERR_UNSUPPORTED;
  static bool raised = false;   // per template instance
  if (!raised) {
    if (hipFuncSetAttribute((const void*)decode_stack_kernel<M, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
      return MXVL_ERR_LAUNCH;
    raised = true;
  }
  StackArgs* dev_args = (StackArgs*)((char*)a.ws + (size_t)stack_ws(a.rows, a.hidden, a.inter, a.H, a.Hkv, a.D).total * 4);
  hipLaunchKernelGGL(stack_setup_kernel, dim3(1), dim3(1), 0, s, a, dev_args);
  hipLaunchKernelGGL((decode_stack_kernel<M, D>), dim3(grid), dim3(kSW * 64), lds, s, (const StackArgs*)dev_args);
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
template <int M>
static int launch_stack(const StackArgs& a, hipStream_t s) {
  switch (a.D) {
    case 64: return launch_stack_d<M, 64>(a, s);
    case 128: return launch_stack_d<M, 128>(a, s);
    default: return launch_stack_d<M, 256>(a, s);
  }
}

}  // namespace mxvl

using namespace mxvl;

extern "C" {

int64_t mxvl_decode_stack_workspace_bytes(int rows, int hidden, int intermediate, int n_heads, int n_kv_heads, int head_dim) {
  if (rows <= 0 || hidden <= 0 || intermediate <= 0 || n_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0) return 0;
  return (int64_t)stack_ws(rows, hidden, intermediate, n_heads, n_kv_heads, head_dim).total * 4 + 512;   // slots + the argument block
}

int mxvl_decode_stack(const mxvl_decode_stack_desc* d, void* hip_stream) {
  if (!d || !d->layers || !d->x || !d->workspace || !d->cos || !d->sin || !d->slot_table || !d->pos || !d->mask ||
      !d->final_norm_weight || !d->lm_head_weight || !d->logits || !d->sync)
    return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->rows > kMaxRows || d->n_layers <= 0 || d->n_layers > 120 || d->hidden <= 0 || d->intermediate <= 0 ||
      d->vocab <= 0 || d->n_heads <= 0 || d->n_kv_heads <= 0 || d->n_heads % d->n_kv_heads || d->max_len <= 0)
    return MXVL_ERR_SHAPE;
  if (d->head_dim != 64 && d->head_dim != 128 && d->head_dim != 256) return MXVL_ERR_UNSUPPORTED;
  if (d->hidden % 8 || d->intermediate % 8 || d->hidden > 8192) return MXVL_ERR_UNSUPPORTED;   // fused RMSNorm: a row in registers
  StackArgs a;
  a.rows = d->rows; a.hidden = d->hidden; a.inter = d->intermediate; a.H = d->n_heads; a.Hkv = d->n_kv_heads; a.D = d->head_dim;
  a.max_len = d->max_len; a.vocab = d->vocab; a.n_layers = d->n_layers; a.scale = d->scale; a.eps_final = d->final_norm_eps;
  a.layers = (const StackLayer*)d->layers;
  a.x_in = (const uint16_t*)d->x; a.ws = (uint16_t*)d->workspace;
  a.cosv = (const float*)d->cos; a.sinv = (const float*)d->sin; a.slot = (const int*)d->slot_table; a.pos = (const int64_t*)d->pos;
  a.mask = (const int64_t*)d->mask; a.norm_w = (const uint16_t*)d->final_norm_weight; a.lm_head = (const uint16_t*)d->lm_head_weight;
  a.logits = (float*)d->logits; a.sync = (unsigned*)d->sync;
  hipStream_t s = (hipStream_t)hip_stream;
  switch (d->rows) {
    case 1: return launch_stack<1>(a, s);
    case 2: return launch_stack<2>(a, s);
    case 3: return launch_stack<3>(a, s);
    case 4: return launch_stack<4>(a, s);
    default: return launch_stack<8>(a, s);
  }
}

}  // extern "C"
