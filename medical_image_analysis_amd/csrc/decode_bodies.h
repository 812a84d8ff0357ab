// decode_bodies.h -- device-side pieces shared by the per-launch decode kernels (decode.hip) and the persistent whole-stack
// kernel (decode_stack.hip): argument blocks, bf16 helpers, the single-query attention bodies, and the two ways the small
// activation vectors travel between phases (PlainIO / LLIO below).
#pragma once
#include "mxvl_common.h"

namespace mxvl {

constexpr int kMaxRows = 8;

struct GemvArgs {
  int rows, K, N, swiglu, out_f32, ablate;
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

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// weights are read exactly once per token: non-temporal loads keep them from displacing the activations in L2
// (explicit global address space: a pointer that reaches the kernel through a table in memory is "generic" to the compiler,
// and a flat_load counts on lgkmcnt as well -- every LDS wait would then also wait for the weight stream)
__device__ inline uint4 ldnt(const uint16_t* p) {
  const u32x4_t v = __builtin_nontemporal_load((__attribute__((address_space(1))) const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline uint4 ldg16(const uint16_t* p) {
  const u32x4_t v = *(__attribute__((address_space(1))) const u32x4_t*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline uint4 ldw(const uint16_t* row, int kk, int K) {
  return kk < K ? ldnt(row + kk) : make_uint4(0, 0, 0, 0);
}
// kPF-load batch of a weight row at columns k0 + lane * 8 + j * 512.  Rows at least a whole batch long (every real model) take
// the unconditional form: one lane address, immediate offsets, nothing between the loads.
template <int PF>
__device__ inline void ld_batch(const uint16_t* row, int k0, int K, int lane, uint4 (&pre)[PF]) {
  if (k0 + PF * 512 <= K) {      // wave-uniform
    const uint16_t* wl = row + k0 + lane * 8;
#pragma unroll
    for (int j = 0; j < PF; ++j) pre[j] = ldnt(wl + j * 512);
  } else {
#pragma unroll
    for (int j = 0; j < PF; ++j) pre[j] = ldw(row, k0 + lane * 8 + j * 512, K);
  }
}

// ---- how a kernel reads / writes the small activation vectors that travel between the phases of a decode step -----------
// PlainIO: ordinary bf16 arrays (per-launch kernels: the producer is an earlier launch).
struct PlainIO {
  template <int N> __device__ inline void gather(const uint16_t* base, const size_t (&idx)[N], uint16_t (&out)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = base[idx[i]];
  }
  __device__ inline uint4 get8(const uint16_t* base, size_t i) const { return *(const uint4*)(base + i); }
  __device__ inline uint16_t peek(const uint16_t* base, size_t i) const { return base[i]; }
  // called by every lane of a wave with consecutive i
  __device__ inline void put(uint16_t* base, size_t i, uint16_t v, uint32_t) const { base[i] = v; }
};

// LLIO: the persistent stack kernel (decode_stack.hip).  Producer and consumer are different workgroups of the SAME launch, on
// different XCDs: element pairs travel as 8-byte slots {2 x bf16, epoch} written and read with sc0 sc1 accesses (they bypass
// the non-coherent cache levels; an 8-byte store becomes visible as a unit).  A consumer polls the slot until the flag holds
// the epoch of the exchange it waits for -- no grid barrier, no separate flag traffic: one store and one load on the path
// between two phases.  `base` pointers address slots (element i lives in slot i / 2, 8 bytes each).  Polling is bounded:
// on a timeout the error word is raised and every later poll of the workgroup gives up at once (no hang, garbage out).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
constexpr int kPollLimit = 1 << 17;
struct LLIO {
  uint32_t e_in;        // epoch the polled slots must carry
  int* dead;            // LDS word of the workgroup: nonzero = a poll timed out
  unsigned* err;        // device error word
  static __device__ inline u32x2_t slot(const uint16_t* base, size_t i) {
    return *(__attribute__((address_space(1))) const volatile u32x2_t*)((const char*)base + (i >> 1) * 8);
  }
  static __device__ inline u32x4_t slot2(const uint16_t* base, size_t i) {   // slots of elements i .. i + 3 (i % 4 == 0)
    return *(__attribute__((address_space(1))) const volatile u32x4_t*)((const char*)base + (i >> 1) * 8);
  }
  __device__ inline bool give_up(int& spins) const {
    if (++spins > kPollLimit) {
      *(volatile int*)dead = 1;
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    if ((spins & 63) == 0 && *(volatile int*)dead) return true;
    __builtin_amdgcn_s_sleep(2);
    return false;
  }
  template <int N> __device__ inline void gather(const uint16_t* base, const size_t (&idx)[N], uint16_t (&out)[N]) const {
    u32x2_t s[N];
    int spins = 0;
    while (true) {
#pragma unroll
      for (int i = 0; i < N; ++i) s[i] = slot(base, idx[i]);
      bool ok = true;
#pragma unroll
      for (int i = 0; i < N; ++i) ok = ok && s[i].y == e_in;
#if MXVL_EXP & 1
      break;
#endif
      if (ok || give_up(spins)) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = (uint16_t)((idx[i] & 1) ? s[i].x >> 16 : s[i].x);
  }
  __device__ inline uint4 get8(const uint16_t* base, size_t i) const {
    u32x4_t a, b;
    int spins = 0;
    while (true) {
      a = slot2(base, i);
      b = slot2(base, i + 4);
      if ((a.y == e_in && a.w == e_in && b.y == e_in && b.w == e_in) || give_up(spins)) break;
    }
    return make_uint4(a.x, a.z, b.x, b.z);
  }
  // an element every consumer of an EARLIER exchange has already waited for
  __device__ inline uint16_t peek(const uint16_t* base, size_t i) const {
    const uint32_t w = *(__attribute__((address_space(1))) const volatile uint32_t*)((const char*)base + (i >> 1) * 8);
    return (uint16_t)((i & 1) ? w >> 16 : w);
  }
  // called by every lane of a wave with consecutive i (i of lane 0 even): even lanes store the pair {own, neighbour}
  __device__ inline void put(uint16_t* base, size_t i, uint16_t v, uint32_t e_out) const {
    const uint32_t other = (uint32_t)__shfl_down((int)v, 1, 64);
    if (!(i & 1)) {
      const u32x2_t w = {(uint32_t)v | (other << 16), e_out};
      *(__attribute__((address_space(1))) volatile u32x2_t*)((char*)base + (i >> 1) * 8) = w;
    }
  }
};

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
  uint16_t* q_rope;          // optional (rows, H * D): the rotated, UNscaled query (what the hybrid layers' image cross-attention reads)
};

struct CrossAttnArgs {
  int rows, H, Hkv, D, n_keys, kv_rows_div, gate_flags;
  float scale;
  const uint16_t* q_rope;      // (rows, H * D) rotated query (decode_attn_kernel wrote it)
  const uint16_t *k, *v;       // (rows / kv_rows_div, Hkv, n_keys, D) image keys / values
  const uint8_t* key_mask;     // optional (rows / kv_rows_div, n_keys), nonzero = may attend
  const uint8_t* row_on;       // optional (rows / kv_rows_div): 0 = this sample carries no image (its context is zeroed)
  const uint16_t* text_state;  // (rows, H * D) the self-attention output the gate reads and the context is added to
  const uint16_t *gate_w, *gate_b, *warm;   // Linear(hidden, 1) weight (hidden), bias (1), warm-up gate (1, optional)
  uint16_t* out;               // (rows, H * D) text_state + ctx * gate
};

constexpr int kAttnWaves = 8;     // per-launch kernels; the stack kernel runs the same bodies with its 16 waves
// scratch floats the attention bodies need (self-attention additionally keeps max_len slot ids behind them)
constexpr int decode_attn_lds_floats(int D, int nwv) { return 3 * D + 2 * (nwv * 64 / (D / 8)) + (nwv * 64 / (D / 8)) * D; }
constexpr int decode_cross_attn_lds_floats(int D, int nwv) { return D + 2 * (nwv * 64 / (D / 8)) + (nwv * 64 / (D / 8)) * D + nwv; }

// One workgroup per (head, row).  A cached K or V line of D bf16 is read by LPR = D/8 lanes with one 16-byte load
// each, so a wave covers 64/LPR positions per load and the workgroup 8 x 64/LPR; K and V of a position are loaded
// together and folded into a running (max, sum, out[8]) per lane group (one-pass softmax), merged once at the end.
// NWV waves cooperate on one (head h, row m); `sm` = NWV-dependent scratch (see decode_attn_lds_floats).  Ends with the
// stores of its output issued (not waited for).
template <int D, int NWV, class IO>
__device__ inline void decode_attn_body(const AttnArgs& p, const int h, const int m, float* sm, const IO& io, uint32_t e_out = 0) {
  constexpr int LPR = D / 8, RPW = 64 / LPR, NG = NWV * RPW;
  const int T = p.max_len;
  float* sq = sm;                 // [D] rotated query
  float* sk = sq + D;             // [D] rotated new key
  float* sv = sk + D;             // [D] new value
  float* gm = sv + D;             // [NG] group maxima
  float* gl = gm + NG;            // [NG] group sums
  float* go = gl + NG;            // [NG][D] group outputs
  int* ssl = (int*)(go + NG * D); // [T] slot of each position, -1 = masked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int group = p.H / p.Hkv, hk = h / group;
  const int pos = (int)*p.pos;
  const size_t row = (size_t)m * (p.H + 2 * p.Hkv) * D;
  const size_t iq = row + (size_t)h * D, ik = row + (size_t)(p.H + hk) * D, iv = row + (size_t)(p.H + p.Hkv + hk) * D;
  for (int t = tid; t <= pos; t += NWV * 64)
    ssl[t] = p.mask[(size_t)m * T + t] != 0 ? p.slot[(size_t)m * T + t] : -1;
  // RoPE (hybrid_decoder_layer.py:284-322): x*cos + rotate_half(x)*sin, computed in the activation dtype (bf16)
  if (tid < D) {
    const int d = tid, half = D / 2;
    const float c = bf2f(f2bf(p.cosv[(size_t)m * D + d])), s = bf2f(f2bf(p.sinv[(size_t)m * D + d]));
    const int dp = d < half ? d + half : d - half;
    const size_t idx[5] = {iq + d, iq + dp, ik + d, ik + dp, iv + d};
    uint16_t in[5];
    io.template gather<5>(p.qkv, idx, in);      // all five in flight together (the stack kernel polls them)
    const float qd = bf2f(in[0]), qo = d < half ? -bf2f(in[1]) : bf2f(in[1]);
    const float kd = bf2f(in[2]), ko = d < half ? -bf2f(in[3]) : bf2f(in[3]);
    const uint16_t vnew = in[4];
    const float qr = bf2f(f2bf(bf2f(f2bf(qd * c)) + bf2f(f2bf(qo * s))));
    const float kr = bf2f(f2bf(bf2f(f2bf(kd * c)) + bf2f(f2bf(ko * s))));
    sq[d] = qr * p.scale;
    if (p.q_rope) io.put(p.q_rope, (size_t)m * p.H * D + (size_t)h * D + d, f2bf(qr), e_out);
    sk[d] = kr;
    sv[d] = bf2f(vnew);
    if (h % group == 0) {  // one head of the group appends to the cache (slot m owns position pos of beam m)
      const size_t o = (((size_t)m * p.Hkv + hk) * T + pos) * D + d;
      p.kc[o] = f2bf(kr);
      p.vc[o] = vnew;
    }
  }
  __syncthreads();
  const int sub = lane % LPR, g = wave * RPW + lane / LPR;
  float qv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qv[j] = sq[sub * 8 + j];
  float mx = -1e30f, l = 0.0f, o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.0f;
  auto fold = [&](bool live, const float* kf, const float* vf) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(qv[j], kf[j], s);
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
    const float mn = live ? fmaxf(mx, s) : mx;
    const float corr = fast_exp(mx - mn), pr = live ? fast_exp(s - mn) : 0.0f;
    mx = mn;
    l = fmaf(l, corr, pr);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaf(o[j], corr, pr * vf[j]);
  };
  auto unpack = [](const uint4 v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = __builtin_bit_cast(float, w[j] << 16);
      f[2 * j + 1] = __builtin_bit_cast(float, w[j] & 0xffff0000u);
    }
  };
  constexpr int U = 4;
  for (int t0 = g; t0 < pos; t0 += NG * U) {   // cached positions 0 .. pos-1
    uint4 kq[U], vq[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * NG;
      const int sl = t < pos ? ssl[t] : -1;
      live[u] = sl >= 0;
      kq[u] = make_uint4(0, 0, 0, 0);
      vq[u] = kq[u];
      if (live[u]) {
        const size_t a = (((size_t)sl * p.Hkv + hk) * T + t) * D + sub * 8;
        kq[u] = ldg16(p.kc + a);
        vq[u] = ldg16(p.vc + a);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kf[8], vf[8];
      unpack(kq[u], kf);
      unpack(vq[u], vf);
      fold(live[u], kf, vf);
    }
  }
  if (g == 0) {  // the fresh position (always attended: its mask bit was just set)
    float kf[8], vf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { kf[j] = sk[sub * 8 + j]; vf[j] = sv[sub * 8 + j]; }
    fold(ssl[pos] >= 0, kf, vf);
  }
  if (sub == 0) { gm[g] = mx; gl[g] = l; }
#pragma unroll
  for (int j = 0; j < 8; ++j) go[g * D + sub * 8 + j] = o[j];
  __syncthreads();
  if (tid < D) {
    float gmax = -1e30f;
    for (int i = 0; i < NG; ++i) gmax = fmaxf(gmax, gm[i]);
    float num = 0.0f, den = 0.0f;
    for (int i = 0; i < NG; ++i) {
      const float w = fast_exp(gm[i] - gmax);
      num = fmaf(w, go[i * D + tid], num);
      den = fmaf(w, gl[i], den);
    }
    io.put(p.out, (size_t)m * p.H * D + (size_t)h * D + tid, f2bf(num / den), e_out);
  }
}

// Image cross-attention of a hybrid decoder layer for ONE new token per row (EMRRG/models/hybrid_decoder_layer.py:653-697,
// `all2media_cross_attn`): ctx = softmax(q K_img^T * scale + mask) V_img from the layer's RoPE'd query to the image keys /
// values (constant over a generation: projected once when the layer is conditioned), then
//     out = text_state + (row_on * ctx) * gate,   gate = tanh?(w_g . text_state + b_g) * warm_up?,
// every product / sum rounded to bf16 where the reference's bf16 tensor ops round.  One workgroup per (head, row), same
// lane mapping and one-pass softmax as decode_attn_kernel; the gate (a hidden-wide dot product per row) is recomputed by
// every head's workgroup -- 8 KB from L2 -- instead of costing a launch of its own.
template <int D, int NWV, class IO>
__device__ inline void decode_cross_attn_body(const CrossAttnArgs& p, const int h, const int m, float* sm, const IO& io, uint32_t e_out = 0) {
  constexpr int LPR = D / 8, RPW = 64 / LPR, NG = NWV * RPW, NT = NWV * 64;
  float* sq = sm;                 // [D] scaled query
  float* gm = sq + D;             // [NG]
  float* gl = gm + NG;            // [NG]
  float* go = gl + NG;            // [NG][D]
  float* sred = go + NG * D;      // [NWV] gate partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int group = p.H / p.Hkv, hk = h / group;
  const int ms = m / p.kv_rows_div;            // image sample of this row (beams of a sample share it)
  const int hidden = p.H * D;
  // ---- gate -------------------------------------------------------------------------------------------------------
  float part = 0.0f;
  for (int c = tid * 8; c < hidden; c += NT * 8) {
    const uint4 xv = io.get8(p.text_state, (size_t)m * hidden + c);
    const uint4 wv = ldg16(p.gate_w + c);
    part = dot2(xv.x, wv.x, part); part = dot2(xv.y, wv.y, part); part = dot2(xv.z, wv.z, part); part = dot2(xv.w, wv.w, part);
  }
  part = wave_sum(part);
  if (lane == 0) sred[wave] = part;
  if (tid < D) {
    const size_t idx[1] = {(size_t)m * hidden + (size_t)h * D + tid};
    uint16_t in[1];
    io.template gather<1>(p.q_rope, idx, in);
    sq[tid] = bf2f(in[0]) * p.scale;
  }
  __syncthreads();
  float gate = 0.0f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) gate += sred[w];
  gate = bf2f(f2bf(gate + bf2f(p.gate_b[0])));                                  // Linear output, bf16
  if (p.gate_flags & 1) gate = bf2f(f2bf(tanhf(gate)));                          // nn.Tanh in bf16
  if (p.warm) {
    float wu = bf2f(p.warm[0]);
    if (p.gate_flags & 2) wu = bf2f(f2bf(tanhf(wu)));                            // text-only variant: gate * warm.tanh()
    gate = bf2f(f2bf(gate * wu));
  }
  // ---- one-query attention over the image tokens ----------------------------------------------------------------------
  const int sub = lane % LPR, g = wave * RPW + lane / LPR;
  float qv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qv[j] = sq[sub * 8 + j];
  float mx = -1e30f, l = 0.0f, o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.0f;
  const uint16_t* kb = p.k + ((size_t)ms * p.Hkv + hk) * p.n_keys * D + sub * 8;
  const uint16_t* vb = p.v + ((size_t)ms * p.Hkv + hk) * p.n_keys * D + sub * 8;
  const uint8_t* km = p.key_mask ? p.key_mask + (size_t)ms * p.n_keys : nullptr;
  constexpr int U = 4;
  for (int t0 = g; t0 < p.n_keys; t0 += NG * U) {
    uint4 kq[U], vq[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * NG;
      live[u] = t < p.n_keys && (!km || km[t] != 0);
      kq[u] = make_uint4(0, 0, 0, 0);
      vq[u] = kq[u];
      if (live[u]) {
        kq[u] = ldg16(kb + (size_t)t * D);
        vq[u] = ldg16(vb + (size_t)t * D);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t kw[4] = {kq[u].x, kq[u].y, kq[u].z, kq[u].w}, vw[4] = {vq[u].x, vq[u].y, vq[u].z, vq[u].w};
      float s = 0.0f, vf[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s = fmaf(qv[2 * j], __builtin_bit_cast(float, kw[j] << 16), s);
        s = fmaf(qv[2 * j + 1], __builtin_bit_cast(float, kw[j] & 0xffff0000u), s);
        vf[2 * j] = __builtin_bit_cast(float, vw[j] << 16);
        vf[2 * j + 1] = __builtin_bit_cast(float, vw[j] & 0xffff0000u);
      }
#pragma unroll
      for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
      const float mn = live[u] ? fmaxf(mx, s) : mx;
      const float corr = fast_exp(mx - mn), pr = live[u] ? fast_exp(s - mn) : 0.0f;
      mx = mn;
      l = fmaf(l, corr, pr);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(o[j], corr, pr * vf[j]);
    }
  }
  if (sub == 0) { gm[g] = mx; gl[g] = l; }
#pragma unroll
  for (int j = 0; j < 8; ++j) go[g * D + sub * 8 + j] = o[j];
  __syncthreads();
  if (tid < D) {
    float gmax = -1e30f;
    for (int i = 0; i < NG; ++i) gmax = fmaxf(gmax, gm[i]);
    float num = 0.0f, den = 0.0f;
    for (int i = 0; i < NG; ++i) {
      const float w = fast_exp(gm[i] - gmax);
      num = fmaf(w, go[i * D + tid], num);
      den = fmaf(w, gl[i], den);
    }
    float ctx = den > 0.0f ? bf2f(f2bf(num / den)) : 0.0f;                       // attention output, bf16
    if (p.row_on && p.row_on[ms] == 0) ctx = 0.0f;
    const size_t o_idx = (size_t)m * hidden + (size_t)h * D + tid;
    io.put(p.out, o_idx, f2bf(bf2f(io.peek(p.text_state, o_idx)) + bf2f(f2bf(ctx * gate))), e_out);
  }
}

}  // namespace mxvl
