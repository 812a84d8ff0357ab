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
#include <type_traits>
#include <cstdlib>

#include "decode_elt.h"

namespace mxvl {

constexpr int kMaxRows = 8;

struct GemvArgs {
  int rows, K, N, swiglu, out_f32, ablate;
  float eps;
  const uint16_t *x, *g, *W, *W2, *bias, *res;
  void* y;
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// weights are read exactly once per token: non-temporal loads keep them from displacing the activations in L2
__device__ inline uint4 ldnt(const uint16_t* p) {
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline uint4 ldw(const uint16_t* row, int kk, int K) {
  return kk < K ? ldnt(row + kk) : make_uint4(0, 0, 0, 0);
}

// Work unit = one output column n: weight row n of W (with SwiGLU: gate row n of W, then up row n of W2).  Units are
// dealt round-robin to the 16 x gridDim.x waves of the launch, a whole weight row per wave at a time, so every wave
// streams 2*K contiguous bytes with PF 16-byte loads in flight per lane.  The next row's first PF loads are issued
// before the current row's cross-lane reduction, and the very first row's before the RMSNorm prologue, so the HBM
// stream never waits for the prologue or an epilogue.
template <typename E, int M>
__global__ __launch_bounds__(1024) void gemv_kernel(const GemvArgs p) {
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
  // Stage x (or RMSNorm(x)*g, Qwen2RMSNorm hybrid_decoder_layer.py:193-198: bf16(bf16(x*rstd) * g)) in LDS, one 16-byte load per
  // thread per (row, 8192-column block).  Order of the requests (round 3; loads return in order, so the order IS the latency chain):
  // the activations and the norm weight of the first block go out FIRST, the wave's first weight row right behind them, and
  // only then is anything waited for -- x arrives after one round trip and the norm / LDS staging runs while the weight row is
  // still streaming in.  Before, the weight row was requested first and the staging loop's header made hipcc wait for ALL
  // outstanding loads before it even asked for x (a loop-carried write-after-write on the x registers): weight round trip, then
  // x round trip, then the norm weight's -- three serialised latencies in front of the first dot product of every launch.
  auto stage = [&](int kk, bool on, const uint4 (&xr)[M], uint4 gv) {
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
          const float a = E::f((uint16_t)w[j]), b = E::f((uint16_t)(w[j] >> 16));
          s = fmaf(a, a, fmaf(b, b, s));
        }
        s = wave_sum(on ? s : 0.0f);
        if (lane == 0) s_part[m][wave] = s;
      }
      __syncthreads();
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
          const float a = E::rr(E::f((uint16_t)w[j]) * rstd) * E::f((uint16_t)gw[j]);
          const float b = E::rr(E::f((uint16_t)(w[j] >> 16)) * rstd) * E::f((uint16_t)(gw[j] >> 16));
          o[j] = (uint32_t)E::r(a) | ((uint32_t)E::r(b) << 16);
        }
        if (on) *(uint4*)(sx + (size_t)m * K + kk) = make_uint4(o[0], o[1], o[2], o[3]);
      }
    }
  };
  if (MXVL_ABL(p.ablate == 2)) return;
  uint4 pre[PF];
  {
    const int kk = tid * 8;
    const bool on = kk < K;
    // UNCONDITIONAL loads (threads past the row read its first 16 bytes again; their values are never stored and are masked out
    // of the norm's sum): behind a per-lane `if` hipcc waits for a load on the spot
    const int kc = on ? kk : 0;
    uint4 xr[M];
#pragma unroll
    for (int m = 0; m < M; ++m) xr[m] = *(const uint4*)(p.x + (size_t)m * K + kc);
    // branch-free: no norm -> the activations' first bytes stand in for the (unused) norm weight; a wave without a row, or a
    // lane past the end of a short row, re-reads the start of a valid row (every consumer tests kk < K / n_items).  With any
    // branch between the requests hipcc loses count of what is in flight and waits for everything at the join.
    const uint4 gv = *(const uint4*)((p.g ? p.g : p.x) + kc);
    {
      const uint16_t* w = n_items > 0 ? row_ptr(0) : p.W;
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int kj = lane * 8 + j * 512;
        pre[j] = ldnt(w + (kj < K ? kj : 0));
      }
    }
    if (!MXVL_ABL(p.ablate == 3)) stage(kk, on, xr, gv);
  }
  for (int c0 = 8192; c0 < K; c0 += 8192) {      // rows longer than one block (down projection: K = 11008); never with a norm
    const int kk = c0 + tid * 8;
    const bool on = kk < K;
    uint4 xr[M];
#pragma unroll
    for (int m = 0; m < M; ++m) xr[m] = on ? *(const uint4*)(p.x + (size_t)m * K + kk) : make_uint4(0, 0, 0, 0);
    if (!MXVL_ABL(p.ablate == 3)) stage(kk, on, xr, make_uint4(0, 0, 0, 0));
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
          acc[m] = E::dot2(pre[j].x, xv.x, E::dot2(pre[j].y, xv.y, E::dot2(pre[j].z, xv.z, E::dot2(pre[j].w, xv.w, acc[m]))));
        }
      }
    }
    // the rest of a row longer than PF x 512 columns, PF loads at a time: with one load per trip the wave had a single 1 KB
    // request in flight and the down projection (K = 11008) streamed at 3.9 TB/s against 5.0 for the 4096-column rows
    // (profiles/r03_decode_timeline.txt)
    for (int k0 = PF * 512; k0 < K; k0 += PF * 512) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {          // branch-free, as above
        const int kj = k0 + lane * 8 + j * 512;
        pre[j] = ldnt(w + (kj < K ? kj : 0));
      }
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int kk = k0 + lane * 8 + j * 512;
        if (kk < K) {
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const uint4 xv = *(const uint4*)(sx + (size_t)m * K + kk);
            acc[m] = E::dot2(pre[j].x, xv.x, E::dot2(pre[j].y, xv.y, E::dot2(pre[j].z, xv.z, E::dot2(pre[j].w, xv.w, acc[m]))));
          }
        }
      }
    }
    if (item + 1 < n_items) {  // next row's head goes in flight before this row's reduction
      const uint16_t* wn = row_ptr(item + 1);
#pragma unroll
      for (int j = 0; j < PF; ++j) {          // branch-free (lanes past a short row re-read its start; consumers test kk < K)
        const int kj = lane * 8 + j * 512;
        pre[j] = ldnt(wn + (kj < K ? kj : 0));
      }
    }
    const int n = n_first + (p.swiglu ? item >> 1 : item) * TW;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float v = wave_sum(acc[m]);
      const size_t o = (size_t)m * N + n;
      if (p.swiglu) {
        if (!(item & 1)) gate[m] = v;
        else if (lane == 0) {  // bf16(bf16(silu(gate)) * up), gate/up rounded to bf16 first (what the torch modules do)
          const float gte = E::rr(gate[m]), up = E::rr(v);
          ((uint16_t*)p.y)[o] = E::r(E::rr(gte * sigmoid(gte)) * up);
        }
      } else if (lane == 0) {
        if (p.bias) v += E::f(p.bias[n]);
        if (p.res) v = E::rr(v) + E::f(p.res[o]);  // the linear output rounds to bf16 before the residual add
        if (p.out_f32) ((float*)p.y)[o] = v; else ((uint16_t*)p.y)[o] = E::r(v);
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
  uint16_t* q_rope;          // optional (rows, H * D): the rotated, UNscaled query (what the hybrid layers' image cross-attention reads)
  int ablate;                // measurement build only (MXVL_ATTN_ABLATE): the beams kernel returns after 1 the prologue, 3 the cached positions
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

// measurement build only (python -m medical_image_analysis_amd.build --ablate; tools/decode_attn_probe.py): MXVL_ATTN_WAVES = 8 / 16
// pins the workgroup size of the per-row kernel (beams kernel: 8 / 4 waves).  The product library reads no environment: always 0.
static int attn_waves_env() {
  static const int v = MXVL_ABL_ENV("MXVL_ATTN_WAVES");
  return (v == 8 || v == 16) ? v : 0;
}
// Sum over the LPR lanes that share a cache row (LPR = head_dim / 8 consecutive lanes), result in every lane.  DPP moves inside a
// 16-lane row -- quad_perm for the partners at distance 1 and 2, row_half_mirror / row_mirror for the other half of 8 / 16 (every
// lane of a half already holds the same partial sum, so the mirrored partner is as good as the xor one and the additions are the
// same additions: bit-identical to the __shfl_xor butterfly) -- instead of ds_bpermute_b32: four DEPENDENT LDS-crossbar round trips
// per (position, beam) score were the attention kernels' critical path (round 4: 230 shared positions cost 1 us per trip of 32).
template <int LPR>
__device__ __forceinline__ float group_sum(float s) {
  auto mv = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  s += mv(s, std::integral_constant<int, 0xB1>{});                          // quad_perm [1,0,3,2]
  s += mv(s, std::integral_constant<int, 0x4E>{});                          // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) s += mv(s, std::integral_constant<int, 0x141>{});   // row_half_mirror
  if constexpr (LPR >= 16) s += mv(s, std::integral_constant<int, 0x140>{});  // row_mirror
  if constexpr (LPR >= 32) s += __shfl_xor(s, 16, 64);                      // the other row (head_dim 256 only)
  static_assert(LPR == 8 || LPR == 16 || LPR == 32, "8 columns per lane, head_dim 64 / 128 / 256");
  return s;
}

constexpr int kAttnWaves = 8;    // (16 waves x 8 positions per group -- one trip of cache loads for a 358-position report instead of
                                 //  three -- measured slower: 9.7 vs 8.9 us per launch, profiles/r03_decode_timeline.txt)

// One workgroup per (head, row).  A cached K or V line of D bf16 is read by LPR = D/8 lanes with one 16-byte load
// each, so a wave covers 64/LPR positions per load and the workgroup 8 x 64/LPR; K and V of a position are loaded
// together and folded into a running (max, sum, out[8]) per lane group (one-pass softmax), merged once at the end.
template <typename E, int D, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void decode_attn_kernel(const AttnArgs p) {
  constexpr int LPR = D / 8, RPW = 64 / LPR, NG = NW * RPW;
  static_assert((DEPTH & (DEPTH - 1)) == 0, "ring slots are picked with a mask");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.max_len;
  char* ring_all = (char*)sm;     // [NW][DEPTH][K 1 KB | V 1 KB]: landing ring of the cache rows (see the loop)
  float* sq = sm + NW * DEPTH * 512;   // [D] rotated query
  float* sk = sq + D;             // [D] rotated new key
  float* sv = sk + D;             // [D] new value
  float* gm = sv + D;             // [NG] group maxima
  float* gl = gm + NG;            // [NG] group sums
  float* go = gl + NG;            // [NG][D] group outputs
  int* ssl = (int*)(go + NG * D); // [T] slot of each position, -1 = masked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, m = blockIdx.y;
  const int group = p.H / p.Hkv, hk = h / group;
  const int pos = (int)*p.pos;
  const size_t row = (size_t)m * (p.H + 2 * p.Hkv) * D;
  const uint16_t* q = p.qkv + row + (size_t)h * D;
  const uint16_t* kn = p.qkv + row + (size_t)(p.H + hk) * D;
  const uint16_t* vn = p.qkv + row + (size_t)(p.H + p.Hkv + hk) * D;
  // every request of the prologue goes out before anything is used, none behind a per-lane branch: as `mask ? slot : -1` and
  // `d < half ? -q[d + half] : q[d - half]` the loads sat in exec-masked blocks and hipcc waited for each of them in turn -- eight
  // serialised round trips in front of a kernel whose whole body is a handful of them (8.9 us per launch, 32 launches per token)
  // (the first trip of the slot sweep does not wait for *pos: entries past pos are never looked at, and the table has T of them)
  {
    const int t = tid < T ? tid : T - 1;
    const int64_t mk = p.mask[(size_t)m * T + t];
    const int sl = p.slot[(size_t)m * T + t];
    __builtin_amdgcn_sched_barrier(0);
    ssl[t] = mk != 0 ? sl : -1;
  }
  for (int t = tid + NW * 64; t <= pos; t += NW * 64) {
    const int64_t mk = p.mask[(size_t)m * T + t];
    const int sl = p.slot[(size_t)m * T + t];
    ssl[t] = mk != 0 ? sl : -1;
  }
  // RoPE (hybrid_decoder_layer.py:284-322): x*cos + rotate_half(x)*sin, computed in the activation dtype (bf16)
  if (tid < D) {
    const int d = tid, half = D / 2;
    const int dp = d < half ? d + half : d - half;
    const uint16_t rq = q[d], rqo = q[dp], rk = kn[d], rko = kn[dp], rv = vn[d];
    const float rc = p.cosv[(size_t)m * D + d], rs = p.sinv[(size_t)m * D + d];
    const float c = E::rr(rc), s = E::rr(rs);
    const float qd = E::f(rq), qo = d < half ? -E::f(rqo) : E::f(rqo);
    const float kd = E::f(rk), ko = d < half ? -E::f(rko) : E::f(rko);
    const float qr = E::rr(E::rr(qd * c) + E::rr(qo * s));
    const float kr = E::rr(E::rr(kd * c) + E::rr(ko * s));
    sq[d] = qr * p.scale;
    if (p.q_rope) p.q_rope[(size_t)m * p.H * D + (size_t)h * D + d] = E::r(qr);
    sk[d] = kr;
    sv[d] = E::f(rv);
    if (h % group == 0) {  // one head of the group appends to the cache (slot m owns position pos of beam m)
      const size_t o = (((size_t)m * p.Hkv + hk) * T + pos) * D + d;
      p.kc[o] = E::r(kr);
      p.vc[o] = rv;
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
    s = group_sum<LPR>(s);
    const float mn = live ? fmaxf(mx, s) : mx;
    const float corr = fast_exp(mx - mn), pr = live ? fast_exp(s - mn) : 0.0f;
    mx = mn;
    l = fmaf(l, corr, pr);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaf(o[j], corr, pr * vf[j]);
  };
  // The cache rows land in LDS, not in registers: a wave owns a ring of DEPTH trips; a trip = one global_load_lds_dwordx4 of K and
  // one of V, each lane moving the 16 bytes IT reads back (LDS address = lane * 16: no swizzle, no cross-lane traffic, no barrier)
  // -- LDS as the register file of the loads in flight.  At batch 1 x beam 3 this kernel is a chain of memory round trips (96
  // workgroups, nothing else on the chip): four positions per lane group in VGPRs made a 300-position report three dependent
  // trips; with DEPTH = 8 trips requested at once it is one and a bit.  vmcnt retires in order: the wait in front of trip j is the
  // constant 2 * (DEPTH - 1) while the ring is full and 0 on the last DEPTH - 1 trips.
  const int ntrips = (pos + NG - 1) / NG;
  char* ring = ring_all + wave * (DEPTH * 2048);
  const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  const size_t c1 = (size_t)p.Hkv * T * D, lane_off = (size_t)hk * T * D + sub * 8;
  uint32_t livebits = 0;                 // bit (j & 31): trip j of this lane group carries an attended position
  int iss = 0;
  auto issue = [&]() {
    if (iss >= ntrips) return;
    const int t = g + iss * NG;
    const int sl = ssl[t < pos ? t : 0];
    const bool live = t < pos && sl >= 0;    // a masked / out-of-range group re-reads the row's first cache line
    const size_t off = (live ? (size_t)sl * c1 + (size_t)t * D : (size_t)m * c1) + lane_off;
    const int bit = iss & 31;
    livebits = (livebits & ~(1u << bit)) | ((uint32_t)live << bit);
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(iss & (DEPTH - 1)) * 2048u);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst), "v"(p.kc + off), "v"(p.vc + off) : "memory", "scc");
    ++iss;
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the counts below are counts of ring trips only
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) issue();
  for (int j = 0; j < ntrips; ++j) {                       // cached positions 0 .. pos-1
    if (j + DEPTH <= ntrips) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* slot = ring + (j & (DEPTH - 1)) * 2048 + lane * 16;
    const uint4 kq = *(const uint4*)slot, vq = *(const uint4*)(slot + 1024);
    const bool live = (livebits >> (j & 31)) & 1u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // both rows are in registers: the slot may be refilled
    issue();
    float kf[8], vf[8];
    elt_unpack8<E>(kq, kf);
    elt_unpack8<E>(vq, vf);
    fold(live, kf, vf);
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
    p.out[(size_t)m * p.H * D + (size_t)h * D + tid] = E::r(num / den);
  }
}

// The same step for the nb beams of ONE sample in one workgroup (grid: head x sample).  Beams continue a common ancestor: the prompt
// (whose slots the stepper points at one physical copy) and usually the first generated tokens name the SAME cache lines for every
// beam -- 230 of ~300 positions at the reference's settings.  With a workgroup per (head, row) those lines were fetched nb times
// (three workgroups asking for a line at the same moment are three fabric reads: measured, the shared slots alone bought nothing);
// here positions [0, n_shared) -- the longest prefix on which all beams agree -- are loaded ONCE for the nb beams, the rest per
// beam.  18 rows x 32 heads x 300 positions: 88 MB -> 43 MB of cache reads per layer.
// History of the arithmetic (docs/DESIGN_HISTORY.md): a VALU version (16 lanes per cache row, one-pass softmax per position, rows in
// VGPRs, later in an LDS landing ring) went 16.6 -> 14.2 us per layer at 6 x 3 rows and stopped there; the phase ablation
// (profiles/r04_attn_phase_ablation2.txt) showed it bound by VALU THROUGHPUT, chip-wide: 8 FMA + 4 DPP adds + a dozen softmax
// instructions + 8 FMA per (position, beam), replicated over the 16 lanes of a cache row, is ~200 VALU instructions per
// 32-position trip and three beams; 1536 waves x 17 trips of that is ~7 us of all 1024 SIMDs, whatever the memory system does
// (ring depth 2 .. 8 and 8 vs 16 waves measured flat).  So the two products go to MFMA and the softmax is per 16-position TILE:
//   S[pos][beam]   = sum_d K[pos][d] Q[beam][d]      v_mfma_f32_16x16x32_bf16: A = K rows straight from the LDS tile (16-byte reads),
//                                                    B = the nb rotated queries (bf16, rows nb..15 zero), once per kernel
//   O^T[d][beam]  += sum_pos V[pos][d] P[pos][beam]  v_mfma_f32_16x16x16_bf16: A = V^T through ds_read_b64_tr_b16 (hardware
//                                                    transpose read of the row-major V tile), B = P straight from the S
//                                                    accumulator (C layout: lane (beam = l & 15, q = l >> 4) holds positions
//                                                    4 q .. 4 q + 3 -- exactly the k-slice lane (n, q) of a 16x16x16 B operand owns)
// so a lane owns ONE beam column: running max / sum / rescale are lane-local, the only cross-lane step per tile is the max over the
// four q (two ds_bpermute).  P is rounded to bf16 before the second product, as the modules' softmax(...).to(bf16) @ V does.
// A wave owns whole tiles (tile k -> wave k % NW): ns tiles over the positions every beam shares, then nt per beam over its own
// positions (only column `beam` of those is kept).  K and V tiles arrive by LDS-DMA into a two-stage ring per wave, XOR-swizzled
// on the SOURCE address (the DMA writes LDS linearly) so that both the 16-byte fragment reads and the transpose reads spread over
// the banks; explicit vmcnt waits (the compiler cannot see the DMA).  The fresh position of every beam (its K / V are in LDS, not
// in the cache yet for the other heads of a KV group) is one more partial in the final merge.
template <int D> __device__ __forceinline__ int attn_tile_key(int row) {
  constexpr int UPR = D / 8;
  return UPR == 8 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : UPR == 16 ? (((row & 3) << 2) | ((row >> 2) & 3)) : (row & (UPR - 1));
}
template <int D> __device__ __forceinline__ int attn_tile_off(int row, int unit) { return row * (D * 2) + ((unit ^ attn_tile_key<D>(row)) << 4); }

template <typename E, int D, int NB, int NW>
__global__ __launch_bounds__(NW * 64) void decode_attn_beams_mfma_kernel(const AttnArgs p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef elt_u32x4 bf16x8;      // eight 16-bit elements of a lane (E decides what they mean)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int UPR = D / 8, RPI = 64 / UPR;     // 16-byte units per cache row; rows per DMA instruction (1 KB)
  constexpr int TILE = 16, NI = TILE / RPI;      // positions per tile; DMA instructions per K (or V) tile
  constexpr int TB = TILE * D * 2, STAGE = 2 * TB, NST = 2, OPS = 2 * NI;
  constexpr int NKK = D / 32, NDT = D / 16, NT = NW * 64, NWP = NW + 1;
  static_assert(OPS <= 31, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.max_len;
  char* ring_all = (char*)sm;                                   // [NW][NST][K tile | V tile]; after the loop: wo [NB][NWP][D] fp32
  uint16_t* sqb = (uint16_t*)(ring_all + NW * NST * STAGE);     // [16][D] rotated queries, bf16 (rows nb .. 15 zero)
  float* sq = (float*)(sqb + 16 * D);                           // [NB][D] rotated, scaled queries (fresh position)
  float* sk = sq + NB * D;                                      // [NB][D] rotated new keys
  float* sv = sk + NB * D;                                      // [NB][D] new values
  float* wm = sv + NB * D;                                      // [NB][NWP] partial maxima
  float* wl = wm + NB * NWP;                                    // [NB][NWP] partial sums
  int* ssl = (int*)(wl + NB * NWP);                             // [NB][T] slot of each position, -1 = masked
  float* wo = (float*)ring_all;
  __shared__ int s_nsh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, m0 = blockIdx.y * NB;
  const int group = p.H / p.Hkv, hk = h / group;
  const int pos = (int)*p.pos;         // (first USED behind the loads below: the scalar load overlaps them)
  // raw RoPE operands of this thread's (row, dim) items: unconditional loads (clamped index), consumed after the slot sweep
  constexpr int RI = (NB * D + NT - 1) / NT;
  uint16_t rope_q[RI], rope_qo[RI], rope_k[RI], rope_ko[RI], rope_v[RI];
  float rope_c[RI], rope_s[RI];
#pragma unroll
  for (int it = 0; it < RI; ++it) {
    const int i = tid + it * NT < NB * D ? tid + it * NT : 0;
    const int r = i / D, d = i - r * D, half = D / 2, m = m0 + r;
    const int dp = d < half ? d + half : d - half;
    const size_t row = (size_t)m * (p.H + 2 * p.Hkv) * D;
    const uint16_t* q = p.qkv + row + (size_t)h * D;
    const uint16_t* kn = p.qkv + row + (size_t)(p.H + hk) * D;
    const uint16_t* vn = p.qkv + row + (size_t)(p.H + p.Hkv + hk) * D;
    rope_q[it] = q[d]; rope_qo[it] = q[dp]; rope_k[it] = kn[d]; rope_ko[it] = kn[dp]; rope_v[it] = vn[d];
    rope_c[it] = p.cosv[(size_t)m * D + d]; rope_s[it] = p.sinv[(size_t)m * D + d];
  }
  // the first trip of the slot sweep is requested here too, before *pos is needed (entries past pos are never looked at)
  int64_t mk0[NB];
  int sl0[NB];
  {
    const int t = tid < T ? tid : T - 1;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      mk0[r] = p.mask[(size_t)(m0 + r) * T + t];
      sl0[r] = p.slot[(size_t)(m0 + r) * T + t];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  for (int i = tid; i < (16 - NB) * D; i += NT) sqb[NB * D + i] = 0;
  if (tid == 0) s_nsh = pos;
  __syncthreads();
  for (int t = tid; t <= pos; t += NT) {
    int64_t mk[NB];
    int sl[NB];
    if (t == tid) {
#pragma unroll
      for (int r = 0; r < NB; ++r) { mk[r] = mk0[r]; sl[r] = sl0[r]; }
    } else {
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        mk[r] = p.mask[(size_t)(m0 + r) * T + t];
        sl[r] = p.slot[(size_t)(m0 + r) * T + t];
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // all 2 nb loads in flight before the first is looked at (hipcc interleaved them: two round trips)
    bool same = true;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const int v = mk[r] != 0 ? sl[r] : -1;
      ssl[r * T + t] = v;
      same = same && v == (mk[0] != 0 ? sl[0] : -1);
    }
    // the first position on which the beams differ: t grows with the lane, so the lowest set lane of the ballot has the wave's minimum
    const unsigned long long differ = __ballot(!same && t < pos);
    if (differ != 0 && lane == __ffsll(differ) - 1) atomicMin(&s_nsh, t);
  }
  // RoPE (hybrid_decoder_layer.py:284-322) of the nb rows' q / k, cache append by one head of each KV group
#pragma unroll
  for (int it = 0; it < RI; ++it) {
    const int i = tid + it * NT;
    if (i < NB * D) {
      const int r = i / D, d = i - r * D, half = D / 2, m = m0 + r;
      const float c = E::rr(rope_c[it]), sn = E::rr(rope_s[it]);
      const float qd = E::f(rope_q[it]), qo = d < half ? -E::f(rope_qo[it]) : E::f(rope_qo[it]);
      const float kd = E::f(rope_k[it]), ko = d < half ? -E::f(rope_ko[it]) : E::f(rope_ko[it]);
      const float qr = E::rr(E::rr(qd * c) + E::rr(qo * sn));
      const float kr = E::rr(E::rr(kd * c) + E::rr(ko * sn));
      sq[i] = qr * p.scale;
      sqb[i] = E::r(qr);
      if (p.q_rope) p.q_rope[(size_t)m * p.H * D + (size_t)h * D + d] = E::r(qr);
      sk[i] = kr;
      sv[i] = E::f(rope_v[it]);
      if (h % group == 0) {
        const size_t o = (((size_t)m * p.Hkv + hk) * T + pos) * D + d;
        p.kc[o] = E::r(kr);
        p.vc[o] = rope_v[it];
      }
    }
  }
  __syncthreads();
  if (MXVL_ABL(p.ablate == 1)) return;      // launch + slot sweep + RoPE + cache append
  const int nsh = s_nsh;
  const int l16 = lane & 15, q4 = lane >> 4;
  const int ns = (nsh + TILE - 1) / TILE, nt = (pos - nsh + TILE - 1) / TILE, ntiles = ns + NB * nt;
  // tile iterators (wave-uniform): tile k of this wave -> shared (r < 0) tile k, or tile i of beam r
  struct Tile { int k, r, i; };
  auto tile_first = [&](Tile& it) {
    it.k = wave; it.r = -1; it.i = 0;
    if (it.k >= ns && it.k < ntiles) { it.r = 0; it.i = it.k - ns; while (it.i >= nt) { it.i -= nt; ++it.r; } }
  };
  auto tile_next = [&](Tile& it) {
    it.k += NW;
    if (it.k >= ns && it.k < ntiles) {
      if (it.r < 0) { it.r = 0; it.i = it.k - ns; } else it.i += NW;
      while (it.i >= nt) { it.i -= nt; ++it.r; }
    }
  };
  char* ring = ring_all + wave * (NST * STAGE);
  const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
  const uint32_t c1 = (uint32_t)p.Hkv * (uint32_t)T * D;                    // elements per cache row (slot); 32-bit offsets: the launcher checks the cache fits
  const uint32_t hk_off = (uint32_t)hk * (uint32_t)T * D;
  const uint32_t dummy = (uint32_t)m0 * c1 + hk_off;                        // a masked / out-of-range row re-reads the sample's first line
  auto issue = [&](const Tile& it, int stage) {
    if (it.k >= ntiles) return;
    const int t0 = it.r < 0 ? it.k * TILE : nsh + it.i * TILE, lim = it.r < 0 ? nsh : pos;
    const int* sp = ssl + (it.r < 0 ? 0 : it.r) * T;
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)stage * STAGE);
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
      const int row = ii * RPI + lane / UPR, t = t0 + row;
      const bool in = t < lim;
      const int sl = sp[in ? t : 0];
      const bool live = in && sl >= 0;
      const uint32_t off = (live ? (uint32_t)sl * c1 + (uint32_t)t * D + hk_off : dummy) + (((lane % UPR) ^ attn_tile_key<D>(row)) << 3);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                   "s_add_u32 m0, m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(dst + ii * 1024), "v"(p.kc + off), "v"(p.vc + off), "n"(TB) : "memory", "scc");
    }
  };
  // B operand of the first product: lane (beam = l16, q4) owns Q[beam][32 kk + 8 q4 .. + 8)
  bf16x8 qb[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) qb[kk] = *(const bf16x8*)(sqb + l16 * D + kk * 32 + q4 * 8);
  f32x4 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float m_run = -1e30f, l_run = 0.0f;            // this lane's beam column; l_run sums this lane's four rows only until the end
  Tile ti, tc;
  tile_first(ti);
  tc = ti;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the counts below are counts of ring tiles only
  issue(ti, 0);
  tile_next(ti);
  issue(ti, 1);
  tile_next(ti);
  int stage = 0;
  for (; tc.k < ntiles; tile_next(tc), stage ^= 1) {
    if (tc.k + NW < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* kt = ring + stage * STAGE;
    const char* vt = kt + TB;
    typedef __attribute__((address_space(3))) s16x4 lds4;
    bf16x8 kf[NKK];
    s16x4 vf[NDT];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) kf[kk] = *(const bf16x8*)(kt + attn_tile_off<D>(l16, kk * 4 + q4));
    // V^T: within a 16-lane group lane t' supplies the address of row (t' >> 2), columns 4 (t' & 3) .. + 3 of a [4][16] block and
    // receives column t' of its 4 rows -> lane (d = 16 dt + l16, q4) gets V[4 q4 .. 4 q4 + 3][d]
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int col = dt * 16 + 4 * (l16 & 3);
      vf[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(vt + attn_tile_off<D>(q4 * 4 + (l16 >> 2), col >> 3) + (col & 7) * 2));
    }
    const int t0 = tc.r < 0 ? tc.k * TILE : nsh + tc.i * TILE, lim = tc.r < 0 ? nsh : pos;
    const int* sp = ssl + (tc.r < 0 ? 0 : tc.r) * T;
    bool ok[4];
    const bool colok = l16 < NB && (tc.r < 0 || l16 == tc.r);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + q4 * 4 + i;
      ok[i] = colok && t < lim && sp[t < lim ? t : 0] >= 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every fragment of the stage is in registers: it may be refilled
    issue(ti, stage);
    tile_next(ti);
    f32x4 sc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) sc = E::mfma32(kf[kk], qb[kk], sc);
    float s[4], tmax = -1e30f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] = sc[i] * p.scale;
      tmax = ok[i] ? fmaxf(tmax, s[i]) : tmax;
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m_run, tmax), corr = fast_exp(m_run - mn);
    m_run = mn;
    float pr[4], psum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pr[i] = ok[i] ? fast_exp(s[i] - mn) : 0.0f;
      psum += pr[i];
    }
    l_run = fmaf(l_run, corr, psum);
    s16x4 pb;
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = (short)E::r(pr[i]);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) oacc[dt][i] *= corr;
      oacc[dt] = E::mfma16(vf[dt], pb, oacc[dt]);
    }
  }
  if (MXVL_ABL(p.ablate == 3)) return;      // + every cached position
  // ---- merge: the q-lanes of a column by shuffles, the waves (+ the fresh position as partial NW) through LDS ----------------------
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  __syncthreads();                          // every wave is done with its ring: wo reuses the memory
  if (l16 < NB) {
    if (q4 == 0) { wm[l16 * NWP + wave] = m_run; wl[l16 * NWP + wave] = l_run; }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      *(f32x4*)(wo + (l16 * NWP + wave) * D + dt * 16 + q4 * 4) = oacc[dt];
  }
  for (int r = wave; r < NB; r += NW) {     // the fresh position of beam r: score by one wave
    float part = 0.0f;
    for (int d = lane; d < D; d += 64) part = fmaf(sq[r * D + d], sk[r * D + d], part);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    const bool live = ssl[r * T + pos] >= 0;
    if (lane == 0) { wm[r * NWP + NW] = live ? part : -1e30f; wl[r * NWP + NW] = live ? 1.0f : 0.0f; }
    for (int d = lane; d < D; d += 64) wo[(r * NWP + NW) * D + d] = live ? sv[r * D + d] : 0.0f;
  }
  __syncthreads();
  for (int i = tid; i < NB * D; i += NT) {
    const int r = i / D, d = i - r * D;
    float gmax = -1e30f;
#pragma unroll
    for (int w = 0; w < NWP; ++w) gmax = fmaxf(gmax, wm[r * NWP + w]);
    float num = 0.0f, den = 0.0f;
#pragma unroll
    for (int w = 0; w < NWP; ++w) {
      const float c = fast_exp(wm[r * NWP + w] - gmax);
      num = fmaf(c, wo[(r * NWP + w) * D + d], num);
      den = fmaf(c, wl[r * NWP + w], den);
    }
    p.out[(size_t)(m0 + r) * p.H * D + (size_t)h * D + d] = E::r(num / den);
  }
}

// Image cross-attention of a hybrid decoder layer for ONE new token per row (EMRRG/models/hybrid_decoder_layer.py:653-697,
// `all2media_cross_attn`): ctx = softmax(q K_img^T * scale + mask) V_img from the layer's RoPE'd query to the image keys /
// values (constant over a generation: projected once when the layer is conditioned), then
//     out = text_state + (row_on * ctx) * gate,   gate = tanh?(w_g . text_state + b_g) * warm_up?,
// every product / sum rounded to bf16 where the reference's bf16 tensor ops round.  One workgroup per (head, row), same
// lane mapping and one-pass softmax as decode_attn_kernel; the gate (a hidden-wide dot product per row) is recomputed by
// every head's workgroup -- 8 KB from L2 -- instead of costing a launch of its own.
template <typename E, int D>
__global__ __launch_bounds__(kAttnWaves * 64) void decode_cross_attn_kernel(const CrossAttnArgs p) {
  constexpr int LPR = D / 8, RPW = 64 / LPR, NG = kAttnWaves * RPW, NT = kAttnWaves * 64;
  extern __shared__ float sm[];
  float* sq = sm;                 // [D] scaled query
  float* gm = sq + D;             // [NG]
  float* gl = gm + NG;            // [NG]
  float* go = gl + NG;            // [NG][D]
  float* sred = go + NG * D;      // [kAttnWaves] gate partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, m = blockIdx.y;
  const int group = p.H / p.Hkv, hk = h / group;
  const int ms = m / p.kv_rows_div;            // image sample of this row (beams of a sample share it)
  const int hidden = p.H * D;
  // ---- gate -------------------------------------------------------------------------------------------------------
  float part = 0.0f;
  for (int c = tid * 8; c < hidden; c += NT * 8) {
    const uint4 xv = *(const uint4*)(p.text_state + (size_t)m * hidden + c);
    const uint4 wv = *(const uint4*)(p.gate_w + c);
    part = E::dot2(xv.x, wv.x, part); part = E::dot2(xv.y, wv.y, part); part = E::dot2(xv.z, wv.z, part); part = E::dot2(xv.w, wv.w, part);
  }
  part = wave_sum(part);
  if (lane == 0) sred[wave] = part;
  if (tid < D) sq[tid] = E::f(p.q_rope[(size_t)m * hidden + (size_t)h * D + tid]) * p.scale;
  __syncthreads();
  float gate = 0.0f;
#pragma unroll
  for (int w = 0; w < kAttnWaves; ++w) gate += sred[w];
  gate = E::rr(gate + E::f(p.gate_b[0]));                                  // Linear output, bf16
  if (p.gate_flags & 1) gate = E::rr(tanhf(gate));                          // nn.Tanh in bf16
  if (p.warm) {
    float wu = E::f(p.warm[0]);
    if (p.gate_flags & 2) wu = E::rr(tanhf(wu));                            // text-only variant: gate * warm.tanh()
    gate = E::rr(gate * wu);
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
        kq[u] = *(const uint4*)(kb + (size_t)t * D);
        vq[u] = *(const uint4*)(vb + (size_t)t * D);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = 0.0f, kf[8], vf[8];
      elt_unpack8<E>(kq[u], kf);
      elt_unpack8<E>(vq[u], vf);
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(qv[j], kf[j], s);
      s = group_sum<LPR>(s);
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
    float ctx = den > 0.0f ? E::rr(num / den) : 0.0f;                       // attention output, bf16
    if (p.row_on && p.row_on[ms] == 0) ctx = 0.0f;
    const size_t o_idx = (size_t)m * hidden + (size_t)h * D + tid;
    p.out[o_idx] = E::r(E::f(p.text_state[o_idx]) + E::rr(ctx * gate));
  }
}

// Everything a generation step does BEFORE the decoder stack, as one launch (it was 22 small torch kernels, 65 us + 100 us of
// gaps per token inside the hipGraph): the position of the step from the search state's counter, the beam re-ordering of the
// slot table (a thread owns a column: it reads the column of every parent row, then writes it), the new position's slot and
// mask bit, the token embeddings, and the RoPE cos / sin rows from a table the caller filled with its own rotary module.
struct PrologueArgs {
  int rows, hidden, max_len, D, prompt_len, table_len;
  const int64_t *tok, *beam, *cur, *n_real;
  const uint16_t* embed;
  const float *cos_t, *sin_t;
  int* slot;
  int64_t* mask;
  uint16_t* x;
  float *cosv, *sinv;
  int64_t* pos;
};
// Workgroup b of G owns the slot-table columns t = b * 256 + tid (+ G * 256 ...): a column is read for every parent row, then
// written, so the in-place re-ordering needs no second table and no cross-workgroup order.  MAXR bounds the per-thread column
// (registers): 8 for the GEMV row counts, 32 / 80 for the batched ones (6 x 3 ... 16 x 5 rows).
template <int MAXR>
__global__ __launch_bounds__(256) void decode_prologue_kernel(const PrologueArgs p) {
  __shared__ int s_parent[MAXR];
  const int tid = threadIdx.x, gtid = blockIdx.x * 256 + tid, gsz = gridDim.x * 256;
  const int64_t cur = *p.cur;
  const int pos = (int)(cur + p.prompt_len - 1), step = (int)(cur - 1);
  if (tid < MAXR) s_parent[tid] = tid < p.rows ? (int)p.beam[tid] : 0;
  __syncthreads();
  for (int t = gtid; t < p.max_len; t += gsz) {
    int v[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) v[r] = r < p.rows ? p.slot[(size_t)s_parent[r] * p.max_len + t] : 0;
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
      if (r < p.rows) p.slot[(size_t)r * p.max_len + t] = t == pos ? r : v[r];
  }
  if (gtid < p.rows && pos < p.max_len) p.mask[(size_t)gtid * p.max_len + pos] = 1;
  const int per_row = p.hidden / 8;
  for (int i = gtid; i < p.rows * per_row; i += gsz) {
    const int r = i / per_row, c = (i - r * per_row) * 8;
    *(uint4*)(p.x + (size_t)r * p.hidden + c) = *(const uint4*)(p.embed + (size_t)p.tok[r] * p.hidden + c);
  }
  for (int i = gtid; i < p.rows * p.D; i += gsz) {
    const int r = i / p.D, d = i - r * p.D;
    int64_t q = p.n_real[r] + step;
    q = q < 0 ? 0 : (q >= p.table_len ? p.table_len - 1 : q);
    p.cosv[i] = p.cos_t[(size_t)q * p.D + d];
    p.sinv[i] = p.sin_t[(size_t)q * p.D + d];
  }
  if (gtid == 0) *p.pos = pos;
}

static int dec_check() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

template <typename E, int M>
static int launch_gemv(const GemvArgs& a, hipStream_t s) {
  const int grid = std::max(1, std::min(256, (a.N + 15) / 16));
  const size_t lds = (size_t)M * a.K * sizeof(uint16_t);
  if (lds > 150 * 1024) return MXVL_ERR_UNSUPPORTED;  // rows * K bf16 must fit one CU's LDS
  // per call: the attribute belongs to the (kernel, device) pair, and a process may drive several devices
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)gemv_kernel<E, M>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
    return MXVL_ERR_LAUNCH;
  hipLaunchKernelGGL((gemv_kernel<E, M>), dim3(grid), dim3(1024), lds, s, a);
  return MXVL_OK;
}

template <typename E>
static int launch_gemv_rows(const GemvArgs& a, hipStream_t s) {
  switch (a.rows) {
    case 1: return launch_gemv<E, 1>(a, s);
    case 2: return launch_gemv<E, 2>(a, s);
    case 3: return launch_gemv<E, 3>(a, s);
    case 4: return launch_gemv<E, 4>(a, s);
    case 5: return launch_gemv<E, 5>(a, s);
    case 6: return launch_gemv<E, 6>(a, s);
    case 7: return launch_gemv<E, 7>(a, s);
    default: return launch_gemv<E, 8>(a, s);
  }
}

template <typename E>
static int launch_decode_attn(const AttnArgs& a, int beams, hipStream_t s) {
  if (beams > 1) {       // the beams of a sample share a workgroup (and every cache line they have in common)
    if (beams > 5 || a.rows % beams != 0) return MXVL_ERR_UNSUPPORTED;
    if ((uint64_t)a.rows * a.Hkv * a.max_len * a.D >= (1ull << 31)) return MXVL_ERR_UNSUPPORTED;   // 32-bit cache offsets
    const int nb = beams;
    // matrix-core kernel: 8 waves (4 when head_dim 256 or a long slot table would not leave room for the ring)
    const size_t stage = (size_t)2 * 16 * a.D * 2;
    auto lds_of = [&](int nw) {
      return (size_t)nw * 2 * stage + (size_t)16 * a.D * 2 + sizeof(float) * ((size_t)3 * nb * a.D + 2 * nb * (nw + 1) + (size_t)nb * a.max_len);
    };
    int nw = a.D <= 128 ? 8 : 4;
    if (attn_waves_env() == 8 || attn_waves_env() == 16) nw = attn_waves_env() == 16 ? 4 : 8;   // (probe knob: 16 = the 4-wave shape)
    if (nw == 8 && lds_of(8) > 160 * 1024) nw = 4;
    const size_t lds = lds_of(nw);
    if (lds > 160 * 1024) return MXVL_ERR_UNSUPPORTED;
    const dim3 grid(a.H, a.rows / nb), block(nw * 64);
#define MXVL_ATTN_BEAMS(DD, NB)                                                                                                    \
  do {                                                                                                                             \
    void (*kern)(const AttnArgs) = nw == 8 ? decode_attn_beams_mfma_kernel<E, DD, NB, (DD <= 128 ? 8 : 4)> : decode_attn_beams_mfma_kernel<E, DD, NB, 4>; \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return MXVL_ERR_LAUNCH;                                                                                                      \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);                                                                              \
  } while (0)
#define MXVL_ATTN_BEAMS_D(NB)                                                                                                      \
  switch (a.D) {                                                                                                                   \
    case 64: MXVL_ATTN_BEAMS(64, NB); break;                                                                                       \
    case 128: MXVL_ATTN_BEAMS(128, NB); break;                                                                                     \
    default: MXVL_ATTN_BEAMS(256, NB); break;                                                                                      \
  }
    switch (nb) {
      case 2: MXVL_ATTN_BEAMS_D(2); break;
      case 3: MXVL_ATTN_BEAMS_D(3); break;
      case 4: MXVL_ATTN_BEAMS_D(4); break;
      default: MXVL_ATTN_BEAMS_D(5); break;
    }
#undef MXVL_ATTN_BEAMS_D
#undef MXVL_ATTN_BEAMS
    return dec_check();
  }
  const int nw = attn_waves_env() ? attn_waves_env() : kAttnWaves;
  const int NG = nw * 64 / (a.D / 8);
  // ring depth (measured, profiles/r04_attn_row_ring_probe*.txt): 4 trips in flight while one workgroup per CU is all the grid asks
  // for -- 8 buy nothing at 96 workgroups (the kernel's floor there is launch + three dependent round trips), 2 lose 0.4 us on a
  // 231-position report -- and 2 beyond, where several workgroups per CU hide each other's round trips (10 % at 18+ rows)
  const size_t lds0 = sizeof(float) * ((size_t)3 * a.D + 2 * NG + (size_t)NG * a.D + a.max_len);
  int depth = (size_t)a.H * a.rows <= 256 ? 4 : 2;
  while (depth > 2 && lds0 + (size_t)nw * depth * 2048 > 160 * 1024) depth /= 2;
  const size_t lds = lds0 + (size_t)nw * depth * 2048;
  if (lds > 160 * 1024) return MXVL_ERR_UNSUPPORTED;
  const dim3 grid(a.H, a.rows), block(nw * 64);
#define MXVL_ATTN_ROW(DD) \
  do { \
    void (*kern)(const AttnArgs) = nw == 16 ? (depth == 8 ? decode_attn_kernel<E, DD, 16, 4> : depth == 4 ? decode_attn_kernel<E, DD, 16, 4> : decode_attn_kernel<E, DD, 16, 2>) \
                                            : (depth == 8 ? decode_attn_kernel<E, DD, 8, 8> : depth == 4 ? decode_attn_kernel<E, DD, 8, 4> : decode_attn_kernel<E, DD, 8, 2>); \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return MXVL_ERR_LAUNCH; \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a); \
  } while (0)
  switch (a.D) {
    case 64: MXVL_ATTN_ROW(64); break;
    case 128: MXVL_ATTN_ROW(128); break;
    case 256: MXVL_ATTN_ROW(256); break;
    default: return MXVL_ERR_UNSUPPORTED;
  }
#undef MXVL_ATTN_ROW
  return dec_check();
}

template <typename E>
static void launch_cross_attn(const CrossAttnArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  switch (a.D) {
    case 64: hipLaunchKernelGGL((decode_cross_attn_kernel<E, 64>), grid, block, lds, s, a); break;
    case 128: hipLaunchKernelGGL((decode_cross_attn_kernel<E, 128>), grid, block, lds, s, a); break;
    default: hipLaunchKernelGGL((decode_cross_attn_kernel<E, 256>), grid, block, lds, s, a); break;
  }
}

int decode_gemm_dispatch(const mxvl_gemv_desc* d, hipStream_t s);   // decode_gemm.hip: 9..80 rows on the matrix cores
int decode_gemm_plan(const mxvl_gemv_desc* d, int32_t* out);         // ... and which of its kernels would take a descriptor

}  // namespace mxvl

using namespace mxvl;

extern "C" {

/* The dispatch of mxvl_decode_gemv for a descriptor, nothing launched (no GPU needed): out[0] = 2 the per-row GEMV kernel (<= 8 rows,
 * k_splits == 0), 1 = decode_gemm_wide_kernel, 0 = the K-split matrix-core kernels; for 1: out[1..4] = waves per workgroup, weight
 * tiles per wave, ring stages, workgroups.  Same argument checks and error codes as the launch. */
int mxvl_decode_gemm_plan(const mxvl_gemv_desc* d, int32_t* out) {
  if (!d || !out || !d->x || !d->W || (!d->y && !d->split_acc)) return MXVL_ERR_NULL;
  out[0] = out[1] = out[2] = out[3] = out[4] = 0;
  if (d->rows > kMaxRows || d->k_splits != 0 || d->split_acc) return decode_gemm_plan(d, out);
  if (d->rows <= 0 || d->K <= 0 || d->N <= 0) return MXVL_ERR_SHAPE;
  out[0] = 2;
  return MXVL_OK;
}

int mxvl_decode_gemv(const mxvl_gemv_desc* d, void* hip_stream) {
  if (!d || !d->x || !d->W || (!d->y && !d->split_acc)) return MXVL_ERR_NULL;
  if (!decode_dtype_ok(d->dtype)) return MXVL_ERR_DTYPE;
  // k_splits != 0 asks for the matrix-core kernels at any row count (1 = no split); 0 = by row count
  if (d->rows > kMaxRows || d->k_splits != 0 || d->split_acc) return decode_gemm_dispatch(d, (hipStream_t)hip_stream);
  if (d->rows <= 0 || d->K <= 0 || d->N <= 0) return MXVL_ERR_SHAPE;
  if (d->K % 8 != 0) return MXVL_ERR_UNSUPPORTED;  // 16-byte weight loads
  if (d->norm_weight && d->K > 8192) return MXVL_ERR_UNSUPPORTED;  // fused RMSNorm keeps a whole row in registers
  if (d->swiglu && (!d->W2 || d->out_f32)) return MXVL_ERR_UNSUPPORTED;
  GemvArgs a;
  a.ablate = MXVL_ABL_ENV("MXVL_GEMV_ABLATE");
  a.rows = d->rows; a.K = d->K; a.N = d->N; a.swiglu = d->swiglu; a.out_f32 = d->out_f32; a.eps = d->eps;
  a.x = (const uint16_t*)d->x; a.g = (const uint16_t*)d->norm_weight; a.W = (const uint16_t*)d->W;
  a.W2 = (const uint16_t*)d->W2; a.bias = (const uint16_t*)d->bias; a.res = (const uint16_t*)d->residual; a.y = d->y;
  hipStream_t s = (hipStream_t)hip_stream;
  const int rc = decode_dtype(d->dtype) == MXVL_F16 ? launch_gemv_rows<EltF16>(a, s) : launch_gemv_rows<EltBf16>(a, s);
  return rc != MXVL_OK ? rc : dec_check();
}

int mxvl_decode_attn(const mxvl_decode_attn_desc* d, void* hip_stream) {
  if (!d || !d->qkv || !d->cos || !d->sin || !d->k_cache || !d->v_cache || !d->slot_table || !d->pos || !d->mask || !d->out)
    return MXVL_ERR_NULL;
  if (!decode_dtype_ok(d->dtype)) return MXVL_ERR_DTYPE;
  if (d->rows <= 0 || d->n_heads <= 0 || d->n_kv_heads <= 0 || d->n_heads % d->n_kv_heads != 0) return MXVL_ERR_SHAPE;
  if (d->head_dim != 64 && d->head_dim != 128 && d->head_dim != 256) return MXVL_ERR_UNSUPPORTED;
  AttnArgs a;
  a.rows = d->rows; a.H = d->n_heads; a.Hkv = d->n_kv_heads; a.D = d->head_dim; a.max_len = d->max_len;
  a.scale = d->scale; a.qkv = (const uint16_t*)d->qkv; a.cosv = (const float*)d->cos; a.sinv = (const float*)d->sin;
  a.kc = (uint16_t*)d->k_cache; a.vc = (uint16_t*)d->v_cache; a.slot = (const int*)d->slot_table;
  a.pos = (const int64_t*)d->pos; a.mask = (const int64_t*)d->mask; a.out = (uint16_t*)d->out;
  a.q_rope = (uint16_t*)d->q_rope;
  a.ablate = MXVL_ABL_ENV("MXVL_ATTN_ABLATE");
  const hipStream_t s = (hipStream_t)hip_stream;
  return decode_dtype(d->dtype) == MXVL_F16 ? launch_decode_attn<EltF16>(a, d->beams, s) : launch_decode_attn<EltBf16>(a, d->beams, s);
}

int mxvl_decode_prologue(const mxvl_decode_prologue_desc* d, void* hip_stream) {
  if (!d || !d->tok || !d->beam_src || !d->cur || !d->n_real || !d->embed || !d->cos_table || !d->sin_table || !d->slot_table ||
      !d->mask || !d->x || !d->cos || !d->sin || !d->pos)
    return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->rows > 80 || d->hidden <= 0 || d->hidden % 8 || d->max_len <= 0 || d->head_dim <= 0 || d->table_len <= 0)
    return MXVL_ERR_SHAPE;
  PrologueArgs a;
  a.rows = d->rows; a.hidden = d->hidden; a.max_len = d->max_len; a.D = d->head_dim; a.prompt_len = d->prompt_len; a.table_len = d->table_len;
  a.tok = (const int64_t*)d->tok; a.beam = (const int64_t*)d->beam_src; a.cur = (const int64_t*)d->cur; a.n_real = (const int64_t*)d->n_real;
  a.embed = (const uint16_t*)d->embed; a.cos_t = (const float*)d->cos_table; a.sin_t = (const float*)d->sin_table;
  a.slot = (int*)d->slot_table; a.mask = (int64_t*)d->mask; a.x = (uint16_t*)d->x; a.cosv = (float*)d->cos; a.sinv = (float*)d->sin;
  a.pos = (int64_t*)d->pos;
  hipStream_t s = (hipStream_t)hip_stream;
  // enough workgroups that the embedding rows (rows x hidden x 2 bytes) and the table columns are one trip per thread
  const int work = std::max(a.rows * a.hidden / 8, a.max_len);
  const dim3 grid(std::max(1, std::min(256, (work + 255) / 256)));   // (64 until round 5: at 80 rows the embedding rows took three trips, 35 us)
  if (a.rows <= 8) hipLaunchKernelGGL(decode_prologue_kernel<8>, grid, dim3(256), 0, s, a);
  else if (a.rows <= 32) hipLaunchKernelGGL(decode_prologue_kernel<32>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(decode_prologue_kernel<80>, grid, dim3(256), 0, s, a);
  return dec_check();
}

int mxvl_decode_cross_attn(const mxvl_decode_cross_attn_desc* d, void* hip_stream) {
  if (!d || !d->q_rope || !d->k || !d->v || !d->text_state || !d->gate_weight || !d->gate_bias || !d->out) return MXVL_ERR_NULL;
  if (!decode_dtype_ok(d->dtype)) return MXVL_ERR_DTYPE;
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
  if (decode_dtype(d->dtype) == MXVL_F16) launch_cross_attn<EltF16>(a, grid, block, lds, s);
  else launch_cross_attn<EltBf16>(a, grid, block, lds, s);
  return dec_check();
}

}  // extern "C"
