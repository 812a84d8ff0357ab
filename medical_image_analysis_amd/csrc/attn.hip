// attn.hip -- fused (flash-style) multi-head attention forward + backward on the gfx950 matrix cores.
//
// Replaces, for every attention on the MambaXray-VL hot path, the score-matrix implementations of the reference:
//   CXPMRG_Bench_MambaXray_VL/pretrain/models_pretrain.py:55-83   CrossAttention (softmax(q k^T * scale + mask) v) with the
//                                                                 block-lower-triangular mask of mask_generate (:395-400,
//                                                                 16-token clusters) -- 4 decoder blocks, 4080 tokens
//   HD_Xray_Pretrain_MAE/finetune/DP/models/vit.py:141-163        ViT / MAE self-attention (dense)
//   EMRRG/models/hybrid_decoder_layer.py:25-77, 392-457           causal self-attention (GQA) and text->image cross-attention
//                                                                 with a boolean key mask
// No (Lq x Lk) matrix ever reaches HBM: one workgroup owns 128 query rows (4 waves x 32), walks the key/value sequence in
// 64-key tiles staged in LDS, keeps the running (max, sum) of the online softmax and the output accumulator in registers.
//
// MFMA mapping (v_mfma_f32_32x32x16_{bf16,f16}; v_mfma_f32_32x32x2_f32 for fp32 inputs -- exact fp32, used by the fp32
// parity tests).  Everything is computed TRANSPOSED so that a lane owns one query (forward, dQ) or one key (dK/dV):
//   S^T[key][q]  = sum_d K[key][d] Q[q][d]      A = K rows from LDS (16-byte reads, XOR-swizzled), B = Q fragments in registers
//   O^T[d][q]   += sum_key V[key][d] P[q][key]  A = V^T through ds_read_b64_tr_b16 (hardware transpose read of the row-major
//                                               V tile), B = P straight from the S^T accumulator registers
// C/D layout of the 32x32 MFMA: lane l, register r holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31 -- so lane l and
// lane l^32 together hold the 32 scores of query (l&31) against a 32-key half tile; row max / sum are 16 in-register ops
// plus one exchange with lane l^32, and the rescale of O by exp2(m_old - m_new) is lane-local.  The B operand of the
// second product needs, for k-slot (l>>5, e), the score of "some" key: the accumulator registers are used as they are
// (slot e of step s <-> register 8s+e) and the V^T operand is simply read at the key rows those registers belong to,
// which removes the cvt/permlane shuffle a row-major P would need.
//
// Backward (no atomics, two kernels over the same primitives; P is recomputed from the saved log-sum-exp):
//   attn_bwd_dq_kernel   per query tile:  delta = rowsum(dO * O);  dS^T = P^T * (V dO^T - delta);  dQ^T += K^T dS^T
//   attn_bwd_dkv_kernel  per key tile (loops over the query heads of its KV group and their query tiles):
//                        dV^T += dO^T P,  dS = P * (dO V^T - delta),  dK^T += Q^T dS
// Masks: none, causal (key <= query + Lk - Lq), block-causal (key cluster <= query cluster), optional (batch, Lk) key
// mask and optional additive (Lq, Lk) fp32 bias (generic slow path).  Tiles wholly above the diagonal are never visited.
#include "mxvl_common.h"

namespace mxvl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct AttnArgs {
  int batch, H, Hkv, Lq, Lk, mask_mode, cluster;
  float scale;
  int64_t q_bs, q_hs, q_ts, k_bs, k_hs, k_ts, v_bs, v_hs, v_ts, o_bs, o_hs, o_ts;
  int64_t do_bs, do_hs, do_ts, dq_bs, dq_hs, dq_ts, dk_bs, dk_hs, dk_ts, dv_bs, dv_hs, dv_ts;
  const void *q, *k, *v, *o, *dout;
  void *out, *dq, *dk, *dv;
  float *lse, *delta;          // (batch, H, Lq)
  const uint8_t* kmask;        // (batch, Lk), nonzero = attend
  const float* bias;           // (Lq, Lk) additive, broadcast over batch and heads
  // attention dropout (training-time nn.Dropout on the probabilities: models_pretrain.py:62 attn_drop, hybrid_decoder_layer.py
  // attention_dropout, Blip2 Q-Former attention_probs_dropout_prob): probability (query i, key j) of head (b, h) is kept iff
  // attn_drop_hash(seed, b * H + h, i, j) >= drop_thresh and scaled by drop_scale = 1 / (1 - p).  0 = off.
  uint32_t drop_thresh, drop_seed;
  float drop_scale;
};

// counter-based: a pure function of (seed, head, query, key), so the forward, the dQ pass and the dK / dV pass -- which meet an element
// in different tiles and lanes -- draw the same bit, and a test can rebuild the whole mask on the host (flash_attention.dropout_keep_mask)
__device__ __forceinline__ uint32_t attn_drop_hash(uint32_t seed, uint32_t bh, uint32_t q, uint32_t k) {
  uint32_t x = seed ^ (bh * 0x9E3779B1u);
  x = (x ^ (q * 0x85EBCA77u)) * 0xC2B2AE3Du;
  x ^= k * 0x27D4EB2Fu;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;     // murmur3's finaliser
  return x;
}

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
constexpr float kNegInf = -__builtin_inff();

// ---- operand policies -------------------------------------------------------------------------------------------------
template <typename E> struct Pol16 {
  static constexpr int KSTEP = 16, ESZ = 2;
  typedef uint4 Frag;
  static __device__ __forceinline__ Frag zero() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
    if constexpr (sizeof(E) == 2 && __is_same(E, bf16_t))
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static constexpr int tile_bytes(int rows, int D) { return rows * D * 2; }
  // byte offset of 16-byte unit u of a row: XOR swizzle so that 32 rows read at one unit spread over the LDS banks
  // A 16-lane group of ds_read_b128 must hit 16 different (256-byte bank row offset, 16-byte unit) pairs.  A row is D * 2
  // bytes = UPR units, 16 / UPR rows share one 256-byte bank row: XOR the unit with the index of the row's bank row.
  // D = 64 (UPR = 8): the XOR key is (row bit 1) << 2 | (row bits 3..2).  ds_read_b128 is serviced in the lane groups
  // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) (MI355X_MICROARCH.md LDS table): the 8 same-parity rows of a group get 8
  // different keys, so a_row is conflict-free; and rows r, r + 2 (which share a 128-byte half of the bank row) differ in key
  // bit 2, so the four rows x 64 bytes of one ds_read_b64_tr_b16 half-wave (a_tr) tile the 256-byte bank row exactly -- with the
  // plain key (row >> 1) & 7 those two rows read the same four units: 25 % of the forward's LDS cycles were conflicts
  // (profiles/r02_attn_pmc.txt).
  template <int D> static __device__ __forceinline__ int unit_off(int row, int u) {
    constexpr int UPR = D / 8, SH = UPR >= 16 ? 0 : (UPR == 8 ? 1 : 2);
    // D = 128 (UPR = 16, a row is a whole bank row): rows r .. r + 3 of a transposed read must land in four different 64-byte
    // blocks -> row bits 1..0 go to key bits 3..2; the 16 rows of a b128 group still get 16 different keys.
    const int key = UPR == 8 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))
                  : UPR == 16 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> SH) & (UPR - 1));
    return row * (D * 2) + ((u ^ key) << 4);
  }
  // A operand, row-major rows: lane (row, hi) takes elements [16 ks + 8 hi, +8) of its row
  template <int D> static __device__ __forceinline__ Frag a_row(const char* tile, int row, int ks, int hi) {
    return *(const uint4*)(tile + unit_off<D>(row, 2 * ks + hi));
  }
  // A operand, TRANSPOSED: lane (c = lane & 31, hi) takes column dbase + c of the 8 rows that accumulator registers
  // 8 ks .. 8 ks + 7 of lane half `hi` stand for: rbase + 16 ks + 4 hi + {0..3} and the same + 8.
  // ds_read_b64_tr_b16: within each 16-lane group lane t' supplies the address of row (t' >> 2), columns 4 (t' & 3) .. + 3 of
  // a [4][16] block and receives column t' of its 4 rows.
  template <int D> static __device__ __forceinline__ Frag a_tr(const char* tile, int rbase, int ks, int hi, int dbase, int lane) {
    const int t = lane & 15, g = (lane >> 4) & 1;
    const int col = dbase + 16 * g + 4 * (t & 3);
    const int r0 = rbase + 16 * ks + 4 * hi + (t >> 2);
    const int inner = (col & 7) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(tile + unit_off<D>(r0, col >> 3) + inner));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(tile + unit_off<D>(r0 + 8, col >> 3) + inner));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi4);
    return make_uint4(a.x, a.y, b.x, b.y);
  }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (__is_same(E, bf16_t)) return cvt_pk_bf16(a, b);
    else {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      return __builtin_bit_cast(uint32_t, h2{(_Float16)a, (_Float16)b});
    }
  }
  // B operand from accumulator registers 8 ks .. 8 ks + 7
  static __device__ __forceinline__ Frag b_from_acc(const f32x16& a, int ks) {
    const int o = 8 * ks;
    return make_uint4(pack2(a[o], a[o + 1]), pack2(a[o + 2], a[o + 3]), pack2(a[o + 4], a[o + 5]), pack2(a[o + 6], a[o + 7]));
  }
  static __device__ __forceinline__ Frag ldg(const E* row, int ks, int hi) { return *(const uint4*)(row + 16 * ks + 8 * hi); }
  static __device__ __forceinline__ float el(uint32_t w, int i) {
    if constexpr (__is_same(E, bf16_t)) return __builtin_bit_cast(float, i ? (w & 0xffff0000u) : (w << 16));
    else {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 v = __builtin_bit_cast(h2, w);
      return (float)(i ? v.y : v.x);
    }
  }
  static __device__ __forceinline__ float dot(Frag a, Frag b) {
    const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s = fmaf(el(x[i], 0), el(y[i], 0), fmaf(el(x[i], 1), el(y[i], 1), s));
    return s;
  }
  // global [rows x D] (row stride ts elements) -> registers -> swizzled LDS tile; rows >= nvalid are zero-filled.  Split in two
  // so that the loads of tile t + 1 are in flight while tile t is being multiplied (the write happens after it).
  template <int D, int ROWS, int NT> struct Stage {
    static constexpr int UPR = D / 8, N = ROWS * UPR / NT;
    uint4 v[N];
    int nv;
    // Unconditional loads (rows past the end re-read the last valid row, nvalid >= 1) and the zero-fill at STORE time: behind
    // `if (row < nvalid) v = load` hipcc waits for the load at the join of the branch, i.e. at once -- the tile that was meant to
    // be in flight under the multiplications of the previous one was waited for before they started.
    __device__ __forceinline__ void load(const E* base, int64_t ts, int nvalid, int tid) {
      nv = nvalid;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT, row = c / UPR, u = c % UPR;
        const int rc = row < nvalid ? row : nvalid - 1;
        v[i] = *(const uint4*)(base + (int64_t)rc * ts + u * 8);
      }
    }
    __device__ __forceinline__ void store(char* tile, int tid) const {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT, row = c / UPR, u = c % UPR;
        *(uint4*)(tile + unit_off<D>(row, u)) = row < nv ? v[i] : make_uint4(0, 0, 0, 0);
      }
    }
  };
  // 4 consecutive output elements
  static __device__ __forceinline__ void st4(E* p, float a, float b, float c, float d) { *(uint2*)p = make_uint2(pack2(a, b), pack2(c, d)); }
};

struct Pol32 {
  static constexpr int KSTEP = 2, ESZ = 4;
  typedef float Frag;
  static __device__ __forceinline__ Frag zero() { return 0.f; }
  static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  static constexpr int tile_bytes(int rows, int D) { return rows * (D + 1) * 4; }   // +1 float: column reads hit 32 banks
  template <int D> static __device__ __forceinline__ Frag a_row(const char* tile, int row, int ks, int hi) {
    return ((const float*)tile)[row * (D + 1) + 2 * ks + hi];
  }
  template <int D> static __device__ __forceinline__ Frag a_tr(const char* tile, int rbase, int ks, int hi, int dbase, int lane) {
    return ((const float*)tile)[(rbase + crow(ks, hi)) * (D + 1) + dbase + (lane & 31)];
  }
  static __device__ __forceinline__ Frag b_from_acc(const f32x16& a, int ks) { return a[ks]; }
  static __device__ __forceinline__ Frag ldg(const float* row, int ks, int hi) { return row[2 * ks + hi]; }
  static __device__ __forceinline__ float dot(Frag a, Frag b) { return a * b; }
  template <int D, int ROWS, int NT> struct Stage {
    static constexpr int QPR = D / 4, N = ROWS * QPR / NT;
    float4 v[N];
    int nv;
    __device__ __forceinline__ void load(const float* base, int64_t ts, int nvalid, int tid) {     // see Pol16::Stage
      nv = nvalid;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT, row = c / QPR, u = c % QPR;
        const int rc = row < nvalid ? row : nvalid - 1;
        v[i] = *(const float4*)(base + (int64_t)rc * ts + u * 4);
      }
    }
    __device__ __forceinline__ void store(char* tile, int tid) const {
      float* t = (float*)tile;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT, row = c / QPR, u = c % QPR;
        float* d = t + row * (D + 1) + u * 4;
        const float4 w = row < nv ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
      }
    }
  };
  static __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) { *(float4*)p = make_float4(a, b, c, d); }
};

template <typename E> struct PolOf { typedef Pol16<E> type; };
template <> struct PolOf<float> { typedef Pol32 type; };

// Reductions over the lane pair (l, l ^ 32) that shares an MFMA column: v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of its second -- one VALU op, no LDS round trip.  Inline asm: this toolchain's
// __builtin_amdgcn_permlane32_swap returns its FIRST result in both vector elements (clang emits extractvalue 0 twice).
// The two v_nop-equivalent wait states the swap needs after a VALU write of its operands are inside the string.
__device__ __forceinline__ void swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float pair_max(float v) {
  float a = v, b = v;
  swap32(a, b);      // a = {lo, lo}, b = {hi, hi}
  return fmaxf(a, b);
}
__device__ __forceinline__ float pair_sum(float v) {
  float a = v, b = v;
  swap32(a, b);
  return a + b;
}

// key limit of a query row: keys [0, klim) may be attended (before the optional key mask / bias)
__device__ __forceinline__ int key_limit(const AttnArgs& p, int qrow) {
  if (qrow >= p.Lq) return 0;
  int lim = p.Lk;
  if (p.mask_mode == 1) lim = qrow + (p.Lk - p.Lq) + 1;
  else if (p.mask_mode == 2) lim = (qrow / p.cluster + 1) * p.cluster;
  return lim < 0 ? 0 : (lim > p.Lk ? p.Lk : lim);
}

constexpr int kKT = 64;   // keys per staged tile (forward, dQ) / queries per staged tile (dK/dV)

// ---- forward / dQ -------------------------------------------------------------------------------------------------------
// DQ = false: O, lse.   DQ = true: the dQ pass of the backward (same walk over the key tiles).
// K / V tiles are double-buffered in LDS: the global loads of tile t + 1 are issued before tile t is multiplied and land in
// registers; they are written to the other buffer after the products, one barrier per tile.
// EXTRA = a key mask and / or an additive bias is present: those per-element loads and branches live in their own
// instantiation (inside the plain kernel they serialised the softmax: a branch + dependent load per score).
template <typename E, int D, int NW, bool DQ, bool EXTRA>
__global__ __launch_bounds__(NW * 64, (D >= 128 || sizeof(E) == 4) ? 1 : 2) void attn_q_kernel(const AttnArgs p) {
  typedef typename PolOf<E>::type Pol;
  typedef typename Pol::Frag Frag;
  constexpr int QT = NW * 32, KT = kKT, NKD = D / Pol::KSTEP, NKR = 32 / Pol::KSTEP, NDT = D / 32, NT = NW * 64;
  constexpr int TB = Pol::tile_bytes(KT, D);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][K | V] tiles, then [buf][KT] key-mask addends
  float* sMaskAll = (float*)(smem + 4 * TB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  // grid = (heads x batch, query tiles): the dispatcher walks x fastest, so with a causal mask ALL workgroups of the longest
  // query tile start first and the shortest fill the tail (longest-processing-time order over the whole launch; ordering only
  // inside a head left the chip half empty at the end: 675 -> 470 us for the 64-query forward, profiles/r03_attn_fwd64.txt)
  const int b = (int)blockIdx.x / p.H, h = (int)blockIdx.x % p.H, hk = h / (p.H / p.Hkv);
  const int qt = p.mask_mode ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int q0 = qt * QT, qrow = q0 + wave * 32 + j;
  const bool qv = qrow < p.Lq;
  const int64_t qr = qv ? qrow : 0;
  const float c = p.scale * kLog2e;

  const int klim = key_limit(p, qrow);
  const int qlast = (q0 + QT < p.Lq ? q0 + QT : p.Lq) - 1;
  const int kend = key_limit(p, qlast);            // limits grow with the row index in every mask mode
  const int kfull = key_limit(p, q0);              // keys below this are visible to every row of the workgroup
  const int ntiles = (kend + KT - 1) / KT;

  const E* kb = (const E*)p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs;
  const E* vb = (const E*)p.v + (int64_t)b * p.v_bs + (int64_t)hk * p.v_hs;
  typename Pol::template Stage<D, KT, NT> stK, stV;
  int mraw = 0, mkey = 0;
  // (the key-mask byte: an unconditional load from a clamped position -- of the K tensor itself when there is no mask -- decided
  // at commit; see Pol16::Stage for why no load of the prefetch may sit behind a condition)
  const uint8_t* const mbase = p.kmask ? p.kmask + (int64_t)b * p.Lk : (const uint8_t*)kb;
  auto fetch = [&](int kt) {
    const int k0 = kt * KT;
    stK.load(kb + (int64_t)k0 * p.k_ts, p.k_ts, p.Lk - k0, tid);
    stV.load(vb + (int64_t)k0 * p.v_ts, p.v_ts, p.Lk - k0, tid);
    if constexpr (EXTRA) {
      mkey = k0 + (tid & (KT - 1));
      mraw = mbase[mkey < p.Lk ? mkey : p.Lk - 1];
    }
  };
  auto commit = [&](int buf) {
    stK.store(smem + (2 * buf) * TB, tid);
    stV.store(smem + (2 * buf + 1) * TB, tid);
    if constexpr (EXTRA) {
      if (tid < KT) sMaskAll[buf * KT + tid] = (mkey < p.Lk && (!p.kmask || mraw)) ? 0.f : kNegInf;
    }
  };
  if (ntiles > 0) fetch(0);

  const E* qp = (const E*)p.q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs + qr * p.q_ts;
  Frag qf[NKD];
#pragma unroll
  for (int ks = 0; ks < NKD; ++ks) qf[ks] = Pol::ldg(qp, ks, hi);     // unconditional (qr is clamped to a valid row): all the
  Frag dof[DQ ? NKD : 1];                                              // loads of this prologue are in flight together; rows
  float lse = 0.f, delta = 0.f;                                        // without a query are zeroed after the pin below
  if constexpr (DQ) {
    const E* dop = (const E*)p.dout + (int64_t)b * p.do_bs + (int64_t)h * p.do_hs + qr * p.do_ts;
    const E* op = (const E*)p.o + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + qr * p.o_ts;
    const int64_t ro = ((int64_t)b * p.H + h) * p.Lq + qr;
    Frag of[NKD];
#pragma unroll
    for (int ks = 0; ks < NKD; ++ks) {
      dof[ks] = Pol::ldg(dop, ks, hi);
      of[ks] = Pol::ldg(op, ks, hi);
    }
    lse = p.lse[ro];
#pragma unroll
    for (int ks = 0; ks < NKD; ++ks) delta += Pol::dot(dof[ks], of[ks]);
    delta = pair_sum(delta);
    if (qv && hi == 0) p.delta[ro] = delta;
  }

  f32x16 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  float m = kNegInf, l = 0.f;
  // The Q-side fragments are consumed HERE, before the loop: s_waitcnt insertion is static, and with their loads still pending
  // at the loop header hipcc put `s_waitcnt vmcnt(0)` in front of the first MFMA of the loop BODY -- executed every iteration,
  // right behind the loads of tile kt + 1 (head_dim 128 causal: 321 -> 302 us forward, 924 -> 868 us backward; head_dim 32: 94 -> 79 us)
#pragma unroll
  for (int ks = 0; ks < NKD; ++ks) {
    if constexpr (sizeof(E) == 2) {
      asm volatile("" : "+v"(qf[ks].x), "+v"(qf[ks].y), "+v"(qf[ks].z), "+v"(qf[ks].w));
      if constexpr (DQ) asm volatile("" : "+v"(dof[ks].x), "+v"(dof[ks].y), "+v"(dof[ks].z), "+v"(dof[ks].w));
    } else {
      asm volatile("" : "+v"(qf[ks]));
      if constexpr (DQ) asm volatile("" : "+v"(dof[ks]));
    }
  }
  if constexpr (DQ) asm volatile("" : "+v"(lse), "+v"(delta));
  if (!qv) {
#pragma unroll
    for (int ks = 0; ks < NKD; ++ks) {
      qf[ks] = Pol::zero();
      if constexpr (DQ) dof[ks] = Pol::zero();
    }
    lse = __builtin_inff();
    delta = 0.f;
  }

  if (ntiles > 0) commit(0);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KT, buf = kt & 1;
    const char* sK = smem + (2 * buf) * TB;
    const char* sV = smem + (2 * buf + 1) * TB;
    const float* sMask = sMaskAll + buf * KT;
    const bool more = kt + 1 < ntiles;
    if (more) fetch(kt + 1);

    // every LDS operand of a phase is requested before its first MFMA (one latency per phase, not one per MFMA), and the
    // operands of the SECOND product are requested before the softmax arithmetic, which hides them completely
    f32x16 s[2];
    {
      Frag ka[2][NKD];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) ka[h2][ks] = Pol::template a_row<D>(sK, 32 * h2 + j, ks, hi);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[h2][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) s[h2] = Pol::mma(ka[h2][ks], qf[ks], s[h2]);
      }
    }
    // forward: the V^T operands are requested before the softmax arithmetic, which hides their LDS latency completely (the
    // dQ pass is register-bound -- Q and dO fragments both live -- and reads its K^T operands where it uses them: measured)
    Frag ta[DQ ? 1 : 2][DQ ? 1 : NKR][DQ ? 1 : NDT];
    if constexpr (!DQ) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int ks = 0; ks < NKR; ++ks)
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) ta[h2][ks][dt] = Pol::template a_tr<D>(sV, 32 * h2, ks, hi, 32 * dt, lane);
    }
    // logits in the log2 domain
    const bool masked = EXTRA || (k0 + KT > kfull);
    if (masked) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = 32 * h2 + crow(r, hi), key = k0 + kl;
          float x = s[h2][r] * c;
          if constexpr (EXTRA) {
            if (p.bias && qv && key < p.Lk) x = fmaf(p.bias[qr * p.Lk + key], kLog2e, x);
            x += sMask[kl];
          }
          s[h2][r] = key < klim ? x : kNegInf;
        }
    } else {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[h2][r] *= c;
    }

    if constexpr (!DQ) {
      float mx = kNegInf;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[h2][r]);
      mx = pair_max(mx);
      const float mn = fmaxf(m, mx);
      const float mu = (mn == kNegInf) ? 0.f : mn;
      const float alpha = fast_exp2(m - mu);
      float rs = 0.f;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = fast_exp2(s[h2][r] - mu);
          rs += pv;                                          // the normaliser sums the probabilities BEFORE dropout
          if constexpr (EXTRA) {
            if (p.drop_thresh) {
              const uint32_t key = (uint32_t)(k0 + 32 * h2 + crow(r, hi));
              pv = attn_drop_hash(p.drop_seed, (uint32_t)blockIdx.x, (uint32_t)qrow, key) >= p.drop_thresh ? pv * p.drop_scale : 0.f;
            }
          }
          s[h2][r] = pv;
        }
      rs = pair_sum(rs);
      l = fmaf(l, alpha, rs);
      m = mn;
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
      }
      // O^T += V^T P^T
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int ks = 0; ks < NKR; ++ks) {
          const Frag pb = Pol::b_from_acc(s[h2], ks);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) acc[dt] = Pol::mma(ta[h2][ks][dt], pb, acc[dt]);
        }
    } else {
      // P^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        f32x16 dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) dp = Pol::mma(Pol::template a_row<D>(sV, 32 * h2 + j, ks, hi), dof[ks], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(s[h2][r] - lse);     // masked: exp2(-inf) = 0; rows without keys: lse = +inf
          float dpr = dp[r];
          if constexpr (EXTRA) {
            if (p.drop_thresh) {                           // d(dropped P) -> dP: the kept elements' scale, zero elsewhere
              const uint32_t key = (uint32_t)(k0 + 32 * h2 + crow(r, hi));
              dpr = attn_drop_hash(p.drop_seed, (uint32_t)blockIdx.x, (uint32_t)qrow, key) >= p.drop_thresh ? dpr * p.drop_scale : 0.f;
            }
          }
          s[h2][r] = pv * (dpr - delta);
        }
#pragma unroll
        for (int ks = 0; ks < NKR; ++ks) {
          const Frag db = Pol::b_from_acc(s[h2], ks);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
            acc[dt] = Pol::mma(Pol::template a_tr<D>(sK, 32 * h2, ks, hi, 32 * dt, lane), db, acc[dt]);
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
  }

  if (!qv) return;
  if constexpr (!DQ) {
    const float inv = l > 0.f ? fast_rcp(l) : 0.f;
    E* op = (E*)p.out + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + qr * p.o_ts;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        Pol::st4(op + 32 * dt + 8 * g + 4 * hi, acc[dt][4 * g] * inv, acc[dt][4 * g + 1] * inv, acc[dt][4 * g + 2] * inv, acc[dt][4 * g + 3] * inv);
    if (p.lse && hi == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + qr] = l > 0.f ? m + fast_log2(l) : __builtin_inff();
  } else {
    E* dqp = (E*)p.dq + (int64_t)b * p.dq_bs + (int64_t)h * p.dq_hs + qr * p.dq_ts;
    const float sc = p.scale;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        Pol::st4(dqp + 32 * dt + 8 * g + 4 * hi, acc[dt][4 * g] * sc, acc[dt][4 * g + 1] * sc, acc[dt][4 * g + 2] * sc, acc[dt][4 * g + 3] * sc);
  }
}

// ---- forward, head_dim 64, 16-bit, no key mask / bias: 64 queries per wave ----------------------------------------------------
// attn_q_kernel gives a wave 32 queries, so every wave reads the whole 64-key K and V tile from LDS (16 KB) for 16 MFMAs: 8 waves
// keep the CU's LDS port busy 1760 cycles per round against 1024 MFMA cycles -- the LDS port, not the matrix core, was the first
// wall (13-14 % of the MFMA peak).  Here a wave owns TWO 32-query blocks: every K fragment (ds_read_b128) and every
// V^T fragment (ds_read_b64_tr_b16) feeds two MFMAs, 32 MFMAs per 16 KB of LDS reads.
//   * workgroup = 4 waves = 256 queries, 64-key tiles, THREE LDS stages of [K | V] (48 KB, two workgroups per CU)
//   * K / V tiles arrive by LDS-DMA (global_load_lds_dwordx4 from inline asm, as in gemm_swiglu.hip): no staging registers, no
//     ds_write pass; the XOR swizzle of unit_off<64> is applied on the DMA's SOURCE address (the DMA writes LDS linearly).
//     Tile t + 2 is requested right after the barrier of tile t; the only waits are one counted vmcnt + one barrier per tile.
//     Key rows past Lk are fetched from row Lk - 1 (finite values under a zero probability), never left as stale LDS bits.
//   * softmax: the scale rides in the exponent's FMA (exp2(s * c - m)), the row maximum is taken on the raw scores
//   * a wave skips the arithmetic of a diagonal tile none of its 64 queries can see (it still keeps the barriers)
template <typename E>
__global__ __launch_bounds__(256, 2) void attn_fwd64_kernel(const AttnArgs p) {
  typedef Pol16<E> Pol;
  typedef uint4 Frag;
  constexpr int D = 64, NW = 4, QT = NW * 64, KT = kKT, NKD = 4, NKR = 2, NDT = 2, TB = KT * D * 2, NST = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // grid = (heads x batch, query tiles): the dispatcher walks x fastest, so ALL workgroups of the longest query tile start
  // first and the shortest ones fill the tail (longest-processing-time order over the whole launch, not per head)
  const int b = (int)blockIdx.x / p.H, h = (int)blockIdx.x % p.H, hk = h / (p.H / p.Hkv);
  const int qt = p.mask_mode ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int q0 = qt * QT, wq0 = q0 + wave * 64;
  const float c = p.scale * kLog2e;

  int qrow[2], klim[2];
  bool qv[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = wq0 + 32 * qb + j;
    qv[qb] = qrow[qb] < p.Lq;
    klim[qb] = key_limit(p, qrow[qb]);
  }
  const int qlast = (q0 + QT < p.Lq ? q0 + QT : p.Lq) - 1;
  const int kend = key_limit(p, qlast);            // limits grow with the row index in every mask mode
  const int kfull = key_limit(p, q0);              // keys below this are visible to every row of the workgroup
  const int wlast = (wq0 + 64 < p.Lq ? wq0 + 64 : p.Lq) - 1;
  const int wkend = wq0 < p.Lq ? key_limit(p, wlast) : 0;     // keys at or beyond this are invisible to the whole wave
  const int ntiles = (kend + KT - 1) / KT;

  // ---- LDS-DMA: this wave moves rows 16 wave .. 16 wave + 15 of the K tile and of the V tile (2 x 1 KB each) ----------------
  const E* kb = (const E*)p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs;
  const E* vb = (const E*)p.v + (int64_t)b * p.v_bs + (int64_t)hk * p.v_hs;
  // address = (tile base, wave-uniform, in SGPRs) + (per-lane 32-bit byte offset, the same for every full tile)
  auto dma_row = [&](int i) { return 16 * wave + 8 * i + (lane >> 3); };
  auto dma_unit = [&](int row) {                      // element offset of the 16-byte unit this lane fetches (Pol16::unit_off<64>)
    const int key = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return ((lane & 7) ^ key) * 8;
  };
  unsigned offK[2], offV[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = dma_row(i), unit = dma_unit(row);
    offK[i] = (unsigned)(((int64_t)row * p.k_ts + unit) * 2);
    offV[i] = (unsigned)(((int64_t)row * p.v_ts + unit) * 2);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
  auto issue = [&](int kt) {
    const int k0 = kt * KT;
    const E* gk = kb + (int64_t)k0 * p.k_ts;
    const E* gv = vb + (int64_t)k0 * p.v_ts;
    unsigned o[4] = {offK[0], offK[1], offV[0], offV[1]};
    if (k0 + KT > p.Lk) {                            // wave-uniform: the ragged last tile re-reads row Lk - 1 for the missing rows
#pragma unroll
      for (int i = 0; i < 2; ++i) {      // (recomputed here: the row / unit of a lane are not worth two registers each in the loop)
        const int row = dma_row(i), unit = dma_unit(row);
        const int r = k0 + row < p.Lk ? row : p.Lk - 1 - k0;
        o[i] = (unsigned)(((int64_t)r * p.k_ts + unit) * 2);
        o[2 + i] = (unsigned)(((int64_t)r * p.v_ts + unit) * 2);
      }
    }
    const unsigned dst = lds0 + (unsigned)((kt % NST) * 2 * TB + wave * 2048);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[gk]\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[gk]\n\t"
        "s_add_u32 m0, m0, 0x1c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o2], %[gv]\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o3], %[gv]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [dst] "s"(dst), [gk] "s"(gk), [gv] "s"(gv), [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3])
        : "memory", "scc");
  };
  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);

  Frag qf[2][NKD];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const E* qp = (const E*)p.q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs + (int64_t)(qv[qb] ? qrow[qb] : 0) * p.q_ts;
#pragma unroll
    for (int ks = 0; ks < NKD; ++ks) qf[qb][ks] = qv[qb] ? Pol::ldg(qp, ks, hi) : Pol::zero();
  }
  f32x16 acc[2][NDT];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[qb][dt][r] = 0.f;
  float m[2] = {kNegInf, kNegInf}, l[2] = {0.f, 0.f};
  // NOTE (profiles/r03_attn_wait_ab.txt): with the Q fragment loads still pending at the loop header, hipcc puts a
  // `s_waitcnt vmcnt(0)` in front of the first MFMA of the loop body (s_waitcnt insertion is static), i.e. behind the DMA request
  // of tile kt + 2.  Consuming the fragments before the loop removes that wait -- and measured 2 % SLOWER here and in the dQ pass
  // (374 vs 366 us; two co-resident workgroups cover each other's wait and stay staggered), while the same fix gained 6-19 % in
  // attn_q_kernel and the dK/dV pass.  Left as the compiler emits it.

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KT;
    // tile kt has landed (this wave's share: at most tile kt + 1's four transfers may still be in flight), then everybody's has,
    // and everybody is done reading the stage tile kt + 2 will overwrite
    if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < ntiles) issue(kt + 2);
    if (k0 >= wkend) continue;                       // wave-uniform: nothing in this tile is visible to these 64 queries
    const char* sK = smem + (kt % NST) * 2 * TB;
    const char* sV = sK + TB;

    f32x16 s[2][2];                                  // [key half][query block]
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      Frag ka[NKD];
#pragma unroll
      for (int ks = 0; ks < NKD; ++ks) ka[ks] = Pol::template a_row<D>(sK, 32 * h2 + j, ks, hi);
#pragma unroll
      for (int ks = 0; ks < NKD; ++ks)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)   // first step: C = the inline constant 0, not 16 zeroed registers per accumulator
          s[h2][qb] = Pol::mma(ka[ks], qf[qb][ks], ks == 0 ? zero16 : s[h2][qb]);
    }
    const bool masked = k0 + KT > kfull;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (masked) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = k0 + 32 * h2 + crow(r, hi);
            s[h2][qb][r] = key < klim[qb] ? s[h2][qb][r] : kNegInf;
          }
      }
      float mx = kNegInf;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[h2][qb][r]);
      mx = pair_max(mx) * c;                         // c > 0: the maximum commutes with the scale
      const float mn = fmaxf(m[qb], mx);
      const float mu = (mn == kNegInf) ? 0.f : mn;
      const float alpha = fast_exp2(m[qb] - mu);
      float rs = 0.f;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(fmaf(s[h2][qb][r], c, -mu));
          s[h2][qb][r] = pv;
          rs += pv;
        }
      rs = pair_sum(rs);
      l[qb] = fmaf(l[qb], alpha, rs);
      m[qb] = mn;
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[qb][dt][r] *= alpha;
      }
    }
    // O^T += V^T P^T; the V^T operands of a key half are requested together, one half ahead of their MFMAs (holding all 32
    // registers of them across the softmax pushed the loop into scratch spills, and a spill reload waits on vmcnt = on the DMA)
    Frag ta[2][NKR][NDT];
#pragma unroll
    for (int ks = 0; ks < NKR; ++ks)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) ta[0][ks][dt] = Pol::template a_tr<D>(sV, 0, ks, hi, 32 * dt, lane);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      if (h2 == 0) {
#pragma unroll
        for (int ks = 0; ks < NKR; ++ks)
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) ta[1][ks][dt] = Pol::template a_tr<D>(sV, 32, ks, hi, 32 * dt, lane);
      }
#pragma unroll
      for (int ks = 0; ks < NKR; ++ks) {
        Frag pb[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) pb[qb] = Pol::b_from_acc(s[h2][qb], ks);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) acc[qb][dt] = Pol::mma(ta[h2][ks][dt], pb[qb], acc[qb][dt]);
      }
    }
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    if (!qv[qb]) continue;
    const int64_t qr = qrow[qb];
    const float inv = l[qb] > 0.f ? fast_rcp(l[qb]) : 0.f;
    E* op = (E*)p.out + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + qr * p.o_ts;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        Pol::st4(op + 32 * dt + 8 * g + 4 * hi, acc[qb][dt][4 * g] * inv, acc[qb][dt][4 * g + 1] * inv, acc[qb][dt][4 * g + 2] * inv,
                 acc[qb][dt][4 * g + 3] * inv);
    if (p.lse && hi == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + qr] = l[qb] > 0.f ? m[qb] + fast_log2(l[qb]) : __builtin_inff();
  }
}

// ---- dQ pass, head_dim 64, 16-bit, no key mask / bias: the same 64-queries-per-wave / LDS-DMA structure as attn_fwd64_kernel.
// Per 64-key tile and wave 48 MFMAs (S^T, dP^T, dQ^T for two query blocks) against 24 KB of LDS reads (K rows, V rows, K^T);
// attn_q_kernel<DQ> needs 40 KB for 24.  Also writes delta = rowsum(dO * O) for the dK/dV kernel.
template <typename E>
__global__ __launch_bounds__(256, 2) void attn_dq64_kernel(const AttnArgs p) {
  typedef Pol16<E> Pol;
  typedef uint4 Frag;
  constexpr int D = 64, NW = 4, QT = NW * 64, KT = kKT, NKD = 4, NKR = 2, NDT = 2, TB = KT * D * 2, NST = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // grid = (heads x batch, query tiles): the dispatcher walks x fastest, so ALL workgroups of the longest query tile start
  // first and the shortest ones fill the tail (longest-processing-time order over the whole launch, not per head)
  const int b = (int)blockIdx.x / p.H, h = (int)blockIdx.x % p.H, hk = h / (p.H / p.Hkv);
  const int qt = p.mask_mode ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int q0 = qt * QT, wq0 = q0 + wave * 64;
  const float c = p.scale * kLog2e;

  int qrow[2], klim[2];
  bool qv[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = wq0 + 32 * qb + j;
    qv[qb] = qrow[qb] < p.Lq;
    klim[qb] = key_limit(p, qrow[qb]);
  }
  const int qlast = (q0 + QT < p.Lq ? q0 + QT : p.Lq) - 1;
  const int kend = key_limit(p, qlast);            // limits grow with the row index in every mask mode
  const int kfull = key_limit(p, q0);              // keys below this are visible to every row of the workgroup
  const int wlast = (wq0 + 64 < p.Lq ? wq0 + 64 : p.Lq) - 1;
  const int wkend = wq0 < p.Lq ? key_limit(p, wlast) : 0;     // keys at or beyond this are invisible to the whole wave
  const int ntiles = (kend + KT - 1) / KT;

  // ---- LDS-DMA: this wave moves rows 16 wave .. 16 wave + 15 of the K tile and of the V tile (2 x 1 KB each) ----------------
  const E* kb = (const E*)p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs;
  const E* vb = (const E*)p.v + (int64_t)b * p.v_bs + (int64_t)hk * p.v_hs;
  // address = (tile base, wave-uniform, in SGPRs) + (per-lane 32-bit byte offset, the same for every full tile)
  auto dma_row = [&](int i) { return 16 * wave + 8 * i + (lane >> 3); };
  auto dma_unit = [&](int row) {                      // element offset of the 16-byte unit this lane fetches (Pol16::unit_off<64>)
    const int key = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return ((lane & 7) ^ key) * 8;
  };
  unsigned offK[2], offV[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = dma_row(i), unit = dma_unit(row);
    offK[i] = (unsigned)(((int64_t)row * p.k_ts + unit) * 2);
    offV[i] = (unsigned)(((int64_t)row * p.v_ts + unit) * 2);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
  auto issue = [&](int kt) {
    const int k0 = kt * KT;
    const E* gk = kb + (int64_t)k0 * p.k_ts;
    const E* gv = vb + (int64_t)k0 * p.v_ts;
    unsigned o[4] = {offK[0], offK[1], offV[0], offV[1]};
    if (k0 + KT > p.Lk) {                            // wave-uniform: the ragged last tile re-reads row Lk - 1 for the missing rows
      int lane_r = threadIdx.x;          // (re-derived behind an opaque move: kept from the top, the lane index of this once-per-kernel
      asm volatile("" : "+v"(lane_r));   //  branch was a VGPR stored before the key loop and reloaded here -- scratch)
      lane_r &= 63;
#pragma unroll
      for (int i = 0; i < 2; ++i) {      // (recomputed here: the row / unit of a lane are not worth two registers each in the loop)
        const int row = 16 * wave + 8 * i + (lane_r >> 3);
        const int unit = ((lane_r & 7) ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) * 8;
        const int r = k0 + row < p.Lk ? row : p.Lk - 1 - k0;
        o[i] = (unsigned)(((int64_t)r * p.k_ts + unit) * 2);
        o[2 + i] = (unsigned)(((int64_t)r * p.v_ts + unit) * 2);
      }
    }
    const unsigned dst = lds0 + (unsigned)((kt % NST) * 2 * TB + wave * 2048);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[gk]\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[gk]\n\t"
        "s_add_u32 m0, m0, 0x1c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o2], %[gv]\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o3], %[gv]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [dst] "s"(dst), [gk] "s"(gk), [gv] "s"(gv), [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3])
        : "memory", "scc");
  };
  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);

  Frag qf[2][NKD], dof[2][NKD];
  float lse[2], delta[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int64_t qr = qv[qb] ? qrow[qb] : 0;
    const E* qp = (const E*)p.q + (int64_t)b * p.q_bs + (int64_t)h * p.q_hs + qr * p.q_ts;
    const E* dop = (const E*)p.dout + (int64_t)b * p.do_bs + (int64_t)h * p.do_hs + qr * p.do_ts;
    const E* op = (const E*)p.o + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + qr * p.o_ts;
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKD; ++ks) {
      qf[qb][ks] = qv[qb] ? Pol::ldg(qp, ks, hi) : Pol::zero();
      dof[qb][ks] = qv[qb] ? Pol::ldg(dop, ks, hi) : Pol::zero();
      const Frag of = qv[qb] ? Pol::ldg(op, ks, hi) : Pol::zero();
      dl += Pol::dot(dof[qb][ks], of);
    }
    delta[qb] = pair_sum(dl);                     // rowsum(dO * O): lanes l and l ^ 32 hold the two halves of a row
    const int64_t ro = ((int64_t)b * p.H + h) * p.Lq + qr;
    lse[qb] = qv[qb] ? p.lse[ro] : __builtin_inff();
    if (qv[qb] && hi == 0) p.delta[ro] = delta[qb];
  }
  f32x16 acc[2][NDT];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[qb][dt][r] = 0.f;

  for (int kt = 0; kt < ntiles; ++kt) {
    const int k0 = kt * KT;
    // tile kt has landed (this wave's share: at most tile kt + 1's four transfers may still be in flight), then everybody's has,
    // and everybody is done reading the stage tile kt + 2 will overwrite
    if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < ntiles) issue(kt + 2);
    if (k0 >= wkend) continue;                       // wave-uniform: nothing in this tile is visible to these 64 queries
    const char* sK = smem + (kt % NST) * 2 * TB;
    const char* sV = sK + TB;
    const bool masked = k0 + KT > kfull;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      // P^T = exp2(K Q^T c - lse), dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T -- one 32-key half at a time
      f32x16 sc[2], dp[2];
      {
        Frag ka[NKD];
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) ka[ks] = Pol::template a_row<D>(sK, 32 * h2 + j, ks, hi);
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) sc[qb] = Pol::mma(ka[ks], qf[qb][ks], ks == 0 ? zero16 : sc[qb]);
      }
      {
        Frag va[NKD];
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) va[ks] = Pol::template a_row<D>(sV, 32 * h2 + j, ks, hi);
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) dp[qb] = Pol::mma(va[ks], dof[qb][ks], ks == 0 ? zero16 : dp[qb]);
      }
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = fast_exp2(fmaf(sc[qb][r], c, -lse[qb]));      // rows without keys: lse = +inf -> 0
          if (masked) pv = (k0 + 32 * h2 + crow(r, hi)) < klim[qb] ? pv : 0.f;
          sc[qb][r] = pv * (dp[qb][r] - delta[qb]);
        }
      Frag kt_a[NKR][NDT];
#pragma unroll
      for (int ks = 0; ks < NKR; ++ks)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) kt_a[ks][dt] = Pol::template a_tr<D>(sK, 32 * h2, ks, hi, 32 * dt, lane);
#pragma unroll
      for (int ks = 0; ks < NKR; ++ks) {
        Frag db[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) db[qb] = Pol::b_from_acc(sc[qb], ks);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) acc[qb][dt] = Pol::mma(kt_a[ks][dt], db[qb], acc[qb][dt]);
      }
    }
  }

  // (the epilogue's row / half indices are re-derived from the work-item id behind an opaque move: as values computed at the top they were
  //  the two VGPRs the register allocator stored before the key loop and reloaded here -- the kernel's 12 bytes of scratch, rounds 3-4)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, j_e = lane_e & 31, hi_e = lane_e >> 5;
  const int wq0_e = q0 + (tid_e >> 6) * 64;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow_e = wq0_e + 32 * qb + j_e;
    if (!(qrow_e < p.Lq)) continue;
    E* dqp = (E*)p.dq + (int64_t)b * p.dq_bs + (int64_t)h * p.dq_hs + (int64_t)qrow_e * p.dq_ts;
    const float sc2 = p.scale;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        Pol::st4(dqp + 32 * dt + 8 * g + 4 * hi_e, acc[qb][dt][4 * g] * sc2, acc[qb][dt][4 * g + 1] * sc2, acc[qb][dt][4 * g + 2] * sc2,
                 acc[qb][dt][4 * g + 3] * sc2);
  }
}

// ---- dK / dV ----------------------------------------------------------------------------------------------------------
// One workgroup per 128-key tile (a lane owns a key); walks the query tiles that can see it, Q / dO tiles double-buffered.
// DMAQ (head_dim 64, 16-bit, no key mask / bias, 16-byte aligned q / dout rows; the launcher decides): the Q / dO tile of the
// NEXT query tile, its lse and delta rows arrive by LDS-DMA (global_load_lds_dwordx4 / _dword from inline asm, the XOR swizzle of
// unit_off<64> on the source address as in attn_fwd64_kernel) while the current tile is multiplied; one vmcnt(0) + one barrier per
// tile.  The register-staged form below was meant to do the same, but hipcc puts `s_waitcnt vmcnt(0)` in front of the first MFMA
// of the iteration that issued the loads (profiles/r03_attn_dkv_dma.txt): every tile's latency was exposed.
template <typename E, int D, int NW, bool EXTRA, bool DMAQ = false>
__global__ __launch_bounds__(NW * 64, (D >= 128 || sizeof(E) == 4) ? 1 : 2) void attn_bwd_dkv_kernel(const AttnArgs p) {
  static_assert(!DMAQ || (D == 64 && sizeof(E) == 2 && NW == 4 && !EXTRA), "DMAQ: head_dim 64, 16-bit, 4 waves, no mask / bias");
  typedef typename PolOf<E>::type Pol;
  typedef typename Pol::Frag Frag;
  constexpr int KTW = NW * 32, QT = kKT, NKD = D / Pol::KSTEP, NKR = 32 / Pol::KSTEP, NDT = D / 32, NT = NW * 64;
  constexpr int TB = Pol::tile_bytes(QT, D);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sRow = (float*)(smem + 4 * TB);      // [buf][3][QT]: lse, delta, key limit (int bits)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  // grid = (kv heads x batch, key tiles): key tile 0 is seen by every query tile of a causal mask -- longest first, globally
  const int b = (int)blockIdx.x / p.Hkv, hk = (int)blockIdx.x % p.Hkv, grp = p.H / p.Hkv;
  const int kk0 = blockIdx.y * KTW, krow = kk0 + wave * 32 + j;
  const bool kv = krow < p.Lk;
  const int64_t kr = kv ? krow : 0;
  const float c = p.scale * kLog2e;
  bool key_on = kv;
  if constexpr (EXTRA) key_on = kv && (!p.kmask || p.kmask[(int64_t)b * p.Lk + kr] != 0);

  // first query row that can see a key of this tile (its own first key kk0)
  int qfirst = 0;
  if (p.mask_mode == 1) qfirst = kk0 - (p.Lk - p.Lq);
  else if (p.mask_mode == 2) qfirst = (kk0 / p.cluster) * p.cluster;
  if (qfirst < 0) qfirst = 0;
  const int qt0 = qfirst / QT, nqt = (p.Lq + QT - 1) / QT;
  const int per_head = nqt > qt0 ? nqt - qt0 : 0, nit = per_head * grp;

  typename Pol::template Stage<D, QT, NT> stQ, stO;
  float r_lse = 0.f, r_delta = 0.f;
  int r_q = 0;
  auto fetch = [&](int it) {
    const int hq = hk * grp + it / per_head, q0 = (qt0 + it % per_head) * QT;
    stQ.load((const E*)p.q + (int64_t)b * p.q_bs + (int64_t)hq * p.q_hs + (int64_t)q0 * p.q_ts, p.q_ts, p.Lq - q0, tid);
    stO.load((const E*)p.dout + (int64_t)b * p.do_bs + (int64_t)hq * p.do_hs + (int64_t)q0 * p.do_ts, p.do_ts, p.Lq - q0, tid);
    {   // every thread loads (clamped, unconditional: see Pol16::Stage); threads >= QT and rows past Lq are sorted out at commit
      const int q = q0 + (tid & (QT - 1));
      const int64_t ro = ((int64_t)b * p.H + hq) * p.Lq;
      const int qc = q < p.Lq ? q : p.Lq - 1;
      r_lse = p.lse[ro + qc];
      r_delta = p.delta[ro + qc];
      r_q = q;
    }
  };
  auto commit = [&](int buf) {
    stQ.store(smem + (2 * buf) * TB, tid);
    stO.store(smem + (2 * buf + 1) * TB, tid);
    if (tid < QT) {
      float* r = sRow + buf * 3 * QT;
      const bool ok = r_q < p.Lq;
      r[tid] = ok ? r_lse : __builtin_inff();
      r[QT + tid] = ok ? r_delta : 0.f;
      ((int*)r)[2 * QT + tid] = key_limit(p, r_q);
    }
  };
  // ---- DMAQ: this wave moves rows 16 wave .. 16 wave + 15 of the Q tile and of the dO tile (2 x 1 KB each); wave 0 also the
  // 64 lse and delta entries.  address = (tile base, wave-uniform, SGPRs) + (per-lane 32-bit byte offset)
  const int wq = __builtin_amdgcn_readfirstlane(wave);
  auto dma_row = [&](int i) { return 16 * wq + 8 * i + (lane >> 3); };
  auto dma_unit = [&](int row) { return ((lane & 7) ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) * 8; };   // Pol16::unit_off<64>
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
  const unsigned ldsRow = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)((char*)sRow);
  auto issue = [&](int it) {
    const int hq = hk * grp + it / per_head, q0 = (qt0 + it % per_head) * QT, buf = it & 1;
    const E* gq = (const E*)p.q + (int64_t)b * p.q_bs + (int64_t)hq * p.q_hs + (int64_t)q0 * p.q_ts;
    const E* go = (const E*)p.dout + (int64_t)b * p.do_bs + (int64_t)hq * p.do_hs + (int64_t)q0 * p.do_ts;
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // rows past Lq re-read row Lq - 1 (finite values under lse = +inf, i.e. probability 0)
      const int row = dma_row(i), unit = dma_unit(row);
      const int r = q0 + row < p.Lq ? row : p.Lq - 1 - q0;
      o[i] = (unsigned)(((int64_t)r * p.q_ts + unit) * 2);
      o[2 + i] = (unsigned)(((int64_t)r * p.do_ts + unit) * 2);
    }
    const unsigned dst = lds0 + (unsigned)(2 * buf * TB + wq * 2048);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o0], %[gq]\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o1], %[gq]\n\t"
        "s_add_u32 m0, m0, 0x1c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o2], %[go]\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[o3], %[go]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [dst] "s"(dst), [gq] "s"(gq), [go] "s"(go), [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3])
        : "memory", "scc");
    if (wq == 0) {
      const int q = q0 + lane;
      const unsigned oq = (unsigned)((q < p.Lq ? q : p.Lq - 1) * 4);
      const float* gl = p.lse + ((int64_t)b * p.H + hq) * p.Lq;
      const float* gd = p.delta + ((int64_t)b * p.H + hq) * p.Lq;
      const unsigned dr = ldsRow + (unsigned)(buf * 3 * QT * 4);
      asm volatile(
          "s_mov_b32 %[keep], m0\n\t"
          "s_mov_b32 m0, %[dr]\n\ts_nop 0\n\tglobal_load_lds_dword %[oq], %[gl]\n\t"
          "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %[oq], %[gd]\n\t"
          "s_mov_b32 m0, %[keep]"
          : [keep] "=&s"(keep)
          : [dr] "s"(dr), [gl] "s"(gl), [gd] "s"(gd), [oq] "v"(oq)
          : "memory", "scc");
    }
  };
  auto landed = [&](int it) {      // after this wave's vmcnt(0): key limits, and lse = +inf / delta = 0 for query rows past Lq
    if (wq == 0) {
      const int q = (qt0 + it % per_head) * QT + lane;
      float* r = sRow + (it & 1) * 3 * QT;
      ((int*)r)[2 * QT + lane] = key_limit(p, q);
      if (q >= p.Lq) { r[lane] = __builtin_inff(); r[QT + lane] = 0.f; }
    }
  };
  if (nit > 0) {
    if constexpr (DMAQ) issue(0); else fetch(0);
  }

  const E* kp = (const E*)p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs + kr * p.k_ts;
  const E* vp = (const E*)p.v + (int64_t)b * p.v_bs + (int64_t)hk * p.v_hs + kr * p.v_ts;
  Frag kf[NKD], vf[NKD];
#pragma unroll
  for (int ks = 0; ks < NKD; ++ks) {     // unconditional (kr is clamped to a valid row); lanes without a key are zeroed after the pin below
    kf[ks] = Pol::ldg(kp, ks, hi);
    vf[ks] = Pol::ldg(vp, ks, hi);
  }
  f32x16 dk[NDT], dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  // The K / V fragments are consumed here, on every path into the loop: s_waitcnt insertion is static, and with their loads
  // possibly pending at the loop header (the nit == 0 path skips the waits below) hipcc put a vmcnt(0) in front of the FIRST
  // MFMA of the loop body -- executed every iteration, it waited for the tile requested a few instructions earlier.
#pragma unroll
  for (int ks = 0; ks < NKD; ++ks) {
    if constexpr (sizeof(E) == 2) {
      asm volatile("" : "+v"(kf[ks].x), "+v"(kf[ks].y), "+v"(kf[ks].z), "+v"(kf[ks].w));
      asm volatile("" : "+v"(vf[ks].x), "+v"(vf[ks].y), "+v"(vf[ks].z), "+v"(vf[ks].w));
    } else {
      asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
    }
  }
  if (!kv) {
#pragma unroll
    for (int ks = 0; ks < NKD; ++ks) { kf[ks] = Pol::zero(); vf[ks] = Pol::zero(); }
  }
  if (nit > 0) {
    if constexpr (DMAQ) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      landed(0);
    } else {
      commit(0);
    }
  }
  __syncthreads();

  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1, q0 = (qt0 + it % per_head) * QT;
    const char* sQ = smem + (2 * buf) * TB;
    const char* sdO = smem + (2 * buf + 1) * TB;
    const float* sLse = sRow + buf * 3 * QT;
    const float* sDelta = sLse + QT;
    const int* sKlim = (const int*)(sLse + 2 * QT);
    const bool more = it + 1 < nit;
    if (more) {
      if constexpr (DMAQ) issue(it + 1); else fetch(it + 1);       // DMAQ: buffer buf ^ 1 was last read before the barrier above
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < NKD; ++ks) {
        s = Pol::mma(Pol::template a_row<D>(sQ, 32 * h2 + j, ks, hi), kf[ks], s);
        dp = Pol::mma(Pol::template a_row<D>(sdO, 32 * h2 + j, ks, hi), vf[ks], dp);
      }
      // per-query scalars of the 16 rows this lane's registers stand for: 4 x 4 consecutive rows -> 16-byte LDS reads, all
      // requested before the first use (a scalar read + branch per score serialised this loop: 3.36 -> 2.13 ms)
      float rl[16], rd[16];
      int rk[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int base = 32 * h2 + 8 * g4 + 4 * hi;
        const float4 a = *(const float4*)(sLse + base), d4 = *(const float4*)(sDelta + base);
        const int4 k4 = *(const int4*)(sKlim + base);
        rl[4 * g4] = a.x; rl[4 * g4 + 1] = a.y; rl[4 * g4 + 2] = a.z; rl[4 * g4 + 3] = a.w;
        rd[4 * g4] = d4.x; rd[4 * g4 + 1] = d4.y; rd[4 * g4 + 2] = d4.z; rd[4 * g4 + 3] = d4.w;
        rk[4 * g4] = k4.x; rk[4 * g4 + 1] = k4.y; rk[4 * g4 + 2] = k4.z; rk[4 * g4 + 3] = k4.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = s[r] * c;
        if constexpr (EXTRA) {
          const int ql = 32 * h2 + crow(r, hi);
          if (p.bias && q0 + ql < p.Lq && kv) x = fmaf(p.bias[(int64_t)(q0 + ql) * p.Lk + kr], kLog2e, x);
        }
        const bool on = key_on && krow < rk[r];
        const float pv = on ? fast_exp2(x - rl[r]) : 0.f;
        float pd = pv, dpr = dp[r];
        if constexpr (EXTRA) {
          if (p.drop_thresh) {
            const int hq = hk * grp + it / per_head;
            const bool keep = attn_drop_hash(p.drop_seed, (uint32_t)(b * p.H + hq), (uint32_t)(q0 + 32 * h2 + crow(r, hi)), (uint32_t)krow) >= p.drop_thresh;
            pd = keep ? pv * p.drop_scale : 0.f;           // dV sees the dropped probabilities, dS the undropped ones
            dpr = keep ? dpr * p.drop_scale : 0.f;
          }
        }
        s[r] = pd;
        dp[r] = pv * (dpr - rd[r]);
      }
#pragma unroll
      for (int ks = 0; ks < NKR; ++ks) {
        const Frag pb = Pol::b_from_acc(s, ks), db = Pol::b_from_acc(dp, ks);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          dv[dt] = Pol::mma(Pol::template a_tr<D>(sdO, 32 * h2, ks, hi, 32 * dt, lane), pb, dv[dt]);
          dk[dt] = Pol::mma(Pol::template a_tr<D>(sQ, 32 * h2, ks, hi, 32 * dt, lane), db, dk[dt]);
        }
      }
    }
    if (more) {
      if constexpr (DMAQ) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed(it + 1);
      } else {
        commit(buf ^ 1);
      }
    }
    __syncthreads();
  }
  if (!kv) return;
  E* dkp = (E*)p.dk + (int64_t)b * p.dk_bs + (int64_t)hk * p.dk_hs + kr * p.dk_ts;
  E* dvp = (E*)p.dv + (int64_t)b * p.dv_bs + (int64_t)hk * p.dv_hs + kr * p.dv_ts;
  const float sc = p.scale;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * dt + 8 * g + 4 * hi;
      Pol::st4(dkp + d, dk[dt][4 * g] * sc, dk[dt][4 * g + 1] * sc, dk[dt][4 * g + 2] * sc, dk[dt][4 * g + 3] * sc);
      Pol::st4(dvp + d, dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
    }
}

// ---- host -------------------------------------------------------------------------------------------------------------
static thread_local int g_attn_hip_error = 0;

static int attn_launch_check() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_attn_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}

template <typename K>
static int raise_lds(K kern, size_t lds) {
  if (lds > 64 * 1024) {
    if (lds > 160 * 1024) return MXVL_ERR_UNSUPPORTED;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MXVL_ERR_LAUNCH;
  }
  return MXVL_OK;
}

template <typename E, int D, bool DQ, bool EXTRA>
static int launch_q1(const AttnArgs& a, hipStream_t s) {
  typedef typename PolOf<E>::type Pol;
  constexpr int NW = 4;
  const size_t lds = 4 * (size_t)Pol::tile_bytes(kKT, D) + 2 * kKT * sizeof(float);
  auto kern = attn_q_kernel<E, D, NW, DQ, EXTRA>;
  int rc = raise_lds(kern, lds);
  if (rc != MXVL_OK) return rc;
  dim3 grid(a.H * a.batch, (a.Lq + NW * 32 - 1) / (NW * 32), 1);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
  return attn_launch_check();
}
template <typename E, bool DQ>
static int launch_fwd64(const AttnArgs& a, hipStream_t s) {
  const size_t lds = 3 * 2 * (size_t)(kKT * 64 * 2);
  dim3 grid(a.H * a.batch, (a.Lq + 255) / 256, 1);
  if constexpr (DQ) hipLaunchKernelGGL(attn_dq64_kernel<E>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(attn_fwd64_kernel<E>, grid, dim3(256), lds, s, a);
  return attn_launch_check();
}
template <typename E, int D, bool DQ>
static int launch_q(const AttnArgs& a, hipStream_t s) {
  if constexpr (D == 64 && sizeof(E) == 2) {
    // 16-byte aligned rows (the DMA moves 16-byte units) and no key mask / bias: the 64-queries-per-wave kernels
    const bool aligned = a.k_ts % 8 == 0 && a.v_ts % 8 == 0 && a.k_hs % 8 == 0 && a.v_hs % 8 == 0 && a.k_bs % 8 == 0 && a.v_bs % 8 == 0 &&
                         ((uintptr_t)a.k % 16) == 0 && ((uintptr_t)a.v % 16) == 0;
    if (!a.kmask && !a.bias && !a.drop_thresh && aligned && a.Lk > 0) return launch_fwd64<E, DQ>(a, s);
  }
  return (a.kmask || a.bias || a.drop_thresh) ? launch_q1<E, D, DQ, true>(a, s) : launch_q1<E, D, DQ, false>(a, s);
}

template <typename E, int D, bool EXTRA, bool DMAQ = false>
static int launch_dkv1(const AttnArgs& a, hipStream_t s);
template <typename E, int D>
static int launch_dkv(const AttnArgs& a, hipStream_t s) {
  if constexpr (D == 64 && sizeof(E) == 2) {      // Q / dO tiles by LDS-DMA: 16-byte aligned rows, 32-bit in-tile offsets
    const bool aligned = a.q_ts % 8 == 0 && a.do_ts % 8 == 0 && a.q_hs % 8 == 0 && a.do_hs % 8 == 0 && a.q_bs % 8 == 0 && a.do_bs % 8 == 0 &&
                         ((uintptr_t)a.q % 16) == 0 && ((uintptr_t)a.dout % 16) == 0 && a.q_ts < (1 << 24) && a.do_ts < (1 << 24);
    if (!a.kmask && !a.bias && !a.drop_thresh && aligned && a.Lq > 0) return launch_dkv1<E, D, false, true>(a, s);
  }
  return (a.kmask || a.bias || a.drop_thresh) ? launch_dkv1<E, D, true>(a, s) : launch_dkv1<E, D, false>(a, s);
}
template <typename E, int D, bool EXTRA, bool DMAQ>
static int launch_dkv1(const AttnArgs& a, hipStream_t s) {
  typedef typename PolOf<E>::type Pol;
  constexpr int NW = 4;
  const size_t lds = 4 * (size_t)Pol::tile_bytes(kKT, D) + 6 * kKT * sizeof(float);
  auto kern = attn_bwd_dkv_kernel<E, D, NW, EXTRA, DMAQ>;
  int rc = raise_lds(kern, lds);
  if (rc != MXVL_OK) return rc;
  dim3 grid(a.Hkv * a.batch, (a.Lk + NW * 32 - 1) / (NW * 32), 1);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
  return attn_launch_check();
}

template <typename E>
static int dispatch_attn(const AttnArgs& a, int D, int what, hipStream_t s) {   // what: 0 fwd, 1 dq, 2 dkv
  switch (D) {
    case 32: return what == 0 ? launch_q<E, 32, false>(a, s) : what == 1 ? launch_q<E, 32, true>(a, s) : launch_dkv<E, 32>(a, s);
    case 64: return what == 0 ? launch_q<E, 64, false>(a, s) : what == 1 ? launch_q<E, 64, true>(a, s) : launch_dkv<E, 64>(a, s);
    // head_dim 128 (Llama-2-7B / Qwen: the hybrid decoder's training-time self- and image cross-attention): one wave per SIMD,
    // the 512-entry unified register file holds the Q / dO (dQ pass) or K / V (dK/dV pass) fragments next to 8 accumulators
    case 128: return what == 0 ? launch_q<E, 128, false>(a, s) : what == 1 ? launch_q<E, 128, true>(a, s) : launch_dkv<E, 128>(a, s);
    // head_dim 256 (Gemma-sized decoders; the decode kernels have a <256> instantiation): the prompt pass only -- forward, 16-bit
    case 256:
      if constexpr (sizeof(E) == 2) { if (what == 0) return launch_q<E, 256, false>(a, s); }
      return MXVL_ERR_UNSUPPORTED;
    default: return MXVL_ERR_UNSUPPORTED;
  }
}

static int fill_args(const mxvl_attn_desc* d, AttnArgs& a) {
  if (!d || !d->q || !d->k || !d->v) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_F32 && d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->batch <= 0 || d->n_heads <= 0 || d->n_kv_heads <= 0 || d->seqlen_q <= 0 || d->seqlen_k <= 0) return MXVL_ERR_SHAPE;
  if (d->n_heads % d->n_kv_heads != 0) return MXVL_ERR_SHAPE;
  if (d->mask_mode < 0 || d->mask_mode > 2 || (d->mask_mode == 2 && d->cluster <= 0)) return MXVL_ERR_SHAPE;
  const int64_t esz = d->io_dtype == MXVL_F32 ? 4 : 2, al = 16 / esz;   // rows must start on 16-byte boundaries
  const int64_t st[] = {d->q_bs, d->q_hs, d->q_ts, d->k_bs, d->k_hs, d->k_ts, d->v_bs, d->v_hs, d->v_ts};
  for (int64_t x : st) if (x < 0 || x % al != 0) return MXVL_ERR_STRIDE;
  const void* ps[] = {d->q, d->k, d->v};
  for (const void* q : ps) if ((uintptr_t)q % 16 != 0) return MXVL_ERR_STRIDE;
  a.batch = d->batch; a.H = d->n_heads; a.Hkv = d->n_kv_heads; a.Lq = d->seqlen_q; a.Lk = d->seqlen_k;
  a.mask_mode = d->mask_mode; a.cluster = d->cluster > 0 ? d->cluster : 1; a.scale = d->scale;
  a.q_bs = d->q_bs; a.q_hs = d->q_hs; a.q_ts = d->q_ts; a.k_bs = d->k_bs; a.k_hs = d->k_hs; a.k_ts = d->k_ts;
  a.v_bs = d->v_bs; a.v_hs = d->v_hs; a.v_ts = d->v_ts; a.o_bs = d->o_bs; a.o_hs = d->o_hs; a.o_ts = d->o_ts;
  a.q = d->q; a.k = d->k; a.v = d->v; a.out = d->out; a.o = d->out; a.lse = (float*)d->lse;
  a.kmask = (const uint8_t*)d->key_mask; a.bias = (const float*)d->bias;
  a.drop_thresh = 0; a.drop_seed = d->dropout_seed; a.drop_scale = 1.0f;
  if (d->dropout_p != 0.0f) {
    if (!(d->dropout_p > 0.0f && d->dropout_p < 1.0f)) return MXVL_ERR_SHAPE;
    const double t = (double)d->dropout_p * 4294967296.0;
    a.drop_thresh = t < 1.0 ? 1u : (t > 4294967295.0 ? 4294967295u : (uint32_t)t);
    a.drop_scale = 1.0f / (1.0f - d->dropout_p);
  }
  a.dout = nullptr; a.dq = a.dk = a.dv = nullptr; a.delta = nullptr;
  a.do_bs = a.do_hs = a.do_ts = a.dq_bs = a.dq_hs = a.dq_ts = a.dk_bs = a.dk_hs = a.dk_ts = a.dv_bs = a.dv_hs = a.dv_ts = 0;
  return MXVL_OK;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_attn_fwd(const mxvl_attn_desc* d, void* hip_stream) {
  AttnArgs a;
  int rc = fill_args(d, a);
  if (rc != MXVL_OK) return rc;
  if (!d->out) return MXVL_ERR_NULL;
  const int64_t al = d->io_dtype == MXVL_F32 ? 4 : 8;
  if (d->o_bs % al || d->o_hs % al || d->o_ts % al || (uintptr_t)d->out % 16) return MXVL_ERR_STRIDE;
  hipStream_t s = (hipStream_t)hip_stream;
  switch (d->io_dtype) {
    case MXVL_F32: return dispatch_attn<float>(a, d->head_dim, 0, s);
    case MXVL_BF16: return dispatch_attn<bf16_t>(a, d->head_dim, 0, s);
    default: return dispatch_attn<f16_t>(a, d->head_dim, 0, s);
  }
}

extern "C" int mxvl_attn_bwd(const mxvl_attn_bwd_desc* d, void* hip_stream) {
  if (!d) return MXVL_ERR_NULL;
  AttnArgs a;
  int rc = fill_args(&d->fwd, a);
  if (rc != MXVL_OK) return rc;
  if (!d->fwd.out || !d->fwd.lse || !d->dout || !d->dq || !d->dk || !d->dv || !d->delta) return MXVL_ERR_NULL;
  const int64_t al = d->fwd.io_dtype == MXVL_F32 ? 4 : 8;
  const int64_t st[] = {d->fwd.o_bs, d->fwd.o_hs, d->fwd.o_ts, d->dout_bs, d->dout_hs, d->dout_ts, d->dq_bs, d->dq_hs, d->dq_ts,
                        d->dk_bs, d->dk_hs, d->dk_ts, d->dv_bs, d->dv_hs, d->dv_ts};
  for (int64_t x : st) if (x < 0 || x % al != 0) return MXVL_ERR_STRIDE;
  const void* ps[] = {d->fwd.out, d->dout, d->dq, d->dk, d->dv};
  for (const void* q : ps) if ((uintptr_t)q % 16 != 0) return MXVL_ERR_STRIDE;
  a.dout = d->dout; a.dq = d->dq; a.dk = d->dk; a.dv = d->dv; a.delta = (float*)d->delta;
  a.do_bs = d->dout_bs; a.do_hs = d->dout_hs; a.do_ts = d->dout_ts; a.dq_bs = d->dq_bs; a.dq_hs = d->dq_hs; a.dq_ts = d->dq_ts;
  a.dk_bs = d->dk_bs; a.dk_hs = d->dk_hs; a.dk_ts = d->dk_ts; a.dv_bs = d->dv_bs; a.dv_hs = d->dv_hs; a.dv_ts = d->dv_ts;
  hipStream_t s = (hipStream_t)hip_stream;
  const int D = d->fwd.head_dim;
  for (int what = 1; what <= 2; ++what) {   // dQ (also writes delta) first, dK/dV second: same stream, in order
    switch (d->fwd.io_dtype) {
      case MXVL_F32: rc = dispatch_attn<float>(a, D, what, s); break;
      case MXVL_BF16: rc = dispatch_attn<bf16_t>(a, D, what, s); break;
      default: rc = dispatch_attn<f16_t>(a, D, what, s); break;
    }
    if (rc != MXVL_OK) return rc;
  }
  return MXVL_OK;
}
