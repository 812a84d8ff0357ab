// gemm_swiglu.hip -- the SwiGLU input projection as ONE MFMA kernel for gfx950 (MI355X, CDNA4):
//
//     ab = x [w1; w2]^T + [b1 | b2]          h = silu(ab[:, :H]) * ab[:, H:]
//
// Replaces `self.w1(x)`, `self.w2(x)`, `self.act(x1) * x2` of the reference's SwiGLU
// (CXPMRG_Bench_MambaXray_VL/arm/Finetuning/models_mamba.py:59-83; pretrain/models_pretrain twin) -- in round 2 one library
// GEMM over the merged weight followed by mxvl::swiglu_kernel, which re-read the 713 MB pre-activation tensor of an ARM-large
// layer (65 280 tokens x 5460) to write the 356 MB gate output.  Here the gate is the GEMM's epilogue: h is formed from the
// fp32 accumulators, ab leaves only when the caller wants it for the backward pass (training), and nothing is re-read.
//
// Tiling.  One PERSISTENT workgroup per CU (8 waves, 2 per SIMD) walks tiles of 256 tokens x 128 gate columns, i.e. 256 x 256
// MFMA tiles: the B operand tile is [128 rows of w1 | the SAME 128 rows of w2], so the two pre-activations a gate element needs
// sit in the same lane and register of two accumulator sets.  K is walked in 64-wide steps, A/B tiles double-buffered in LDS
// (2 x 64 KB), filled by LDS-DMA (global_load_lds_dwordx4 from inline asm: no staging VGPRs, no ds_write pass, and no compiler
// vmcnt(0) in front of the next ds_read) one step ahead -- across tile boundaries too: the first stage of the next tile lands
// while the last step of this one is multiplied and its epilogue runs.
// K step = 4 sub-steps of [LOAD 6 ds_read_b128] barrier [COMPUTE 8 MFMA] barrier with the two waves of every SIMD one segment
// apart (waves 4-7 take one extra barrier up front): one wave feeds the matrix pipe while its partner reads LDS / issues DMA
// (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md 5, the 8-phase template's idea at 4 phases per K step).
// LDS rows are 128 bytes; the 16-byte units of a row are XOR-swizzled with key(row) = (row bit 1) << 2 | (row bits 3..2) --
// the key csrc/attn.hip measured conflict-free for ds_read_b128 on 128-byte rows -- applied on the DMA's SOURCE address (the
// DMA writes LDS linearly) and on the fragment reads.
// MFMA mapping: v_mfma_f32_32x32x16_{bf16,f16}, computed TRANSPOSED (A operand = weight rows, B operand = tokens), so a lane
// owns one token and 4 consecutive output columns per accumulator quad.  Wave (wm, wn) = 128 tokens x 32 gate columns: 4 token
// tiles x {a, b} = 8 accumulators (128 VGPRs) that START at the bias.
// Epilogue, no LDS and no barrier: rounding and the gate in registers, lanes l / l + 32 trade 8-byte quads
// (v_permlane32_swap) so that every lane owns 16 contiguous bytes of its token row, two 16-byte stores per accumulator.
// Measured (profiles/r03_gemm_swiglu_bench.txt), ARM-large layer 65 280 x 1024 -> 2 x 2730: K loop 1.3-1.4 PFLOP/s; h only
// (inference) 0.73 ms vs 0.85 ms for library GEMM + gate kernel; h + ab (training) 0.85 ms = parity -- the 713 MB of
// pre-activations cost the library GEMM 0.19 ms too.
#include "mxvl_common.h"

namespace mxvl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gs_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void gs_lds_void;
typedef __attribute__((address_space(1))) const void gs_g_void;
struct __attribute__((packed, aligned(4))) gs_u4_a4 { uint32_t x, y, z, w; };     // output rows are only 4-byte aligned when H % 8 != 0 (2730)

struct GemmSwigluArgs {
  int M, K, H, bias_f32, ntm, ntn, ntm_x;
  int64_t x_rs, w_rs, ab_rs, h_rs;     // row strides in elements
  const void *x, *w, *bias;
  void *ab, *h;
  // MODE 1 (backward): x = dy (M, K), w = w3^T (H, K); ab is READ, dab (M, 2H; row stride h_rs) and partial (2 ntm, 2H) fp32 are written
  float* partial;
};

constexpr int GS_BM = 256, GS_BN = 128, GS_BK = 64, GS_NT = 512, GS_GROUP_M = 4, GS_SLOTS = 2;
constexpr int GS_ROWB = GS_BK * 2;                           // bytes per LDS row (128)
constexpr int GS_A_BYTES = GS_BM * GS_ROWB;                  // 32 KB
constexpr int GS_STAGE = GS_A_BYTES + 2 * GS_BN * GS_ROWB;   // 64 KB per K step, double-buffered

// 128-byte rows: the XOR key csrc/attn.hip measured conflict-free for ds_read_b128 (SQ_LDS_BANK_CONFLICT = 0)
__device__ __forceinline__ int gs_key(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

template <typename E>
__device__ __forceinline__ f32x16 gs_mma(uint4 a, uint4 b, f32x16 c) {
  if constexpr (__is_same(E, bf16_t))
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gs_bf16x8, a), __builtin_bit_cast(gs_bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gs_f16x8, a), __builtin_bit_cast(gs_f16x8, b), c, 0, 0, 0);
}
template <typename E>
__device__ __forceinline__ uint32_t gs_pack2(float a, float b) {
  if constexpr (__is_same(E, bf16_t)) return cvt_pk_bf16(a, b);
  else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, h2{(_Float16)a, (_Float16)b});
  }
}

// lanes l <-> l + 32 exchange the upper half of `a` with the lower half of `b` (one VALU op; the builtin of this toolchain returns
// its first result twice, so inline asm; `s_nop 1` = the two wait states a swap needs after a VALU write of its operands)
__device__ __forceinline__ void gs_swap32(uint32_t& a, uint32_t& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// MODE 0: the forward above.  MODE 1 (round 5): the BACKWARD of the MLP's first half fused into the dgrad GEMM that feeds it --
//     d_h = dy w3        (M, K = C) x (K, H): the B tile is 256 rows of w3^T, one 256 x 256 output tile of d_h per workgroup
//     d[a | b] = swiglu'(a, b) d_h    in the epilogue: ab is read and dab written straight from / to the accumulator layout
//     partial[2 mt + wm][2H] = column sums of the rounded d[a | b] over the workgroup's 128-token halves (the bias gradients)
// d_h never exists in memory: the round-4 step wrote it (359 MB per ARM-large layer), and mxvl_swiglu_bwd_colsum read it back
// together with ab and wrote dab -- 4 % of the step in a pass that is now this GEMM's epilogue (same formulas, same roundings:
// d_h is rounded to the io dtype before it is used, the sums are of the rounded gradients).
// (MODE 1 epilogue) the same two instructions as gs_swap32 / cvt_pk_bf16 WITHOUT `volatile`: their operands carry every dependency, and
// as volatile statements they may not be reordered against each other -- the bf16 epilogue then kept 35 more VGPRs alive than the
// fp16 one (whose packing is plain C) and spilled them
__device__ __forceinline__ void gs_swap32_nv(uint32_t& a, uint32_t& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <typename E>
__device__ __forceinline__ uint32_t gs_pack2_nv(float a, float b) {
  if constexpr (__is_same(E, bf16_t)) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, h2{(_Float16)a, (_Float16)b});
  }
}

template <typename E, int MODE = 0>
__global__ __launch_bounds__(GS_NT, 2) void gemm_swiglu_kernel(const GemmSwigluArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int j = lane & 31, hi = lane >> 5;

  // ---- persistent workgroup: tiles blockIdx.x, + gridDim.x, ... ; consecutive ids sit on different XCDs (id % 8), so XCD x walks
  // the token tiles x, x + 8, ... in groups of GS_GROUP_M x all column tiles: the tiles resident on an XCD at one time share their
  // x / weight K slices through its L2
  const int xcd = blockIdx.x & 7, lstride = gridDim.x >> 3;
  const int per_group = GS_GROUP_M * p.ntn, nlocal = p.ntm_x * p.ntn;
  int local = blockIdx.x >> 3;
  int m0 = 0, n0 = 0;
  auto locate = [&](int loc) -> bool {           // tile coordinates of local index loc; false: beyond the last token tile
    const int grp = loc / per_group, li = loc - grp * per_group;
    const int gm0 = grp * GS_GROUP_M;
    const int gsz = (p.ntm_x - gm0) < GS_GROUP_M ? (p.ntm_x - gm0) : GS_GROUP_M;
    const int mt = (gm0 + li % gsz) * 8 + xcd;
    m0 = mt * GS_BM;
    n0 = (li / gsz) * (MODE == 0 ? GS_BN : 2 * GS_BN);
    return mt < p.ntm;
  };
  // the 8th XCD-slice of a ragged grid has holes: skip them (wave-uniform)
  while (local < nlocal && !locate(local)) local += lstride;
  if (local >= nlocal) return;

  // ---- LDS-DMA sources: wave w < 4 fills x rows [64 w, 64 w + 64), wave w >= 4 the weight-tile rows [64 (w - 4), ...) ------
  // one call = 8 rows x 128 bytes: lane i writes LDS (row 8 c + i / 8, unit i % 8) and therefore READS unit (i % 8) ^ key(row)
  const char* src[8];
  auto sources = [&]() {
    const int rl = lane >> 3, ul = lane & 7;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int row = (wave & 3) * 64 + c * 8 + rl;       // row inside the 256-row A tile / the 256-row B tile
      const int su = ul ^ gs_key(row);
      if (wave < 4) {
        int tk = m0 + row;
        tk = tk < p.M ? tk : p.M - 1;
        src[c] = (const char*)p.x + ((int64_t)tk * p.x_rs + su * 8) * 2;
      } else if constexpr (MODE == 0) {
        int col = n0 + (row & (GS_BN - 1));
        col = col < p.H ? col : p.H - 1;
        src[c] = (const char*)p.w + ((int64_t)((row >> 7) * p.H + col) * p.w_rs + su * 8) * 2;
      } else {                                   // MODE 1 / 2
        int col = n0 + row;                      // 256 consecutive rows of w3^T: acc_a = hidden columns n0 .. + 127, acc_b = + 128 .. + 255
        col = col < p.H ? col : p.H - 1;
        src[c] = (const char*)p.w + ((int64_t)col * p.w_rs + su * 8) * 2;
      }
    }
  };
  const int dma_dst = (wave < 4 ? 0 : GS_A_BYTES) + (wave & 3) * 64 * GS_ROWB;      // wave-uniform byte offset inside a stage
  // The DMA is issued from inline asm: behind __builtin_amdgcn_global_load_lds hipcc orders every later ds_read after the
  // transfer (it cannot tell the LDS bytes apart) with an `s_waitcnt vmcnt(0)` -- the whole HBM / L2 latency exposed in every K
  // step, which is what held the first version of this kernel at 1.0 PFLOP/s.  The asm statement is opaque: the only waits are
  // the explicit ones below.  M0 carries the wave-uniform LDS byte address; `s_nop 0` covers the M0 write -> LDS-DMA hazard.
  auto issue = [&](int buf, int kt) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem) + buf * GS_STAGE + dma_dst;
    const int64_t koff = (int64_t)kt * GS_ROWB;
    const char* g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) g[c] = src[c] + koff;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g0], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g1], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g2], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g3], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g4], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g5], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g6], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g7], off\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [dst] "s"(dst), [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]),
          [g6] "v"(g[6]), [g7] "v"(g[7])
        : "memory", "scc");
  };

  // ---- fragment addresses ------------------------------------------------------------------------------------------------
  const int kq = gs_key(j);
  int uoff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) uoff[ks] = ((2 * ks + hi) ^ kq) << 4;
  const int offFa = GS_A_BYTES + (wn * 32 + j) * GS_ROWB;          // w1 rows of the tile
  const int offFb = offFa + GS_BN * GS_ROWB;                       // the same rows of w2
  const int offT = (wm * 128 + j) * GS_ROWB;                       // token tile t: + t * 32 rows

  // accumulators start at the bias: lane (j, hi) of wave (wm, wn) holds, for token wm * 128 + t * 32 + j, the tile columns
  // wn * 32 + 8 g + 4 hi + {0..3} in registers 4 g .. 4 g + 3 of acc_a[t] (w1 part) and acc_b[t] (w2 part).  ONE branch on the
  // bias dtype around 32 independent loads (a dtype test per element serialised 32 L2 round trips per tile).
  f32x16 acc_a[4], acc_b[4];
  float ba[4][4], bb[4][4];
  auto bias_fetch = [&]() {
    int col[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = n0 + wn * 32 + 8 * g + 4 * hi + i;
        col[g][i] = c < p.H ? c : p.H - 1;
      }
    if (MODE == 1 || !p.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ba[g][i] = 0.f; bb[g][i] = 0.f; }
    } else if (p.bias_f32) {
      const float* bp = (const float*)p.bias;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ba[g][i] = bp[col[g][i]]; bb[g][i] = bp[MODE == 2 ? (col[g][i] + 128 < p.H ? col[g][i] + 128 : p.H - 1) : p.H + col[g][i]]; }
    } else {
      const uint16_t* bp = (const uint16_t*)p.bias;
      uint16_t ra[4][4], rb[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[g][i] = bp[col[g][i]]; rb[g][i] = bp[MODE == 2 ? (col[g][i] + 128 < p.H ? col[g][i] + 128 : p.H - 1) : p.H + col[g][i]]; }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ba[g][i] = Io<E>::ld((const E*)&ra[g][i]); bb[g][i] = Io<E>::ld((const E*)&rb[g][i]); }
    }
  };
  auto acc_init = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc_a[t][4 * g + i] = ba[g][i]; acc_b[t][4 * g + i] = bb[g][i]; }
  };

  // epilogue of one accumulator set: io-dtype rounding, the two hi halves of a lane pair trade 8-byte quads so that every lane
  // owns 16 contiguous bytes (columns 16 q + 8 hi .. + 7 of its token row), two 16-byte stores per accumulator -- no LDS, no
  // barrier: the next tile's first stage is already landing in the LDS ring while these stores go out
  auto store_acc = [&](E* base, int64_t rs, int tm0, int tn0, int cbase, const float (&v)[16], int t) {
    const int tkn = tm0 + wm * 128 + t * 32 + j;
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = gs_pack2<E>(v[2 * q], v[2 * q + 1]);      // w[2 g], w[2 g + 1]: the quad of group g
#pragma unroll
    for (int q = 0; q < 2; ++q) {                      // groups (2 q, 2 q + 1)
      gs_swap32(w[4 * q], w[4 * q + 2]);
      gs_swap32(w[4 * q + 1], w[4 * q + 3]);
      const int cl = wn * 32 + 16 * q + 8 * hi;        // first of this lane's 8 tile columns
      if (tkn < p.M && tn0 + cl < p.H) {
        E* dst = base + (int64_t)tkn * rs + cbase + tn0 + cl;
        if (tn0 + cl + 8 <= p.H) {
          *(gs_u4_a4*)dst = gs_u4_a4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (tn0 + cl + e < p.H) ((uint16_t*)dst)[e] = (uint16_t)(w[4 * q + (e >> 1)] >> ((e & 1) * 16));
        }
      }
    }
  };

  // K loop as a two-phase ping-pong between the two waves of every SIMD (MI355X_MICROARCH.md "Two waves per SIMD"): a K step is
  // 4 sub-steps of [LOAD: 6 ds_read_b128 fragments] barrier [COMPUTE: 8 MFMAs] barrier, and waves 4-7 (the SIMD partners of waves
  // 0-3) run ONE segment behind -- they take one extra barrier up front -- so that on every SIMD one wave feeds the matrix pipe
  // while the other reads LDS / issues the next stage's DMA, instead of both loading and then both queueing MFMAs.
  // The stage after the current one -- the next K step, or step 0 of this workgroup's NEXT tile -- is requested in the LOAD
  // segment of sub-step 0 (the late group finished reading that buffer one segment before the early group gets there) and waited
  // for (vmcnt(0), then the segment's barrier) in the LOAD segment of sub-step 3.
  const int nk = p.K / GS_BK;
  const int late = wave >> 2;
  sources();
  issue(0, 0);
  bias_fetch();
  acc_init();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (late) __builtin_amdgcn_s_barrier();
  int gk = 0;                                   // K steps done by this workgroup: stage buffer = gk & 1
  for (;;) {
    const int tm0 = m0, tn0 = n0;               // the tile being accumulated (m0 / n0 move on to the next tile in its last step)
    int nxt = local + lstride;
    bool has_next = false;
    for (int kt = 0; kt < nk; ++kt, ++gk) {
      const char* st = smem + (gk & 1) * GS_STAGE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // LOAD segment
        const uint4 fa = *(const uint4*)(st + offFa + uoff[ks]);
        const uint4 fb = *(const uint4*)(st + offFb + uoff[ks]);
        uint4 tk[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) tk[t] = *(const uint4*)(st + offT + t * 32 * GS_ROWB + uoff[ks]);
        if (ks == 0) {
          if (kt + 1 < nk) {
            issue((gk + 1) & 1, kt + 1);
          } else {
            while (nxt < nlocal && !locate(nxt)) nxt += lstride;
            has_next = nxt < nlocal;
            if (has_next) {
              sources();
              issue((gk + 1) & 1, 0);
            }
          }
        }
        if (ks == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // COMPUTE segment
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc_a[t] = gs_mma<E>(fa, tk[t], acc_a[t]);
          acc_b[t] = gs_mma<E>(fb, tk[t], acc_b[t]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    // ---- epilogue of tile (tm0, tn0): the next tile's bias is requested first, the stores hide its latency ------------------
    if constexpr (MODE == 2) {
      // plain GEMM: y (M, N = H) = x w^T (+ bias); acc_a = columns tn0 .. + 127, acc_b = tn0 + 128 .. + 255 of the 256-wide tile
      if (has_next) bias_fetch();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float va[16], vb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { va[r] = acc_a[t][r]; vb[r] = acc_b[t][r]; }
        store_acc((E*)p.h, p.h_rs, tm0, tn0, 0, va, t);
        store_acc((E*)p.h, p.h_rs, tm0, tn0 + 128, 0, vb, t);
      }
    } else if constexpr (MODE == 0) {
    if (has_next) bias_fetch();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float va[16], vb[16], vh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { va[r] = acc_a[t][r]; vb[r] = acc_b[t][r]; vh[r] = silu(va[r]) * vb[r]; }
      if (p.ab) {
        store_acc((E*)p.ab, p.ab_rs, tm0, tn0, 0, va, t);
        store_acc((E*)p.ab, p.ab_rs, tm0, tn0, p.H, vb, t);
      }
      store_acc((E*)p.h, p.h_rs, tm0, tn0, 0, vh, t);
    }
    } else {
      // 8 (column half X, token tile t) units, X outermost (one half's 32 column sums live at a time: with both the bf16
      // instantiation spilled 70 VGPRs); the pre-activations of unit u + 1 are requested before unit u is computed (two
      // register sets of 4 x 16 bytes).  A lane loads the 16 contiguous bytes it would store (columns 16 q + 8 hi .. + 7 of its token
      // row) and the v_permlane32_swap pair of store_acc -- an involution -- turns them into the accumulator layout.
      const E* abp = (const E*)p.ab;
      uint32_t wa[2][8], wb[2][8];
      auto fetch = [&](int u, int set) {
        const int t = u & 3, X = u >> 2;
        const int tkn = tm0 + wm * 128 + t * 32 + j;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int col = tn0 + X * 128 + wn * 32 + 16 * q + 8 * hi;
          const bool ok = tkn < p.M && col < p.H;          // H % 8 == 0: a lane's 8 columns are inside or outside as a whole
          const E* src_a = abp + (int64_t)(ok ? tkn : 0) * p.ab_rs + (ok ? col : 0);
          const gs_u4_a4 ra = *(const gs_u4_a4*)src_a, rb = *(const gs_u4_a4*)(src_a + p.H);
          wa[set][4 * q] = ok ? ra.x : 0u; wa[set][4 * q + 1] = ok ? ra.y : 0u; wa[set][4 * q + 2] = ok ? ra.z : 0u; wa[set][4 * q + 3] = ok ? ra.w : 0u;
          wb[set][4 * q] = ok ? rb.x : 0u; wb[set][4 * q + 1] = ok ? rb.y : 0u; wb[set][4 * q + 2] = ok ? rb.z : 0u; wb[set][4 * q + 3] = ok ? rb.w : 0u;
        }
      };
      auto unpack2 = [](uint32_t w, float& lo, float& hi_) {
        if constexpr (__is_same(E, bf16_t)) {
          lo = __builtin_bit_cast(float, w << 16);
          hi_ = __builtin_bit_cast(float, w & 0xffff0000u);
        } else {
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          const h2 hv = __builtin_bit_cast(h2, w);
          lo = (float)hv.x;
          hi_ = (float)hv.y;
        }
      };
      auto rnd = [](float v) -> float {
        if constexpr (__is_same(E, bf16_t)) {             // round-to-nearest-even in integer arithmetic: schedulable (the v_cvt_pk_bf16_f32
          uint32_t u = __builtin_bit_cast(uint32_t, v);     // asm statement pinned 37 more VGPRs across the unit: spills)
          u += 0x7fffu + ((u >> 16) & 1u);
          return __builtin_bit_cast(float, u & 0xffff0000u);
        } else return (float)(_Float16)v;
      };
      auto red32 = [](float v) -> float {                 // 32 token lanes of each hi half -> lanes 31 / 63
        v += dpp<DPP_ROW_SHR(1)>(0.0f, v);
        v += dpp<DPP_ROW_SHR(2)>(0.0f, v);
        v += dpp<DPP_ROW_SHR(4)>(0.0f, v);
        v += dpp<DPP_ROW_SHR(8)>(0.0f, v);
        v += dpp<DPP_ROW_BCAST15, 0xa>(0.0f, v);           // lane 15 of rows 0 / 2 into rows 1 / 3
        return v;
      };
      float csa[16], csb[16];                             // column sums over this wave's 128 tokens of the current half X
      fetch(0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = u & 3, X = u >> 2, set = u & 1;
        if (t == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { csa[r] = 0.0f; csb[r] = 0.0f; }
        }
        if (u + 1 < 8) fetch(u + 1, set ^ 1);
        const int tkn_u = tm0 + wm * 128 + t * 32 + j;
#pragma unroll
        for (int q = 0; q < 2; ++q) {                     // the 8 columns 16 q + 8 hi .. + 7 of this lane's token row, loaded and stored as 16 bytes
          gs_swap32_nv(wa[set][4 * q], wa[set][4 * q + 2]);
          gs_swap32_nv(wa[set][4 * q + 1], wa[set][4 * q + 3]);
          gs_swap32_nv(wb[set][4 * q], wb[set][4 * q + 2]);
          gs_swap32_nv(wb[set][4 * q + 1], wb[set][4 * q + 3]);
          uint32_t pda[4], pdb[4];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {                // accumulator groups 2 q and 2 q + 1
            const int g = 2 * q + gg;
            float a4[4], b4[4], da4[4], db4[4];
            unpack2(wa[set][2 * g], a4[0], a4[1]);
            unpack2(wa[set][2 * g + 1], a4[2], a4[3]);
            unpack2(wb[set][2 * g], b4[0], b4[1]);
            unpack2(wb[set][2 * g + 1], b4[2], b4[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 4 * g + i;
              const float gv = rnd(X ? acc_b[t][r] : acc_a[t][r]);      // d_h as the io-dtype tensor the unfused step materialised
              const float sg = sigmoid(a4[i]);
              da4[i] = rnd(gv * b4[i] * (sg * (1.0f + a4[i] * (1.0f - sg))));
              db4[i] = rnd(gv * (a4[i] * sg));
              csa[r] += da4[i];
              csb[r] += db4[i];
            }
            pda[2 * gg] = gs_pack2_nv<E>(da4[0], da4[1]); pda[2 * gg + 1] = gs_pack2_nv<E>(da4[2], da4[3]);
            pdb[2 * gg] = gs_pack2_nv<E>(db4[0], db4[1]); pdb[2 * gg + 1] = gs_pack2_nv<E>(db4[2], db4[3]);
          }
          gs_swap32_nv(pda[0], pda[2]); gs_swap32_nv(pda[1], pda[3]);
          gs_swap32_nv(pdb[0], pdb[2]); gs_swap32_nv(pdb[1], pdb[3]);
          const int col = tn0 + X * 128 + wn * 32 + 16 * q + 8 * hi;
          if (tkn_u < p.M && col < p.H) {
            E* dst = (E*)p.h + (int64_t)tkn_u * p.h_rs + col;
            *(gs_u4_a4*)dst = gs_u4_a4{pda[0], pda[1], pda[2], pda[3]};
            *(gs_u4_a4*)(dst + p.H) = gs_u4_a4{pdb[0], pdb[1], pdb[2], pdb[3]};
          }
        }
        if (t == 3 && p.partial) {       // the half's column sums: four row_shr adds + row_bcast15, lanes 31 / 63 write
          float* prow = p.partial + (int64_t)(2 * (tm0 / GS_BM) + wm) * 2 * p.H;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float sa4[4], sb4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { sa4[i] = red32(csa[4 * g + i]); sb4[i] = red32(csb[4 * g + i]); }
            const int col = tn0 + X * 128 + wn * 32 + 8 * g + 4 * hi;
            if (j == 31 && col < p.H) {
              *(float4*)(prow + col) = make_float4(sa4[0], sa4[1], sa4[2], sa4[3]);
              *(float4*)(prow + p.H + col) = make_float4(sb4[0], sb4[1], sb4[2], sb4[3]);
            }
          }
        }
      }
    }
    if (!has_next) break;
    local = nxt;
    acc_init();
  }
  if (!late) __builtin_amdgcn_s_barrier();     // the early group waits for the late group's last COMPUTE segment: equal barrier counts
}

static thread_local int g_gs_hip_error = 0;

template <typename E, int MODE = 0>
static int launch_gemm_swiglu(const GemmSwigluArgs& a, hipStream_t s) {
  auto kern = gemm_swiglu_kernel<E, MODE>;
  const size_t lds = (size_t)GS_SLOTS * GS_STAGE;
  // per call: the attribute belongs to the (kernel, device) pair, and a process may drive several devices
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MXVL_ERR_LAUNCH;
  // one persistent workgroup per CU (256 CUs; LDS and the 2-waves-per-SIMD register budget admit exactly one), fewer for small grids
  const int tiles = 8 * a.ntm_x * a.ntn;
  hipLaunchKernelGGL(kern, dim3(tiles < 256 ? tiles : 256), dim3(GS_NT), lds, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_gs_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_gemm_swiglu_fwd(const mxvl_gemm_swiglu_desc* d, void* hip_stream) {
  if (!d || !d->x || !d->weight || !d->h) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->M <= 0 || d->K <= 0 || d->H <= 0) return MXVL_ERR_SHAPE;
  if (d->K % GS_BK != 0) return MXVL_ERR_UNSUPPORTED;                    // whole 64-wide K steps (the reference: 768 / 1024 / 1536)
  if (d->bias && d->bias_dtype != MXVL_F32 && d->bias_dtype != d->io_dtype) return MXVL_ERR_DTYPE;
  if (d->x_rs % 8 || d->w_rs % 8 || (uintptr_t)d->x % 16 || (uintptr_t)d->weight % 16) return MXVL_ERR_STRIDE;   // 16-byte DMA pieces
  if (d->x_rs < d->K || d->w_rs < d->K || d->h_rs < d->H || (d->ab && d->ab_rs < 2 * (int64_t)d->H)) return MXVL_ERR_STRIDE;
  GemmSwigluArgs a;
  a.M = d->M; a.K = d->K; a.H = d->H; a.bias_f32 = d->bias_dtype == MXVL_F32 ? 1 : 0;
  a.ntm = (d->M + GS_BM - 1) / GS_BM; a.ntn = (d->H + GS_BN - 1) / GS_BN; a.ntm_x = (a.ntm + 7) / 8;
  a.x_rs = d->x_rs; a.w_rs = d->w_rs; a.ab_rs = d->ab_rs; a.h_rs = d->h_rs;
  a.x = d->x; a.w = d->weight; a.bias = d->bias; a.ab = d->ab; a.h = d->h; a.partial = nullptr;
  hipStream_t s = (hipStream_t)hip_stream;
  return d->io_dtype == MXVL_BF16 ? launch_gemm_swiglu<bf16_t>(a, s) : launch_gemm_swiglu<f16_t>(a, s);
}

/* see include/mxvl.h: mxvl_gemm_swiglu_bwd */
extern "C" int mxvl_gemm_swiglu_bwd(const mxvl_gemm_swiglu_bwd_desc* d, void* hip_stream) {
  if (!d || !d->dy || !d->w3t || !d->ab || !d->dab) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->M <= 0 || d->K <= 0 || d->H <= 0) return MXVL_ERR_SHAPE;
  if (d->K % GS_BK != 0 || d->H % 8 != 0) return MXVL_ERR_UNSUPPORTED;
  if (d->dy_rs % 8 || d->w_rs % 8 || (uintptr_t)d->dy % 16 || (uintptr_t)d->w3t % 16) return MXVL_ERR_STRIDE;
  if (d->ab_rs % 2 || d->dab_rs % 2 || (uintptr_t)d->ab % 4 || (uintptr_t)d->dab % 4) return MXVL_ERR_STRIDE;
  if (d->dy_rs < d->K || d->w_rs < d->K || d->ab_rs < 2 * (int64_t)d->H || d->dab_rs < 2 * (int64_t)d->H) return MXVL_ERR_STRIDE;
  if (d->partial && ((uintptr_t)d->partial % 16 || d->H % 4)) return MXVL_ERR_STRIDE;
  GemmSwigluArgs a;
  a.M = d->M; a.K = d->K; a.H = d->H; a.bias_f32 = 0;
  a.ntm = (d->M + GS_BM - 1) / GS_BM; a.ntn = (d->H + 2 * GS_BN - 1) / (2 * GS_BN); a.ntm_x = (a.ntm + 7) / 8;
  a.x_rs = d->dy_rs; a.w_rs = d->w_rs; a.ab_rs = d->ab_rs; a.h_rs = d->dab_rs;
  a.x = d->dy; a.w = d->w3t; a.bias = nullptr; a.ab = (void*)d->ab; a.h = d->dab; a.partial = (float*)d->partial;
  hipStream_t s = (hipStream_t)hip_stream;
  return d->io_dtype == MXVL_BF16 ? launch_gemm_swiglu<bf16_t, 1>(a, s) : launch_gemm_swiglu<f16_t, 1>(a, s);
}

extern "C" int mxvl_gemm_swiglu_bwd_partials(int M) { return M > 0 ? 2 * ((M + GS_BM - 1) / GS_BM) : 0; }

/* see include/mxvl.h: mxvl_gemm_nt */
extern "C" int mxvl_gemm_nt(const mxvl_gemm_nt_desc* d, void* hip_stream) {
  if (!d || !d->a || !d->b || !d->c) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->M <= 0 || d->K <= 0 || d->N <= 0) return MXVL_ERR_SHAPE;
  if (d->K % GS_BK != 0 || d->N % 8 != 0) return MXVL_ERR_UNSUPPORTED;
  if (d->bias && d->bias_dtype != MXVL_F32 && d->bias_dtype != d->io_dtype) return MXVL_ERR_DTYPE;
  if (d->a_rs % 8 || d->b_rs % 8 || (uintptr_t)d->a % 16 || (uintptr_t)d->b % 16) return MXVL_ERR_STRIDE;
  if (d->c_rs % 2 || (uintptr_t)d->c % 4 || d->a_rs < d->K || d->b_rs < d->K || d->c_rs < d->N) return MXVL_ERR_STRIDE;
  GemmSwigluArgs a;
  a.M = d->M; a.K = d->K; a.H = d->N; a.bias_f32 = d->bias_dtype == MXVL_F32 ? 1 : 0;
  a.ntm = (d->M + GS_BM - 1) / GS_BM; a.ntn = (d->N + 2 * GS_BN - 1) / (2 * GS_BN); a.ntm_x = (a.ntm + 7) / 8;
  a.x_rs = d->a_rs; a.w_rs = d->b_rs; a.ab_rs = 0; a.h_rs = d->c_rs;
  a.x = d->a; a.w = d->b; a.bias = d->bias; a.ab = nullptr; a.h = d->c; a.partial = nullptr;
  hipStream_t s = (hipStream_t)hip_stream;
  return d->io_dtype == MXVL_BF16 ? launch_gemm_swiglu<bf16_t, 2>(a, s) : launch_gemm_swiglu<f16_t, 2>(a, s);
}
