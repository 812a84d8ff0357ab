// gemm_tn.hip -- the weight-gradient GEMM of the training step as ONE MFMA kernel for gfx950 (MI355X, CDNA4):
//
//     C (M, N) fp32  (+)=  A^T B        A (K, M), B (K, N) token-major bf16 / fp16, K = tokens (65 280 per GPU at the headline shape)
//
// Replaces, for the TOKEN-MAJOR `nn.Linear` layers of the step (CXPMRG_Bench_MambaXray_VL/arm/Finetuning/models_mamba.py:59-83 SwiGLU
// w1 | w2 | w3; pretrain/models_pretrain.py:45-83 the decoder's q / kv / proj / fc1 / fc2; the ViT-MAE blocks), what autograd's
// `grad_weight = dy^T x` was in the round-2..5 step: a batched library GEMM over S token slices into S fp32 planes plus an `aten::sum`
// over the planes (235 reduce launches and 4.7 ms of the 213 ms step, profiles/r06_step_eager_pretrain.txt; the batched GEMMs
// themselves ran 0.84-1.09 PFLOP/s).  in_proj / out_proj / x_proj / dt_proj (mamba_simple.py:76,91) hold one operand channel-major
// (K-contiguous): the library runs those at 0.95-1.13 PFLOP/s and keeps them.  Where the kernel wins is a measured rule on the host
// side (selective_scan_interface.gemm_tn_wins: token axis >= 32 000, a tile count that fills an XCD's workgroups).
//
// Both operands are K-major (the reduction axis is the SLOW one), which is the layout the NT kernel of gemm_swiglu.hip cannot
// take: an MFMA lane needs 8 consecutive k of ONE output row.  gfx950's transpose read does that on the way out of the LDS:
//   * a K step's A tile is 32 k-rows x 256 m (512 bytes per row), the B tile 32 x 256 n; both are filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KB = two k-rows per instruction, no staging VGPRs) into a ring of four 32 KB stages, THREE K steps
//     ahead of the MFMAs and across unit boundaries (counted vmcnt waits; the first form -- two 64 KB stages, one step ahead -- left
//     0.75 K steps of latency cover and measured 5-12 % slower, profiles/r06_wgrad_tn_bench.txt);
//   * a fragment = two ds_read_b64_tr_b16: each 16-lane group reads a [4 k][16 m] block, lane t' supplies the address of row
//     t' >> 2, columns 4 (t' & 3) .. + 3, and receives column t' of the four rows.  The k order inside an MFMA is whatever the
//     reads deliver -- A and B fragments are read the same way, the sum over k does not care;
//   * bank conflicts: a half-wave of a transposed read touches 4 k-rows x 64 bytes; rows are 512 bytes apart (same bank offset),
//     so the 16-byte units of a row are XOR-swizzled with (k & 3) << 2 -- the four rows then tile the 256-byte bank row exactly.
//     The DMA writes the LDS linearly, so the swizzle is applied to its SOURCE address (as in gemm_swiglu.hip) and to the reads,
//     where it is a per-lane constant (k & 3 == t' >> 2).
// Tiling: 256 x 256 output tiles, 8 waves as 2 (m) x 4 (n), a wave = 128 m x 64 n = 8 accumulators of v_mfma_f32_32x32x16
// (A operand = the A matrix: D rows are m, D columns n, a lane owns one n and 4 x 4 consecutive m); K step = 2 sub-steps of
// [LOAD 12 ds_read_b64_tr_b16] barrier [COMPUTE 8 MFMA] barrier with the two waves of every SIMD one segment apart (the schedule of
// gemm_swiglu.hip).  Measured and dropped (profiles/r06_wgrad_tn_bench.txt): requesting the fragments of sub-step s + 1 at the top of
// the COMPUTE segment of sub-step s (two register sets, 218 VGPRs) -- 17-20 % SLOWER at every shape: the reads then compete with the
// partner wave's LOAD segment for the LDS instead of filling the gap the partner's MFMAs leave; ONE segment pair per K step (24 reads,
// 16 MFMAs, half the barriers) -- 4-6 % slower.
// Split over K.  M x N <= 5504 x 1024 gives 16..88 tiles for 256 CUs, so the token axis is cut into 8 q slices: XCD x takes the
// slices x, x + 8, ... and its 32 persistent workgroups walk (slice, tile) units in step -- the workgroups resident on one XCD
// read the SAME k range of different tiles, i.e. they share the A / B slices through that XCD's L2 (a tile's operands are
// 64 KB per K step and workgroup: 5.9 GB per w1|w2 wgrad without the sharing, 0.85 GB of tensor).  A unit's accumulators leave
// as fp32 atomics into C (coalesced: a register of a half-wave is 32 consecutive n of one row): no partial planes, no reduce.
#include "mxvl_common.h"

namespace mxvl {

typedef float tn_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tn_f16x8 __attribute__((ext_vector_type(8)));
typedef short tn_s16x4 __attribute__((ext_vector_type(4)));

struct GemmTnArgs {
  int M, N, K, ntm, ntn, ntiles, nk, slices, n_fast, nfull, ntail, tailf;
  int64_t a_rs, b_rs, c_rs;     // row strides in elements
  const void *a, *b;
  float* c;
};

constexpr int TN_BM = 256, TN_BN = 256, TN_BK = 32, TN_NT = 512, TN_NST = 4;
constexpr int TN_ROWB = TN_BM * 2;                    // bytes per LDS k-row (512)
constexpr int TN_A_BYTES = TN_BK * TN_ROWB;           // 16 KB
constexpr int TN_STAGE = 2 * TN_A_BYTES;              // 32 KB per K step; a ring of TN_NST = 4 stages, three K steps in flight

template <typename E>
__device__ __forceinline__ tn_f32x16 tn_mma(uint4 a, uint4 b, tn_f32x16 c) {
  if constexpr (__is_same(E, bf16_t))
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tn_bf16x8, a), __builtin_bit_cast(tn_bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tn_f16x8, a), __builtin_bit_cast(tn_f16x8, b), c, 0, 0, 0);
}

// two transposed reads = the 8 k-values (rows r .. r + 3, r + 4 .. r + 7 of the tile image) of this lane's output row
__device__ __forceinline__ uint4 tn_frag(const char* p) {
  typedef __attribute__((address_space(3))) tn_s16x4 lds4;
  const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)p);
  const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)(p + 4 * TN_ROWB));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return make_uint4(a.x, a.y, b.x, b.y);
}

template <typename E>
__global__ __launch_bounds__(TN_NT, 2) void gemm_tn_kernel(const GemmTnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int j = lane & 31, hi = lane >> 5;

  // ---- persistent workgroup: XCD x (= blockIdx.x % 8) owns the K slices x, x + 8, ...; its workgroups take the units
  // (slice-major, tile-minor) w, w + 32, ... so that one round of an XCD is 32 tiles over the same k range
  const int xcd = blockIdx.x & 7, lstride = gridDim.x >> 3;
  // The last, partly filled round of an XCD (ntail < 32 units) is cut into tailf K parts each, so that its rounds are full again.
  const int nunits = p.nfull + p.ntail * p.tailf;
  if ((int)(blockIdx.x >> 3) >= nunits) return;
  auto locate = [&](int u, int& m0, int& n0, int& k0, int& k1) {     // tile origin and K-step range of unit u
    int part = 0;
    if (u >= p.nfull) {
      const int v = u - p.nfull;
      part = v / p.ntail;
      u = p.nfull + (v - part * p.ntail);
    }
    const int sl = xcd + 8 * (u / p.ntiles), tile = u % p.ntiles;
    int mt, nt;
    if (p.n_fast) { mt = tile / p.ntn; nt = tile - mt * p.ntn; }
    else { nt = tile / p.ntm; mt = tile - nt * p.ntm; }
    m0 = mt * TN_BM;
    n0 = nt * TN_BN;
    k0 = (int)((int64_t)sl * p.nk / p.slices);
    k1 = (int)((int64_t)(sl + 1) * p.nk / p.slices);
    if (u >= p.nfull) {
      const int len = k1 - k0;
      k1 = k0 + (part + 1) * len / p.tailf;
      k0 = k0 + part * len / p.tailf;
    }
  };

  // ---- the prefetch cursor walks this workgroup's (unit, K step) sequence three steps ahead of the MFMAs, across unit boundaries.
  // LDS-DMA sources: waves 0-3 fill the A tile, waves 4-7 the B tile; call c of wave (w & 3) = k-rows 2 (4 (w & 3) + c) + {0, 1}.
  // Lane i writes LDS (row 2 cc + (i >> 5), 16-byte slot i & 31) and therefore READS unit (i & 31) ^ ((row & 3) << 2) of that row.
  const char* src[4];
  const int64_t rs_w = wave < 4 ? p.a_rs : p.b_rs;
  int pf_unit = blockIdx.x >> 3, pf_k = 0, pf_end = 0, pf_count = 0;
  bool pf_live = true;
  auto sources = [&](int m0, int n0) {
    const int rl = lane >> 5, sl = lane & 31;
    const int lim = wave < 4 ? p.M : p.N, org = wave < 4 ? m0 : n0;
    const char* base = (const char*)(wave < 4 ? p.a : p.b);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 2 * (4 * (wave & 3) + c) + rl;
      const int u = sl ^ ((row & 3) << 2);
      int col = org + 8 * u;
      col = col < lim ? col : lim - 8;                    // beyond the matrix: any valid 16 bytes (those outputs are never stored)
      src[c] = base + ((int64_t)row * rs_w + col) * 2;
    }
  };
  const int dma_dst = (wave < 4 ? 0 : TN_A_BYTES) + (wave & 3) * 4 * 1024;      // wave-uniform byte offset inside a stage
  // opaque asm (see gemm_swiglu.hip): the only waits on the DMA are the explicit ones below
  auto issue = [&](int buf, int kt) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem) + buf * TN_STAGE + dma_dst;
    const int64_t koff = (int64_t)kt * TN_BK * rs_w * 2;
    const char* g[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) g[c] = src[c] + koff;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g0], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g1], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g2], off\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g3], off\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [dst] "s"(dst), [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3])
        : "memory", "scc");
  };
  auto pf_begin = [&]() {
    int m0, n0;
    locate(pf_unit, m0, n0, pf_k, pf_end);
    sources(m0, n0);
  };
  auto pf_advance = [&]() {                      // request the next K step of the sequence into the next ring stage
    issue(pf_count & (TN_NST - 1), pf_k);
    ++pf_count;
    if (++pf_k == pf_end) {
      pf_unit += lstride;
      if (pf_unit < nunits) pf_begin();
      else pf_live = false;
    }
  };
  // stage `need` has landed once at most (pf_count - need - 1) newer requests of this wave are outstanding (4 instructions each, in
  // order); `drain`: fp32 atomics of an epilogue sit in the same counter -- wait for everything
  auto wait_stage = [&](int need, bool drain) {
    const int newer = pf_count - need - 1;
    if (drain || newer <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };

  // ---- fragment addresses: 16-lane group (lane >> 4) = (k half `hi`) x (row half `rh`); lane t' of it addresses k-row
  // 8 hi + (t' >> 2) (+ 16 ks, + 4 for the second read) and the 8 bytes of columns 16 rh + 4 (t' & 3) .. + 3 of its 32-row tile
  const int tq = lane & 15, rh = (lane >> 4) & 1;
  const int key = (tq >> 2) << 2;
  int offA[4], offB[2];
  {
    const int kro = (8 * hi + (tq >> 2)) * TN_ROWB, inner = (tq & 1) * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = wm * 128 + t * 32 + 16 * rh + 4 * (tq & 3);
      offA[t] = kro + (((col >> 3) ^ key) << 4) + inner;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = wn * 64 + t * 32 + 16 * rh + 4 * (tq & 3);
      offB[t] = TN_A_BYTES + kro + (((col >> 3) ^ key) << 4) + inner;
    }
  }

  tn_f32x16 acc[4][2];
  auto acc_zero = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
  };

  const int late = wave >> 2;
  pf_begin();
#pragma unroll 1
  for (int i = 0; i < TN_NST - 1 && pf_live; ++i) pf_advance();
  acc_zero();
  wait_stage(0, false);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (late) __builtin_amdgcn_s_barrier();
  int gk = 0;                                   // K steps done by this workgroup: ring stage = gk & 3
  bool drain = false;
  for (int unit = blockIdx.x >> 3; unit < nunits; unit += lstride) {
    int tm0, tn0, kbeg, kend;
    locate(unit, tm0, tn0, kbeg, kend);
    for (int kt = kbeg; kt < kend; ++kt, ++gk) {
      const char* st = smem + (gk & (TN_NST - 1)) * TN_STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // LOAD segment
        uint4 fa[4], fb[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = tn_frag(st + offA[t] + ks * 16 * TN_ROWB);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t] = tn_frag(st + offB[t] + ks * 16 * TN_ROWB);
        // the stage read in step gk - 1 is free: the late group finished reading it one segment before the early group gets here
        if (ks == 0 && pf_live) pf_advance();
        if (ks == 1) { wait_stage(gk + 1, drain); drain = false; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // COMPUTE segment
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t][0] = tn_mma<E>(fa[t], fb[0], acc[t][0]);
          acc[t][1] = tn_mma<E>(fa[t], fb[1], acc[t][1]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    // ---- epilogue of unit (tm0, tn0): register 4 g + e of lane (j, hi) is C[tm0 + wm 128 + t 32 + 8 g + 4 hi + e][tn0 + wn 64 + u 32 + j];
    // one atomic instruction = two rows x 32 consecutive floats.  The next unit's first stages are already landing.
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int n = tn0 + wn * 64 + u * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = tm0 + wm * 128 + t * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
          if (m < p.M && n < p.N) atomicAdd(p.c + (int64_t)m * p.c_rs + n, acc[t][u][r]);
        }
      }
    drain = true;
    acc_zero();
  }
  if (!late) __builtin_amdgcn_s_barrier();     // the early group waits for the late group's last COMPUTE segment: equal barrier counts
}

// ---- mxvl_colsum: the bias gradient of a token-major linear, out[c] (fp32 partial rows) = sum over rows of x[r][c] ------------------------
// (autograd's `grad_bias = dy.sum(0)`: torch's reduce kernel runs the tall (tokens, C) 16-bit case at 1.4-1.7 TB/s.)  A wave owns 512
// columns (16 bytes per lane) of one of R row groups, four rows in flight; partial (R, C) fp32, the caller adds the R rows.
template <typename E>
__global__ __launch_bounds__(256) void colsum_kernel(const E* __restrict__ x, float* __restrict__ partial, int rows, int C, int64_t rs, int R) {
  const int lane = threadIdx.x & 63;
  const int tiles = (C + 511) / 512;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (int64_t)tiles * R) return;
  const int tile = (int)(w % tiles), rg = (int)(w / tiles);
  const int c = tile * 512 + lane * 8;
  if (c >= C) return;                                  // C % 8 == 0: a lane's 8 columns are inside or outside as a whole
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
  auto add8 = [&](const uint4 v) {
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (__is_same(E, bf16_t)) {
        acc[2 * k] += __builtin_bit_cast(float, wd[k] << 16);
        acc[2 * k + 1] += __builtin_bit_cast(float, wd[k] & 0xffff0000u);
      } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 hv = __builtin_bit_cast(h2, wd[k]);
        acc[2 * k] += (float)hv.x;
        acc[2 * k + 1] += (float)hv.y;
      }
    }
  };
  const E* base = x + c;
  int r = rg;
  for (; r + 3 * R < rows; r += 4 * R) {               // four independent 16-byte loads in flight per lane
    const uint4 v0 = *(const uint4*)(base + (int64_t)r * rs), v1 = *(const uint4*)(base + (int64_t)(r + R) * rs);
    const uint4 v2 = *(const uint4*)(base + (int64_t)(r + 2 * R) * rs), v3 = *(const uint4*)(base + (int64_t)(r + 3 * R) * rs);
    add8(v0); add8(v1); add8(v2); add8(v3);
  }
  for (; r < rows; r += R) add8(*(const uint4*)(base + (int64_t)r * rs));
  float* pr = partial + (int64_t)rg * C + c;
  *(float4*)pr = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *(float4*)(pr + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// slices per XCD: the q in 1..4 with the fewest (rounds x K steps per slice), at least 8 K steps per slice
constexpr int TN_EPI_STEPS = 20;      // a unit's atomic epilogue + ring refill, in K steps (measured: ~20 us per extra round at 1 us per step)
static int tn_slices_per_xcd(int ntiles, int nk, int forced) {
  if (forced < 0) forced = -forced;
  if (forced >= 1 && forced <= 8 && nk / (8 * forced) >= 1) return forced;
  int best = 1;
  int64_t best_cost = -1;
  for (int q = 1; q <= 4; ++q) {
    if (nk / (8 * q) < 16 && q > 1) break;
    const int64_t rounds = (q * ntiles + 31) / 32, ksl = (nk + 8 * q - 1) / (8 * q);
    const int64_t cost = rounds * (ksl + TN_EPI_STEPS);               // + the atomic epilogue / refill of every unit
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = q; }
  }
  return best;
}

}  // namespace mxvl

using namespace mxvl;

/* see include/mxvl.h: mxvl_gemm_tn */
extern "C" int mxvl_gemm_tn(const mxvl_gemm_tn_desc* d, void* hip_stream) {
  if (!d || !d->a || !d->b || !d->c) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->M <= 0 || d->K <= 0 || d->N <= 0) return MXVL_ERR_SHAPE;
  if (d->K % 64 != 0 || d->K / 64 < 8 || d->M % 8 != 0 || d->N % 8 != 0) return MXVL_ERR_UNSUPPORTED;
  if (d->a_rs % 8 || d->b_rs % 8 || (uintptr_t)d->a % 16 || (uintptr_t)d->b % 16 || (uintptr_t)d->c % 4) return MXVL_ERR_STRIDE;
  if (d->a_rs < d->M || d->b_rs < d->N || d->c_rs < d->N) return MXVL_ERR_STRIDE;
  hipStream_t s = (hipStream_t)hip_stream;
  if (!d->accumulate) {
    if (hipMemset2DAsync(d->c, (size_t)d->c_rs * 4, 0, (size_t)d->N * 4, (size_t)d->M, s) != hipSuccess) return MXVL_ERR_LAUNCH;
  }
  GemmTnArgs a;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.ntm = (d->M + TN_BM - 1) / TN_BM; a.ntn = (d->N + TN_BN - 1) / TN_BN; a.ntiles = a.ntm * a.ntn;
  a.nk = d->K / TN_BK;
  a.slices = 8 * tn_slices_per_xcd(a.ntiles, a.nk, d->slices_per_xcd);
  a.n_fast = a.ntn <= a.ntm ? 1 : 0;
  {
    const int nunits = (a.slices >> 3) * a.ntiles, ksl = a.nk / a.slices;
    a.ntail = nunits % 32;
    a.nfull = nunits - a.ntail;
    a.tailf = 1;
    int64_t best = -1;
    for (int f = 1; f <= 4 && a.ntail > 0 && ksl / f >= 4; ++f) {
      const int64_t cost = (int64_t)((a.ntail * f + 31) / 32) * ((ksl + f - 1) / f + TN_EPI_STEPS);
      if (best < 0 || cost < best) { best = cost; a.tailf = f; }
    }
    if (d->slices_per_xcd < 0) a.tailf = 1;        /* measurement: forced slice count without the tail split */
  }
  a.a_rs = d->a_rs; a.b_rs = d->b_rs; a.c_rs = d->c_rs;
  a.a = d->a; a.b = d->b; a.c = (float*)d->c;
  const size_t lds = (size_t)TN_NST * TN_STAGE;
  const void* kern = d->io_dtype == MXVL_BF16 ? (const void*)gemm_tn_kernel<bf16_t> : (const void*)gemm_tn_kernel<f16_t>;
  // per call: the attribute belongs to the (kernel, device) pair, and a process may drive several devices
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MXVL_ERR_LAUNCH;
  if (d->io_dtype == MXVL_BF16) hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(256), dim3(TN_NT), lds, s, a);
  else hipLaunchKernelGGL(gemm_tn_kernel<f16_t>, dim3(256), dim3(TN_NT), lds, s, a);
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

/* see include/mxvl.h: mxvl_colsum */
extern "C" int mxvl_colsum_partials(int rows, int cols) {
  if (rows <= 0 || cols <= 0 || cols % 8) return 0;
  const int tiles = (cols + 511) / 512;
  int R = 2048 / tiles;                                 /* ~ 8 waves per CU over the whole chip */
  if (R > (rows + 15) / 16) R = (rows + 15) / 16;       /* at least 16 rows per group */
  return R < 1 ? 1 : R;
}

extern "C" int mxvl_colsum(const void* x, void* partial, int rows, int cols, int64_t row_stride, int n_partials, int io_dtype, void* hip_stream) {
  if (!x || !partial) return MXVL_ERR_NULL;
  if (io_dtype != MXVL_BF16 && io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (rows <= 0 || cols <= 0) return MXVL_ERR_SHAPE;
  if (cols % 8 || n_partials != mxvl_colsum_partials(rows, cols)) return MXVL_ERR_UNSUPPORTED;
  if (row_stride < cols || row_stride % 8 || (uintptr_t)x % 16 || (uintptr_t)partial % 16) return MXVL_ERR_STRIDE;
  const int tiles = (cols + 511) / 512;
  const int64_t waves = (int64_t)tiles * n_partials;
  const dim3 grid((unsigned)((waves + 3) / 4));
  hipStream_t s = (hipStream_t)hip_stream;
  if (io_dtype == MXVL_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (float*)partial, rows, cols, row_stride, n_partials);
  else hipLaunchKernelGGL(colsum_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)x, (float*)partial, rows, cols, row_stride, n_partials);
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
