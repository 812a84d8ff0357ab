// decode_gemm.h -- the decoder step's projections for 9..80 rows (batch x beams) on the matrix cores, gfx950 only.
//
// The reference decodes reports at val_batch_size 6 x beam 3 = 18 rows (CXPMRG_Bench_MambaXray_VL/launch/launch_mambaclip_chexpert.sh:23),
// 8 x 3 = 24 (launch_mambaclip_test_cheXpert.sh:26) and 16 x 5 = 80 (launch_mambaclip_test_iu.sh:26-27) through
// models/MambaXrayVL_DownStream.py:292-301 -> HF generate -> one cuBLAS GEMM per nn.Linear and token.  At those row counts a
// projection is still a stream of its bf16 weight matrix (13.5 GB per token for Llama-2-7B) against 18..80 activation rows that
// live in L2: HBM-bound, ~2 % of the MFMA rate.  So the kernel is shaped by the weight stream, not by the matrix cores:
//
//   y[m][n] = epi( sum_k W[n][k] * x[m][k] )        W (N, K) bf16 row-major (nn.Linear), x (rows, K) bf16, fp32 accumulate
//
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand (16 output columns n x 32 k) and the activations as B
//     (32 k x 16 rows m): the A fragment of lane l is 16 contiguous bytes of weight row n0 + l%16 at k0 + 8*(l/16) -- it is
//     loaded from HBM straight into the MFMA's registers (non-temporal 16-byte loads, 16 rows x 64 contiguous bytes per
//     instruction), never through LDS.  B fragments are the same pattern on the activation rows (L2 hits).
//   * a workgroup owns R 16-column tiles over the WHOLE K; its NW waves split K, so the weight rows of a tile are read as NW
//     contiguous segments, PF k-steps (PF x (R + MT) 16-byte loads per lane) in flight per wave, no barrier and no LDS in the loop.
//     One LDS reduction over the NW partial tiles at the end, then the epilogue of mxvl_decode_gemv (bias, bf16 rounding before
//     the residual add, SwiGLU across the gate / up tiles of one workgroup, fp32 logits).  No split-K across workgroups: a
//     cross-workgroup seam costs 5-13 us (agent-scope fences) on kernels that run 7-40 us.
//   * activation traffic from L2 per weight byte = MT / R (MT = 16-row activation tiles): R is chosen by the host from N.
#pragma once
#include <type_traits>
#include <utility>
#include "decode_elt.h"

namespace mxvl {

typedef float dg_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int dg_u32x4 __attribute__((ext_vector_type(4)));

struct DecodeGemmArgs {
  int rows, K, N, swiglu, out_f32;
  const uint16_t *x, *W, *W2, *bias, *res;
  void* y;
  float* split_acc;     // K split over gridDim.y workgroups: fp32 (gridDim.y, rows, N) PARTIAL sums, slice blockIdx.y fully written by
                        // plain stores; the consumer (mxvl_decode_rmsnorm) adds the slices in a fixed order -- deterministic, unlike the
                        // fp32 atomics of round 4 whose order decided the last bit of every residual row
  const uint16_t* g;    // RMSNorm gain (K) of the activation rows, fused (NORM instantiations of the LDS-DMA kernel); else NULL
  float eps;
  float g_scale, g_inv;   // NORM: power-of-two scale on the gain (mxvl_gemv_desc.norm_gain_scale) and its reciprocal, folded into rstd
};

// Reduction over the waves' K split + the epilogue of mxvl_decode_gemv, one round per output tile (SwiGLU: per gate / up pair).
// acc[r][mt] is lane (col = l%16 -> activation row, 4 * (l/16) + v -> weight row) of the 16x16 tile (weight tile r, row tile mt).
// s_ss (NORM): [NW][MT * 16] sums of squares of the activation rows over each wave's K slice -- RMSNorm's statistics, gathered
// from the fragments the MFMAs consumed; the epilogue scales a row's sums by rstd = rsqrt(mean(x^2) + eps).
template <typename E, int MT, int R, int NW>
__device__ __forceinline__ void dg_reduce_epilogue(const DecodeGemmArgs& p, const dg_f32x4 (&acc)[R][MT], float* dg_red, int n0,
                                                   const float* s_ss = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = p.N;
  constexpr int TPR = 2;                                   // tiles per round (the second one only with SwiGLU)
  const int rounds = p.swiglu ? R / 2 : R;
  for (int rd = 0; rd < rounds; ++rd) {
    if (rd) __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ts = p.swiglu ? (r >= R / 2 ? 1 : 0) : 0;
      const int tr = p.swiglu ? (r % (R / 2 > 0 ? R / 2 : 1)) : r;
      if (tr == rd) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          *(dg_f32x4*)(dg_red + ((size_t)((wave * TPR + ts) * MT + mt) * 64 + lane) * 4) = acc[r][mt];
      }
    }
    __syncthreads();
    for (int e = tid; e < MT * 256; e += NW * 64) {
      const int m = e >> 4, nl = e & 15;
      const int n = n0 + rd * 16 + nl;
      if (m >= p.rows || n >= N) continue;
      const int mt = m >> 4, src_lane = (m & 15) + 16 * (nl >> 2), v = nl & 3;
      float s0 = 0.0f, s1 = 0.0f;
      const float* part = dg_red + ((size_t)mt * 64 + src_lane) * 4 + v;
#pragma unroll
      for (int w = 0; w < NW; ++w) s0 += part[(size_t)(w * TPR) * MT * 256];
      if (p.swiglu) {
#pragma unroll
        for (int w = 0; w < NW; ++w) s1 += part[(size_t)(w * TPR + 1) * MT * 256];
      }
      if (s_ss) {
        float ss = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) ss += s_ss[w * MT * 16 + m];
        const float rstd = rsqrtf(ss / (float)p.K + p.eps) * p.g_inv;
        s0 *= rstd;
        s1 *= rstd;
      }
      const size_t o = (size_t)m * N + n;
      if (p.split_acc) {
        p.split_acc[(size_t)blockIdx.y * p.rows * N + o] = s0;
      } else if (p.swiglu) {   // bf16(bf16(silu(gate)) * up), gate / up rounded to bf16 first (what the torch modules do)
        const float gte = E::rr(s0), up = E::rr(s1);
        ((uint16_t*)p.y)[o] = E::r(E::rr(gte * sigmoid(gte)) * up);
      } else {
        float val = s0;
        if (p.bias) val += E::f(p.bias[n]);
        if (p.res) val = E::rr(val) + E::f(p.res[o]);   // the linear output rounds to bf16 before the residual add
        if (p.out_f32) ((float*)p.y)[o] = val; else ((uint16_t*)p.y)[o] = E::r(val);
      }
    }
  }
}

// MT: 16-row activation tiles (rows <= 16 * MT), R: 16-column weight tiles per workgroup (with SwiGLU: R/2 gate tiles + the
// R/2 up tiles of the same columns), NW: waves per workgroup (they split K), PF: k-steps in flight per wave.
template <typename E, int MT, int R, int NW, int PF>
__global__ __launch_bounds__(NW * 64) void decode_gemm_kernel(const DecodeGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float dg_red[];   // [NW][TPR][MT][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N;
  const int cols_per_wg = (p.swiglu ? R / 2 : R) * 16;
  const int n0 = blockIdx.x * cols_per_wg;

  // per-lane row bases (rows past the end re-read the last valid row: their results are never stored)
  const uint16_t* wrow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int t = p.swiglu ? (r % (R / 2 > 0 ? R / 2 : 1)) : r;
    const uint16_t* base = (p.swiglu && r >= R / 2) ? p.W2 : p.W;
    int n = n0 + t * 16 + l16;
    n = n < N ? n : N - 1;
    wrow[r] = base + (size_t)n * K + q * 8;
  }
  const uint16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = mt * 16 + l16;
    m = m < p.rows ? m : p.rows - 1;
    xrow[mt] = p.x + (size_t)m * K + q * 8;
  }
  // k-steps of 32 columns, dealt to the waves in contiguous runs
  const int steps = (K + 31) >> 5;
  const int nsl = NW * gridDim.y;                  // K slices: the waves of the gridDim.y workgroups that share these columns
  const int spw = (steps + nsl - 1) / nsl;
  const int s_begin = (blockIdx.y * NW + wave) * spw;
  const int s_end = s_begin + spw < steps ? s_begin + spw : steps;
  const int iters = (spw + PF - 1) / PF;        // the same trip count for every wave

  dg_u32x4 a[PF][R], b[PF][MT];
  dg_f32x4 acc[R][MT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[r][mt] = dg_f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // every load is unconditional (an invalid step re-reads the row's first bytes and its weight fragment is zeroed before the
  // MFMA): behind a per-lane or run-time-uniform branch hipcc's static s_waitcnt placement waits for a load where it is issued
  auto issue = [&](int slot, int s) {
    const bool ok = s < s_end && (s * 32 + q * 8) < K;
    const int off = ok ? s * 32 : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) a[slot][r] = __builtin_nontemporal_load((const dg_u32x4*)(wrow[r] + off));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[slot][mt] = *(const dg_u32x4*)(xrow[mt] + off);
  };
  auto consume = [&](int slot, int s) {
    const bool ok = s < s_end && (s * 32 + q * 8) < K;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const dg_u32x4 av = ok ? a[slot][r] : dg_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[r][mt] = E::mfma32(av, b[slot][mt], acc[r][mt]);
    }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    issue(j, s_begin + j);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int i = 0; i + 1 < iters; ++i) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int s = s_begin + i * PF + j;
      consume(j, s);
      issue(j, s + PF);
      __builtin_amdgcn_sched_barrier(0);      // keep the slots' requests in slot order: loads return in order, and hipcc otherwise
    }                                         // clusters them by operand, which puts slot 0's activations behind every weight load
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) consume(j, s_begin + (iters - 1) * PF + j);

  dg_reduce_epilogue<E, MT, R, NW>(p, acc, dg_red, n0);
}

// ---- K % 64 == 0: the weight stream through LDS-DMA -------------------------------------------------------------------------------
// Same decomposition (a workgroup owns R tiles over its K slice, the waves split it, one reduction at the end), but the weight
// tiles never touch a VGPR on their way in: a wave owns a ring of PF stages in LDS, a stage = R tiles x (16 rows x 128 bytes = 64
// columns); one global_load_lds_dwordx4 covers 8 rows x 128 contiguous bytes (lane i: row i / 8, 16-byte unit i % 8, stored at
// unit ^ key(row) so the ds_read_b128 fragment reads are conflict-free).  Measured on the weight stream alone
// (tools/ubench/wstream_probe.hip, N = 32000): 6.0-6.3 TB/s against 5.5 for the 16 rows x 64 bytes of the direct loads, and the
// registers the direct ring needed go to R = 4 tiles (activation re-reads per weight byte = MT / 4).
// The activation fragments of a stage are requested right behind its DMA, ALSO from inline asm: vmcnt counts DMA and loads in
// issue order, and hipcc -- which cannot see the DMA -- would wait for a compiler-visible fragment load with a count that drains
// the whole ring.  So every vector-memory wait of the loop is the explicit one in front of a stage's first MFMA.
__device__ __forceinline__ int dg_key(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

template <int N> __device__ __forceinline__ void dg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): loop bodies whose wait counts are immediates
template <typename F, int... I> __device__ __forceinline__ void dg_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void dg_static_for(F&& f) {
  dg_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// NORM (round 5): RMSNorm of the activation rows fused into the projection that consumes them -- y = rstd[m] * sum_k W[n][k] (g[k] x[m][k]):
// the gain is applied to the activation fragments on their way into the B operand (two more 16-byte requests per stage: the gain's
// k-slices), the squares of the RAW fragments are summed beside the MFMAs (the VALU idles in this kernel), and the row's
// rsqrt(mean(x^2) + eps) scales the sums in the epilogue.  That removes mxvl_decode_rmsnorm's 65 launches per token (4.8 us each:
// a tenth of the batch-1 token) without the round-4 prologue variant's cost (a memory round trip + two barriers in front of every
// workgroup's first MFMA while its weight ring is full: rejected, 3.13 -> 3.39 ms).  Rounding: the modules compute
// bf16(bf16(x * rstd) * g) before the product; here g x is rounded once and rstd stays in fp32 -- one rounding less, not bit-equal.
template <typename E, int MT, int R, int NW, int PF, bool NORM = false>
__global__ __launch_bounds__(NW * 64) void decode_gemm_dma_kernel(const DecodeGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char dg_smem[];
  constexpr int STAGE = R * 2048, OPS = 2 * R + 2 * MT + (NORM ? 2 : 0);          // vector-memory operations per stage
  static_assert((PF - 1) * OPS <= 63, "vmcnt is a 6-bit field");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N;
  const int cols_per_wg = (p.swiglu ? R / 2 : R) * 16;
  const int n0 = blockIdx.x * cols_per_wg;
  char* ring = dg_smem + wave * PF * STAGE;
  const char* src[R][2];                    // DMA sources: call h of tile r covers rows 8 h .. 8 h + 7
  {
    const int rl = lane >> 3, ul = lane & 7;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = p.swiglu ? (r % (R / 2 > 0 ? R / 2 : 1)) : r;
      const uint16_t* base = (p.swiglu && r >= R / 2) ? p.W2 : p.W;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 8 + rl;
        int n = n0 + t * 16 + row;
        n = n < N ? n : N - 1;
        src[r][h] = (const char*)(base + (size_t)n * K) + ((ul ^ dg_key(row)) << 4);
      }
    }
  }
  const uint16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = mt * 16 + l16;
    m = m < p.rows ? m : p.rows - 1;
    xrow[mt] = p.x + (size_t)m * K + q * 8;
  }
  const int chunks = K >> 6;                       // 64 columns = 128 bytes of a weight row
  const int nsl = NW * gridDim.y;
  const int cpw = (chunks + nsl - 1) / nsl;
  const int c_begin = (blockIdx.y * NW + wave) * cpw;
  const int c_end = c_begin + cpw < chunks ? c_begin + cpw : chunks;
  const int n_it = (cpw + PF - 1) / PF;            // the same trip count for every wave

  dg_f32x4 acc[R][MT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[r][mt] = dg_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  dg_u32x4 xb[PF][2][MT];
  dg_u32x4 gb[PF][2];                              // NORM: the gain's k-slices of a stage
  float ss[MT];                                    // NORM: sum of squares of row mt * 16 + l16 over this lane's k-slices
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ss[mt] = 0.0f;
  const uint16_t* grow = NORM ? p.g + q * 8 : nullptr;

  auto issue = [&](int slot, int c) {
    const int cc = c < c_end ? c : 0;              // a chunk past the wave's run re-reads chunk 0; its activations are zeroed
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + slot * STAGE));
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const char* g0 = src[r][0] + (size_t)cc * 128;
      const char* g1 = src[r][1] + (size_t)cc * 128;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\t"
                   "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(dst + r * 2048), "v"(g0), "v"(g1) : "memory", "scc");
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(xb[slot][ks][mt]) : "v"(xrow[mt] + cc * 64 + ks * 32) : "memory");
    if constexpr (NORM) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(gb[slot][ks]) : "v"(grow + cc * 64 + ks * 32) : "memory");
    }
  };
  auto consume = [&](int slot, int c) {
    // (the wait for this stage was issued by the caller: its count depends on how many younger stages are in flight)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        asm volatile("" : "+v"(xb[slot][ks][mt]));           // the fragments are defined HERE, behind the wait
        if (c >= c_end) xb[slot][ks][mt] = dg_u32x4{0u, 0u, 0u, 0u};
      }
    if constexpr (NORM) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        asm volatile("" : "+v"(gb[slot][ks]));
        float gf[8];
        elt_unpack8<E>(make_uint4(gb[slot][ks].x, gb[slot][ks].y, gb[slot][ks].z, gb[slot][ks].w), gf);
#pragma unroll
        for (int j = 0; j < 8; ++j) gf[j] *= p.g_scale;      // a power of two: exact (fp16 range of g * x)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float xf[8];
          const dg_u32x4 raw = xb[slot][ks][mt];
          elt_unpack8<E>(make_uint4(raw.x, raw.y, raw.z, raw.w), xf);
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ss[mt] = fmaf(xf[2 * j], xf[2 * j], fmaf(xf[2 * j + 1], xf[2 * j + 1], ss[mt]));
            o[j] = (uint32_t)E::r(xf[2 * j] * gf[2 * j]) | ((uint32_t)E::r(xf[2 * j + 1] * gf[2 * j + 1]) << 16);
          }
          xb[slot][ks][mt] = dg_u32x4{o[0], o[1], o[2], o[3]};
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int unit = (ks * 4 + q) ^ dg_key(l16);
        const dg_u32x4 av = *(const dg_u32x4*)(ring + slot * STAGE + r * 2048 + l16 * 128 + (unit << 4));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[r][mt] = E::mfma32(av, xb[slot][ks][mt], acc[r][mt]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot's fragment reads have returned: it may be refilled
  };
#pragma unroll
  for (int j = 0; j < PF; ++j) issue(j, c_begin + j);
  for (int i = 0; i + 1 < n_it; ++i) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int c = c_begin + i * PF + j;
      dg_wait_vm<(PF - 1) * OPS>();
      consume(j, c);
      issue(j, c + PF);
    }
  }
  {
    const int c = c_begin + (n_it - 1) * PF;
    if constexpr (PF >= 1) { dg_wait_vm<(PF - 1) * OPS>(); consume(0, c); }
    if constexpr (PF >= 2) { dg_wait_vm<(PF - 2) * OPS>(); consume(1, c + 1); }
    if constexpr (PF >= 3) { dg_wait_vm<(PF - 3) * OPS>(); consume(2, c + 2); }
    if constexpr (PF >= 4) { dg_wait_vm<(PF - 4) * OPS>(); consume(3, c + 3); }
    static_assert(PF <= 4, "tail is written out for PF <= 4");
  }
  __syncthreads();                                            // every wave is done with its ring: the reduction reuses the memory
  if constexpr (NORM) {
    float* s_ss = (float*)dg_smem + (size_t)NW * 2 * MT * 256;  // behind the reduction tiles
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v = ss[mt];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q == 0) s_ss[wave * MT * 16 + mt * 16 + l16] = v;
    }
    dg_reduce_epilogue<E, MT, R, NW>(p, acc, (float*)dg_smem, n0, s_ss);   // (its first barrier orders the s_ss writes)
  } else {
    dg_reduce_epilogue<E, MT, R, NW>(p, acc, (float*)dg_smem, n0);
  }
}

// ---- 17..80 rows (MT >= 2): the waves split N, the activations are shared through LDS (round 5) ------------------------------------
// With the waves of a workgroup splitting K every wave fetches its own activation fragments from L2: MT / R bytes per weight byte,
// 1.25x the weight stream at 80 rows -- and at MT = 5 only two stages fit the registers.  Measured in round 4
// (profiles/r04_decode_gemm_bench.txt): 80 rows ran the projections at 1.3-3.2 TB/s where hipBLASLt reaches 2.6-4.2, the 48- and
// 80-row tokens at 0.30 / 0.23 of the HBM roofline.  Here a workgroup of NW waves owns NW x R column tiles over ITS K range
// (gridDim.y ranges: o_proj / down_proj write partial planes as before); wave w streams the R tiles of its own columns through a
// private LDS ring exactly as above, and the activation tile of a 64-column chunk (MT x 16 rows x 128 bytes, same XOR swizzle) is
// brought in ONCE per workgroup by LDS-DMA (its 2 MT one-kilobyte pieces dealt round-robin to the waves) and read by every wave as
// the B operand: MT / (NW R) activation bytes per weight byte, no fragment registers, a ring as deep as LDS holds (5..8 stages,
// decode_gemm.hip wide_pf).  One s_barrier per chunk: behind it every wave's pieces of stage i have landed, and every wave is done
// with the slot of stage i - 1, which is refilled right away.  No cross-wave reduction: a wave owns the whole K range of its columns
// and runs the epilogue from registers (a lane's four columns of a row as ONE 16- / 8-byte store).
// What bounds a launch (tools/cu_stream_probe.hip + the kernel with pieces switched off, profiles/r05_cu_stream_probe.txt,
// r05_decode_gemm_wide_attribution.txt): ONE CU moves at most ~55 GB/s of full 128-byte lines through its load path, LDS-DMA or
// registers, hit or miss, weights and activations alike (half-used lines -- the MFMA A layout fetched straight into registers, 64
// bytes per row and instruction -- halve it), HBM gives ~6.5-7 TB/s to >= 128 streaming CUs.  So the activation bytes a CU has to
// pull next to its weight share, and the CUs a grid leaves idle, are what the 80-row projections pay for: NW = 3 where that fills
// the chip (decode_gemm.hip), and the walk over K staggered per workgroup (below).
template <typename E, int MT, int R, int NW, int PF>
__global__ __launch_bounds__(NW * 64) void decode_gemm_wide_kernel(const DecodeGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char dgw_smem[];
  constexpr int ACT = MT * 2048, WST = R * 2048, STAGE = ACT + NW * WST;
  constexpr int AI = (2 * MT + NW - 1) / NW;           // activation DMA instructions per wave and stage (a few waves repeat the last piece)
  constexpr int OPS = 2 * R + AI;
  static_assert(PF >= 2 && (PF - 1) * OPS <= 63, "vmcnt is a 6-bit field");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, q = lane >> 4;
  const int K = p.K, N = p.N;
  const int cols_per_wave = (p.swiglu ? R / 2 : R) * 16;
  const int n0 = (blockIdx.x * NW + wave) * cols_per_wave;       // this wave's first output column
  const int rl = lane >> 3, ul = lane & 7;
  const char* wsrc[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int t = p.swiglu ? (r % (R / 2 > 0 ? R / 2 : 1)) : r;
    const uint16_t* base = (p.swiglu && r >= R / 2) ? p.W2 : p.W;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = h * 8 + rl;
      int n = n0 + t * 16 + row;
      n = n < N ? n : N - 1;
      wsrc[r][h] = (const char*)(base + (size_t)n * K) + ((ul ^ dg_key(row)) << 4);
    }
  }
  const char* asrc[AI];
  unsigned adst[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int a = wave + NW * i;
    a = a < 2 * MT ? a : 2 * MT - 1;
    const int row = a * 8 + rl;                                    // row of the padded MT x 16 tile
    const int m = row < p.rows ? row : p.rows - 1;
    asrc[i] = (const char*)(p.x + (size_t)m * K) + ((ul ^ dg_key(row & 15)) << 4);
    adst[i] = (unsigned)a * 1024u;
  }
  const int chunks = K >> 6;
  const int cpw = (chunks + (int)gridDim.y - 1) / (int)gridDim.y;
  const int c_begin = blockIdx.y * cpw;
  const int c_end = c_begin + cpw < chunks ? c_begin + cpw : chunks;
  const int n_it = c_end - c_begin;

  dg_f32x4 acc[R][MT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[r][mt] = dg_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dgw_smem;

  // Workgroup b starts its walk at chunk (5 b) % n_it of its range and wraps: with everybody at chunk 0 every request in flight on the
  // chip shares address bits 7..12 (rows are K * 2 bytes apart) and the HBM channels take turns instead of working side by side --
  // tools/cu_stream_probe.hip, profiles/r05_cu_stream_probe.txt: 192 workgroups streaming 100 MB reach 5.5 TB/s in lock-step, 6.2 staggered.
  const int stag = n_it > 0 ? (int)((blockIdx.x * 5u) % (unsigned)n_it) : 0;
  auto issue = [&](int slot, int c) {
    int j = c - c_begin + stag;
    j = j >= n_it ? j - n_it : j;
    int cc = c < c_end ? c_begin + j : c_begin;       // past the range: a harmless repeat into a slot nobody reads (the counts stay uniform)
    cc = cc < chunks ? cc : chunks - 1;
    const unsigned sbase = lds0 + (unsigned)slot * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(sbase + adst[i]), "v"(asrc[i] + (size_t)cc * 128) : "memory", "scc");
    }
    const unsigned wdst = sbase + ACT + (unsigned)wave * WST;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\t"
                   "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(wdst + r * 2048), "v"(wsrc[r][0] + (size_t)cc * 128), "v"(wsrc[r][1] + (size_t)cc * 128) : "memory", "scc");
    }
  };
  auto consume = [&](int slot) {
    const char* act = dgw_smem + slot * STAGE;
    const char* wt = act + ACT + wave * WST;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int unit = ((ks * 4 + q) ^ dg_key(l16)) << 4;
      dg_u32x4 bf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bf[mt] = *(const dg_u32x4*)(act + (mt * 16 + l16) * 128 + unit);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const dg_u32x4 av = *(const dg_u32x4*)(wt + r * 2048 + l16 * 128 + unit);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = E::mfma32(av, bf[mt], acc[r][mt]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int j = 0; j < PF - 1; ++j) issue(j, c_begin + j);
  int slot = 0, fill = PF - 1;
  const int n_main = n_it - (PF - 1);                  // iterations that still have a stage to ask for
  for (int i = 0; i < n_main; ++i) {
    dg_wait_vm<(PF - 2) * OPS>();                      // stage i has landed (this wave's pieces); PF - 2 younger stages stay in flight
    __builtin_amdgcn_s_barrier();                      // ... and everybody's; every wave is done with the slot of stage i - 1
    asm volatile("" ::: "memory");
    issue(fill, c_begin + i + PF - 1);
    consume(slot);
    slot = slot + 1 == PF ? 0 : slot + 1;
    fill = fill + 1 == PF ? 0 : fill + 1;
  }
  // the last PF - 1 stages: nothing left to ask for (the deep rings used to re-request a chunk per iteration here, a tenth of the
  // launch's LDS-DMA instructions at 64 chunks, a third at 16), the wait counts shrink with the ring
  const int i_tail = n_main > 0 ? n_main : 0;
  dg_static_for<PF - 1>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if (i_tail + t < n_it) {                            // (uniform)
      dg_wait_vm<(PF - 2 - t) * OPS>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      consume(slot);
      slot = slot + 1 == PF ? 0 : slot + 1;
    }
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (ranges shorter than the ring: the prologue's repeats may not land after the workgroup is gone)

  // ---- epilogue from registers: lane (l16 -> activation row, 4 q + v -> column of the tile) -------------------------------------------
  const int tiles = p.swiglu ? R / 2 : R;
#pragma unroll
  for (int t = 0; t < (R > 1 ? R : 1); ++t) {
    if (t >= tiles) break;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + l16;
      if (m >= p.rows) continue;
      const int nq = n0 + t * 16 + 4 * q;
      if ((N & 3) == 0 && nq + 3 < N) {             // the lane's four columns as one store (16 bytes of a plane, 8 of a 16-bit row)
        const size_t o = (size_t)m * N + nq;
        const dg_f32x4 s0 = acc[t][mt];
        if (p.split_acc) {
          *(dg_f32x4*)(p.split_acc + (size_t)blockIdx.y * p.rows * N + o) = s0;
        } else if (p.swiglu) {
          const dg_f32x4 s1 = acc[(t + R / 2) < R ? t + R / 2 : t][mt];
          uint16_t h[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float gte = E::rr(s0[v]), up = E::rr(s1[v]);
            h[v] = E::r(E::rr(gte * sigmoid(gte)) * up);
          }
          *(uint2*)((uint16_t*)p.y + o) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        } else {
          float val[4] = {s0[0], s0[1], s0[2], s0[3]};
          if (p.bias) {
            const uint2 bw = *(const uint2*)(p.bias + nq);
            val[0] += E::lo(bw.x); val[1] += E::hi(bw.x); val[2] += E::lo(bw.y); val[3] += E::hi(bw.y);
          }
          if (p.res) {
            const uint2 rw = *(const uint2*)(p.res + o);
            val[0] = E::rr(val[0]) + E::lo(rw.x); val[1] = E::rr(val[1]) + E::hi(rw.x);
            val[2] = E::rr(val[2]) + E::lo(rw.y); val[3] = E::rr(val[3]) + E::hi(rw.y);
          }
          if (p.out_f32) *(dg_f32x4*)((float*)p.y + o) = dg_f32x4{val[0], val[1], val[2], val[3]};
          else *(uint2*)((uint16_t*)p.y + o) = make_uint2((uint32_t)E::r(val[0]) | ((uint32_t)E::r(val[1]) << 16), (uint32_t)E::r(val[2]) | ((uint32_t)E::r(val[3]) << 16));
        }
        continue;
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = n0 + t * 16 + 4 * q + v;
        if (n >= N) continue;
        const float s0 = acc[t][mt][v];
        const size_t o = (size_t)m * N + n;
        if (p.split_acc) {
          p.split_acc[(size_t)blockIdx.y * p.rows * N + o] = s0;
        } else if (p.swiglu) {
          const float s1 = acc[(t + R / 2) < R ? t + R / 2 : t][mt][v];
          const float gte = E::rr(s0), up = E::rr(s1);
          ((uint16_t*)p.y)[o] = E::r(E::rr(gte * sigmoid(gte)) * up);
        } else {
          float val = s0;
          if (p.bias) val += E::f(p.bias[n]);
          if (p.res) val = E::rr(val) + E::f(p.res[o]);
          if (p.out_f32) ((float*)p.y)[o] = val; else ((uint16_t*)p.y)[o] = E::r(val);
        }
      }
    }
  }
}

// RMSNorm of the activation rows ahead of a projection (Qwen2RMSNorm / LlamaRMSNorm, EMRRG/models/hybrid_decoder_layer.py:185-199):
// y = bf16( bf16(x * rsqrt(mean(x^2) + eps)) * g ), statistics in fp32.  One workgroup per row.  (At <= 8 rows the GEMV kernel
// does this in its prologue, redundantly per workgroup; at 18..80 rows that redundancy would cost more than the weight stream.)
// Fold mode (acc != NULL): the row is first completed from a K-split projection's fp32 sums -- x = bf16(acc) + residual, the two
// roundings of `residual + linear(...)` in the modules -- written to x_out, and acc is cleared for the next projection.  That is the
// epilogue of o_proj / down_proj (N = 4096: too few columns to fill 256 CUs without splitting K) in the kernel that has to follow
// them anyway, instead of a cross-workgroup seam (agent-scope fences, 5-13 us) inside a 7-20 us projection.
struct RmsNormArgs {
  int rows, K, splits;
  float eps;
  const uint16_t *x, *g;
  uint16_t* y;
  float* acc;
  const uint16_t* res;
  uint16_t* x_out;
};
// (Tried and removed, profiles/r04_decode_fused_norm_rejected_timeline.txt: for rows <= 8 the norm -- and the fold -- in the PROLOGUE of the
//  consuming projection, rows in LDS, no RMSNorm launches at all: 164 instead of 229 launches per token, but every workgroup of
//  every projection then starts with a memory round trip + two barriers while its weight ring is already full -- gate / up 32 -> 42 us,
//  qkv 17.5 -> 24 us, the token 3.13 -> 3.39 ms.  A launch boundary costs less than a stalled stream on 256 CUs.)
// 1024 threads per row, 8 columns per thread and trip (K <= 16384 in two trips): EVERY load of the row -- the fp32 sums, the
// residual, the gain -- is requested before anything is stored or reduced; the first version walked the row in 256-thread trips of
// load -> store -> load, three to four serialised round trips in a kernel that runs 65 times per token (4.9 us each in the
// step's trace, a tenth of the batch-1 token).
// S: planes of the K-split sums as a compile-time constant (1, 2, 4: every load of every plane is requested up front -- as a run-time
// loop the planes were three more DEPENDENT round trips, +1.9 us on each of the 65 folds of a token); 0: any count, run-time loop
template <typename E, int S = 1>
__global__ __launch_bounds__(1024) void decode_rmsnorm_kernel(const RmsNormArgs p) {
  constexpr int NT = 1024, MAXV = 2, SU = S > 0 ? S : 1;
  __shared__ float s_part[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = blockIdx.x;
  uint4 xr[MAXV], gr[MAXV], rr[MAXV];
  float4 a0[MAXV], a1[MAXV];
  bool on[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int k = (i * NT + tid) * 8;
    on[i] = k < p.K;
    const int kc = on[i] ? k : 0;                       // unconditional loads (clamped): nothing waits behind a branch
    gr[i] = *(const uint4*)(p.g + kc);
    if (p.acc) {
      const float* ap = p.acc + (size_t)m * p.K + kc;
      const size_t slice = (size_t)p.rows * p.K;
      float4 b0[SU], b1[SU];
#pragma unroll
      for (int sp = 0; sp < SU; ++sp) {                // all planes of this thread's columns in flight at once
        b0[sp] = *(const float4*)(ap + sp * slice);
        b1[sp] = *(const float4*)(ap + sp * slice + 4);
      }
      rr[i] = *(const uint4*)(p.res + (size_t)m * p.K + kc);
      a0[i] = b0[0];
      a1[i] = b1[0];
#pragma unroll
      for (int sp = 1; sp < SU; ++sp) {                // summed in the order 0, 1, 2, ...: deterministic
        a0[i].x += b0[sp].x; a0[i].y += b0[sp].y; a0[i].z += b0[sp].z; a0[i].w += b0[sp].w;
        a1[i].x += b1[sp].x; a1[i].y += b1[sp].y; a1[i].z += b1[sp].z; a1[i].w += b1[sp].w;
      }
      if constexpr (S == 0) {
        for (int sp = 1; sp < p.splits; ++sp) {
          const float4 c0 = *(const float4*)(ap + sp * slice), c1 = *(const float4*)(ap + sp * slice + 4);
          a0[i].x += c0.x; a0[i].y += c0.y; a0[i].z += c0.z; a0[i].w += c0.w;
          a1[i].x += c1.x; a1[i].y += c1.y; a1[i].z += c1.z; a1[i].w += c1.w;
        }
      }
    } else {
      xr[i] = *(const uint4*)(p.x + (size_t)m * p.K + kc);
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (p.acc) {
      const float av[8] = {a0[i].x, a0[i].y, a0[i].z, a0[i].w, a1[i].x, a1[i].y, a1[i].z, a1[i].w};
      const uint32_t rw[4] = {rr[i].x, rr[i].y, rr[i].z, rr[i].w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = E::rr(av[2 * j]) + E::f((uint16_t)rw[j]);
        const float hi = E::rr(av[2 * j + 1]) + E::f((uint16_t)(rw[j] >> 16));
        o[j] = (uint32_t)E::r(lo) | ((uint32_t)E::r(hi) << 16);
      }
      xr[i] = make_uint4(o[0], o[1], o[2], o[3]);
      if (on[i]) {
        const int k = (i * NT + tid) * 8;
        *(uint4*)(p.x_out + (size_t)m * p.K + k) = xr[i];
      }
    }
    if (!on[i]) xr[i] = make_uint4(0, 0, 0, 0);
    const uint32_t w[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = E::f((uint16_t)w[j]), b = E::f((uint16_t)(w[j] >> 16));
      s = fmaf(a, a, fmaf(b, b, s));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) s_part[wave] = s;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) tot += s_part[w];
  const float rstd = rsqrtf(tot / (float)p.K + p.eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (on[i]) {
      const int k = (i * NT + tid) * 8;
      const uint32_t w[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w}, gw[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = E::rr(E::f((uint16_t)w[j]) * rstd) * E::f((uint16_t)gw[j]);
        const float b = E::rr(E::f((uint16_t)(w[j] >> 16)) * rstd) * E::f((uint16_t)(gw[j] >> 16));
        o[j] = (uint32_t)E::r(a) | ((uint32_t)E::r(b) << 16);
      }
      *(uint4*)(p.y + (size_t)m * p.K + k) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

}  // namespace mxvl
