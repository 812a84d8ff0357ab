// decode_gemm.hip -- host side of the 9..80-row decoder projections (kernels: decode_gemm.h) and of mxvl_decode_rmsnorm.
// mxvl_decode_gemv (decode.hip) forwards here for rows > 8, so the stepper has ONE projection entry for every row count the
// reference's launch scripts decode at (MambaXrayVL_DownStream.py:292-301; 3 .. 80 rows).
#include "decode_gemm.h"

namespace mxvl {

// ---- launchers ----------------------------------------------------------------------------------------------------------------------
// K % 64 == 0 (every real decoder: 4096 / 11008 / 3584 / 18944): LDS-DMA weight stream, 4 waves x R tiles, two workgroups per CU.
template <typename E, int MT, int R, int NW, int PF>
static int launch_dma(const DecodeGemmArgs& a, int splits, hipStream_t s) {
  const int cols_per_wg = (a.swiglu ? R / 2 : R) * 16;
  const dim3 grid((a.N + cols_per_wg - 1) / cols_per_wg, splits);
  const size_t ring = (size_t)NW * PF * R * 2048, red = (size_t)NW * 2 * MT * 1024 + (a.g ? (size_t)NW * MT * 64 : 0);
  const size_t lds = ring > red ? ring : red;
  // NORM needs (PF - 1) * (OPS + 2) <= 63 outstanding operations: the ring is one stage shallower where it would not fit
  constexpr bool norm_fits = (PF - 1) * (2 * R + 2 * MT + 2) <= 63;
  void (*kern)(const DecodeGemmArgs) = decode_gemm_dma_kernel<E, MT, R, NW, PF>;
  if (a.g) {
    if constexpr (norm_fits) kern = decode_gemm_dma_kernel<E, MT, R, NW, PF, true>;
    else kern = decode_gemm_dma_kernel<E, MT, R, NW, (PF > 2 ? PF - 1 : 2), true>;
  }
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MXVL_ERR_LAUNCH;   // per call: the attribute is per device
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
  return MXVL_OK;
}

// any K % 8 == 0: fragments loaded straight into the MFMA operands
template <typename E, int MT, int R>
static int launch_direct(const DecodeGemmArgs& a, int splits, hipStream_t s) {
  constexpr int NW = 8, PF = (R + MT <= 4) ? 4 : ((R + MT <= 6) ? 3 : 2);
  const int cols_per_wg = (a.swiglu ? R / 2 : R) * 16;
  const dim3 grid((a.N + cols_per_wg - 1) / cols_per_wg, splits);
  const size_t lds = (size_t)NW * 2 * MT * 1024;
  auto kern = decode_gemm_kernel<E, MT, R, NW, PF>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MXVL_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
  return MXVL_OK;
}

// MT >= 2 (17..80 rows), K % 64 == 0, >= 64 workgroups: the waves split N, activations shared through LDS (decode_gemm_wide_kernel)
// ring depth of the wide kernel: as many stages as 150 KB of LDS hold (one workgroup per CU is all these grids ask for), capped by the
// 6-bit vmcnt field and at 8 -- against the first version's 3 stages: 48-row token +5 %, 80-row +0.7 % (profiles/r05_decode_gemm_ring_ab.txt)
constexpr int wide_pf(int MT, int R, int NW) {
  const int stage = MT * 2048 + NW * R * 2048, ops = 2 * R + (2 * MT + NW - 1) / NW;
  int pf = (150 * 1024) / stage;
  while (pf > 2 && (pf - 1) * ops > 63) --pf;
  return pf > 8 ? 8 : (pf < 2 ? 2 : pf);
}
static int g_wide_pf3 = 0;     // (A/B: mxvl_set_decode_gemm_wide(3) = the 3-stage ring of the first version)
static int g_wide_nw4 = 0;     // (A/B: mxvl_set_decode_gemm_wide(4) = four waves per workgroup everywhere)
static int g_wide_min_mt = 2;  // wide from 17 rows on (A/B: mode 5 = from 33 rows on, as first shipped; mode 6 = from 1 row on)
static int g_wide_min_g = 64;  // ... for grids of at least 64 workgroups (A/B: mode 5 = 160)

template <typename E, int MT, int R, int NW, int PF>
static int launch_wide_pf(const DecodeGemmArgs& a, int splits, hipStream_t s) {
  const int cols_per_wg = NW * (a.swiglu ? R / 2 : R) * 16;
  const dim3 grid((a.N + cols_per_wg - 1) / cols_per_wg, splits);
  const size_t lds = (size_t)PF * (MT * 2048 + NW * R * 2048);
  auto kern = decode_gemm_wide_kernel<E, MT, R, NW, PF>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MXVL_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
  return MXVL_OK;
}

template <typename E, int MT, int R, int NW = 4>
static int launch_wide(const DecodeGemmArgs& a, int splits, hipStream_t s) {
  if constexpr (NW == 4) { if (g_wide_pf3) return launch_wide_pf<E, MT, R, 4, 3>(a, splits, s); }
  return launch_wide_pf<E, MT, R, NW, wide_pf(MT, R, NW)>(a, splits, s);
}

static int g_decode_gemm_wide = 1;        // mxvl_set_decode_gemm_wide: the A/B switch of tools / bench (default on; 2: also at 17..32 rows)

// The wide kernel's plan for a launch, or false: the K-split kernels keep it.  Shared by the launch below and by mxvl_decode_gemm_plan
// (the dispatch as a pure function of the descriptor: tests/test_abi.py pins it for the reference's decoder shapes without a GPU).
// What bounds these launches is what ONE CU can pull through its load path, weights and activations together (~50 GB/s of
// full-line LDS-DMA, tools/cu_stream_probe.hip -- HBM needs 27 GB/s from each of 256 CUs): a workgroup of NW waves x R tiles
// moves 1 + MT / (NW R) bytes per weight byte, and the launch is as long as its busiest CU.  So: the (NW, R) with the least
//   (1 + MT / (NW R)) x rounds / workgroups,      rounds = ceil(workgroups / 256)   (the LDS ring leaves one workgroup per CU)
// -- three waves per workgroup when that fills the chip: 230 workgroups for Llama's gate / up (172 with four waves), 256 for qkv (192).
// From 17 rows and 64 workgroups on (profiles/r05_decode_gemm_wide_rows_ab.txt: against the K-split kernels the 18-row token
// +7 %, the 24-row token +11 %, Qwen1.5-1.8B's 96..128-workgroup projections +4 % at 80 rows; the first version -- 3-stage
// ring, aligned walks -- had lost to them below 160 workgroups and at 17..32 rows).  At 1..16 rows the K-split kernel keeps the
// launches: it carries the fused RMSNorm, and without it the wide kernel is +2 % (Llama) / 0 % (Qwen) there.
struct WidePlan { int nw, r; long g; };
static bool wide_plan(int MT, const DecodeGemmArgs& a, int splits, WidePlan& pl) {
  if (!(g_decode_gemm_wide && MT >= g_wide_min_mt && a.K % 64 == 0 && a.K >= 256 && !a.g)) return false;
  pl = WidePlan{0, 0, 0};
  double best = 1e30;
  for (int nw : {4, 3}) {
    if (nw == 3 && g_wide_nw4) continue;
    for (int r : {1, 2, 4}) {
      if (a.swiglu && (r & 1)) continue;
      if (nw == 3 && r == 4) continue;
      const int cols = nw * (a.swiglu ? r / 2 : r) * 16;
      const long g = (long)((a.N + cols - 1) / cols) * splits;
      const double est = (1.0 + (double)MT / (nw * r)) * (double)((g + 255) / 256) / (double)g;
      if (est < best - 1e-12) { best = est; pl = WidePlan{nw, r, g}; }
    }
  }
  return pl.g >= g_wide_min_g;
}

template <typename E, int MT>
static int launch_decode_gemm(const DecodeGemmArgs& a, int splits, hipStream_t s) {
  {
    WidePlan pl;
    if (wide_plan(MT, a, splits, pl)) {
      if (pl.nw == 3) return pl.r == 2 ? launch_wide<E, MT, 2, 3>(a, splits, s) : launch_wide<E, MT, 1, 3>(a, splits, s);
      switch (pl.r) {
        case 4: return launch_wide<E, MT, 4>(a, splits, s);
        case 2: return launch_wide<E, MT, 2>(a, splits, s);
        default: return launch_wide<E, MT, 1>(a, splits, s);
      }
    }
  }
  // R = weight tiles per workgroup.  Two effects, both measured (tools/decode_gemm_bench.py A/B at 18 rows, profiles/r04_decode_gemm_r_sweep.txt):
  // (1) a CU streams ~28 GB/s whatever it holds, so the workgroups must fill whole rounds of the 256 CUs -- 344 workgroups (gate / up
  //     at R = 4: 88 CUs with two, 168 with one) run at 4.7 TB/s, 688 (R = 2) at 5.3, 256 (qkv at R = 3) at 5.2 against 4.3 for 384;
  // (2) the activation re-read from L2 per weight byte is MT / R: at equal fill R = 4 beats R = 2 by 6 % (lm_head 5.73 vs 5.37 TB/s).
  // score = fill of the last round x 1 / (1 + 0.06 MT / R); SwiGLU needs an even R (gate and up tiles of the same columns).
  const int tiles = (a.N + 15) / 16;
  auto score = [&](int R) {
    const long wgs = (long)(a.swiglu ? (tiles + R / 2 - 1) / (R / 2) : (tiles + R - 1) / R) * splits;
    const long rounds = (wgs + 255) / 256;
    return (double)wgs / (double)(rounds * 256) / (1.0 + 0.06 * MT / R);
  };
  if (a.K % 64 == 0 && a.K >= 256) {
    constexpr int PFN = MT <= 2 ? 4 : 2;       // narrow workgroups: deeper rings while the activation fragments fit the registers
    int best = a.swiglu ? 2 : 1;
    for (int R : {2, 3, 4}) {
      if (a.swiglu && (R & 1)) continue;
      if (score(R) > score(best) + 1e-9) best = R;
    }
    switch (best) {
      case 4: return launch_dma<E, MT, 4, 4, 2>(a, splits, s);
      case 3: return launch_dma<E, MT, 3, 4, 2>(a, splits, s);
      case 2: return launch_dma<E, MT, 2, 4, PFN>(a, splits, s);
      default: return launch_dma<E, MT, 1, 8, PFN>(a, splits, s);
    }
  }
  if (a.swiglu) return tiles >= 768 ? launch_direct<E, MT, 4>(a, splits, s) : launch_direct<E, MT, 2>(a, splits, s);
  if (tiles >= 1536) return launch_direct<E, MT, 4>(a, splits, s);
  if (tiles >= 512) return launch_direct<E, MT, 2>(a, splits, s);
  return launch_direct<E, MT, 1>(a, splits, s);
}

template <typename E>
static int launch_decode_gemm_rows(const DecodeGemmArgs& a, int splits, hipStream_t s) {
  switch ((a.rows + 15) / 16) {
    case 1: return launch_decode_gemm<E, 1>(a, splits, s);
    case 2: return launch_decode_gemm<E, 2>(a, splits, s);
    case 3: return launch_decode_gemm<E, 3>(a, splits, s);
    case 4: return launch_decode_gemm<E, 4>(a, splits, s);
    default: return launch_decode_gemm<E, 5>(a, splits, s);
  }
}

static int decode_gemm_args(const mxvl_gemv_desc* d, DecodeGemmArgs& a, int& splits);

int decode_gemm_dispatch(const mxvl_gemv_desc* d, hipStream_t s) {
  DecodeGemmArgs a;
  int splits = 1;
  const int chk = decode_gemm_args(d, a, splits);
  if (chk != MXVL_OK) return chk;
  const int rc = decode_dtype(d->dtype) == MXVL_F16 ? launch_decode_gemm_rows<EltF16>(a, splits, s) : launch_decode_gemm_rows<EltBf16>(a, splits, s);
  if (rc != MXVL_OK) return rc;
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

// the wide kernel's ring depth for a plan, as launch_wide picks it
static int wide_plan_pf(int MT, const WidePlan& pl) {
  if (pl.nw == 4 && g_wide_pf3) return 3;
  int pf = 2;
  for (int mt = 1; mt <= 5; ++mt) for (int r : {1, 2, 4}) for (int nw : {3, 4}) if (mt == MT && r == pl.r && nw == pl.nw) pf = wide_pf(mt, r, nw);
  return pf;
}

int decode_gemm_plan(const mxvl_gemv_desc* d, int32_t* out) {
  DecodeGemmArgs a;
  int splits = 1;
  const int chk = decode_gemm_args(d, a, splits);
  if (chk != MXVL_OK) return chk;
  const int MT = (a.rows + 15) / 16 > 5 ? 5 : (a.rows + 15) / 16;
  WidePlan pl;
  const bool wide = wide_plan(MT, a, splits, pl);
  out[0] = wide ? 1 : 0;
  out[1] = wide ? pl.nw : 0;
  out[2] = wide ? pl.r : 0;
  out[3] = wide ? wide_plan_pf(MT, pl) : 0;
  out[4] = wide ? (int32_t)pl.g : 0;
  return MXVL_OK;
}

// argument checks of mxvl_decode_gemv + the kernel-side argument block (nothing is launched here)
static int decode_gemm_args(const mxvl_gemv_desc* d, DecodeGemmArgs& a, int& splits) {
  if (!decode_dtype_ok(d->dtype)) return MXVL_ERR_DTYPE;
  if (d->rows <= 0 || d->rows > 80 || d->K < 32 || d->N <= 0) return MXVL_ERR_SHAPE;
  if (d->K % 8 != 0) return MXVL_ERR_UNSUPPORTED;                      // 16-byte fragments
  if (d->swiglu && (!d->W2 || d->out_f32)) return MXVL_ERR_UNSUPPORTED;
  // RMSNorm fused into the projection (ABI v8): the LDS-DMA kernel only (K % 64 == 0); elsewhere rows come from mxvl_decode_rmsnorm
  if (d->norm_weight && (d->K % 64 != 0 || d->K < 256 || d->split_acc)) return MXVL_ERR_UNSUPPORTED;
  if ((long long)d->N * d->K > 0x7fffffffLL * 16) return MXVL_ERR_SHAPE;
  a.rows = d->rows; a.K = d->K; a.N = d->N; a.swiglu = d->swiglu; a.out_f32 = d->out_f32;
  a.x = (const uint16_t*)d->x; a.W = (const uint16_t*)d->W; a.W2 = (const uint16_t*)d->W2;
  a.bias = (const uint16_t*)d->bias; a.res = (const uint16_t*)d->residual; a.y = d->y;
  a.split_acc = (float*)d->split_acc;
  a.g = (const uint16_t*)d->norm_weight; a.eps = d->eps;
  a.g_scale = 1.0f; a.g_inv = 1.0f;
  if (d->norm_weight && d->norm_gain_scale != 0.0f) {
    int e = 0;
    const float m = frexpf(d->norm_gain_scale, &e);
    if (m != 0.5f || e < -60 || e > 60) return MXVL_ERR_UNSUPPORTED;    // a positive power of two, or nothing
    a.g_scale = d->norm_gain_scale; a.g_inv = 1.0f / d->norm_gain_scale;
  }
  splits = 1;
  if (!a.split_acc && d->k_splits > 1) return MXVL_ERR_UNSUPPORTED;    // a split needs the accumulator
  if (a.split_acc) {        // the caller folds the fp32 sums itself (mxvl_decode_rmsnorm): no epilogue here
    if (d->swiglu || d->bias || d->residual || d->out_f32 || d->k_splits < 1 || d->k_splits > 16) return MXVL_ERR_UNSUPPORTED;
    splits = d->k_splits;
  }
  return MXVL_OK;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_decode_rmsnorm(const mxvl_rmsnorm_desc* d, void* hip_stream) {
  if (!d || !d->weight || !d->y) return MXVL_ERR_NULL;
  if (d->acc ? (!d->residual || !d->x_out) : !d->x) return MXVL_ERR_NULL;
  if (!decode_dtype_ok(d->dtype)) return MXVL_ERR_DTYPE;
  if (d->rows <= 0 || d->K <= 0) return MXVL_ERR_SHAPE;
  if (d->K % 8 != 0 || d->K > 16384) return MXVL_ERR_UNSUPPORTED;
  RmsNormArgs a;
  a.rows = d->rows; a.K = d->K; a.eps = d->eps;
  a.splits = d->acc_splits > 0 ? d->acc_splits : 1;
  if (a.splits > 16) return MXVL_ERR_UNSUPPORTED;
  a.x = (const uint16_t*)d->x; a.g = (const uint16_t*)d->weight; a.y = (uint16_t*)d->y;
  a.acc = (float*)d->acc; a.res = (const uint16_t*)d->residual; a.x_out = (uint16_t*)d->x_out;
  const bool f16 = decode_dtype(d->dtype) == MXVL_F16;
  const dim3 grid(d->rows), block(1024);
  hipStream_t s = (hipStream_t)hip_stream;
#define MXVL_RMSNORM(SS) do { if (f16) hipLaunchKernelGGL((decode_rmsnorm_kernel<EltF16, SS>), grid, block, 0, s, a); \
                              else hipLaunchKernelGGL((decode_rmsnorm_kernel<EltBf16, SS>), grid, block, 0, s, a); } while (0)
  switch (a.acc ? a.splits : 1) {
    case 1: MXVL_RMSNORM(1); break;
    case 2: MXVL_RMSNORM(2); break;
    case 4: MXVL_RMSNORM(4); break;
    case 8: MXVL_RMSNORM(8); break;
    default: MXVL_RMSNORM(0); break;
  }
#undef MXVL_RMSNORM
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

/* diagnostic / A-B switch (tools, bench.py --decode-gemm): 0 = the K-split kernels at every row count (round 4), 1 = waves split N with
 * LDS-shared activations at 17..80 rows for grids of >= 64 workgroups (default; 2 = the same), 3 = the wide kernel as first measured
 * (33..80 rows, >= 160 workgroups, 3-stage ring, four waves per workgroup), 4 = as 1 with four waves per workgroup everywhere,
 * 5 = as 1 at 33..80 rows and >= 160 workgroups only, 6 = as 1 from one row on */
extern "C" int mxvl_set_decode_gemm_wide(int on) {
  static int mode = 1;
  const int was = mode;
  mode = on < 0 ? 0 : on;
  mxvl::g_wide_pf3 = on == 3;
  mxvl::g_wide_nw4 = on == 3 || on == 4;
  mxvl::g_wide_min_g = (on == 3 || on == 5) ? 160 : 64;
  mxvl::g_wide_min_mt = (on == 3 || on == 5) ? 3 : (on == 6 ? 1 : 2);
  mxvl::g_decode_gemm_wide = on <= 0 ? 0 : 1;
  return was;
}
