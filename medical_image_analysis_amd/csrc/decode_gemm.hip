// decode_gemm.hip -- host side of the 9..80-row decoder projections (kernels: decode_gemm.h) and of mxvl_decode_rmsnorm.
// mxvl_decode_gemv (decode.hip) forwards here for rows > 8, so the stepper has ONE projection entry for every row count the
// reference's launch scripts decode at (MambaXrayVL_DownStream.py:292-301; 3 .. 80 rows).
#include "decode_gemm.h"

namespace mxvl {

constexpr int kGemmWaves = 8;

template <int MT, int R>
static int launch_decode_gemm(const DecodeGemmArgs& a, hipStream_t s) {
  constexpr int PF = (R + MT <= 4) ? 4 : ((R + MT <= 6) ? 3 : 2);
  const int cols_per_wg = (a.swiglu ? R / 2 : R) * 16;
  const int grid = (a.N + cols_per_wg - 1) / cols_per_wg;
  const size_t lds = (size_t)kGemmWaves * 2 * MT * 64 * 4 * sizeof(float);
  auto kern = decode_gemm_kernel<MT, R, kGemmWaves, PF>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return MXVL_ERR_LAUNCH;   // per call: the attribute is per device
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemmWaves * 64), lds, s, a);
  return MXVL_OK;
}

template <int MT>
static int launch_decode_gemm_r(const DecodeGemmArgs& a, hipStream_t s) {
  // R = weight tiles per workgroup: the activation re-read from L2 per weight byte is MT / R, the number of workgroups
  // N / (16 R): keep at least ~1.5 workgroups per CU (256 CUs), then take the widest R
  const int tiles = (a.N + 15) / 16;
  if (a.swiglu) return tiles >= 768 ? launch_decode_gemm<MT, 4>(a, s) : launch_decode_gemm<MT, 2>(a, s);
  if (tiles >= 1536) return launch_decode_gemm<MT, 4>(a, s);
  if (tiles >= 512) return launch_decode_gemm<MT, 2>(a, s);
  return launch_decode_gemm<MT, 1>(a, s);
}

int decode_gemm_dispatch(const mxvl_gemv_desc* d, hipStream_t s) {
  if (d->rows <= 0 || d->rows > 80 || d->K < 32 || d->N <= 0) return MXVL_ERR_SHAPE;
  if (d->K % 8 != 0) return MXVL_ERR_UNSUPPORTED;                      // 16-byte fragments
  if (d->norm_weight) return MXVL_ERR_UNSUPPORTED;                      // rows > 8: mxvl_decode_rmsnorm runs ahead of the projection
  if (d->swiglu && (!d->W2 || d->out_f32)) return MXVL_ERR_UNSUPPORTED;
  if ((long long)d->N * d->K > 0x7fffffffLL * 16) return MXVL_ERR_SHAPE;
  DecodeGemmArgs a;
  a.rows = d->rows; a.K = d->K; a.N = d->N; a.swiglu = d->swiglu; a.out_f32 = d->out_f32;
  a.x = (const uint16_t*)d->x; a.W = (const uint16_t*)d->W; a.W2 = (const uint16_t*)d->W2;
  a.bias = (const uint16_t*)d->bias; a.res = (const uint16_t*)d->residual; a.y = d->y;
  int rc;
  switch ((d->rows + 15) / 16) {
    case 1: rc = launch_decode_gemm_r<1>(a, s); break;
    case 2: rc = launch_decode_gemm_r<2>(a, s); break;
    case 3: rc = launch_decode_gemm_r<3>(a, s); break;
    case 4: rc = launch_decode_gemm_r<4>(a, s); break;
    default: rc = launch_decode_gemm_r<5>(a, s); break;
  }
  if (rc != MXVL_OK) return rc;
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_decode_rmsnorm(const mxvl_rmsnorm_desc* d, void* hip_stream) {
  if (!d || !d->x || !d->weight || !d->y) return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->K <= 0) return MXVL_ERR_SHAPE;
  if (d->K % 8 != 0 || d->K > 16384) return MXVL_ERR_UNSUPPORTED;
  RmsNormArgs a;
  a.rows = d->rows; a.K = d->K; a.eps = d->eps;
  a.x = (const uint16_t*)d->x; a.g = (const uint16_t*)d->weight; a.y = (uint16_t*)d->y;
  hipLaunchKernelGGL(decode_rmsnorm_kernel, dim3(d->rows), dim3(256), 0, (hipStream_t)hip_stream, a);
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
