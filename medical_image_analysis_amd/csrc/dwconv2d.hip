// dwconv2d.hip -- depthwise KxK (K = 3) 2-D convolution (+ optional SiLU) of VMamba's SS2D block, forward and backward, gfx950.
//
// Replaces `self.act(self.conv2d(x))` with nn.Conv2d(d_inner, d_inner, 3, padding=1, groups=d_inner)
// (R2GenCSR/VMamba/classification/models/vmamba.py:746-755 constructor, :1121-1123 call).  MIOpen serves this shape with its
// naive direct kernels on gfx950 (measured on the R2GenCSR encoder, batch 32: weight-gradient 16.9 ms + data-gradient 3.1 ms +
// forward 1.8 ms of a 76 ms training step).  The op is pure HBM traffic: every (batch, channel) plane is read once into LDS
// with its halo, the nine taps and the activation are applied from LDS, and the backward produces dx, dweight and dbias in
// the same pass (recomputing the pre-activation from the x tile it already holds).
//   forward : 2 * elt * B*C*H*W bytes;   backward: 3 * elt * B*C*H*W (+ 40 bytes per plane of fp32 atomics)
#include <algorithm>

#include "mxvl_common.h"

namespace mxvl {

constexpr int kDwMaxPlane = 64 * 64;   // largest H*W handled in one LDS tile (SS2D stages are 56x56 .. 7x7 at 224x224)

struct DwArgs {
  int B, C, H, W, silu, planes_per_wg;
  const void *x, *dy;
  const float *w, *bias;   // (C, 9), (C) fp32
  void *y, *dx;
  float *dw, *dbias;
};

// LDS tile of one plane with a 1-pixel zero halo: (H+2) x (W+2)
template <typename io_t>
__device__ inline void load_plane(float* tile, const io_t* src, int H, int W, int tid, int nthreads) {
  const int PW = W + 2, n = (H + 2) * PW;
  for (int i = tid; i < n; i += nthreads) {
    const int r = i / PW, c = i - r * PW;
    const int h = r - 1, w = c - 1;
    tile[i] = (h >= 0 && h < H && w >= 0 && w < W) ? Io<io_t>::ld(src + (size_t)h * W + w) : 0.0f;
  }
}

template <typename io_t>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const DwArgs p) {
  extern __shared__ float smem[];
  const int H = p.H, W = p.W, HW = H * W, PW = W + 2, tsz = (H + 2) * PW;
  const int P = p.planes_per_wg;
  const int nthr = 256 / P;                       // threads per plane
  const int sub = threadIdx.x / nthr, tid = threadIdx.x - sub * nthr;
  const long plane = (long)blockIdx.x * P + sub;
  const bool live = plane < (long)p.B * p.C;
  float* tile = smem + sub * tsz;
  const int c = live ? (int)(plane % p.C) : 0;
  if (live) load_plane<io_t>(tile, (const io_t*)p.x + plane * HW, H, W, tid, nthr);
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = p.w[c * 9 + k];
  const float bias = p.bias ? p.bias[c] : 0.0f;
  __syncthreads();
  if (!live) return;
  io_t* dst = (io_t*)p.y + plane * HW;
  for (int i = tid; i < HW; i += nthr) {
    const int h = i / W, w = i - h * W;
    const float* t = tile + h * PW + w;           // top-left of the 3x3 window (halo offset folded in)
    float acc = bias;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc = fmaf(wk[ky * 3 + kx], t[ky * PW + kx], acc);
    Io<io_t>::st(dst + i, p.silu ? silu(acc) : acc);
  }
}

template <typename io_t>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const DwArgs p) {
  extern __shared__ float smem[];
  __shared__ float red[4][10];
  const int H = p.H, W = p.W, HW = H * W, PW = W + 2, tsz = (H + 2) * PW;
  const long plane = blockIdx.x;                  // one plane per workgroup (the reductions are per channel)
  const int c = (int)(plane % p.C);
  float* tx = smem;                               // x with halo
  float* tg = smem + tsz;                         // d(pre-activation) with halo
  const int tid = threadIdx.x;
  load_plane<io_t>(tx, (const io_t*)p.x + plane * HW, H, W, tid, 256);
  load_plane<io_t>(tg, (const io_t*)p.dy + plane * HW, H, W, tid, 256);
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = p.w[c * 9 + k];
  const float bias = p.bias ? p.bias[c] : 0.0f;
  __syncthreads();
  float dwa[9], dba = 0.0f;
#pragma unroll
  for (int k = 0; k < 9; ++k) dwa[k] = 0.0f;
  // phase A: dpre = dy * silu'(pre) in place (interior only; the halo stays zero), dweight / dbias partial sums
  for (int i = tid; i < HW; i += 256) {
    const int h = i / W, w = i - h * W;
    const float* t = tx + h * PW + w;
    float g = tg[(h + 1) * PW + w + 1];
    if (p.silu) {
      float pre = bias;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) pre = fmaf(wk[ky * 3 + kx], t[ky * PW + kx], pre);
      const float s = sigmoid(pre);
      g *= s * fmaf(pre, 1.0f - s, 1.0f);
    }
    dba += g;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) dwa[ky * 3 + kx] = fmaf(g, t[ky * PW + kx], dwa[ky * 3 + kx]);
    tg[(h + 1) * PW + w + 1] = g;                 // only this thread touches this element in phase A
  }
  __syncthreads();
  // phase B: dx[h][w] = sum_k w[ky][kx] * dpre[h + 1 - ky][w + 1 - kx]
  io_t* dst = (io_t*)p.dx + plane * HW;
  for (int i = tid; i < HW; i += 256) {
    const int h = i / W, w = i - h * W;
    const float* t = tg + h * PW + w;
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc = fmaf(wk[ky * 3 + kx], t[(2 - ky) * PW + (2 - kx)], acc);
    Io<io_t>::st(dst + i, acc);
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    float v = k < 9 ? dwa[k] : dba;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (tid < 10) {
    const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid < 9) unsafeAtomicAdd(p.dw + c * 9 + tid, v);
    else if (p.dbias) unsafeAtomicAdd(p.dbias + c, v);
  }
}

static int dw_check(int B, int C, int H, int W, int K, int dtype) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return MXVL_ERR_SHAPE;
  if (K != 3 || (long)H * W > kDwMaxPlane) return MXVL_ERR_UNSUPPORTED;
  if (dtype != MXVL_F32 && dtype != MXVL_BF16 && dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  return MXVL_OK;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" {

int mxvl_dwconv2d_fwd(const void* x, const void* weight, const void* bias, void* y, int batch, int channels, int height,
                      int width, int ksize, int io_dtype, int silu_on, void* hip_stream) {
  if (!x || !weight || !y) return MXVL_ERR_NULL;
  int rc = dw_check(batch, channels, height, width, ksize, io_dtype);
  if (rc != MXVL_OK) return rc;
  DwArgs a{};
  a.B = batch; a.C = channels; a.H = height; a.W = width; a.silu = silu_on;
  a.x = x; a.w = (const float*)weight; a.bias = (const float*)bias; a.y = y;
  const int hw = height * width;
  a.planes_per_wg = hw >= 512 ? 1 : hw >= 256 ? 2 : hw >= 128 ? 4 : 8;   // small planes share a workgroup
  const long planes = (long)batch * channels;
  const int grid = (int)((planes + a.planes_per_wg - 1) / a.planes_per_wg);
  const size_t lds = sizeof(float) * (size_t)a.planes_per_wg * (height + 2) * (width + 2);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (io_dtype) {
    case MXVL_F32: hipLaunchKernelGGL(dwconv_fwd_kernel<float>, dim3(grid), dim3(256), lds, s, a); break;
    case MXVL_BF16: hipLaunchKernelGGL(dwconv_fwd_kernel<bf16_t>, dim3(grid), dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL(dwconv_fwd_kernel<f16_t>, dim3(grid), dim3(256), lds, s, a); break;
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

int mxvl_dwconv2d_bwd(const void* x, const void* weight, const void* bias, const void* dy, void* dx, void* dweight,
                      void* dbias, int batch, int channels, int height, int width, int ksize, int io_dtype, int silu_on,
                      void* hip_stream) {
  if (!x || !weight || !dy || !dx || !dweight) return MXVL_ERR_NULL;
  if (bias && !dbias) return MXVL_ERR_NULL;
  int rc = dw_check(batch, channels, height, width, ksize, io_dtype);
  if (rc != MXVL_OK) return rc;
  DwArgs a{};
  a.B = batch; a.C = channels; a.H = height; a.W = width; a.silu = silu_on;
  a.x = x; a.dy = dy; a.w = (const float*)weight; a.bias = (const float*)bias; a.dx = dx;
  a.dw = (float*)dweight; a.dbias = (float*)dbias;
  const long planes = (long)batch * channels;
  const size_t lds = sizeof(float) * 2 * (size_t)(height + 2) * (width + 2);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (io_dtype) {
    case MXVL_F32: hipLaunchKernelGGL(dwconv_bwd_kernel<float>, dim3((unsigned)planes), dim3(256), lds, s, a); break;
    case MXVL_BF16: hipLaunchKernelGGL(dwconv_bwd_kernel<bf16_t>, dim3((unsigned)planes), dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL(dwconv_bwd_kernel<f16_t>, dim3((unsigned)planes), dim3(256), lds, s, a); break;
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

}  // extern "C"
