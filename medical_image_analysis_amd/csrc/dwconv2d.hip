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
#include <initializer_list>

#include "mxvl_common.h"

namespace mxvl {

constexpr int kDwMaxPlane = 64 * 64;   // largest H*W handled in one LDS tile (SS2D stages are 56x56 .. 7x7 at 224x224)

// A workgroup owns ONE channel and `nb` batch elements of it (grid = channels x batch parts): the nine taps and the bias sit in
// registers, dweight / dbias add up in registers over all its planes (ten atomics per WORKGROUP; the round-1 kernels spent a
// workgroup, two barriers, ten wave reductions and ten atomics on every 14 x 14 plane: 122 us for 38 MB at the third VSSM stage),
// and P planes go through the LDS per pass so that a 196-pixel plane does not leave a quarter of the threads idle.  Planes are read
// and written as flat V-element vectors (V = 4: 8 bytes of 16-bit data / 16 of fp32; V = 1 for H * W % 4 != 0 or unaligned bases);
// the halo words of the tiles are zeroed once and never written again.
struct DwArgs {
  int B, C, H, W, silu, P, nb;
  uint32_t magW, magHW, magLV;   // floor(2^32 / d) + 1 for W, H*W, H*W / V (dw_div: exact while n * d < 2^32)
  const void *x, *dy;
  const float *w, *bias;   // (C, 9), (C) fp32
  void *y, *dx;
  float *dw, *dbias;
};
static inline uint32_t dw_magic(int d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull); }
__device__ __forceinline__ int dw_div(int n, int d, uint32_t mag) { return d == 1 ? n : (int)__umulhi((uint32_t)n, mag); }

template <typename io_t, int V> struct DwVec { typedef io_t type; };
template <> struct DwVec<float, 4> { typedef float4 type; };
template <> struct DwVec<bf16_t, 4> { typedef uint2 type; };
template <> struct DwVec<f16_t, 4> { typedef uint2 type; };

// np planes (batch elements b0 .. b0 + np - 1 of one channel, `bstride` elements apart) into tiles of (H+2) x (W+2) floats
template <typename io_t, int V>
__device__ inline void dw_load(float* tiles, const io_t* base, size_t bstride, int np, const DwArgs& p, int tsz, int tid) {
  typedef typename DwVec<io_t, V>::type vec_t;
  const int W = p.W, LV = p.H * W / V, PW = W + 2;
  for (int i = tid; i < np * LV; i += 256) {
    const int q = dw_div(i, LV, p.magLV), l0 = (i - q * LV) * V;
    io_t tmp[V];
    *(vec_t*)tmp = *(const vec_t*)(base + (size_t)q * bstride + l0);
    int h = dw_div(l0, W, p.magW), w = l0 - h * W;
    float* t = tiles + q * tsz + (h + 1) * PW + (w + 1);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      *t++ = Io<io_t>::ld(tmp + e);
      if (++w == W) { w = 0; t += 2; }              // next row: over the two halo words
    }
  }
}

template <typename io_t, int V>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const DwArgs p) {
  typedef typename DwVec<io_t, V>::type vec_t;
  extern __shared__ float smem[];
  const int H = p.H, W = p.W, HW = H * W, PW = W + 2, tsz = (H + 2) * PW, LV = HW / V;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int b_begin = blockIdx.y * p.nb, b_end = b_begin + p.nb < p.B ? b_begin + p.nb : p.B;
  for (int i = tid; i < p.P * tsz; i += 256) smem[i] = 0.0f;
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = p.w[c * 9 + k];
  const float bias = p.bias ? p.bias[c] : 0.0f;
  const size_t bstride = (size_t)p.C * HW;
  for (int b0 = b_begin; b0 < b_end; b0 += p.P) {
    const int np = b_end - b0 < p.P ? b_end - b0 : p.P;
    const size_t off = ((size_t)b0 * p.C + c) * HW;
    __syncthreads();                                // the tiles are free (first pass: zeroed)
    dw_load<io_t, V>(smem, (const io_t*)p.x + off, bstride, np, p, tsz, tid);
    __syncthreads();
    for (int i = tid; i < np * LV; i += 256) {
      const int q = dw_div(i, LV, p.magLV), l0 = (i - q * LV) * V;
      int h = dw_div(l0, W, p.magW), w = l0 - h * W;
      const float* t = smem + q * tsz + h * PW + w; // top-left of the 3x3 window (halo offset folded in)
      io_t o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc = fmaf(wk[ky * 3 + kx], t[ky * PW + kx], acc);
        Io<io_t>::st(o + e, p.silu ? silu(acc) : acc);
        ++t;
        if (++w == W) { w = 0; t += 2; }
      }
      *(vec_t*)((io_t*)p.y + off + (size_t)q * bstride + l0) = *(const vec_t*)o;
    }
  }
}

template <typename io_t, int V>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const DwArgs p) {
  typedef typename DwVec<io_t, V>::type vec_t;
  extern __shared__ float smem[];
  __shared__ float red[4][10];
  const int H = p.H, W = p.W, HW = H * W, PW = W + 2, tsz = (H + 2) * PW, LV = HW / V;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int b_begin = blockIdx.y * p.nb, b_end = b_begin + p.nb < p.B ? b_begin + p.nb : p.B;
  float* tx = smem;                               // x with halo, P tiles
  float* tg = smem + p.P * tsz;                   // d(pre-activation) with halo, P tiles
  for (int i = tid; i < 2 * p.P * tsz; i += 256) smem[i] = 0.0f;
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = p.w[c * 9 + k];
  const float bias = p.bias ? p.bias[c] : 0.0f;
  float dwa[9], dba = 0.0f;
#pragma unroll
  for (int k = 0; k < 9; ++k) dwa[k] = 0.0f;
  const size_t bstride = (size_t)p.C * HW;
  for (int b0 = b_begin; b0 < b_end; b0 += p.P) {
    const int np = b_end - b0 < p.P ? b_end - b0 : p.P;
    const size_t off = ((size_t)b0 * p.C + c) * HW;
    __syncthreads();                              // the tiles are free (first pass: zeroed)
    dw_load<io_t, V>(tx, (const io_t*)p.x + off, bstride, np, p, tsz, tid);
    dw_load<io_t, V>(tg, (const io_t*)p.dy + off, bstride, np, p, tsz, tid);
    __syncthreads();
    // phase A: dpre = dy * silu'(pre) in place (interior only; the halo stays zero), dweight / dbias partial sums
    for (int i = tid; i < np * HW; i += 256) {
      const int q = dw_div(i, HW, p.magHW), l = i - q * HW;
      const int h = dw_div(l, W, p.magW), w = l - h * W;
      const float* t = tx + q * tsz + h * PW + w;
      float* gq = tg + q * tsz + (h + 1) * PW + w + 1;
      float g = *gq;
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
      *gq = g;                                    // only this thread touches this element in phase A
    }
    __syncthreads();
    // phase B: dx[h][w] = sum_k w[ky][kx] * dpre[h + 1 - ky][w + 1 - kx]
    for (int i = tid; i < np * LV; i += 256) {
      const int q = dw_div(i, LV, p.magLV), l0 = (i - q * LV) * V;
      int h = dw_div(l0, W, p.magW), w = l0 - h * W;
      const float* t = tg + q * tsz + h * PW + w;
      io_t o[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float acc = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc = fmaf(wk[ky * 3 + kx], t[(2 - ky) * PW + (2 - kx)], acc);
        Io<io_t>::st(o + e, acc);
        ++t;
        if (++w == W) { w = 0; t += 2; }
      }
      *(vec_t*)((io_t*)p.dx + off + (size_t)q * bstride + l0) = *(const vec_t*)o;
    }
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

// planes per pass (<= 32 KB of tiles, <= 16) and batch elements per workgroup (>= 1024 workgroups where the problem has them)
static void dw_geometry(DwArgs& a, int tiles_per_plane, bool v4) {
  const long tile = (long)sizeof(float) * tiles_per_plane * (a.H + 2) * (a.W + 2);
  long P = 32 * 1024 / tile;
  P = P < 1 ? 1 : P > 16 ? 16 : P;
  int parts = (1024 + a.C - 1) / a.C;
  if (parts > a.B) parts = a.B;
  a.nb = (a.B + parts - 1) / parts;
  if (P > a.nb) P = a.nb;
  a.P = (int)P;
  const int HW = a.H * a.W;
  a.magW = dw_magic(a.W); a.magHW = dw_magic(HW); a.magLV = dw_magic(v4 ? HW / 4 : HW);
}
static bool dw_vec4(int HW, std::initializer_list<const void*> ptrs) {
  if (HW % 4 != 0) return false;
  for (const void* q : ptrs)
    if (((uintptr_t)q) % 16 != 0) return false;
  return true;
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
  const bool v4 = dw_vec4(height * width, {x, y});
  dw_geometry(a, 1, v4);
  const dim3 grid(channels, (batch + a.nb - 1) / a.nb);
  const size_t lds = sizeof(float) * (size_t)a.P * (height + 2) * (width + 2);
  hipStream_t s = (hipStream_t)hip_stream;
#define MXVL_DW_LAUNCH(K, T)                                                                                   \
  do {                                                                                                         \
    if (v4) hipLaunchKernelGGL((K<T, 4>), grid, dim3(256), lds, s, a);                                         \
    else hipLaunchKernelGGL((K<T, 1>), grid, dim3(256), lds, s, a);                                            \
  } while (0)
  switch (io_dtype) {
    case MXVL_F32: MXVL_DW_LAUNCH(dwconv_fwd_kernel, float); break;
    case MXVL_BF16: MXVL_DW_LAUNCH(dwconv_fwd_kernel, bf16_t); break;
    default: MXVL_DW_LAUNCH(dwconv_fwd_kernel, f16_t); break;
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
  const bool v4 = dw_vec4(height * width, {x, dy, dx});
  dw_geometry(a, 2, v4);
  const dim3 grid(channels, (batch + a.nb - 1) / a.nb);
  const size_t lds = sizeof(float) * 2 * (size_t)a.P * (height + 2) * (width + 2);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (io_dtype) {
    case MXVL_F32: MXVL_DW_LAUNCH(dwconv_bwd_kernel, float); break;
    case MXVL_BF16: MXVL_DW_LAUNCH(dwconv_bwd_kernel, bf16_t); break;
    default: MXVL_DW_LAUNCH(dwconv_bwd_kernel, f16_t); break;
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

}  // extern "C"
