// image_preprocess.hip -- chest X-ray -> encoder input on the GPU: Pillow-exact 8-bit resize + rescale + normalise, gfx950.
//
// Replaces `AutoImageProcessor(...)(img, return_tensors="pt", size=input_size).pixel_values[0]`
// (CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py:17-26, called per image from :70-76), i.e. Pillow's
// `Image.resize` (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc,
// ImagingResampleVertical_8bpc; the reference pins Pillow==10.1.0) followed by transformers' rescale (x/255 via float64) and
// normalize ((x - mean)/std in float32).  Integer work end to end: results are BIT-EXACT with the CPU path.
//   * coefficients: double-precision host code below (no GPU involved), rounded to 22-bit fixed point like Pillow;
//     stored tap-major (tap, out) so neighbouring output pixels read neighbouring words;
//   * horizontal pass: one workgroup per source row, the row is staged once in LDS (the only read of the big image:
//     3 bytes/pixel), every thread produces output pixels from LDS -> uint8 intermediate (in_h, out_w, 3) like Pillow's;
//   * vertical pass + normalisation: one thread per output pixel, taps broadcast through scalar loads, the byte ->
//     float map (rescale + normalise of all 256 byte values per channel, computed by the caller exactly as
//     transformers does) is a 3x256 LDS table, so the float result carries no rounding of its own.
// HBM traffic = 3*in_h*in_w read + 2*3*in_h*out_w (intermediate, L2-resident) + elt*3*out_h*out_w written.
#include <cmath>

#include "mxvl_common.h"

namespace mxvl {

constexpr int kPrecisionBits = 32 - 8 - 2;   // Resample.c PRECISION_BITS
constexpr int kImgThreads = 256;
constexpr int kMaxRowBytes = 60 * 1024;      // LDS row buffer of the horizontal pass

// ---- Resample.c filters (double precision, no contraction: must round exactly like the C library build) -------------
#pragma clang fp contract(off)
static double filter_weight(int kind, double x) {
  if (x < 0.0) x = -x;
  if (kind == MXVL_RESAMPLE_BILINEAR) return x < 1.0 ? 1.0 - x : 0.0;
  const double a = -0.5;   // bicubic
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
static double filter_support(int kind) { return kind == MXVL_RESAMPLE_BILINEAR ? 1.0 : 2.0; }

struct ImgArgs {
  int in_h, in_w, out_h, out_w, ksize_h, ksize_v;
  const uint8_t* src;
  const int32_t *bounds_h, *kk_h, *bounds_v, *kk_v;
  const float* lut;
  uint8_t* tmp;
  void* out;
};

__device__ inline int clip8(int v) {
  v >>= kPrecisionBits;                      // arithmetic shift, then clamp (Resample.c clip8 lookup table)
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(kImgThreads) void resample_h_kernel(const ImgArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t row[];
  const int y = blockIdx.x, tid = threadIdx.x;
  const int nbytes = p.in_w * 3;
  const uint8_t* rowp = p.src + (size_t)y * nbytes;
  const int mis = (int)((uintptr_t)rowp & 3);            // LDS byte i mirrors address rowp - mis + i (word aligned)
  const int nw = (mis + nbytes + 3) >> 2;
  for (int w = tid; w < nw; w += kImgThreads) {
    const int b0 = w * 4;
    if (b0 >= mis && b0 + 4 <= mis + nbytes) {
      *(uint32_t*)(row + b0) = *(const uint32_t*)(rowp - mis + b0);
    } else {                                             // first / last word: never touch bytes outside the row
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int b = b0 + q;
        if (b >= mis && b < mis + nbytes) row[b] = rowp[b - mis];
      }
    }
  }
  __syncthreads();
  uint8_t* dst = p.tmp + (size_t)y * p.out_w * 3;
  for (int ox = tid; ox < p.out_w; ox += kImgThreads) {
    const int xmin = p.bounds_h[2 * ox], n = p.bounds_h[2 * ox + 1];
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    const uint8_t* px = row + mis + xmin * 3;
    for (int x = 0; x < n; ++x) {
      const int k = p.kk_h[(size_t)x * p.out_w + ox];
      s0 += (int)px[0] * k;
      s1 += (int)px[1] * k;
      s2 += (int)px[2] * k;
      px += 3;
    }
    dst[ox * 3 + 0] = (uint8_t)clip8(s0);
    dst[ox * 3 + 1] = (uint8_t)clip8(s1);
    dst[ox * 3 + 2] = (uint8_t)clip8(s2);
  }
}

template <typename out_t>
__global__ __launch_bounds__(kImgThreads) void resample_v_norm_kernel(const ImgArgs p) {
  __shared__ float lut[3 * 256];
  const int tid = threadIdx.x, oy = blockIdx.y;
  for (int i = tid; i < 3 * 256; i += kImgThreads) lut[i] = p.lut[i];
  __syncthreads();
  const int ox = blockIdx.x * kImgThreads + tid;
  if (ox >= p.out_w) return;
  const int ymin = p.bounds_v[2 * oy], n = p.bounds_v[2 * oy + 1];
  const size_t pitch = (size_t)p.out_w * 3;
  const uint8_t* px = p.tmp + (size_t)ymin * pitch + (size_t)ox * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int yy = 0; yy < n; ++yy) {
    const int k = p.kk_v[(size_t)yy * p.out_h + oy];       // uniform over the workgroup
    s0 += (int)px[0] * k;
    s1 += (int)px[1] * k;
    s2 += (int)px[2] * k;
    px += pitch;
  }
  const size_t plane = (size_t)p.out_h * p.out_w;
  out_t* o = (out_t*)p.out + (size_t)oy * p.out_w + ox;
  Io<out_t>::st(o, lut[clip8(s0)]);
  Io<out_t>::st(o + plane, lut[256 + clip8(s1)]);
  Io<out_t>::st(o + 2 * plane, lut[512 + clip8(s2)]);
}

}  // namespace mxvl

using namespace mxvl;

extern "C" {

// precompute_coeffs' ksize for the whole-image box; 1 when in_size == out_size (Pillow skips that pass; the identity
// tap 2^22 reproduces every byte exactly)
int mxvl_resample_ksize(int in_size, int out_size, int filter) {
  if (in_size <= 0 || out_size <= 0) return MXVL_ERR_SHAPE;
  if (filter != MXVL_RESAMPLE_BILINEAR && filter != MXVL_RESAMPLE_BICUBIC) return MXVL_ERR_UNSUPPORTED;
  if (in_size == out_size) return 1;
  double filterscale = (double)((float)in_size - 0.0f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = filter_support(filter) * filterscale;
  return (int)std::ceil(support) * 2 + 1;
}

int mxvl_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk) {
  const int ksize = mxvl_resample_ksize(in_size, out_size, filter);
  if (ksize < 0) return ksize;
  if (!bounds || !kk) return MXVL_ERR_NULL;
  for (size_t i = 0; i < (size_t)ksize * out_size; ++i) kk[i] = 0;
  if (in_size == out_size) {
    for (int xx = 0; xx < out_size; ++xx) { bounds[2 * xx] = xx; bounds[2 * xx + 1] = 1; kk[xx] = 1 << kPrecisionBits; }
    return MXVL_OK;
  }
  const double scale = (double)((float)in_size - 0.0f) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = filter_support(filter) * filterscale;
  const double ss = 1.0 / filterscale;
  double* w = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      w[x] = filter_weight(filter, (x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) w[x] /= ww;
      const double v = w[x];
      kk[(size_t)x * out_size + xx] = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  delete[] w;
  return MXVL_OK;
}

int mxvl_image_preprocess(const mxvl_image_desc* d, void* hip_stream) {
  if (!d) return MXVL_ERR_NULL;
  if (!d->src || !d->bounds_h || !d->kk_h || !d->bounds_v || !d->kk_v || !d->lut || !d->tmp || !d->out) return MXVL_ERR_NULL;
  if (d->in_h <= 0 || d->in_w <= 0 || d->out_h <= 0 || d->out_w <= 0 || d->ksize_h <= 0 || d->ksize_v <= 0) return MXVL_ERR_SHAPE;
  if (d->out_dtype != MXVL_F32 && d->out_dtype != MXVL_BF16 && d->out_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if ((long)d->in_w * 3 + 8 > kMaxRowBytes) return MXVL_ERR_UNSUPPORTED;
  ImgArgs a;
  a.in_h = d->in_h; a.in_w = d->in_w; a.out_h = d->out_h; a.out_w = d->out_w; a.ksize_h = d->ksize_h; a.ksize_v = d->ksize_v;
  a.src = (const uint8_t*)d->src; a.bounds_h = (const int32_t*)d->bounds_h; a.kk_h = (const int32_t*)d->kk_h;
  a.bounds_v = (const int32_t*)d->bounds_v; a.kk_v = (const int32_t*)d->kk_v; a.lut = (const float*)d->lut;
  a.tmp = (uint8_t*)d->tmp; a.out = d->out;
  hipStream_t s = (hipStream_t)hip_stream;
  const size_t lds = ((size_t)d->in_w * 3 + 8 + 15) & ~(size_t)15;
  hipLaunchKernelGGL(resample_h_kernel, dim3(d->in_h), dim3(kImgThreads), lds, s, a);
  const dim3 grid((d->out_w + kImgThreads - 1) / kImgThreads, d->out_h);
  switch (d->out_dtype) {
    case MXVL_F32: hipLaunchKernelGGL(resample_v_norm_kernel<float>, grid, dim3(kImgThreads), 0, s, a); break;
    case MXVL_BF16: hipLaunchKernelGGL(resample_v_norm_kernel<bf16_t>, grid, dim3(kImgThreads), 0, s, a); break;
    default: hipLaunchKernelGGL(resample_v_norm_kernel<f16_t>, grid, dim3(kImgThreads), 0, s, a); break;
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

}  // extern "C"
