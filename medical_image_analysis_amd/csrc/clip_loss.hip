// clip_loss.hip -- stage-2 MambaXray-CLIP contrastive step as ONE kernel: L2-normalise image / text features, cosine
// logits * exp(logit_scale), symmetric cross-entropy, and every gradient of it.
//
// Replaces the eager sequence of CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py:133-148
//   image_features /= norm; text_features /= norm; logits = scale * img @ txt.t(); loss = (CE(logits) + CE(logits.t())) / 2
// (~25 small kernels forward + backward at batch 48 x 512: launch-bound).  The problem is tiny (2 x 98 KB of features, a
// 48 x 48 logit matrix): one workgroup keeps the logits and their gradient in LDS; features are read through L1/L2.
// Outputs the loss and, for d(loss) = 1, d(image_features), d(text_features), d(logit_scale); the autograd wrapper scales
// them by the incoming gradient.  fp32 features (the projections' outputs are cast up, as the reference's .float() loss).
#include "mxvl_common.h"

namespace mxvl {

struct ClipArgs {
  int n, dim;
  const float* logit_scale;   // device scalar: the PARAMETER (log of the multiplier)
  const float *img, *txt;
  float *loss, *dimg, *dtxt, *dscale;
};

__device__ inline float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(1024) void clip_loss_kernel(const ClipArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = p.n, P = p.dim, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = 16;
  float* sL = sm;                 // [n][n] logits
  float* sG = sL + n * n;         // [n][n] d loss / d logits
  float* inv_i = sG + n * n;      // [n] 1 / |img_i|
  float* inv_t = inv_i + n;       // [n]
  float* lse_r = inv_t + n;       // [n] row log-sum-exp (image -> text)
  float* lse_c = lse_r + n;       // [n] column log-sum-exp (text -> image)
  float* red = lse_c + n;         // [NW] scratch
  const float s = __expf(*p.logit_scale);

  for (int r = wave; r < 2 * n; r += NW) {            // row norms: one wave per row
    const float* x = (r < n ? p.img + (size_t)r * P : p.txt + (size_t)(r - n) * P);
    float a = 0.f;
    for (int k = lane; k < P; k += 64) a = fmaf(x[k], x[k], a);
    a = wave_sum64(a);
    if (lane == 0) (r < n ? inv_i[r] : inv_t[r - n]) = rsqrtf(a);
  }
  __syncthreads();
  for (int e = wave; e < n * n; e += NW) {             // logits: one wave per (i, j)
    const int i = e / n, j = e - i * n;
    const float *x = p.img + (size_t)i * P, *y = p.txt + (size_t)j * P;
    float a = 0.f;
    for (int k = lane; k < P; k += 64) a = fmaf(x[k], y[k], a);
    a = wave_sum64(a);
    if (lane == 0) sL[e] = s * a * inv_i[i] * inv_t[j];
  }
  __syncthreads();
  if (tid < 2 * n) {                                   // log-sum-exp of every row and every column
    const int r = tid < n ? tid : tid - n;
    float mx = -__builtin_inff();
    for (int k = 0; k < n; ++k) mx = fmaxf(mx, tid < n ? sL[r * n + k] : sL[k * n + r]);
    float a = 0.f;
    for (int k = 0; k < n; ++k) a += __expf((tid < n ? sL[r * n + k] : sL[k * n + r]) - mx);
    (tid < n ? lse_r : lse_c)[r] = mx + __logf(a);
  }
  __syncthreads();
  float part = 0.f, dsc = 0.f;                          // loss and d/d(logit_scale)
  const float w = 0.5f / (float)n;
  for (int e = tid; e < n * n; e += 1024) {
    const int i = e / n, j = e - i * n;
    const float l = sL[e];
    const float g = w * (__expf(l - lse_r[i]) + __expf(l - lse_c[j]) - (i == j ? 2.f : 0.f));
    sG[e] = g;
    dsc = fmaf(g, l, dsc);                              // d logits / d logit_scale = logits  (multiplier = exp(param))
    if (i == j) part += w * ((lse_r[i] - l) + (lse_c[i] - l));
  }
  part = wave_sum64(part);
  dsc = wave_sum64(dsc);
  if (lane == 0) { red[wave] = part; red[NW + wave] = dsc; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < NW; ++k) { a += red[k]; b += red[NW + k]; }
    *p.loss = a;
    *p.dscale = b;
  }
  // d normalised features, then back through x / |x|:  dx = (dn - n (n . dn)) / |x|
  for (int r = wave; r < 2 * n; r += NW) {
    const bool is_img = r < n;
    const int i = is_img ? r : r - n;
    const float* x = (is_img ? p.img : p.txt) + (size_t)i * P;
    float* dx = (is_img ? p.dimg : p.dtxt) + (size_t)i * P;
    const float ix = is_img ? inv_i[i] : inv_t[i];
    float dot = 0.f;
    for (int k = lane; k < P; k += 64) {
      float a = 0.f;
      for (int j = 0; j < n; ++j) {
        const float g = is_img ? sG[i * n + j] : sG[j * n + i];
        const float y = (is_img ? p.txt : p.img)[(size_t)j * P + k] * (is_img ? inv_t[j] : inv_i[j]);
        a = fmaf(g, y, a);
      }
      a *= s;
      dx[k] = a;                     // d n_k, finished below
      dot = fmaf(a, x[k] * ix, dot);
    }
    dot = wave_sum64(dot);
    for (int k = lane; k < P; k += 64) dx[k] = (dx[k] - x[k] * ix * dot) * ix;   // same lane wrote dx[k]: no hazard
  }
}

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_clip_loss(const float* image_features, const float* text_features, const float* logit_scale, int batch, int dim,
                              float* loss, float* d_image, float* d_text, float* d_logit_scale, void* hip_stream) {
  if (!image_features || !text_features || !logit_scale || !loss || !d_image || !d_text || !d_logit_scale) return MXVL_ERR_NULL;
  if (batch <= 0 || dim <= 0) return MXVL_ERR_SHAPE;
  const size_t lds = sizeof(float) * ((size_t)2 * batch * batch + 4 * batch + 32);
  if (lds > 64 * 1024) return MXVL_ERR_UNSUPPORTED;      // batch <= 89 (the reference trains stage 2 at batch 48)
  ClipArgs a{batch, dim, logit_scale, image_features, text_features, loss, d_image, d_text, d_logit_scale};
  hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(1024), lds, (hipStream_t)hip_stream, a);
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
