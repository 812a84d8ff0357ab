// mae_ops.hip -- the index / loss glue of ViT-MAE pre-training as HIP kernels (gfx950).
//
// Replaces the eager torch sequences of HD_Xray_Pretrain_MAE/pretrain/models/mae.py:
//   :157-182, :184-253   random_masking / random_masking_yiliao: `torch.gather(x, 1, ids_keep[..., None].repeat(1, 1, D))`
//   :280-305             forward_decoder: cat(x[:, 1:], mask_tokens) -> gather(ids_restore) -> cat(cls, .) -> + decoder_pos_embed
//   :129-141, :307-323   patchify + per-patch normalisation (mean, unbiased var, eps 1e-6) + mean squared error per patch
// and their backward passes (autograd's gather-backward is a zero fill + scatter_add; the loss chain is ~10 element-wise
// kernels over the (N, L, p*p) target).  Two kernels:
//
//   row_gather_kernel    out[n, r, :] = (idx[n, r] >= 0 ? src[n, idx[n, r], :] : fill[:]) (+ add[r, :])
//                        One kernel covers the four uses -- encoder masking (idx = ids_keep), its backward (idx = position of
//                        token l among the kept ones, or -1 -> zero row), the decoder un-shuffle (idx = 1 + ids_restore or -1 ->
//                        mask token, row 0 = cls, add = decoder_pos_embed) and its backward (idx = 1 + ids_keep).  The host
//                        builds the small (N, rows) int32 index tensors; every output element is ONE copy (+ ONE add, the same
//                        fp32 add torch performs): bit-exact with the reference expressions.  16-byte accesses, a wave per row.
//   patch_loss_kernel    loss[n, l] = mean_j (pred[n, l, j] - tgt[n, l, j])^2 with tgt = patchify(img) (normalised per patch when
//                        norm_pix_loss): one workgroup per patch reads its p x p pixels from the image once (fp32 two-pass
//                        statistics), never materialising the (N, L, p*p) target; the backward recomputes the target the same
//                        way and writes d pred = dloss * 2 (pred - tgt) / (p*p).
// HBM-bound: algorithmic bytes = read + write of every row once (gather), image + pred (+ dpred) once (loss).
#include <algorithm>

#include "mxvl_common.h"

namespace mxvl {

struct RowGatherArgs {
  int N, rows_out, rows_src, D;
  int64_t src_bs, out_bs;                // batch strides in elements (rows are contiguous, D elements each)
  const void* src;
  const float *fill, *add;               // fp32: fill (D) or null -> zeros; add (rows_out, D) or null
  const int* idx;                        // (N, rows_out)
  void* out;
};

// one wave per output row, 4 elements per lane per trip.  S = source dtype, O = output dtype: S == O without `add` is a bit copy;
// otherwise fp32 in between (what torch's type promotion of `cat(bf16, fp32 mask_token) + fp32 pos_embed` computes, and the cast
// autograd applies to a gradient that flows back into a 16-bit tensor).
template <typename S, typename O>
__global__ __launch_bounds__(256) void row_gather_kernel(const RowGatherArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= (int64_t)p.N * p.rows_out) return;
  const int n = (int)(row / p.rows_out), r = (int)(row - (int64_t)n * p.rows_out);
  const int s = p.idx[row];
  const S* srow = s >= 0 ? (const S*)p.src + (int64_t)n * p.src_bs + (int64_t)s * p.D : nullptr;
  O* orow = (O*)p.out + (int64_t)n * p.out_bs + (int64_t)r * p.D;
  const float* arow = p.add ? p.add + (int64_t)r * p.D : nullptr;
  const bool vec = (p.D % 4 == 0) && (!srow || (uintptr_t)srow % (4 * sizeof(S)) == 0) && ((uintptr_t)orow % (4 * sizeof(O)) == 0) &&
                   (!arow || (uintptr_t)arow % 16 == 0) && (!p.fill || (uintptr_t)p.fill % 16 == 0);
  if (vec) {
    for (int c = lane * 4; c < p.D; c += 256) {
      if constexpr (__is_same(S, O)) {
        if (srow && !arow) {                       // pure row copy: bits, not values
          if constexpr (sizeof(S) == 4) *(uint4*)(orow + c) = *(const uint4*)(srow + c);
          else *(uint2*)(orow + c) = *(const uint2*)(srow + c);
          continue;
        }
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (srow) v = ld4<S>(srow + c);
      else if (p.fill) v = *(const float4*)(p.fill + c);
      if (arow) {
        const float4 a = *(const float4*)(arow + c);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      st4<O>(orow + c, v);
    }
  } else {
    for (int c = lane; c < p.D; c += 64) {
      if constexpr (__is_same(S, O)) {
        if (srow && !arow) { orow[c] = srow[c]; continue; }
      }
      float v = srow ? Io<S>::ld(srow + c) : (p.fill ? p.fill[c] : 0.0f);
      if (arow) v += arow[c];
      Io<O>::st(orow + c, v);
    }
  }
}

struct PatchLossArgs {
  int N, C, HW, p, gw, norm;             // image (N, C, HW, HW) fp32, patch p, gw = HW / p patches per row
  const float* img;
  const void* pred;                      // (N, L, p*p*C) io dtype
  const float* dloss;                    // backward: (N, L)
  float* loss;                           // forward: (N, L)
  void* dpred;                           // backward: (N, L, p*p*C) io dtype
};

__device__ inline float block_sum256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// one workgroup (256 threads) per patch.  Element j of the patch vector = (py * p + px) * C + c ('nchpwq->nhwpqc', mae.py:139)
template <typename E, bool BWD>
__global__ __launch_bounds__(256) void patch_loss_kernel(const PatchLossArgs p) {
  __shared__ float red[4];
  const int l = blockIdx.x, n = blockIdx.y;
  const int gy = l / p.gw, gx = l - gy * p.gw;
  const int P = p.p * p.p * p.C;
  const float* img = p.img + (int64_t)n * p.C * p.HW * p.HW;
  auto pixel = [&](int j) {
    const int c = j % p.C, q = j / p.C, py = q / p.p, px = q - py * p.p;
    return img[((int64_t)c * p.HW + gy * p.p + py) * p.HW + gx * p.p + px];
  };
  float mean = 0.f, rstd = 1.f;
  if (p.norm) {
    float s = 0.f;
    for (int j = threadIdx.x; j < P; j += 256) s += pixel(j);
    mean = block_sum256(s, red) / (float)P;
    float q = 0.f;
    for (int j = threadIdx.x; j < P; j += 256) { const float d = pixel(j) - mean; q += d * d; }
    const float var = block_sum256(q, red) / (float)(P - 1);           // torch.var: unbiased
    rstd = 1.0f / sqrtf(var + 1.0e-6f);
  }
  const E* pr = (const E*)p.pred + ((int64_t)n * gridDim.x + l) * P;
  if constexpr (!BWD) {
    float acc = 0.f;
    for (int j = threadIdx.x; j < P; j += 256) {
      const float d = Io<E>::ld(pr + j) - (pixel(j) - mean) * rstd;
      acc += d * d;
    }
    acc = block_sum256(acc, red);
    if (threadIdx.x == 0) p.loss[(int64_t)n * gridDim.x + l] = acc / (float)P;
  } else {
    const float g = p.dloss[(int64_t)n * gridDim.x + l] * 2.0f / (float)P;
    E* dp = (E*)p.dpred + ((int64_t)n * gridDim.x + l) * P;
    for (int j = threadIdx.x; j < P; j += 256) Io<E>::st(dp + j, g * (Io<E>::ld(pr + j) - (pixel(j) - mean) * rstd));
  }
}

static int mae_check() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}


// patch_cols_kernel: the im2col of a convolution whose kernel equals its stride (patch embeddings: SmallPatchEmbed conv1 16x16 / s16 on
// a 1280^2 X-ray, HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:21-43; ViT PatchEmbed, finetune/DP/models/vit.py:186-208) --
//   cols[n, i * gw + j, (c * p + di) * p + dj] = img[n, c, i * p + di, j * p + dj]
// = `img.reshape(N, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)`, written in the GEMM's input dtype (the autocast cast rides along).
// As a torch permute-copy this is a gather of p-element runs: the generic strided-copy kernel moved the 419 M pixels of a 256-image
// MAE batch at 0.6 TB/s (5.3 ms, twice per step).  Here a workgroup owns the p image rows of one (image, channel, patch row): they
// come in as whole 1 KB wave loads, cross an LDS tile, and leave as whole p * p-element patch rows -- both sides coalesced.
struct PatchColsArgs {
  int N, C, H, W, p, gw;
  const void* img;
  void* cols;
};

template <typename S, typename O>
__global__ __launch_bounds__(256) void patch_cols_kernel(const PatchColsArgs a) {
  constexpr int CW = 256, LD = CW + 4;               // columns per chunk; LDS row stride (16-byte aligned, off the power of two)
  extern __shared__ __attribute__((aligned(16))) float pc_tile[];   // [p][LD]
  const int tid = threadIdx.x;
  const int i = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int p = a.p, W = a.W, K = a.C * p * p;
  const S* src = (const S*)a.img + (((int64_t)n * a.C + c) * a.H + (int64_t)i * p) * W;
  O* dst = (O*)a.cols + ((int64_t)n * (a.H / p) + i) * a.gw * (int64_t)K + (int64_t)c * p * p;
  const int q_per_row = p / 4, u_per_patch = p * q_per_row, ppc = CW / p;   // 4-element units per patch row of the image / per patch
  for (int x0 = 0; x0 < W; x0 += CW) {
    __syncthreads();
    for (int u = tid; u < p * (CW / 4); u += 256) {          // p rows x 64 units of 4 columns
      const int r = u / (CW / 4), x = (u - r * (CW / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 + x < W) v = ld4<S>(src + (int64_t)r * W + x0 + x);
      *(float4*)(pc_tile + r * LD + x) = v;
    }
    __syncthreads();
    const int j0 = x0 / p;
    for (int u = tid; u < ppc * u_per_patch; u += 256) {
      const int jj = u / u_per_patch, e = u - jj * u_per_patch;
      const int di = e / q_per_row, q = e - di * q_per_row;
      if (j0 + jj < a.gw) st4<O>(dst + (int64_t)(j0 + jj) * K + di * p + q * 4, *(const float4*)(pc_tile + di * LD + jj * p + q * 4));
    }
  }
}

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_row_gather(const void* src, const int32_t* idx, const float* fill, const float* add, void* out, int N, int rows_src,
                               int rows_out, int D, int64_t src_bs, int64_t out_bs, int src_dtype, int out_dtype, void* hip_stream) {
  if (!src || !idx || !out) return MXVL_ERR_NULL;
  if (N <= 0 || rows_out <= 0 || rows_src <= 0 || D <= 0) return MXVL_ERR_SHAPE;
  auto ok = [](int d) { return d == MXVL_F32 || d == MXVL_BF16 || d == MXVL_F16; };
  if (!ok(src_dtype) || !ok(out_dtype)) return MXVL_ERR_DTYPE;
  if (src_dtype != out_dtype && src_dtype != MXVL_F32 && out_dtype != MXVL_F32) return MXVL_ERR_DTYPE;   // 16-bit <-> 16-bit of another kind: no use
  RowGatherArgs a;
  a.N = N; a.rows_out = rows_out; a.rows_src = rows_src; a.D = D;
  a.src_bs = src_bs; a.out_bs = out_bs; a.src = src; a.fill = fill; a.add = add; a.idx = idx; a.out = out;
  const int64_t rows = (int64_t)N * rows_out;
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)hip_stream;
#define MXVL_RG(S, O) hipLaunchKernelGGL((row_gather_kernel<S, O>), grid, dim3(256), 0, s, a)
  if (src_dtype == MXVL_F32 && out_dtype == MXVL_F32) MXVL_RG(float, float);
  else if (src_dtype == MXVL_BF16 && out_dtype == MXVL_BF16) MXVL_RG(bf16_t, bf16_t);
  else if (src_dtype == MXVL_F16 && out_dtype == MXVL_F16) MXVL_RG(f16_t, f16_t);
  else if (src_dtype == MXVL_BF16) MXVL_RG(bf16_t, float);
  else if (src_dtype == MXVL_F16) MXVL_RG(f16_t, float);
  else if (out_dtype == MXVL_BF16) MXVL_RG(float, bf16_t);
  else MXVL_RG(float, f16_t);
#undef MXVL_RG
  return mae_check();
}

extern "C" int mxvl_patch_loss(const void* img, const void* pred, const void* dloss, void* loss, void* dpred, int N, int C, int HW,
                               int patch, int norm_pix, int io_dtype, void* hip_stream) {
  if (!img || !pred || (!loss && !dpred) || (dpred && !dloss)) return MXVL_ERR_NULL;
  if (N <= 0 || C <= 0 || HW <= 0 || patch <= 0 || HW % patch != 0) return MXVL_ERR_SHAPE;
  if (io_dtype != MXVL_F32 && io_dtype != MXVL_BF16 && io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (norm_pix && patch * patch * C < 2) return MXVL_ERR_SHAPE;
  PatchLossArgs a;
  a.N = N; a.C = C; a.HW = HW; a.p = patch; a.gw = HW / patch; a.norm = norm_pix ? 1 : 0;
  a.img = (const float*)img; a.pred = pred; a.dloss = (const float*)dloss; a.loss = (float*)loss; a.dpred = dpred;
  dim3 grid(a.gw * a.gw, N);
  hipStream_t s = (hipStream_t)hip_stream;
  const bool bwd = dpred != nullptr;
#define MXVL_PL(E) \
  do { if (bwd) hipLaunchKernelGGL((patch_loss_kernel<E, true>), grid, dim3(256), 0, s, a); \
       else hipLaunchKernelGGL((patch_loss_kernel<E, false>), grid, dim3(256), 0, s, a); } while (0)
  switch (io_dtype) {
    case MXVL_F32: MXVL_PL(float); break;
    case MXVL_BF16: MXVL_PL(bf16_t); break;
    default: MXVL_PL(f16_t); break;
  }
#undef MXVL_PL
  return mae_check();
}

// ---- window rows of a CHANNELS-LAST feature map, the activation in front of it fused (ABI v11) ---------------------------------------
// SmallPatchEmbed's second convolution (HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:21-41: conv 16x16/s16 -> ReLU -> conv 4x4/s4 -> ReLU
// -> conv 1x1) reads the ReLU of the first one's output, a (256, 80, 80, 1024) map at the reference's 1280 x 1280 input: as torch
// expressions that was a ReLU pass, a strided copy into window rows, and in backward the inverse copy and threshold_backward -- four
// passes over a 3.3 GB tensor.  Here: cols[n][(i, j)][(di, dj, c)] = act(x[n][i k + di][j k + dj][c]) in one pass (16-byte units along the
// channel axis: a window row is k runs of k C contiguous elements), and the backward dx = dcols (inverse placement) * act'(x) in one pass
// from the PRE-activation map, which is the only tensor kept (the post-ReLU map never exists).
struct WindowArgs {
  int N, H, W, C, k, relu, backward;
  const void *x, *g;     // forward: x; backward: x (pre-activation, may be null without relu) and g = dcols
  void* out;             // forward: cols; backward: dx
};
template <typename io_t>
__global__ __launch_bounds__(256) void window_cols_kernel(const WindowArgs p) {
  constexpr int V = 16 / (int)sizeof(io_t);
  const int cv = p.C / V, gw = p.W / p.k;
  const int64_t total = (int64_t)p.N * p.H * p.W * cv;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    // idx walks the map (n, h, w, c-vector): reads of x (and stores of dx) are fully coalesced; the window side moves k C-element runs
    const int c8 = (int)(idx % cv);
    int64_t t = idx / cv;
    const int w = (int)(t % p.W);
    t /= p.W;
    const int h = (int)(t % p.H), n = (int)(t / p.H);
    const int i = h / p.k, di = h - i * p.k, j = w / p.k, dj = w - j * p.k;
    const int64_t col = ((((int64_t)n * (p.H / p.k) + i) * gw + j) * (p.k * p.k) + di * p.k + dj) * p.C + (int64_t)c8 * V;
    const int64_t src = idx * V;
    io_t v[V];
    if (!p.backward) {
      *(uint4*)v = *(const uint4*)((const io_t*)p.x + src);
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < V; ++e) Io<io_t>::st(v + e, fmaxf(Io<io_t>::ld(v + e), 0.0f));
      }
      *(uint4*)((io_t*)p.out + col) = *(const uint4*)v;
    } else {
      *(uint4*)v = *(const uint4*)((const io_t*)p.g + col);
      if (p.relu) {
        io_t xv[V];
        *(uint4*)xv = *(const uint4*)((const io_t*)p.x + src);
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (!(Io<io_t>::ld(xv + e) > 0.0f)) Io<io_t>::st(v + e, 0.0f);      // threshold_backward: the gradient passes where x > 0
      }
      *(uint4*)((io_t*)p.out + src) = *(const uint4*)v;
    }
  }
}

extern "C" int mxvl_window_cols(const void* x, const void* dcols, void* out, int N, int H, int W, int C, int k, int relu, int backward,
                                int io_dtype, void* hip_stream) {
  if (!out || (backward ? (!dcols || (relu && !x)) : !x)) return MXVL_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || H % k != 0 || W % k != 0) return MXVL_ERR_SHAPE;
  if (io_dtype != MXVL_F32 && io_dtype != MXVL_BF16 && io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  const int V = io_dtype == MXVL_F32 ? 4 : 8;
  if (C % V != 0 || (uintptr_t)out % 16 != 0 || (x && (uintptr_t)x % 16 != 0) || (dcols && (uintptr_t)dcols % 16 != 0)) return MXVL_ERR_UNSUPPORTED;
  WindowArgs a;
  a.N = N; a.H = H; a.W = W; a.C = C; a.k = k; a.relu = relu ? 1 : 0; a.backward = backward ? 1 : 0; a.x = x; a.g = dcols; a.out = out;
  const int64_t total = (int64_t)N * H * W * (C / V);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (io_dtype) {
    case MXVL_F32: hipLaunchKernelGGL(window_cols_kernel<float>, dim3(grid), dim3(256), 0, s, a); break;
    case MXVL_BF16: hipLaunchKernelGGL(window_cols_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(window_cols_kernel<f16_t>, dim3(grid), dim3(256), 0, s, a); break;
  }
  return mae_check();
}

extern "C" int mxvl_patch_cols(const void* img, void* cols, int N, int C, int H, int W, int patch, int in_dtype, int out_dtype,
                               void* hip_stream) {
  if (!img || !cols) return MXVL_ERR_NULL;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || patch <= 0 || H % patch != 0 || W % patch != 0) return MXVL_ERR_SHAPE;
  if (patch % 4 != 0 || 256 % patch != 0 || N > 65535 || C > 65535) return MXVL_ERR_UNSUPPORTED;
  auto ok = [](int d) { return d == MXVL_F32 || d == MXVL_BF16 || d == MXVL_F16; };
  if (!ok(in_dtype) || !ok(out_dtype)) return MXVL_ERR_DTYPE;
  if (in_dtype != out_dtype && in_dtype != MXVL_F32) return MXVL_ERR_DTYPE;      // a copy, or the autocast down-cast of an fp32 image
  const int esz_in = in_dtype == MXVL_F32 ? 4 : 2, esz_out = out_dtype == MXVL_F32 ? 4 : 2;
  if ((uintptr_t)img % (4 * esz_in) != 0 || (uintptr_t)cols % (4 * esz_out) != 0) return MXVL_ERR_UNSUPPORTED;
  PatchColsArgs a;
  a.N = N; a.C = C; a.H = H; a.W = W; a.p = patch; a.gw = W / patch; a.img = img; a.cols = cols;
  const dim3 grid(H / patch, C, N);
  const size_t lds = sizeof(float) * (size_t)patch * (256 + 4);
  hipStream_t s = (hipStream_t)hip_stream;
#define MXVL_PC(S, O) hipLaunchKernelGGL((patch_cols_kernel<S, O>), grid, dim3(256), lds, s, a)
  if (in_dtype == MXVL_F32 && out_dtype == MXVL_F32) MXVL_PC(float, float);
  else if (in_dtype == MXVL_F32 && out_dtype == MXVL_BF16) MXVL_PC(float, bf16_t);
  else if (in_dtype == MXVL_F32) MXVL_PC(float, f16_t);
  else if (in_dtype == MXVL_BF16) MXVL_PC(bf16_t, bf16_t);
  else MXVL_PC(f16_t, f16_t);
#undef MXVL_PC
  return mae_check();
}
