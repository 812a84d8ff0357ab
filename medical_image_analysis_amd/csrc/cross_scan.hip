// cross_scan.hip -- the 4-direction 2-D scan orderings of VMamba's SS2D block for gfx950.
//
// Replaces the pure-PyTorch CrossScan / CrossMerge autograd functions (R2GenCSR/VMamba/classification/models/
// vmamba.py:25-67; Triton twins in csm_triton.py).  The reference builds xs with four strided copies (flatten,
// transpose+flatten, two flips) and merges with two flips, a transpose and three adds -- ~10 passes over the tensor.
// Both are pure HBM traffic, so each is ONE kernel here that reads every input element once and writes every output
// element once; the transposed directions go through a padded 32x32 LDS tile so that reads and writes are both
// coalesced 128-byte segments, and the flipped directions are the same tile written to mirrored addresses.
//   cross_scan : x (B,C,H,W)  -> xs (B,4,C,L)   xs[0]=row-major, xs[1]=column-major, xs[2]=flip(xs[0]), xs[3]=flip(xs[1])
//   cross_merge: ys (B,4,C,L) -> y (B,C,L)      y = (ys0 + flip(ys2)) + transpose(ys1 + flip(ys3))
// cross_merge is also CrossScan's backward and cross_scan is CrossMerge's backward (vmamba.py:37-44, 59-67).
// The adds are rounded to the io dtype after each step, exactly the three tensor adds of the reference: bit-exact.
#include "mxvl_common.h"

namespace mxvl {

constexpr int kTile = 32;

template <typename io_t>
__global__ __launch_bounds__(256) void cross_scan_kernel(const io_t* __restrict__ x, io_t* __restrict__ xs, int planes, int C, int H, int W) {
  __shared__ float tile[kTile][kTile + 1];
  using io = Io<io_t>;
  const int L = H * W;
  const int tw = (W + kTile - 1) / kTile;
  const int h0 = (blockIdx.x / tw) * kTile, w0 = (blockIdx.x % tw) * kTile;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {   // plane = b * C + c
  const int b = plane / C, c = plane - b * C;
  const io_t* src = x + (size_t)plane * L;
  io_t* o0 = xs + ((size_t)(b * 4 + 0) * C + c) * L;
  io_t* o1 = xs + ((size_t)(b * 4 + 1) * C + c) * L;
  io_t* o2 = xs + ((size_t)(b * 4 + 2) * C + c) * L;
  io_t* o3 = xs + ((size_t)(b * 4 + 3) * C + c) * L;
#pragma unroll
  for (int r = ty; r < kTile; r += 8) {
    const int h = h0 + r, w = w0 + tx;
    if (h < H && w < W) {
      const float v = io::ld(src + (size_t)h * W + w);
      tile[r][tx] = v;
      const int l = h * W + w;
      io::st(o0 + l, v);
      io::st(o2 + (L - 1 - l), v);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < kTile; r += 8) {
    const int w = w0 + r, h = h0 + tx;            // column-major position l = w*H + h: contiguous along h
    if (h < H && w < W) {
      const float v = tile[tx][r];
      const int l = w * H + h;
      io::st(o1 + l, v);
      io::st(o3 + (L - 1 - l), v);
    }
  }
  __syncthreads();
  }
}

template <typename io_t>
__device__ inline float rnd(float v) {  // value after a store + load in the io dtype (what a torch tensor add leaves)
  if constexpr (sizeof(io_t) == 4) return v;
  io_t t;
  Io<io_t>::st(&t, v);
  return Io<io_t>::ld(&t);
}

template <typename io_t>
__global__ __launch_bounds__(256) void cross_merge_kernel(const io_t* __restrict__ ys, io_t* __restrict__ y, int planes, int C, int H, int W) {
  __shared__ float tile[kTile][kTile + 1];
  using io = Io<io_t>;
  const int L = H * W;
  const int tw = (W + kTile - 1) / kTile;
  const int h0 = (blockIdx.x / tw) * kTile, w0 = (blockIdx.x % tw) * kTile;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
  const int b = plane / C, c = plane - b * C;
  const io_t* i0 = ys + ((size_t)(b * 4 + 0) * C + c) * L;
  const io_t* i1 = ys + ((size_t)(b * 4 + 1) * C + c) * L;
  const io_t* i2 = ys + ((size_t)(b * 4 + 2) * C + c) * L;
  const io_t* i3 = ys + ((size_t)(b * 4 + 3) * C + c) * L;
#pragma unroll
  for (int r = ty; r < kTile; r += 8) {           // column-major pair, read contiguous along h
    const int w = w0 + r, h = h0 + tx;
    if (h < H && w < W) {
      const int l = w * H + h;
      tile[tx][r] = rnd<io_t>(io::ld(i1 + l) + io::ld(i3 + (L - 1 - l)));
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < kTile; r += 8) {
    const int h = h0 + r, w = w0 + tx;
    if (h < H && w < W) {
      const int l = h * W + w;
      const float a = rnd<io_t>(io::ld(i0 + l) + io::ld(i2 + (L - 1 - l)));
      io::st(y + (size_t)plane * L + l, a + tile[r][tx]);
    }
  }
  __syncthreads();
  }
}

template <typename io_t>
static int launch_cross(bool merge, const void* in, void* out, int B, int C, int H, int W, hipStream_t s) {
  const int planes = B * C;
  const dim3 grid(((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile), planes < 65535 ? planes : 65535);
  if (merge)
    hipLaunchKernelGGL(cross_merge_kernel<io_t>, grid, dim3(256), 0, s, (const io_t*)in, (io_t*)out, planes, C, H, W);
  else
    hipLaunchKernelGGL(cross_scan_kernel<io_t>, grid, dim3(256), 0, s, (const io_t*)in, (io_t*)out, planes, C, H, W);
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

static int cross_dispatch(bool merge, const void* in, void* out, int B, int C, int H, int W, int dtype, void* stream) {
  if (!in || !out) return MXVL_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return MXVL_ERR_SHAPE;
  if ((long long)B * C > 2147483647LL || (long long)H * W > 2147483647LL) return MXVL_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case MXVL_F32: return launch_cross<float>(merge, in, out, B, C, H, W, s);
    case MXVL_BF16: return launch_cross<bf16_t>(merge, in, out, B, C, H, W, s);
    case MXVL_F16: return launch_cross<f16_t>(merge, in, out, B, C, H, W, s);
    default: return MXVL_ERR_DTYPE;
  }
}

}  // namespace mxvl

extern "C" {

int mxvl_cross_scan(const void* x, void* xs, int batch, int channels, int height, int width, int io_dtype, void* hip_stream) {
  return mxvl::cross_dispatch(false, x, xs, batch, channels, height, width, io_dtype, hip_stream);
}

int mxvl_cross_merge(const void* ys, void* y, int batch, int channels, int height, int width, int io_dtype, void* hip_stream) {
  return mxvl::cross_dispatch(true, ys, y, batch, channels, height, width, io_dtype, hip_stream);
}

}  // extern "C"
