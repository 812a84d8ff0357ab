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

// ---- planes that fit the LDS (every stage of the reference's VSSMs: 56x56 ... 7x7) ---------------------------------------------
// The 32 x 32 tiles above leave 81 % of a workgroup idle on a 14 x 14 plane and move 2 bytes per lane.  Here a workgroup owns P
// consecutive channel planes of one batch element -- contiguous in x / y and, per direction, in xs / ys -- brings them into the LDS
// with flat V-element vector loads (V = 4: 8 bytes of 16-bit data, 16 of fp32; V = 1 when H * W is not a multiple of 4 or a base is
// not aligned) and writes every output as flat vectors gathered from the LDS: all four orders are permutations inside a plane.  Same
// arithmetic as the tiles (the three rounded adds of the reference), bit for bit.
struct CrossGeom {
  int C, H, W, L, P, LV;          // LV = L / V vectors per plane
  uint32_t magLV, magH, magW;     // floor(2^32 / d) + 1: exact quotients for n * d < 2^32 (n < P * L here); d = 1 has no such word
};
__device__ __forceinline__ int cross_div(int n, int d, uint32_t mag) { return d == 1 ? n : (int)__umulhi((uint32_t)n, mag); }

template <typename io_t, int V> struct CrossVec { typedef io_t type; };
template <> struct CrossVec<float, 4> { typedef float4 type; };
template <> struct CrossVec<bf16_t, 4> { typedef uint2 type; };
template <> struct CrossVec<f16_t, 4> { typedef uint2 type; };

template <typename io_t, int V>
__global__ __launch_bounds__(256) void cross_scan_flat_kernel(const io_t* __restrict__ x, io_t* __restrict__ xs, const CrossGeom g) {
  typedef typename CrossVec<io_t, V>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char cross_smem[];
  io_t* s = (io_t*)cross_smem;                         // [Pn][L]
  const int b = blockIdx.y, c0 = blockIdx.x * g.P;
  const int Pn = g.C - c0 < g.P ? g.C - c0 : g.P;
  const int L = g.L, LV = g.LV, H = g.H, W = g.W, nvec = Pn * LV;
  const io_t* src = x + ((size_t)b * g.C + c0) * L;
  for (int v = threadIdx.x; v < nvec; v += 256) *(vec_t*)(s + (size_t)v * V) = *(const vec_t*)(src + (size_t)v * V);
  __syncthreads();
  const size_t dir = (size_t)g.C * L;                  // elements between two directions of one batch element
  io_t* dst = xs + ((size_t)b * 4 * g.C + c0) * L;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    const int p = cross_div(v, LV, g.magLV), l0 = (v - p * LV) * V;
    const io_t* sp = s + p * L;
    io_t* o = dst + (size_t)p * L + l0;
    io_t t0[V], t1[V], t2[V], t3[V];
    // out1[l] = x[h * W + w] with l = w * H + h; out3[l] = out1[L - 1 - l]
    int w1 = cross_div(l0, H, g.magH), h1 = l0 - w1 * H;
    const int lr = L - 1 - l0;
    int w3 = cross_div(lr, H, g.magH), h3 = lr - w3 * H;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      t0[i] = sp[l0 + i];
      t2[i] = sp[lr - i];
      t1[i] = sp[h1 * W + w1];
      t3[i] = sp[h3 * W + w3];
      if (++h1 == H) { h1 = 0; ++w1; }
      if (--h3 < 0) { h3 = H - 1; --w3; }
    }
    *(vec_t*)o = *(const vec_t*)t0;
    *(vec_t*)(o + dir) = *(const vec_t*)t1;
    *(vec_t*)(o + 2 * dir) = *(const vec_t*)t2;
    *(vec_t*)(o + 3 * dir) = *(const vec_t*)t3;
  }
}

template <typename io_t, int V>
__global__ __launch_bounds__(256) void cross_merge_flat_kernel(const io_t* __restrict__ ys, io_t* __restrict__ y, const CrossGeom g) {
  typedef typename CrossVec<io_t, V>::type vec_t;
  using io = Io<io_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char cross_smem[];
  io_t* s = (io_t*)cross_smem;                         // [4][Pn][L]
  const int b = blockIdx.y, c0 = blockIdx.x * g.P;
  const int Pn = g.C - c0 < g.P ? g.C - c0 : g.P;
  const int L = g.L, LV = g.LV, H = g.H, W = g.W, nvec = Pn * LV, PL = Pn * L;
  const size_t dir = (size_t)g.C * L;
  const io_t* src = ys + ((size_t)b * 4 * g.C + c0) * L;
  // the four direction planes of a vector position are requested together, then written to the LDS: direction by direction every
  // load sat behind its own wait (one memory round trip per direction, read off the ISA)
  for (int v = threadIdx.x; v < nvec; v += 256) {
    vec_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = *(const vec_t*)(src + k * dir + (size_t)v * V);
#pragma unroll
    for (int k = 0; k < 4; ++k) *(vec_t*)(s + k * PL + (size_t)v * V) = r[k];
  }
  __syncthreads();
  io_t* dst = y + ((size_t)b * g.C + c0) * L;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    const int p = cross_div(v, LV, g.magLV), l0 = (v - p * LV) * V;
    const io_t* s0 = s + p * L;
    const io_t* s1 = s0 + PL;
    const io_t* s2 = s1 + PL;
    const io_t* s3 = s2 + PL;
    int h = cross_div(l0, W, g.magW), w = l0 - h * W;      // l = h * W + w
    io_t t[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int l = l0 + i, lt = w * H + h;
      const float a = rnd<io_t>(io::ld(s0 + l) + io::ld(s2 + (L - 1 - l)));
      const float c = rnd<io_t>(io::ld(s1 + lt) + io::ld(s3 + (L - 1 - lt)));
      io::st(t + i, a + c);
      if (++w == W) { w = 0; ++h; }
    }
    *(vec_t*)(dst + (size_t)p * L + l0) = *(const vec_t*)t;
  }
}

// planes per workgroup: ~16 KB (scan) / ~48 KB (merge: four directions) of LDS, >= 1024 workgroups where the problem has them
static int cross_planes_per_wg(bool merge, int B, int C, int L, int esz) {
  const long plane = (long)L * esz * (merge ? 4 : 1);
  if (plane > 60 * 1024) return 0;                     // does not fit: the tiles
  long P = (merge ? 48 * 1024 : 16 * 1024) / plane;
  if (P < 1) P = 1;
  if (P > 16) P = 16;
  while (P > 1 && (long)B * ((C + P - 1) / P) < 1024) --P;
  return (int)P;
}

template <typename io_t>
static int launch_cross(bool merge, const void* in, void* out, int B, int C, int H, int W, hipStream_t s) {
  const int L = H * W;
  const int P = cross_planes_per_wg(merge, B, C, L, (int)sizeof(io_t));
  if (P > 0 && B <= 65535 && (long)P * L * L < (1l << 32)) {
    const bool v4 = L % 4 == 0 && ((uintptr_t)in) % 16 == 0 && ((uintptr_t)out) % 16 == 0;
    CrossGeom g;
    g.C = C; g.H = H; g.W = W; g.L = L; g.P = P; g.LV = v4 ? L / 4 : L;
    g.magLV = (uint32_t)((1ull << 32) / (uint64_t)g.LV + 1ull);
    g.magH = (uint32_t)((1ull << 32) / (uint64_t)H + 1ull);
    g.magW = (uint32_t)((1ull << 32) / (uint64_t)W + 1ull);
    const dim3 grid((C + P - 1) / P, B);
    const size_t lds = (size_t)P * L * sizeof(io_t) * (merge ? 4 : 1);
    if (merge) {
      if (v4) hipLaunchKernelGGL((cross_merge_flat_kernel<io_t, 4>), grid, dim3(256), lds, s, (const io_t*)in, (io_t*)out, g);
      else hipLaunchKernelGGL((cross_merge_flat_kernel<io_t, 1>), grid, dim3(256), lds, s, (const io_t*)in, (io_t*)out, g);
    } else {
      if (v4) hipLaunchKernelGGL((cross_scan_flat_kernel<io_t, 4>), grid, dim3(256), lds, s, (const io_t*)in, (io_t*)out, g);
      else hipLaunchKernelGGL((cross_scan_flat_kernel<io_t, 1>), grid, dim3(256), lds, s, (const io_t*)in, (io_t*)out, g);
    }
    return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
  }
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
