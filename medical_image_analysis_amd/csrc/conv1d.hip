// conv1d.hip -- depthwise causal conv1d (+SiLU) forward/backward, and the two single-token decode
// steps of the Mamba mixer, for gfx950.
//
// Replaces (reference paths relative to CXPMRG_Bench_MambaXray_VL/arm/Finetuning/):
//   mxvl_conv1d_fwd/bwd   causal_conv1d_fn (third-party wheel, call site mamba_simple.py:676-681); semantics =
//                         the in-repo fallback act(conv1d(x)[..., :L]) with nn.Conv1d(D, D, W, groups=D,
//                         padding=W-1) (mamba_simple.py:78-86, 672-673)
//   mxvl_conv1d_update    causal_conv1d_update (call site :732-738), fallback roll + dot (:724-730)
//   mxvl_state_update     selective_state_update (call site :757-759), fallback :748-755
// All are HBM-bound streaming kernels: algorithmic bytes = 2*elt*B*D*L (fwd), 3*elt*B*D*L (bwd).
// (B,D,L) rows are L-contiguous and usually NOT 16-byte aligned (L = 197), so a thread owns 4 consecutive
// steps of one row and reads its 3-step halo through L1; consecutive threads cover consecutive steps.
#include <algorithm>

#include "mxvl_common.h"

namespace mxvl {

constexpr int kMaxW = 8;

struct ConvArgs {
  int batch, dim, L, W, silu, vec;
  int64_t x_bs, x_ds, y_bs, y_ds, dy_bs, dy_ds, dx_bs, dx_ds;
  const void *x, *dy;
  const float *w, *bias;
  void *y, *dx;
  float *dw, *dbias;
};

// runtime-indexed register arrays would go to scratch: select chain instead (generic-width path only)
__device__ inline float pick(const float (&w)[kMaxW], int idx) {
  float v = 0.0f;
#pragma unroll
  for (int k = 0; k < kMaxW; ++k) v = (k == idx) ? w[k] : v;
  return v;
}

// WT = compile-time tap count (4 is Mamba's d_conv); WT = 0 reads the width at run time (<= kMaxW)
template <typename io_t, int WT>
__global__ __launch_bounds__(256) void conv1d_fwd_kernel(const ConvArgs p) {
  using io = Io<io_t>;
  const int nq = (p.L + 3) / 4;
  const int64_t total = (int64_t)p.batch * p.dim * nq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % nq);
    const int64_t rowi = idx / nq;
    const int d = (int)(rowi % p.dim), b = (int)(rowi / p.dim);
    const io_t* xr = (const io_t*)p.x + (int64_t)b * p.x_bs + (int64_t)d * p.x_ds;
    io_t* yr = (io_t*)p.y + (int64_t)b * p.y_bs + (int64_t)d * p.y_ds;
    const int t0 = q * 4, W = WT ? WT : p.W;
    float w[kMaxW];
#pragma unroll
    for (int k = 0; k < kMaxW; ++k) w[k] = k < W ? p.w[(int64_t)d * W + k] : 0.0f;
    const float bias = p.bias ? p.bias[d] : 0.0f;
    float xv[kMaxW + 3];  // x[t0-(W-1) .. t0+3]
#pragma unroll
    for (int k = 0; k < kMaxW + 3; ++k) {
      const int t = t0 - (W - 1) + k;
      xv[k] = (k < W + 3 && t >= 0 && t < p.L) ? io::ld(xr + t) : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (t0 + i < p.L) {
        float acc = bias;
#pragma unroll
        for (int k = 0; k < kMaxW; ++k)
          if (k < W) acc = fmaf(w[k], xv[i + k], acc);
        io::st(yr + t0 + i, p.silu ? silu(acc) : acc);
      }
    }
  }
}

// Aligned rows (row starts and strides multiples of 8 elements, W <= 5): a thread owns 8 consecutive steps, reads them
// as two 4-element vectors plus ONE 4-element vector of halo, writes two vectors -- 3 loads + 2 stores per 8 outputs
// instead of 11 + 8 two-byte accesses (the scalar kernel above is load-issue bound: 87 us at B8 D1024 L4080 bf16).
template <typename io_t, int WT>
__global__ __launch_bounds__(256) void conv1d_fwd_vec_kernel(const ConvArgs p) {
  static_assert(WT >= 1 && WT <= 5, "halo of one 4-vector");
  const int nq = (p.L + 7) / 8;
  const int64_t total = (int64_t)p.batch * p.dim * nq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % nq);
    const int64_t rowi = idx / nq;
    const int d = (int)(rowi % p.dim), b = (int)(rowi / p.dim);
    const io_t* xr = (const io_t*)p.x + (int64_t)b * p.x_bs + (int64_t)d * p.x_ds;
    io_t* yr = (io_t*)p.y + (int64_t)b * p.y_bs + (int64_t)d * p.y_ds;
    const int t0 = q * 8;
    float w[WT];
#pragma unroll
    for (int k = 0; k < WT; ++k) w[k] = p.w[(int64_t)d * WT + k];
    const float bias = p.bias ? p.bias[d] : 0.0f;
    float xv[12];   // x[t0-4 .. t0+8)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // every load is issued, from an in-range address, and the halo is zeroed by a select afterwards: as `cond ? load : 0` each
    // guarded load sat in its own basic block behind an `s_waitcnt vmcnt(0)` -- three serialised memory round trips per thread
    // (read off the ISA, round 6)
    const bool has_h = t0 > 0, has_1 = t0 + 4 < p.L;
    const float4 h_r = ld4<io_t>(xr + (has_h ? t0 - 4 : t0));
    const float4 a0 = ld4<io_t>(xr + t0);                       // L % 4 == 0 on this path: [t0, t0+4) is in range
    const float4 a1_r = ld4<io_t>(xr + (has_1 ? t0 + 4 : t0));
    const float4 h = has_h ? h_r : z4, a1 = has_1 ? a1_r : z4;
    xv[0] = h.x; xv[1] = h.y; xv[2] = h.z; xv[3] = h.w;
    xv[4] = a0.x; xv[5] = a0.y; xv[6] = a0.z; xv[7] = a0.w;
    xv[8] = a1.x; xv[9] = a1.y; xv[10] = a1.z; xv[11] = a1.w;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float acc = bias;
#pragma unroll
      for (int k = 0; k < WT; ++k) acc = fmaf(w[k], xv[i + 4 - (WT - 1) + k], acc);
      o[i] = p.silu ? silu(acc) : acc;
    }
    st4<io_t>(yr + t0, make_float4(o[0], o[1], o[2], o[3]));
    if (t0 + 4 < p.L) st4<io_t>(yr + t0 + 4, make_float4(o[4], o[5], o[6], o[7]));
  }
}

// Backward for aligned rows (the vector path's conditions, W = WT <= 5, SiLU or not): a thread owns 8 consecutive steps and
// works from registers only -- 4 vector loads of x ([t0-4, t0+12): the pre-activation of steps t0 .. t0+10 is recomputed),
// 3 of dy, 2 vector stores of dx; the overlapping halves of neighbouring threads' loads are L1 hits.  dweight / dbias: wave
// shuffle + one LDS hop + ONE fp32 atomic per (d, k) per workgroup.  No LDS staging, one barrier.  (The LDS-staged kernel
// below needs 3 barriers per 1024 steps and stores dx two bytes at a time: 292 us at B16 D1024 L4080 bf16 = 2.75 TB/s.)
// per-thread body of the vector backward: 8 steps t0 .. t0+7 of one row from registers; adds into dw_acc / db_acc
template <typename io_t, int WT>
__device__ __forceinline__ void conv_bwd_vec_body(const io_t* xr, const io_t* gr, io_t* dxr, int t0, int L, const float (&w)[WT],
                                                  float bias, int silu_on, float (&dw_acc)[WT], float& db_acc) {
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float xv[16], g[12];   // x[t0-4 .. t0+12), dy[t0 .. t0+12)
  // all seven loads issued from in-range addresses, the halo zeroed by selects afterwards (guarded loads serialise: see the forward)
  const bool has_h = t0 > 0, has_1 = t0 + 4 < L, has_2 = t0 + 8 < L;
  const int o1 = has_1 ? t0 + 4 : t0, o2 = has_2 ? t0 + 8 : t0;
  const float4 x0r = ld4<io_t>(xr + (has_h ? t0 - 4 : t0)), x1 = ld4<io_t>(xr + t0), x2r = ld4<io_t>(xr + o1), x3r = ld4<io_t>(xr + o2);
  const float4 g0 = ld4<io_t>(gr + t0), g1r = ld4<io_t>(gr + o1), g2r = ld4<io_t>(gr + o2);
  const float4 x0 = has_h ? x0r : z4, x2 = has_1 ? x2r : z4, x3 = has_2 ? x3r : z4, g1 = has_1 ? g1r : z4, g2 = has_2 ? g2r : z4;
  xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
  xv[8] = x2.x; xv[9] = x2.y; xv[10] = x2.z; xv[11] = x2.w; xv[12] = x3.x; xv[13] = x3.y; xv[14] = x3.z; xv[15] = x3.w;
  g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
  g[8] = g2.x; g[9] = g2.y; g[10] = g2.z; g[11] = g2.w;
  // d(pre-activation) of steps t0 .. t0 + 7 + (WT - 1); pre[t] = bias + sum_k w[k] x[t - (WT-1) + k], x[t0 + i] = xv[4 + i]
  if (silu_on) {
#pragma unroll
    for (int i = 0; i < 8 + WT - 1; ++i) {
      float pre = bias;
#pragma unroll
      for (int k = 0; k < WT; ++k) pre = fmaf(w[k], xv[4 + i - (WT - 1) + k], pre);
      const float sgm = sigmoid(pre);
      g[i] *= sgm * fmaf(pre, 1.0f - sgm, 1.0f);
    }
  }
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int m = 0; m < WT; ++m) acc = fmaf(w[WT - 1 - m], g[i + m], acc);   // dx[s] = sum_m w[W-1-m] dpre[s+m]; dpre beyond L is 0
    o[i] = acc;
    db_acc += g[i];
#pragma unroll
    for (int k = 0; k < WT; ++k) dw_acc[k] = fmaf(g[i], xv[4 + i - (WT - 1) + k], dw_acc[k]);
  }
  st4<io_t>(dxr + t0, make_float4(o[0], o[1], o[2], o[3]));
  if (t0 + 4 < L) st4<io_t>(dxr + t0 + 4, make_float4(o[4], o[5], o[6], o[7]));
}

template <typename io_t, int WT>
__global__ __launch_bounds__(128) void conv1d_bwd_vec_kernel(const ConvArgs p) {
  static_assert(WT >= 1 && WT <= 5, "halo of one 4-vector on each side");
  constexpr int TILE = 1024;
  const int L = p.L, d = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * TILE + threadIdx.x * 8;
  __shared__ float red[2][WT + 1];
  const io_t* xr = (const io_t*)p.x + (int64_t)b * p.x_bs + (int64_t)d * p.x_ds;
  const io_t* gr = (const io_t*)p.dy + (int64_t)b * p.dy_bs + (int64_t)d * p.dy_ds;
  io_t* dxr = (io_t*)p.dx + (int64_t)b * p.dx_bs + (int64_t)d * p.dx_ds;
  float w[WT];
#pragma unroll
  for (int k = 0; k < WT; ++k) w[k] = p.w[(int64_t)d * WT + k];
  const float bias = p.bias ? p.bias[d] : 0.0f;
  float dw_acc[WT], db_acc = 0.0f;
#pragma unroll
  for (int k = 0; k < WT; ++k) dw_acc[k] = 0.0f;
  if (t0 < L) conv_bwd_vec_body<io_t, WT>(xr, gr, dxr, t0, L, w, bias, p.silu, dw_acc, db_acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= WT; ++k) {
    float v = (k < WT) ? dw_acc[k] : db_acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x <= WT) {
    const int k = threadIdx.x;
    const float v = red[0][k] + red[1][k];
    if (k < WT) unsafeAtomicAdd(p.dw + (int64_t)d * WT + k, v);
    else if (p.dbias) unsafeAtomicAdd(p.dbias + d, v);
  }
}

// The same register-only body for SHORT aligned rows (L <= 512: the 197 -> 200-step rows of the 224x224 encoders): a workgroup
// owns one channel and as many batch rows as its 256 threads cover at 8 steps per thread (10 rows at L = 200), one reduction +
// atomic set per workgroup.  The LDS-staged short-row kernel below moves two bytes per load: 213 us at B64 D4096 L200 bf16
// (1.5 TB/s).
template <typename io_t, int WT>
__global__ __launch_bounds__(256) void conv1d_bwd_vec_rows_kernel(const ConvArgs p, int TPR, int RW) {
  const int L = p.L, d = blockIdx.y;
  const int s_ = threadIdx.x / TPR, c = threadIdx.x - s_ * TPR;
  const int b = blockIdx.x * RW + s_, t0 = c * 8;
  __shared__ float red[4][WT + 1];
  float w[WT];
#pragma unroll
  for (int k = 0; k < WT; ++k) w[k] = p.w[(int64_t)d * WT + k];
  const float bias = p.bias ? p.bias[d] : 0.0f;
  float dw_acc[WT], db_acc = 0.0f;
#pragma unroll
  for (int k = 0; k < WT; ++k) dw_acc[k] = 0.0f;
  if (s_ < RW && b < p.batch && t0 < L) {
    const io_t* xr = (const io_t*)p.x + (int64_t)b * p.x_bs + (int64_t)d * p.x_ds;
    const io_t* gr = (const io_t*)p.dy + (int64_t)b * p.dy_bs + (int64_t)d * p.dy_ds;
    io_t* dxr = (io_t*)p.dx + (int64_t)b * p.dx_bs + (int64_t)d * p.dx_ds;
    conv_bwd_vec_body<io_t, WT>(xr, gr, dxr, t0, L, w, bias, p.silu, dw_acc, db_acc);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= WT; ++k) {
    float v = (k < WT) ? dw_acc[k] : db_acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x <= WT) {
    const int k = threadIdx.x;
    const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    if (k < WT) unsafeAtomicAdd(p.dw + (int64_t)d * WT + k, v);
    else if (p.dbias) unsafeAtomicAdd(p.dbias + d, v);
  }
}

// Backward: one workgroup per (1024-step tile, d, b).  x and dy tiles (+ halos) are read ONCE, coalesced, into LDS;
// phase A turns dy into d(pre-activation) in place, phase B forms dx[s] = sum_m w[W-1-m] * dpre[s+m] and the
// per-thread dw / dbias partial sums, reduced in the block and added with one fp32 atomic per (d, k) per block.
// (The first version re-loaded 7+4 scalars per element straight from global memory: 452 us at B8 D1024 L4080 bf16
// against a 50 us HBM bound -- load-issue bound.)
template <typename io_t, int WT>
__global__ __launch_bounds__(256) void conv1d_bwd_kernel(const ConvArgs p) {
  using io = Io<io_t>;
  constexpr int TILE = 1024, PT = 4;
  const int W = WT ? WT : p.W, L = p.L;
  const int t0 = blockIdx.x * TILE, d = blockIdx.y, b = blockIdx.z;
  __shared__ float sx[TILE + 2 * (kMaxW - 1)];   // x[t0-(W-1) .. t0+TILE+(W-1))
  __shared__ __attribute__((aligned(16))) float sg[TILE + (kMaxW - 1)];       // dy, then dpre, for t in [t0, t0+TILE+W-1)
  __shared__ float red[4][kMaxW + 1];
  const io_t* xr = (const io_t*)p.x + (int64_t)b * p.x_bs + (int64_t)d * p.x_ds;
  const io_t* gr = (const io_t*)p.dy + (int64_t)b * p.dy_bs + (int64_t)d * p.dy_ds;
  io_t* dxr = (io_t*)p.dx + (int64_t)b * p.dx_bs + (int64_t)d * p.dx_ds;
  float w[kMaxW];
#pragma unroll
  for (int k = 0; k < kMaxW; ++k) w[k] = k < W ? p.w[(int64_t)d * W + k] : 0.0f;
  const float bias = p.bias ? p.bias[d] : 0.0f;

  if (p.vec && t0 + TILE <= L) {   // aligned full tile: body as 4-element vectors, halos as scalars
    {
      const int i = threadIdx.x * 4;
      const float4 xv = ld4<io_t>(xr + t0 + i), gv = ld4<io_t>(gr + t0 + i);
      float* dxs = sx + (W - 1) + i;      // sx is not 16-byte aligned at W-1: scalar LDS stores
      dxs[0] = xv.x; dxs[1] = xv.y; dxs[2] = xv.z; dxs[3] = xv.w;
      *(float4*)(sg + i) = gv;
    }
    if (threadIdx.x < 2 * (W - 1)) {
      const int i = threadIdx.x < W - 1 ? threadIdx.x : TILE + threadIdx.x;   // left halo | right halo of sx
      const int t = t0 - (W - 1) + i;
      sx[i] = (t >= 0 && t < L) ? io::ld(xr + t) : 0.0f;
    } else if (threadIdx.x < 3 * (W - 1)) {
      const int i = TILE + threadIdx.x - 2 * (W - 1);
      sg[i] = (t0 + i < L) ? io::ld(gr + t0 + i) : 0.0f;
    }
  } else {
    for (int i = threadIdx.x; i < TILE + 2 * (W - 1); i += 256) {
      const int t = t0 - (W - 1) + i;
      sx[i] = (t >= 0 && t < L) ? io::ld(xr + t) : 0.0f;
    }
    for (int i = threadIdx.x; i < TILE + (W - 1); i += 256) {
      const int t = t0 + i;
      sg[i] = (t < L) ? io::ld(gr + t) : 0.0f;
    }
  }
  __syncthreads();
  if (p.silu) {  // phase A: dpre[t] = dy[t] * silu'(pre[t]); pre[t] uses sx[(t-t0) .. (t-t0)+W-1]
    for (int i = threadIdx.x; i < TILE + (W - 1); i += 256) {
      float pre = bias;
#pragma unroll
      for (int k = 0; k < kMaxW; ++k)
        if (k < W) pre = fmaf(w[k], sx[i + k], pre);
      const float sgm = sigmoid(pre);
      sg[i] *= sgm * fmaf(pre, 1.0f - sgm, 1.0f);
    }
    __syncthreads();
  }
  float dw_acc[kMaxW], db_acc = 0.0f;
#pragma unroll
  for (int k = 0; k < kMaxW; ++k) dw_acc[k] = 0.0f;
#pragma unroll
  for (int q = 0; q < PT; ++q) {
    const int i = threadIdx.x + q * 256;  // consecutive threads -> consecutive steps (coalesced store)
    const int sidx = t0 + i;
    if (sidx < L) {
      float dxs = 0.0f;
#pragma unroll
      for (int m = 0; m < kMaxW; ++m) {
        if (m < W) {
          float wr;
          if constexpr (WT > 0) wr = w[(WT - 1 - m) & (kMaxW - 1)];
          else wr = pick(w, W - 1 - m);
          dxs = fmaf(wr, sg[i + m], dxs);  // sg beyond L is zero
        }
      }
      io::st(dxr + sidx, dxs);
      const float g = sg[i];
      db_acc += g;
#pragma unroll
      for (int k = 0; k < kMaxW; ++k)
        if (k < W) dw_acc[k] = fmaf(g, sx[i + k], dw_acc[k]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= kMaxW; ++k) {
    float v = (k < kMaxW) ? dw_acc[k] : db_acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x <= kMaxW) {
    const int k = threadIdx.x;
    const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    if (k < W) unsafeAtomicAdd(p.dw + (int64_t)d * W + k, v);
    if (k == kMaxW && p.dbias) unsafeAtomicAdd(p.dbias + d, v);
  }
}

// Short rows (L + W - 1 <= 512, e.g. the 197-token sequences of the 224x224 encoders): one workgroup per 1024-step tile would
// leave most of it empty and pay a block reduction + 5 atomics per row (820 us at B64 D4096 L197).  Here a workgroup takes
// S = 1024 / (L + W - 1) batch rows of ONE channel at once -- rows side by side in LDS, each with its causal zero halo --
// and reduces dweight / dbias once for all of them.
template <typename io_t, int WT>
__global__ __launch_bounds__(256) void conv1d_bwd_short_kernel(const ConvArgs p, int S) {
  using io = Io<io_t>;
  constexpr int CAP = 1024 + 8;
  const int L = p.L, RS = L + (WT - 1);            // LDS row stride: x carries WT-1 leading zeros, dpre WT-1 trailing zeros
  __shared__ float sx[CAP];
  __shared__ float sg[CAP];
  __shared__ float red[4][kMaxW + 1];
  const int d = blockIdx.y, b0 = blockIdx.x * S;
  const int rows = min(S, p.batch - b0);
  const int n = rows * L;
  float w[WT];
#pragma unroll
  for (int k = 0; k < WT; ++k) w[k] = p.w[(int64_t)d * WT + k];
  const float bias = p.bias ? p.bias[d] : 0.0f;
  for (int i = threadIdx.x; i < rows * RS; i += 256) {   // halos
    const int s_ = i / RS, c = i - s_ * RS;
    if (c < WT - 1) sx[s_ * RS + c] = 0.0f;
    if (c >= L) sg[s_ * RS + c] = 0.0f;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int s_ = i / L, t = i - s_ * L;
    const int64_t bo = (int64_t)(b0 + s_);
    sx[s_ * RS + (WT - 1) + t] = io::ld((const io_t*)p.x + bo * p.x_bs + (int64_t)d * p.x_ds + t);
    sg[s_ * RS + t] = io::ld((const io_t*)p.dy + bo * p.dy_bs + (int64_t)d * p.dy_ds + t);
  }
  __syncthreads();
  if (p.silu) {
    for (int i = threadIdx.x; i < n; i += 256) {
      const int s_ = i / L, t = i - s_ * L;
      const float* xr = sx + s_ * RS + t;
      float pre = bias;
#pragma unroll
      for (int k = 0; k < WT; ++k) pre = fmaf(w[k], xr[k], pre);
      const float sgm = sigmoid(pre);
      sg[s_ * RS + t] *= sgm * fmaf(pre, 1.0f - sgm, 1.0f);
    }
    __syncthreads();
  }
  float dw_acc[WT], db_acc = 0.0f;
#pragma unroll
  for (int k = 0; k < WT; ++k) dw_acc[k] = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int s_ = i / L, t = i - s_ * L;
    const float* gr = sg + s_ * RS + t;
    const float* xr = sx + s_ * RS + t;
    float dxs = 0.0f;
#pragma unroll
    for (int m = 0; m < WT; ++m) dxs = fmaf(w[WT - 1 - m], gr[m], dxs);
    io::st((io_t*)p.dx + (int64_t)(b0 + s_) * p.dx_bs + (int64_t)d * p.dx_ds + t, dxs);
    const float g = gr[0];
    db_acc += g;
#pragma unroll
    for (int k = 0; k < WT; ++k) dw_acc[k] = fmaf(g, xr[k], dw_acc[k]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= WT; ++k) {
    float v = (k < WT) ? dw_acc[k] : db_acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x <= WT) {
    const int k = threadIdx.x;
    const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    if (k < WT) unsafeAtomicAdd(p.dw + (int64_t)d * WT + k, v);
    else if (p.dbias) unsafeAtomicAdd(p.dbias + d, v);
  }
}

template <typename io_t>
__global__ __launch_bounds__(256) void conv1d_update_kernel(const io_t* x, io_t* state, const float* w,
                                                            const float* bias, io_t* y, int batch, int dim,
                                                            int W, int silu_on) {
  using io = Io<io_t>;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)batch * dim) return;
  const int d = (int)(idx % dim);
  io_t* st = state + idx * W;
  float acc = bias ? bias[d] : 0.0f;
  for (int k = 0; k + 1 < W; ++k) {
    const float v = io::ld(st + k + 1);
    io::st(st + k, v);
    acc = fmaf(v, w[(int64_t)d * W + k], acc);
  }
  const float xn = io::ld(x + idx);
  io::st(st + W - 1, xn);
  acc = fmaf(xn, w[(int64_t)d * W + W - 1], acc);
  io::st(y + idx, silu_on ? silu(acc) : acc);
}

template <typename io_t>
__global__ __launch_bounds__(256) void state_update_kernel(float* state, const io_t* x, const io_t* dt, const float* A,
                                                           const io_t* B, const io_t* C, const float* D, const io_t* z,
                                                           const float* dt_bias, io_t* out, int batch, int dim, int N,
                                                           int softplus_on) {
  using io = Io<io_t>;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)batch * dim) return;
  const int d = (int)(idx % dim), b = (int)(idx / dim);
  float dtv = io::ld(dt + idx) + (dt_bias ? dt_bias[d] : 0.0f);
  if (softplus_on) dtv = softplus(dtv);
  const float xv = io::ld(x + idx);
  float* st = state + idx * N;
  float y = 0.0f;
  for (int n = 0; n < N; ++n) {
    const float h = fmaf(st[n], fast_exp(dtv * A[(int64_t)d * N + n]), xv * dtv * io::ld(B + (int64_t)b * N + n));
    st[n] = h;
    y = fmaf(h, io::ld(C + (int64_t)b * N + n), y);
  }
  if (D) y = fmaf(D[d], xv, y);
  if (z) y *= silu(io::ld(z + idx));
  io::st(out + idx, y);
}

// every row start a multiple of 4 elements and 16 / 8 bytes (fp32 / half) aligned
static bool rows_aligned(const void* base, int64_t bs, int64_t ds, int io_dtype) {
  const uintptr_t al = io_dtype == MXVL_F32 ? 16 : 8;
  return base && ((uintptr_t)base % al) == 0 && bs % 4 == 0 && ds % 4 == 0;
}

static int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

static int conv_check(const mxvl_conv1d_desc* d) {
  if (!d || !d->x || !d->weight) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_F32 && d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->batch <= 0 || d->dim <= 0 || d->seqlen <= 0 || d->width <= 0) return MXVL_ERR_SHAPE;
  if (d->width > kMaxW) return MXVL_ERR_UNSUPPORTED;
  if (d->x_bs < 0 || d->x_ds < 0) return MXVL_ERR_STRIDE;
  return MXVL_OK;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" {

int mxvl_conv1d_fwd(const mxvl_conv1d_desc* d, void* hip_stream) {
  int rc = conv_check(d);
  if (rc != MXVL_OK) return rc;
  if (!d->y) return MXVL_ERR_NULL;
  ConvArgs a{};
  a.batch = d->batch; a.dim = d->dim; a.L = d->seqlen; a.W = d->width; a.silu = d->silu;
  a.x_bs = d->x_bs; a.x_ds = d->x_ds; a.y_bs = d->y_bs; a.y_ds = d->y_ds;
  a.x = d->x; a.w = (const float*)d->weight; a.bias = (const float*)d->bias; a.y = d->y;
  const bool vec = a.W == 4 && a.L % 4 == 0 && rows_aligned(d->x, d->x_bs, d->x_ds, d->io_dtype) &&
                   rows_aligned(d->y, d->y_bs, d->y_ds, d->io_dtype);
  const int64_t total = (int64_t)a.batch * a.dim * (vec ? (a.L + 7) / 8 : (a.L + 3) / 4);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
  hipStream_t s = (hipStream_t)hip_stream;
#define MXVL_CONV_FWD(T) \
  do { \
    if (vec) hipLaunchKernelGGL((conv1d_fwd_vec_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, a); \
    else if (a.W == 4) hipLaunchKernelGGL((conv1d_fwd_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((conv1d_fwd_kernel<T, 0>), dim3(blocks), dim3(256), 0, s, a); \
  } while (0)
  switch (d->io_dtype) {
    case MXVL_F32: MXVL_CONV_FWD(float); break;
    case MXVL_BF16: MXVL_CONV_FWD(bf16_t); break;
    default: MXVL_CONV_FWD(f16_t); break;
  }
  return check_launch();
}

int mxvl_conv1d_bwd(const mxvl_conv1d_bwd_desc* d, void* hip_stream) {
  if (!d) return MXVL_ERR_NULL;
  int rc = conv_check(&d->fwd);
  if (rc != MXVL_OK) return rc;
  if (!d->dy || !d->dx || !d->dweight) return MXVL_ERR_NULL;
  if (d->fwd.bias && !d->dbias) return MXVL_ERR_NULL;
  ConvArgs a{};
  a.batch = d->fwd.batch; a.dim = d->fwd.dim; a.L = d->fwd.seqlen; a.W = d->fwd.width; a.silu = d->fwd.silu;
  a.x_bs = d->fwd.x_bs; a.x_ds = d->fwd.x_ds; a.dy_bs = d->dy_bs; a.dy_ds = d->dy_ds; a.dx_bs = d->dx_bs; a.dx_ds = d->dx_ds;
  a.x = d->fwd.x; a.w = (const float*)d->fwd.weight; a.bias = (const float*)d->fwd.bias;
  a.dy = d->dy; a.dx = d->dx; a.dw = (float*)d->dweight; a.dbias = (float*)d->dbias;
  a.vec = (rows_aligned(d->fwd.x, d->fwd.x_bs, d->fwd.x_ds, d->fwd.io_dtype) &&
           rows_aligned(d->dy, d->dy_bs, d->dy_ds, d->fwd.io_dtype)) ? 1 : 0;
  dim3 grid((a.L + 1023) / 1024, a.dim, a.batch);
  hipStream_t s = (hipStream_t)hip_stream;
  const int S = (a.W == 4 && a.L + 3 <= 512 && a.dim <= 65535) ? 1024 / (a.L + 3) : 0;   // batch rows per workgroup of the short-row kernel
  const dim3 sgrid(S ? (a.batch + S - 1) / S : 1, a.dim);
  // register-only vector kernel: aligned x / dy / dx rows of a multiple of 4 steps, d_conv 4, long rows
  const bool bvec = a.vec && a.W == 4 && a.L % 4 == 0 && a.L > 512 && rows_aligned(d->dx, d->dx_bs, d->dx_ds, d->fwd.io_dtype);
  // short aligned rows: the register-only body, several batch rows per workgroup
  const bool svec = a.vec && a.W == 4 && a.L % 4 == 0 && a.L <= 512 && a.dim <= 65535 && rows_aligned(d->dx, d->dx_bs, d->dx_ds, d->fwd.io_dtype);
  const int TPR = (a.L + 7) / 8, RW = svec ? 256 / TPR : 0;          // L <= 512 -> TPR <= 64, RW >= 4
  const dim3 vgrid(RW > 0 ? (a.batch + RW - 1) / RW : 1, a.dim);
#define MXVL_CONV_BWD(T) \
  do { \
    if (svec && RW >= 2) hipLaunchKernelGGL((conv1d_bwd_vec_rows_kernel<T, 4>), vgrid, dim3(256), 0, s, a, TPR, RW); \
    else if (S >= 2) hipLaunchKernelGGL((conv1d_bwd_short_kernel<T, 4>), sgrid, dim3(256), 0, s, a, S); \
    else if (bvec) hipLaunchKernelGGL((conv1d_bwd_vec_kernel<T, 4>), grid, dim3(128), 0, s, a); \
    else if (a.W == 4) hipLaunchKernelGGL((conv1d_bwd_kernel<T, 4>), grid, dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((conv1d_bwd_kernel<T, 0>), grid, dim3(256), 0, s, a); \
  } while (0)
  switch (d->fwd.io_dtype) {
    case MXVL_F32: MXVL_CONV_BWD(float); break;
    case MXVL_BF16: MXVL_CONV_BWD(bf16_t); break;
    default: MXVL_CONV_BWD(f16_t); break;
  }
  return check_launch();
}

int mxvl_conv1d_update(const void* x, void* conv_state, const void* weight, const void* bias, void* y, int batch,
                       int dim, int width, int io_dtype, int silu_on, void* hip_stream) {
  if (!x || !conv_state || !weight || !y) return MXVL_ERR_NULL;
  if (batch <= 0 || dim <= 0 || width <= 0) return MXVL_ERR_SHAPE;
  const int blocks = (int)(((int64_t)batch * dim + 255) / 256);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (io_dtype) {
    case MXVL_F32:
      hipLaunchKernelGGL(conv1d_update_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)conv_state,
                         (const float*)weight, (const float*)bias, (float*)y, batch, dim, width, silu_on);
      break;
    case MXVL_BF16:
      hipLaunchKernelGGL(conv1d_update_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)conv_state,
                         (const float*)weight, (const float*)bias, (bf16_t*)y, batch, dim, width, silu_on);
      break;
    case MXVL_F16:
      hipLaunchKernelGGL(conv1d_update_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, (const f16_t*)x, (f16_t*)conv_state,
                         (const float*)weight, (const float*)bias, (f16_t*)y, batch, dim, width, silu_on);
      break;
    default: return MXVL_ERR_DTYPE;
  }
  return check_launch();
}

int mxvl_state_update(void* state, const void* x, const void* dt, const void* A, const void* B, const void* C,
                      const void* D, const void* z, const void* dt_bias, void* out, int batch, int dim, int dstate,
                      int io_dtype, int dt_softplus, void* hip_stream) {
  if (!state || !x || !dt || !A || !B || !C || !out) return MXVL_ERR_NULL;
  if (batch <= 0 || dim <= 0 || dstate <= 0) return MXVL_ERR_SHAPE;
  const int blocks = (int)(((int64_t)batch * dim + 255) / 256);
  hipStream_t s = (hipStream_t)hip_stream;
  switch (io_dtype) {
    case MXVL_F32:
      hipLaunchKernelGGL(state_update_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)state, (const float*)x,
                         (const float*)dt, (const float*)A, (const float*)B, (const float*)C, (const float*)D,
                         (const float*)z, (const float*)dt_bias, (float*)out, batch, dim, dstate, dt_softplus);
      break;
    case MXVL_BF16:
      hipLaunchKernelGGL(state_update_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (float*)state, (const bf16_t*)x,
                         (const bf16_t*)dt, (const float*)A, (const bf16_t*)B, (const bf16_t*)C, (const float*)D,
                         (const bf16_t*)z, (const float*)dt_bias, (bf16_t*)out, batch, dim, dstate, dt_softplus);
      break;
    case MXVL_F16:
      hipLaunchKernelGGL(state_update_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, (float*)state, (const f16_t*)x,
                         (const f16_t*)dt, (const float*)A, (const f16_t*)B, (const f16_t*)C, (const float*)D,
                         (const f16_t*)z, (const float*)dt_bias, (f16_t*)out, batch, dim, dstate, dt_softplus);
      break;
    default: return MXVL_ERR_DTYPE;
  }
  return check_launch();
}

}  // extern "C"
