// fused_norm_act.hip -- the HBM-bound glue between the GEMMs of an ARM / VisionMamba block, for gfx950.
//
//   add + LayerNorm   h = x + branch;  n = LN(h)            (models_mamba.py:110-116: x + mixer(norm1(x)), x + mlp(norm2(x));
//                                                            the reference's fused_add_norm Triton path does the same pairing)
//   SwiGLU gate       y = silu(a) * b,  [a | b] = one GEMM   (models_mamba.py:59-83: act(w1 x) * w2 x)
//
// In eager PyTorch under bf16 autocast each residual+norm is 3 kernels and 5 tensor passes (fp32 add, fp32 LayerNorm,
// bf16 cast for the next GEMM) and its backward 4 more; here it is one kernel forward (read x, branch; write h, n) and
// one backward (read dn, dh, h; write dx, dbranch; per-workgroup partial dgamma/dbeta).  A row is owned by one wave:
// 16-byte loads, values stay in registers, statistics in fp32 (two-pass variance), one DPP/shuffle reduction each.
#include <algorithm>

#include "mxvl_common.h"

namespace mxvl {

struct NormArgs {
  int rows, C;
  float eps;
  const void *x, *br;          // residual stream in, branch output (may be null)
  const float *gamma, *beta;   // (C) fp32; beta may be null
  void *h, *n;                 // x + br (written only when br != null), LN(h)
  float *mean, *rstd;          // (rows)
};
struct NormBwdArgs {
  int rows, C;
  const void *dn, *dh, *h;     // grad of n, grad of h (may be null), saved h
  const float *gamma, *mean, *rstd;
  void *dx, *dbr;              // grad wrt x (res dtype), optional copy in the branch dtype
  float *pgamma, *pbeta;       // (gridDim.x, C) partial sums
  float* pdbr;                 // optional (gridDim.x, C): column sums of the branch gradient as written (= the bias gradient of the
                               // linear that produced the branch: its backward no longer re-reads rows x C to sum it)
};

// sum over the LPR lanes that own one row (LPR = 64: the wave; 32 / 16: two / four rows per wave)
template <int LPR = 64> __device__ inline float wsum(float v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// sum over the 64 / LPR rows of a wave that share a column (lanes l, l + LPR, ...)
template <int LPR> __device__ inline float rsum(float v) {
#pragma unroll
  for (int off = 32; off >= LPR; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T> __device__ inline void ld4v(const T* p, float (&v)[4]) {
  const float4 a = ld4<T>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <typename T> __device__ inline void st4v(T* p, const float (&v)[4]) { st4<T>(p, make_float4(v[0], v[1], v[2], v[3])); }

// C = 4 K LPR: a row is owned by LPR lanes (64: one row per wave; 32 / 16 -- rows of 128 / 64 columns and their odd multiples: VMamba's
// first stage and patch embedding, the constructors' default embed_dim 192 -- two / four rows per wave), every lane owns K groups of 4
// consecutive columns: column (k * LPR + lane % LPR) * 4
template <typename res_t, typename br_t, typename out_t, int K, int LPR = 64>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const NormArgs p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane % LPR, sub = lane / LPR;
  const int C = p.C;
  float g[K][4], bta[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c = (k * LPR + sl) * 4;
    ld4v<float>(p.gamma + c, g[k]);
    if (p.beta) ld4v<float>(p.beta + c, bta[k]);
    else { bta[k][0] = bta[k][1] = bta[k][2] = bta[k][3] = 0.0f; }
  }
  for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < p.rows; row0 += gridDim.x * 4 * RPW) {
    const int row_i = row0 + sub;
    const bool live = row_i < p.rows;                  // the last wave of a narrow-row launch may own fewer than RPW rows
    const int row = live ? row_i : p.rows - 1;         // (its idle lanes re-read the last row and store nothing)
    float v[K][4];
    const res_t* xr = (const res_t*)p.x + (size_t)row * C;
#pragma unroll
    for (int k = 0; k < K; ++k) ld4v<res_t>(xr + (k * LPR + sl) * 4, v[k]);
    if (p.br) {
      const br_t* br = (const br_t*)p.br + (size_t)row * C;
      res_t* hr = (res_t*)p.h + (size_t)row * C;
      // ALL branch loads first, then the stores of the new stream: interleaved (load k, store k, load k + 1, ...) the compiler must
      // order each load behind the store in front of it (the pointers may alias) and waits for both -- K serialised memory round trips
      // per row, stores included (read off the ISA, round 6)
      float b[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k) ld4v<br_t>(br + (k * LPR + sl) * 4, b[k]);
#pragma unroll
      for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[k][j] += b[k][j];
          if constexpr (sizeof(res_t) == 2) {  // the stream itself is half precision: normalise what is stored
            res_t t; Io<res_t>::st(&t, v[k][j]); v[k][j] = Io<res_t>::ld(&t);
          }
        }
      }
      if (live) {
#pragma unroll
        for (int k = 0; k < K; ++k) st4v<res_t>(hr + (k * LPR + sl) * 4, v[k]);
      }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    const float mean = wsum<LPR>(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[k][j] - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wsum<LPR>(q) / (float)C + p.eps);
    out_t* nr = (out_t*)p.n + (size_t)row * C;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaf((v[k][j] - mean) * rstd, g[k][j], bta[k][j]);
      if (live) st4v<out_t>(nr + (k * LPR + sl) * 4, o);
    }
    if (sl == 0 && live) { p.mean[row] = mean; p.rstd[row] = rstd; }
  }
}

template <typename T> __device__ inline float ln_round_trip(float v) {   // value after a round trip through dtype T
  if constexpr (sizeof(T) == 4) return v;
  else if constexpr (__is_same(T, bf16_t)) return __builtin_bit_cast(float, cvt_pk_bf16(v, 0.0f) << 16);
  else return (float)(_Float16)v;
}

template <typename res_t, typename br_t, typename out_t, int K, int LPR = 64>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const NormBwdArgs p) {
  constexpr int RPW = 64 / LPR;
  __shared__ float sred[2][4][K * LPR * 4];   // [gamma|beta][wave][column]; re-used for the branch-gradient sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane % LPR, sub = lane / LPR;
  const int C = p.C;
  float g[K][4], ag[K][4], ab[K][4], ad[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    ld4v<float>(p.gamma + (k * LPR + sl) * 4, g[k]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ag[k][j] = 0.0f; ab[k][j] = 0.0f; ad[k][j] = 0.0f; }
  }
  for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < p.rows; row0 += gridDim.x * 4 * RPW) {
    const int row_i = row0 + sub;
    const bool live = row_i < p.rows;
    const int row = live ? row_i : p.rows - 1;
    const float mean = p.mean[row], rstd = p.rstd[row];
    float xh[K][4], dy[K][4];
    const res_t* hr = (const res_t*)p.h + (size_t)row * C;
    const out_t* dr = (const out_t*)p.dn + (size_t)row * C;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ld4v<res_t>(hr + (k * LPR + sl) * 4, xh[k]);
      ld4v<out_t>(dr + (k * LPR + sl) * 4, dy[k]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!live) dy[k][j] = 0.0f;                    // an idle sub-row adds nothing to the column sums
        xh[k][j] = (xh[k][j] - mean) * rstd;
        ag[k][j] = fmaf(dy[k][j], xh[k][j], ag[k][j]);
        ab[k][j] += dy[k][j];
        dy[k][j] *= g[k][j];
        s1 += dy[k][j];
        s2 = fmaf(dy[k][j], xh[k][j], s2);
      }
    }
    // the incoming residual gradient is requested BEFORE the row reduction (it depends on nothing computed here): one memory round
    // trip per row instead of two
    float rin[K][4];
    if (p.dh) {
#pragma unroll
      for (int k = 0; k < K; ++k) ld4v<res_t>((const res_t*)p.dh + (size_t)row * C + (k * LPR + sl) * 4, rin[k]);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) rin[k][j] = 0.0f;
    }
    const float m1 = wsum<LPR>(s1) / (float)C, m2 = wsum<LPR>(s2) / (float)C;
    res_t* dxr = (res_t*)p.dx + (size_t)row * C;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rstd * (dy[k][j] - m1 - xh[k][j] * m2) + rin[k][j];
      if (live) {
        st4v<res_t>(dxr + (k * LPR + sl) * 4, o);
        if (p.dbr) st4v<br_t>((br_t*)p.dbr + (size_t)row * C + (k * LPR + sl) * 4, o);
      }
      if (p.pdbr && live) {      // the sum of the values the consumer will read (the branch dtype's rounding of dx)
#pragma unroll
        for (int j = 0; j < 4; ++j) ad[k][j] += p.dbr ? ln_round_trip<br_t>(o[j]) : ln_round_trip<res_t>(o[j]);
      }
    }
  }
  if constexpr (RPW > 1) {       // the rows of a wave that share a column: one sum per column before the cross-wave tree
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) { ag[k][j] = rsum<LPR>(ag[k][j]); ab[k][j] = rsum<LPR>(ab[k][j]); ad[k][j] = rsum<LPR>(ad[k][j]); }
  }
  if (sub == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      *(float4*)&sred[0][wave][(k * LPR + sl) * 4] = make_float4(ag[k][0], ag[k][1], ag[k][2], ag[k][3]);
      *(float4*)&sred[1][wave][(k * LPR + sl) * 4] = make_float4(ab[k][0], ab[k][1], ab[k][2], ab[k][3]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    p.pgamma[(size_t)blockIdx.x * C + c] = (sred[0][0][c] + sred[0][1][c]) + (sred[0][2][c] + sred[0][3][c]);
    p.pbeta[(size_t)blockIdx.x * C + c] = (sred[1][0][c] + sred[1][1][c]) + (sred[1][2][c] + sred[1][3][c]);
  }
  if (p.pdbr) {
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) *(float4*)&sred[0][wave][(k * LPR + sl) * 4] = make_float4(ad[k][0], ad[k][1], ad[k][2], ad[k][3]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      p.pdbr[(size_t)blockIdx.x * C + c] = (sred[0][0][c] + sred[0][1][c]) + (sred[0][2][c] + sred[0][3][c]);
  }
}

// ---- any other row width: a 64-thread workgroup (ONE wave) per row, element accesses, the row walked three times (it stays in the
// vector cache: a row of this path is at most a few kilobytes).  Same arithmetic and rounding points as the kernels above -- fp32 two-pass
// statistics, the residual stream normalised as stored -- so a model whose width is not one of the built instantiations runs the same
// function, only slower: the package has no library LayerNorm to fall back to (include/mxvl.h).
template <typename res_t, typename br_t, typename out_t>
__global__ __launch_bounds__(64) void add_ln_fwd_any_kernel(const NormArgs p) {
  const int lane = threadIdx.x, C = p.C;
  for (int row = blockIdx.x; row < p.rows; row += gridDim.x) {
    const res_t* xr = (const res_t*)p.x + (size_t)row * C;
    const res_t* src = xr;
    float s = 0.0f;
    if (p.br) {
      const br_t* br = (const br_t*)p.br + (size_t)row * C;
      res_t* hr = (res_t*)p.h + (size_t)row * C;
      for (int c = lane; c < C; c += 64) {
        const float v = ln_round_trip<res_t>(Io<res_t>::ld(xr + c) + Io<br_t>::ld(br + c));
        Io<res_t>::st(hr + c, v);                       // (read back below by the lane that wrote it)
        s += v;
      }
      src = hr;
    } else {
      for (int c = lane; c < C; c += 64) s += Io<res_t>::ld(xr + c);
    }
    const float mean = wsum<64>(s) / (float)C;
    float q = 0.0f;
    for (int c = lane; c < C; c += 64) { const float d = Io<res_t>::ld(src + c) - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wsum<64>(q) / (float)C + p.eps);
    out_t* nr = (out_t*)p.n + (size_t)row * C;
    for (int c = lane; c < C; c += 64)
      Io<out_t>::st(nr + c, fmaf((Io<res_t>::ld(src + c) - mean) * rstd, p.gamma[c], p.beta ? p.beta[c] : 0.0f));
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
  }
}

// backward: workgroup w = partial row w of dgamma | dbeta | d-branch sums, which it owns alone (column c always meets the same lane): the
// first row it walks stores, the following ones add -- plain read-modify-writes, no zero fill, no atomics
template <typename res_t, typename br_t, typename out_t>
__global__ __launch_bounds__(64) void add_ln_bwd_any_kernel(const NormBwdArgs p) {
  const int lane = threadIdx.x, C = p.C;
  float* pg = p.pgamma + (size_t)blockIdx.x * C;
  float* pb = p.pbeta + (size_t)blockIdx.x * C;
  float* pd = p.pdbr ? p.pdbr + (size_t)blockIdx.x * C : nullptr;
  bool first = true;
  for (int row = blockIdx.x; row < p.rows; row += gridDim.x, first = false) {
    const float mean = p.mean[row], rstd = p.rstd[row];
    const res_t* hr = (const res_t*)p.h + (size_t)row * C;
    const out_t* dr = (const out_t*)p.dn + (size_t)row * C;
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = lane; c < C; c += 64) {
      const float xh = (Io<res_t>::ld(hr + c) - mean) * rstd, dy = Io<out_t>::ld(dr + c);
      pg[c] = first ? dy * xh : fmaf(dy, xh, pg[c]);
      pb[c] = first ? dy : pb[c] + dy;
      const float dg = dy * p.gamma[c];
      s1 += dg;
      s2 = fmaf(dg, xh, s2);
    }
    const float m1 = wsum<64>(s1) / (float)C, m2 = wsum<64>(s2) / (float)C;
    res_t* dxr = (res_t*)p.dx + (size_t)row * C;
    for (int c = lane; c < C; c += 64) {
      const float xh = (Io<res_t>::ld(hr + c) - mean) * rstd, dg = Io<out_t>::ld(dr + c) * p.gamma[c];
      float o = rstd * (dg - m1 - xh * m2);
      if (p.dh) o += Io<res_t>::ld((const res_t*)p.dh + (size_t)row * C + c);
      Io<res_t>::st(dxr + c, o);
      if (p.dbr) Io<br_t>::st((br_t*)p.dbr + (size_t)row * C + c, o);
      if (pd) {
        const float r = p.dbr ? ln_round_trip<br_t>(o) : ln_round_trip<res_t>(o);
        pd[c] = first ? r : pd[c] + r;
      }
    }
  }
}

// ---- SwiGLU gate: ab (rows, 2H) -> y (rows, H) = silu(ab[:, :H]) * ab[:, H:] ------------------------------------------
// H need not be a multiple of 8 (2730 for ARM-large): rows are only 4-byte aligned in bf16.  A wave walks a row in
// 512-column tiles; a lane owns the column PAIRS lane*2 + k*128 (k = 0..3), so every load instruction of the wave is
// one contiguous 256-byte (bf16) / 512-byte (fp32) segment and four of them are in flight per operand.
template <typename io_t> struct Pair;
template <> struct Pair<float> {
  __device__ static inline void ld(const float* p, float& a, float& b) { const float2 v = *(const float2*)p; a = v.x; b = v.y; }
  __device__ static inline void st(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }
};
template <> struct Pair<bf16_t> {
  __device__ static inline void ld(const bf16_t* p, float& a, float& b) {
    const uint32_t v = *(const uint32_t*)p;
    a = __builtin_bit_cast(float, v << 16); b = __builtin_bit_cast(float, v & 0xffff0000u);
  }
  __device__ static inline void st(bf16_t* p, float a, float b) { *(uint32_t*)p = cvt_pk_bf16(a, b); }
};
template <> struct Pair<f16_t> {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  __device__ static inline void ld(const f16_t* p, float& a, float& b) { const h2 v = *(const h2*)p; a = (float)v.x; b = (float)v.y; }
  __device__ static inline void st(f16_t* p, float a, float b) { *(h2*)p = h2{(_Float16)a, (_Float16)b}; }
};

template <typename io_t>
__device__ inline float rnd_io(float v) {  // value after a round trip through the io dtype
  if constexpr (sizeof(io_t) == 4) return v;
  io_t t;
  Io<io_t>::st(&t, v);
  return Io<io_t>::ld(&t);
}

// H must be even (pairs); odd H goes through the scalar tail kernel below
template <typename io_t, bool BWD>
__global__ __launch_bounds__(256) void swiglu_kernel(const io_t* __restrict__ ab, const io_t* __restrict__ dy, io_t* __restrict__ out,
                                                      int rows, int H) {
  using P = Pair<io_t>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles = (H + 511) / 512;
  const long total = (long)rows * tiles;
  for (long t = (long)blockIdx.x * 4 + wave; t < total; t += (long)gridDim.x * 4) {
    const int r = (int)(t / tiles), c0 = (int)(t - (long)r * tiles) * 512 + lane * 2;
    const io_t* a = ab + (size_t)r * 2 * H;
    float av[4][2], bv[4][2], gv[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * 128;
      av[k][0] = av[k][1] = bv[k][0] = bv[k][1] = gv[k][0] = gv[k][1] = 0.0f;
      if (c < H) {
        P::ld(a + c, av[k][0], av[k][1]);
        P::ld(a + H + c, bv[k][0], bv[k][1]);
        if constexpr (BWD) P::ld(dy + (size_t)r * H + c, gv[k][0], gv[k][1]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * 128;
      if (c >= H) continue;
      if constexpr (!BWD) {
        // torch rounds act(w1 x) to the io dtype before the product
        P::st(out + (size_t)r * H + c, rnd_io<io_t>(silu(av[k][0])) * bv[k][0], rnd_io<io_t>(silu(av[k][1])) * bv[k][1]);
      } else {
        float da[2], db[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float sg = sigmoid(av[k][j]);
          da[j] = gv[k][j] * bv[k][j] * (sg * (1.0f + av[k][j] * (1.0f - sg)));   // d silu(a) = s (1 + a (1 - s))
          db[j] = gv[k][j] * (av[k][j] * sg);
        }
        io_t* d = out + (size_t)r * 2 * H;
        P::st(d + c, da[0], da[1]);
        P::st(d + H + c, db[0], db[1]);
      }
    }
  }
}

// Backward that also leaves the column sums of d[a|b] (the bias gradient of the w1|w2 GEMM that produced ab): a wave keeps
// ONE column tile (<= 512 wide) and walks the rows rg, rg + R, ...; the sums of what it stored (rounded to the io dtype, like a
// separate reduction over dab would see them) stay in 16 registers per lane and leave as row rg of `partial` (R, 2H) fp32.
// Saves the second full read of dab (713 MB per ARM-large layer at 16 x 4080 tokens) that `dab.sum(0)` costs.
template <typename io_t>
__global__ __launch_bounds__(256) void swiglu_bwd_colsum_kernel(const io_t* __restrict__ ab, const io_t* __restrict__ dy,
                                                                 io_t* __restrict__ out, float* __restrict__ partial, int rows,
                                                                 int H, int R) {
  using P = Pair<io_t>;
  const int lane = threadIdx.x & 63;
  const int tiles = (H + 511) / 512;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long)tiles * R) return;
  const int tile = (int)(w % tiles), rg = (int)(w / tiles);
  // equal-width column tiles (even, <= 512): with fixed 512-wide tiles the last tile of H = 2730 holds 170 columns and the
  // waves that own it would idle two thirds of the time
  const int tw = ((H + tiles - 1) / tiles + 1) & ~1;
  const int c0 = tile * tw + lane * 2;
  const int cend = min(H, (tile + 1) * tw);
  float sa[4][2], sb[4][2];
#pragma unroll
  for (int k = 0; k < 4; ++k) sa[k][0] = sa[k][1] = sb[k][0] = sb[k][1] = 0.0f;
  for (int r = rg; r < rows; r += R) {
    const io_t* a = ab + (size_t)r * 2 * H;
    float av[4][2], bv[4][2], gv[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * 128;
      av[k][0] = av[k][1] = bv[k][0] = bv[k][1] = gv[k][0] = gv[k][1] = 0.0f;
      if (c < cend) {
        P::ld(a + c, av[k][0], av[k][1]);
        P::ld(a + H + c, bv[k][0], bv[k][1]);
        P::ld(dy + (size_t)r * H + c, gv[k][0], gv[k][1]);
      }
    }
    io_t* d = out + (size_t)r * 2 * H;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * 128;
      if (c >= cend) continue;
      float da[2], db[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float sg = sigmoid(av[k][j]);
        da[j] = rnd_io<io_t>(gv[k][j] * bv[k][j] * (sg * (1.0f + av[k][j] * (1.0f - sg))));
        db[j] = rnd_io<io_t>(gv[k][j] * (av[k][j] * sg));
        sa[k][j] += da[j];
        sb[k][j] += db[j];
      }
      P::st(d + c, da[0], da[1]);
      P::st(d + H + c, db[0], db[1]);
    }
  }
  float* pr = partial + (size_t)rg * 2 * H;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + k * 128;
    if (c < cend) {
      pr[c] = sa[k][0]; pr[c + 1] = sa[k][1];
      pr[H + c] = sb[k][0]; pr[H + c + 1] = sb[k][1];
    }
  }
}

// The same backward for 16-bit rows that are 16-byte aligned (H % 8 == 0: every ARM width once the hidden axis is padded to whole
// 64-column tiles, models_mamba.SwiGLU): a lane owns 8 consecutive columns -- one 16-byte load per operand and row, two rows in
// flight, 16 column sums in registers.  Column tiles are equal-width multiples of 8 (<= 512), as above.
template <typename io_t>
__global__ __launch_bounds__(256) void swiglu_bwd_colsum_vec_kernel(const io_t* __restrict__ ab, const io_t* __restrict__ dy,
                                                                     io_t* __restrict__ out, float* __restrict__ partial, int rows,
                                                                     int H, int R) {
  const int lane = threadIdx.x & 63;
  const int tiles = (H + 511) / 512;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long)tiles * R) return;
  const int tile = (int)(w % tiles), rg = (int)(w / tiles);
  const int tw = (((H + tiles - 1) / tiles) + 7) & ~7;
  const int c = tile * tw + lane * 8;
  const bool on = lane * 8 < tw && c < H;          // H % 8 == 0: a lane's 8 columns are all inside or all outside
  float sa[8], sb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sa[k] = sb[k] = 0.0f;
  auto unpack = [](const uint4 v, float (&f)[8]) {
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (sizeof(io_t) == 2 && __is_same(io_t, bf16_t)) {
        f[2 * k] = __builtin_bit_cast(float, wd[k] << 16);
        f[2 * k + 1] = __builtin_bit_cast(float, wd[k] & 0xffff0000u);
      } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 hv = __builtin_bit_cast(h2, wd[k]);
        f[2 * k] = (float)hv.x;
        f[2 * k + 1] = (float)hv.y;
      }
    }
  };
  auto pack = [](const float (&f)[8]) {
    uint32_t wd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (__is_same(io_t, bf16_t)) wd[k] = cvt_pk_bf16(f[2 * k], f[2 * k + 1]);
      else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        wd[k] = __builtin_bit_cast(uint32_t, h2{(_Float16)f[2 * k], (_Float16)f[2 * k + 1]});
      }
    }
    return make_uint4(wd[0], wd[1], wd[2], wd[3]);
  };
  if (on) {
    for (int r0 = rg; r0 < rows; r0 += 2 * R) {
      uint4 av[2], bv[2], gv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = r0 + u * R;
        av[u] = bv[u] = gv[u] = make_uint4(0, 0, 0, 0);
        if (r < rows) {
          const io_t* a = ab + (size_t)r * 2 * H + c;
          av[u] = *(const uint4*)a;
          bv[u] = *(const uint4*)(a + H);
          gv[u] = *(const uint4*)(dy + (size_t)r * H + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = r0 + u * R;
        if (r >= rows) continue;
        float a[8], b[8], g[8], da[8], db[8];
        unpack(av[u], a); unpack(bv[u], b); unpack(gv[u], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float sg = sigmoid(a[k]);
          da[k] = rnd_io<io_t>(g[k] * b[k] * (sg * (1.0f + a[k] * (1.0f - sg))));
          db[k] = rnd_io<io_t>(g[k] * (a[k] * sg));
          sa[k] += da[k];
          sb[k] += db[k];
        }
        io_t* d = out + (size_t)r * 2 * H + c;
        *(uint4*)d = pack(da);
        *(uint4*)(d + H) = pack(db);
      }
    }
    float* pr = partial + (size_t)rg * 2 * H + c;
    *(float4*)pr = make_float4(sa[0], sa[1], sa[2], sa[3]);
    *(float4*)(pr + 4) = make_float4(sa[4], sa[5], sa[6], sa[7]);
    *(float4*)(pr + H) = make_float4(sb[0], sb[1], sb[2], sb[3]);
    *(float4*)(pr + H + 4) = make_float4(sb[4], sb[5], sb[6], sb[7]);
  }
}

template <typename io_t, bool BWD>
__global__ __launch_bounds__(256) void swiglu_scalar_kernel(const io_t* __restrict__ ab, const io_t* __restrict__ dy,
                                                             io_t* __restrict__ out, int rows, int H) {
  using io = Io<io_t>;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)rows * H; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / H), c = (int)(i - (size_t)r * H);
    const float av = io::ld(ab + (size_t)r * 2 * H + c), bv = io::ld(ab + (size_t)r * 2 * H + H + c);
    if constexpr (!BWD) {
      io::st(out + i, rnd_io<io_t>(silu(av)) * bv);
    } else {
      const float g = io::ld(dy + i), sg = sigmoid(av);
      io::st(out + (size_t)r * 2 * H + c, g * bv * (sg * (1.0f + av * (1.0f - sg))));
      io::st(out + (size_t)r * 2 * H + H + c, g * (av * sg));
    }
  }
}

template <typename io_t, bool BWD>
static int launch_swiglu(const void* ab, const void* dy, void* out, int rows, int H, hipStream_t s) {
  if (H % 2 == 0) {
    const long work = (long)rows * ((H + 511) / 512);
    const int grid = (int)std::min<long>((work + 3) / 4, 256L * 32);
    hipLaunchKernelGGL((swiglu_kernel<io_t, BWD>), dim3(grid), dim3(256), 0, s, (const io_t*)ab, (const io_t*)dy, (io_t*)out, rows, H);
  } else {
    const size_t work = (size_t)rows * H;
    const int grid = (int)std::min<size_t>((work + 255) / 256, 256 * 64);
    hipLaunchKernelGGL((swiglu_scalar_kernel<io_t, BWD>), dim3(grid), dim3(256), 0, s, (const io_t*)ab, (const io_t*)dy, (io_t*)out, rows, H);
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
template <bool BWD>
static int dispatch_swiglu(const void* ab, const void* dy, void* out, int rows, int H, int dt, hipStream_t s) {
  switch (dt) {
    case MXVL_F32: return launch_swiglu<float, BWD>(ab, dy, out, rows, H, s);
    case MXVL_BF16: return launch_swiglu<bf16_t, BWD>(ab, dy, out, rows, H, s);
    case MXVL_F16: return launch_swiglu<f16_t, BWD>(ab, dy, out, rows, H, s);
    default: return MXVL_ERR_DTYPE;
  }
}

// backward workgroups = partial rows of dgamma | dbeta | d-branch sums the caller reduces: 768 = ONE round of 3 workgroups per CU --
// the 1024-wide instantiation holds 152-162 VGPRs, i.e. three waves per SIMD, so of the 1024 workgroups of rounds 4-6 a quarter ran as
// a second, mostly idle round.  (With 2048 the three (2048, C) partial tensors cost a 41 us reduce_kernel per add+LN backward.)
constexpr int kLnBwdPartials = 768;

template <typename R, typename B, typename O, int K, int LPR = 64>
static void launch_ln(bool bwd, const void* args, int rows, hipStream_t s) {
  // (the backward's grid IS the caller's n_partials = mxvl_add_layernorm_partials(rows), whatever the row width: workgroups that
  // get no row of a narrow-row launch write zero partials)
  const int wgs = std::min((rows + 3) / 4, bwd ? kLnBwdPartials : 2048);
  if (bwd) hipLaunchKernelGGL((add_ln_bwd_kernel<R, B, O, K, LPR>), dim3(wgs), dim3(256), 0, s, *(const NormBwdArgs*)args);
  else hipLaunchKernelGGL((add_ln_fwd_kernel<R, B, O, K, LPR>), dim3(wgs), dim3(256), 0, s, *(const NormArgs*)args);
}
// any other width: one wave per row (the backward's grid is still the caller's n_partials: a workgroup owns a partial row)
template <typename R, typename B, typename O>
static void launch_ln_any(bool bwd, const void* args, int rows, hipStream_t s) {
  if (bwd) hipLaunchKernelGGL((add_ln_bwd_any_kernel<R, B, O>), dim3(std::min((rows + 3) / 4, kLnBwdPartials)), dim3(64), 0, s, *(const NormBwdArgs*)args);
  else hipLaunchKernelGGL((add_ln_fwd_any_kernel<R, B, O>), dim3(std::min(rows, 8192)), dim3(64), 0, s, *(const NormArgs*)args);
}
template <typename R, typename B, typename O>
static int dispatch_k(bool bwd, const void* args, int rows, int C, hipStream_t s) {
  if (C % 256 != 0) {       // narrow rows: 128 k (two rows per wave) or 64 k (four rows per wave), k in {1, 3}
    switch (C) {
      case 128: launch_ln<R, B, O, 1, 32>(bwd, args, rows, s); break;
      case 384: launch_ln<R, B, O, 3, 32>(bwd, args, rows, s); break;
      case 64: launch_ln<R, B, O, 1, 16>(bwd, args, rows, s); break;
      case 192: launch_ln<R, B, O, 3, 16>(bwd, args, rows, s); break;
      default: launch_ln_any<R, B, O>(bwd, args, rows, s); break;
    }
    return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
  }
  switch (C / 256) {
    case 1: launch_ln<R, B, O, 1>(bwd, args, rows, s); break;
    case 2: launch_ln<R, B, O, 2>(bwd, args, rows, s); break;
    case 3: launch_ln<R, B, O, 3>(bwd, args, rows, s); break;
    case 4: launch_ln<R, B, O, 4>(bwd, args, rows, s); break;
    case 6: launch_ln<R, B, O, 6>(bwd, args, rows, s); break;
    case 8: launch_ln<R, B, O, 8>(bwd, args, rows, s); break;
    default: launch_ln_any<R, B, O>(bwd, args, rows, s); break;
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
static int dispatch_ln(bool bwd, const void* args, int rows, int C, int res_dt, int br_dt, int out_dt, hipStream_t s) {
  if (res_dt == MXVL_F32 && br_dt == MXVL_F32 && out_dt == MXVL_F32) return dispatch_k<float, float, float>(bwd, args, rows, C, s);
  if (res_dt == MXVL_F32 && br_dt == MXVL_BF16 && out_dt == MXVL_BF16) return dispatch_k<float, bf16_t, bf16_t>(bwd, args, rows, C, s);
  if (res_dt == MXVL_F32 && br_dt == MXVL_F32 && out_dt == MXVL_BF16) return dispatch_k<float, float, bf16_t>(bwd, args, rows, C, s);
  if (res_dt == MXVL_BF16 && br_dt == MXVL_BF16 && out_dt == MXVL_BF16) return dispatch_k<bf16_t, bf16_t, bf16_t>(bwd, args, rows, C, s);
  // fp16 autocast (the ViT-MAE recipe: HD_Xray_Pretrain_MAE/pretrain/main.py:211-213,317): fp32 residual stream, fp16 branch / n
  if (res_dt == MXVL_F32 && br_dt == MXVL_F16 && out_dt == MXVL_F16) return dispatch_k<float, f16_t, f16_t>(bwd, args, rows, C, s);
  if (res_dt == MXVL_F32 && br_dt == MXVL_F32 && out_dt == MXVL_F16) return dispatch_k<float, float, f16_t>(bwd, args, rows, C, s);
  return MXVL_ERR_DTYPE;
}

}  // namespace mxvl

using namespace mxvl;

extern "C" {

int mxvl_add_layernorm_fwd(const mxvl_add_ln_desc* d, void* hip_stream) {
  if (!d || !d->x || !d->gamma || !d->n || !d->mean || !d->rstd) return MXVL_ERR_NULL;
  if (d->branch && !d->h) return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->cols <= 0) return MXVL_ERR_SHAPE;
  NormArgs a;
  a.rows = d->rows; a.C = d->cols; a.eps = d->eps; a.x = d->x; a.br = d->branch; a.gamma = (const float*)d->gamma;
  a.beta = (const float*)d->beta; a.h = d->h; a.n = d->n; a.mean = (float*)d->mean; a.rstd = (float*)d->rstd;
  return dispatch_ln(false, &a, d->rows, d->cols, d->res_dtype, d->branch_dtype, d->out_dtype, (hipStream_t)hip_stream);
}

int mxvl_add_layernorm_bwd(const mxvl_add_ln_bwd_desc* d, void* hip_stream) {
  if (!d || !d->dn || !d->h || !d->gamma || !d->mean || !d->rstd || !d->dx || !d->partial_dgamma || !d->partial_dbeta) return MXVL_ERR_NULL;
  if (d->rows <= 0 || d->cols <= 0) return MXVL_ERR_SHAPE;
  if (d->n_partials != std::min((d->rows + 3) / 4, kLnBwdPartials)) return MXVL_ERR_SHAPE;
  NormBwdArgs a;
  a.rows = d->rows; a.C = d->cols; a.dn = d->dn; a.dh = d->dh; a.h = d->h; a.gamma = (const float*)d->gamma;
  a.mean = (const float*)d->mean; a.rstd = (const float*)d->rstd; a.dx = d->dx; a.dbr = d->dbranch;
  a.pgamma = (float*)d->partial_dgamma; a.pbeta = (float*)d->partial_dbeta; a.pdbr = (float*)d->partial_dbranch;
  return dispatch_ln(true, &a, d->rows, d->cols, d->res_dtype, d->branch_dtype, d->out_dtype, (hipStream_t)hip_stream);
}

int mxvl_add_layernorm_partials(int rows) { return std::min((rows + 3) / 4, kLnBwdPartials); }

int mxvl_swiglu_fwd(const void* ab, void* y, int rows, int hidden, int io_dtype, void* hip_stream) {
  if (!ab || !y) return MXVL_ERR_NULL;
  if (rows <= 0 || hidden <= 0) return MXVL_ERR_SHAPE;
  return dispatch_swiglu<false>(ab, nullptr, y, rows, hidden, io_dtype, (hipStream_t)hip_stream);
}

int mxvl_swiglu_bwd(const void* ab, const void* dy, void* dab, int rows, int hidden, int io_dtype, void* hip_stream) {
  if (!ab || !dy || !dab) return MXVL_ERR_NULL;
  if (rows <= 0 || hidden <= 0) return MXVL_ERR_SHAPE;
  return dispatch_swiglu<true>(ab, dy, dab, rows, hidden, io_dtype, (hipStream_t)hip_stream);
}

// rows of the partial-sum buffer mxvl_swiglu_bwd_colsum fills (0: hidden is odd, use mxvl_swiglu_bwd and reduce dab yourself)
int mxvl_swiglu_partials(int rows, int hidden) {
  if (rows <= 0 || hidden <= 0 || hidden % 2) return 0;
  const int tiles = (hidden + 511) / 512;
  return std::max(1, std::min(rows, 8192 / tiles));   // ~8192 waves: every SIMD slot of the chip at 8 waves / SIMD
}

int mxvl_swiglu_bwd_colsum(const void* ab, const void* dy, void* dab, void* partial, int n_partials, int rows, int hidden,
                           int io_dtype, void* hip_stream) {
  if (!ab || !dy || !dab || !partial) return MXVL_ERR_NULL;
  if (rows <= 0 || hidden <= 0) return MXVL_ERR_SHAPE;
  if (hidden % 2) return MXVL_ERR_UNSUPPORTED;
  if (n_partials != mxvl_swiglu_partials(rows, hidden)) return MXVL_ERR_SHAPE;
  const int tiles = (hidden + 511) / 512;
  const int grid = (int)(((long)tiles * n_partials + 3) / 4);
  hipStream_t s = (hipStream_t)hip_stream;
  const bool vec = hidden % 8 == 0 && io_dtype != MXVL_F32 && ((uintptr_t)ab % 16) == 0 && ((uintptr_t)dy % 16) == 0 &&
                   ((uintptr_t)dab % 16) == 0 && ((uintptr_t)partial % 16) == 0;
  if (vec) {
    if (io_dtype == MXVL_BF16) hipLaunchKernelGGL(swiglu_bwd_colsum_vec_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)ab, (const bf16_t*)dy, (bf16_t*)dab, (float*)partial, rows, hidden, n_partials);
    else hipLaunchKernelGGL(swiglu_bwd_colsum_vec_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)ab, (const f16_t*)dy, (f16_t*)dab, (float*)partial, rows, hidden, n_partials);
    return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
  }
  switch (io_dtype) {
    case MXVL_F32: hipLaunchKernelGGL(swiglu_bwd_colsum_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)ab, (const float*)dy, (float*)dab, (float*)partial, rows, hidden, n_partials); break;
    case MXVL_BF16: hipLaunchKernelGGL(swiglu_bwd_colsum_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)ab, (const bf16_t*)dy, (bf16_t*)dab, (float*)partial, rows, hidden, n_partials); break;
    case MXVL_F16: hipLaunchKernelGGL(swiglu_bwd_colsum_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const f16_t*)ab, (const f16_t*)dy, (f16_t*)dab, (float*)partial, rows, hidden, n_partials); break;
    default: return MXVL_ERR_DTYPE;
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

}  // extern "C"
