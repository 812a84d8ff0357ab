// scan_n1_short.h -- dstate-1 selective scan for rows of at most 128 steps whose length is no multiple of 4 (VMamba's last stage:
// 7 x 7 = 49 tokens, scan dim 8192; R2GenCSR/VMamba/classification/models/vmamba.py:294-312 at vssm_base_224.yaml's stage 3).
//
// scan_n1.h / scan_n1_bwd.h need 4-element row alignment for their vector accesses; a 49-step row (98 bytes) has none, and the
// general kernels spent a 128-step chunk prologue, an LDS B / C tile and a 16-lane DPP scan per row on it (126 / 332 us forward /
// backward for 77 / 128 MB at batch 32).  Rows this short need no cross-lane scan at all: a LANE owns a row and walks its steps one
// after the other, a wave owns 64 consecutive channels of one (batch element, B / C group).  The wave's rows are ONE contiguous
// range of 64 L elements in u / delta / out (the launcher checks the strides): it comes in and goes out as flat 16-byte vectors
// through the LDS, whatever L is; B_t / C_t are the same for the whole wave (an LDS broadcast); dB_t / dC_t are a DPP sum over the
// wave and one atomic per (wave, step).  One chunk per row: the only checkpoint is the zero state entering step 0.
// Included twice: by scan_fwd.hip (forward, ScanArgs) and by scan_bwd.hip with MXVL_N1_SHORT_BWD defined (backward, ScanBwdArgs).
#pragma once
#include "mxvl_common.h"

namespace mxvl {

// softplus and its derivative from one exponential (mxvl_common.h softplus)
__device__ __forceinline__ void n1s_softplus(float xr, float& sp, float& dsp) {
  const float w = fast_exp(xr), s = 1.0f + w, den = s - 1.0f;
  const float l = fast_log2(s) * 0.6931471805599453f;
  const float r = fmaf(w - den, fmaxf(1.0f - den, 0.0f), l);
  sp = xr > 20.0f ? xr : r;
  dsp = xr > 20.0f ? 1.0f : w * fast_rcp(s);
}
__device__ __forceinline__ void n1s_wave_sync() {       // a wave's own LDS writes before its own LDS reads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// n bytes (a multiple of 16), global <-> this wave's LDS region, as flat 16-byte vectors
__device__ __forceinline__ void n1s_copy_in(void* lds, const void* g, int nbytes, int lane) {
  for (int v = lane; v < nbytes / 16; v += 64) ((uint4*)lds)[v] = ((const uint4*)g)[v];
}
__device__ __forceinline__ void n1s_copy_out(void* g, const void* lds, int nbytes, int lane) {
  for (int v = lane; v < nbytes / 16; v += 64) ((uint4*)g)[v] = ((const uint4*)lds)[v];
}

#ifndef MXVL_N1_SHORT_BWD

// LDS per wave: u, delta (io_t [64][L]), out (fp32 or io_t [64][L]), B, C (float [L] each, padded to 16 bytes)
template <typename io_t, int NW, bool OF32>
__global__ __launch_bounds__(NW * 64) void scan_n1_short_fwd_kernel(const ScanArgs p) {
  using io = Io<io_t>;
  typedef typename std::conditional<OF32, float, io_t>::type out_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char n1s_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, L = p.L;
  const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * 64;           // first row (b * dim + d) of this wave
  if (row0 >= (int64_t)p.batch * p.dim) return;                          // (no workgroup barrier below: every LDS region is a wave's own)
  const int LP = (L + 3) & ~3;
  const size_t per_wave = (size_t)64 * L * (2 * sizeof(io_t) + sizeof(out_t)) + (size_t)2 * LP * sizeof(float);
  unsigned char* base = n1s_smem + wave * per_wave;
  io_t* su = (io_t*)base;
  io_t* sd = su + 64 * L;
  out_t* so = (out_t*)(sd + 64 * L);
  float* sB = (float*)(so + 64 * L);
  float* sC = sB + LP;
  const int b = (int)(row0 / p.dim), d0 = (int)(row0 - (int64_t)b * p.dim), d = d0 + lane;
  const int g = d0 / (p.dim / p.G);
  n1s_copy_in(su, (const io_t*)p.u + row0 * L, 64 * L * (int)sizeof(io_t), lane);
  n1s_copy_in(sd, (const io_t*)p.delta + row0 * L, 64 * L * (int)sizeof(io_t), lane);
  const io_t* Bp = (const io_t*)p.B + (int64_t)b * p.B_bs + (int64_t)g * p.B_gs;
  const io_t* Cp = (const io_t*)p.C + (int64_t)b * p.C_bs + (int64_t)g * p.C_gs;
  for (int t = lane; t < L; t += 64) {
    sB[t] = io::ld(Bp + t);
    sC[t] = io::ld(Cp + t);
  }
  const float A2 = p.A[(int64_t)d * p.A_ds] * kLog2e;
  const float Dv = p.D ? p.D[d] : 0.0f;
  const float bias = p.bias ? p.bias[d] : 0.0f;
  n1s_wave_sync();
  const io_t* ru = su + lane * L;
  const io_t* rd = sd + lane * L;
  out_t* ro = so + lane * L;
  float h = 0.0f;
  for (int t = 0; t < L; ++t) {
    const float uu = io::ld(ru + t);
    float dl = io::ld(rd + t) + bias;
    if (p.softplus) dl = softplus(dl);
    h = fmaf(fast_exp2(dl * A2), h, dl * uu * sB[t]);
    const float y = fmaf(sC[t], h, Dv * uu);
    if constexpr (OF32) ro[t] = y; else io::st(ro + t, y);
  }
  if (p.last_state != nullptr) p.last_state[row0 + lane] = h;
  if (p.ckpt != nullptr) p.ckpt[(row0 + lane) * p.n_ckpt] = 0.0f;        // the state entering the row's only chunk
  n1s_wave_sync();
  n1s_copy_out((out_t*)p.out + row0 * L, so, 64 * L * (int)sizeof(out_t), lane);
}

#else  // ---- backward ------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ float n1s_sum_to_lane63(float v) {
  v += dpp<DPP_ROW_SHR(1)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(2)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(4)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(8)>(0.0f, v);
  v += dpp<DPP_ROW_BCAST15, 0xa>(0.0f, v);
  v += dpp<DPP_ROW_BCAST31, 0xc>(0.0f, v);
  return v;
}

// LDS per wave: u -> du, delta -> ddelta (io_t [64][L], in place), dout (fp32 or io_t [64][L]), h (float [L][64], step-major: a lane's
// column is conflict-free), B, C (float [L] padded)
template <typename io_t, int NW, bool OF32>
__global__ __launch_bounds__(NW * 64) void scan_n1_short_bwd_kernel(const ScanBwdArgs p) {
  using io = Io<io_t>;
  typedef typename std::conditional<OF32, float, io_t>::type g_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char n1s_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, L = p.L;
  const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * 64;
  if (row0 >= (int64_t)p.batch * p.dim) return;
  const int LP = (L + 3) & ~3;
  const size_t per_wave = (size_t)64 * L * (2 * sizeof(io_t) + sizeof(g_t) + sizeof(float)) + (size_t)2 * LP * sizeof(float);
  unsigned char* base = n1s_smem + wave * per_wave;
  io_t* su = (io_t*)base;
  io_t* sd = su + 64 * L;
  g_t* sg = (g_t*)(sd + 64 * L);
  float* sh = (float*)(sg + 64 * L);
  float* sB = sh + 64 * L;
  float* sC = sB + LP;
  const int b = (int)(row0 / p.dim), d0 = (int)(row0 - (int64_t)b * p.dim), d = d0 + lane;
  const int g = d0 / (p.dim / p.G);
  n1s_copy_in(su, (const io_t*)p.u + row0 * L, 64 * L * (int)sizeof(io_t), lane);
  n1s_copy_in(sd, (const io_t*)p.delta + row0 * L, 64 * L * (int)sizeof(io_t), lane);
  n1s_copy_in(sg, (const g_t*)p.dout + row0 * L, 64 * L * (int)sizeof(g_t), lane);
  const io_t* Bp = (const io_t*)p.B + (int64_t)b * p.B_bs + (int64_t)g * p.B_gs;
  const io_t* Cp = (const io_t*)p.C + (int64_t)b * p.C_bs + (int64_t)g * p.C_gs;
  for (int t = lane; t < L; t += 64) {
    sB[t] = io::ld(Bp + t);
    sC[t] = io::ld(Cp + t);
  }
  const float A2 = p.A[(int64_t)d * p.A_ds] * kLog2e;
  const float Aln = A2 * 0.6931471805599453f;            // A = A2 ln 2
  const float Dv = p.D ? p.D[d] : 0.0f;
  const float bias = p.bias ? p.bias[d] : 0.0f;
  n1s_wave_sync();
  io_t* ru = su + lane * L;
  io_t* rd = sd + lane * L;
  const g_t* rg = sg + lane * L;
  {   // forward states
    float h = 0.0f;
    for (int t = 0; t < L; ++t) {
      const float uu = io::ld(ru + t);
      float dl = io::ld(rd + t) + bias;
      if (p.softplus) dl = softplus(dl);
      h = fmaf(fast_exp2(dl * A2), h, dl * uu * sB[t]);
      sh[t * 64 + lane] = h;
    }
  }
  float* dBp = p.dB + (int64_t)b * p.dB_bs + (int64_t)g * p.dB_gs;
  float* dCp = p.dC + (int64_t)b * p.dC_bs + (int64_t)g * p.dC_gs;
  float gg = 0.0f, dA_p = 0.0f, dD_p = 0.0f, db_p = 0.0f;     // gg = a_{t+1} g_{t+1}
  for (int t = L - 1; t >= 0; --t) {
    const float uu = io::ld(ru + t);
    const float xr = io::ld(rd + t) + bias;
    float dl = xr, dsp = 1.0f;
    if (p.softplus) n1s_softplus(xr, dl, dsp);
    float dy;
    if constexpr (OF32) dy = rg[t]; else dy = io::ld(rg + t);
    const float a = fast_exp2(dl * A2);
    const float ht = sh[t * 64 + lane];
    const float hprev = t > 0 ? sh[(t - 1) * 64 + lane] : 0.0f;
    const float Bt = sB[t];
    const float gi = fmaf(sC[t], dy, gg);
    const float ga = gi * a;
    const float gha = ga * hprev;
    const float gd = gi * dl;
    const float sC_t = n1s_sum_to_lane63(dy * ht);
    const float sB_t = n1s_sum_to_lane63(gd * uu);
    if (lane == 63) {
      unsafeAtomicAdd(dBp + t, sB_t);
      unsafeAtomicAdd(dCp + t, sC_t);
    }
    const float dd = fmaf(gi * uu, Bt, gha * Aln) * dsp;
    io::st(ru + t, fmaf(gd, Bt, dy * Dv));                // du over u: step t is not read again
    io::st(rd + t, dd);
    dA_p = fmaf(gha, dl, dA_p);
    dD_p = fmaf(dy, uu, dD_p);
    db_p += dd;
    gg = ga;
  }
  unsafeAtomicAdd(p.dA + (int64_t)d, dA_p);               // dstate 1: dA is (dim, 1)
  if (p.dD) unsafeAtomicAdd(p.dD + d, dD_p);
  if (p.dbias) unsafeAtomicAdd(p.dbias + d, db_p);
  n1s_wave_sync();
  n1s_copy_out((io_t*)p.du + row0 * L, su, 64 * L * (int)sizeof(io_t), lane);
  n1s_copy_out((io_t*)p.ddelta + row0 * L, sd, 64 * L * (int)sizeof(io_t), lane);
}

#endif

}  // namespace mxvl
