// scan_n1_bwd.h -- selective-scan backward for dstate = 1 without z (VMamba / R2GenCSR SS2D; the vendored oflex kernels'
// R2GenCSR/VMamba/kernels/selective_scan/csrc/selective_scan/cusoflex/selective_scan_bwd_kernel_oflex.cuh at dstate 1, reached through
// vmamba.py:300-312).  Included by scan_bwd.hip (ScanBwdArgs).
//
// One state per channel: the N = 16 kernel's machinery (B / C tiles in LDS, flush groups of four states, a barrier per group, the
// 128-step chunk grid that makes a 196-token row pay for 256 steps) was 0.11 of the HBM roofline here.  This kernel has none of it:
//   * a wave owns `rw` consecutive channels of ONE (batch element, B / C group) and walks their rows PASS-MAJOR: for pass k (64 lanes
//     x T consecutive steps, last pass first) it visits its rw rows one after the other, so lane j meets the same time steps in
//     every row of the pass: the dB / dC shares of the wave's rows add up IN REGISTERS (2 T accumulators), B_t / C_t are loaded once
//     per pass, and nothing about a row is per lane (its A, D, delta_bias, checkpoint are wave-uniform);
//   * the forward state is recomputed from the checkpoint that enters the pass (pass boundaries are multiples of 64 T = 256 / 512
//     steps: every one is a 128-step checkpoint of the forward), by the wave-wide DPP scan of scan_n1.h;
//   * the adjoint recurrence g_t = C_t dy_t + a_{t+1} g_{t+1} runs right-to-left: the lanes' adjoint maps are mirrored across the
//     wave (ds_bpermute), scanned with the same prefix scan, mirrored back; the adjoint that leaves a pass to the left waits in LDS
//     (one float per row) for pass k - 1;
//   * dB / dC: the four waves of a workgroup (4 rw channels of the same group) add their register sums through LDS, then one fp32
//     atomic per (t, workgroup, pass); dA / dD / ddelta_bias: a wave sum per (row, pass), three atomics.
// u / delta / dout of row i + 1 are requested before row i is computed.  Algorithmic bytes: elt (3 B D L reads +
// 2 B D L writes) + 2 elt B G L + 8 B G L + checkpoints.
#pragma once
#include "mxvl_common.h"

namespace mxvl {

struct ScanN1BwdGeom {
  int rw;          // rows (channels) per wave; 4 rw divides dim / n_groups
  int do_vec, du_vec, dd_vec;   // dout / du / ddelta rows allow T-element vector access
};

template <typename io_t, int T> struct N1BRaw;
template <> struct N1BRaw<float, 4> { typedef float4 type; };
template <> struct N1BRaw<bf16_t, 4> { typedef uint2 type; };
template <> struct N1BRaw<f16_t, 4> { typedef uint2 type; };
template <> struct N1BRaw<bf16_t, 8> { typedef uint4 type; };
template <> struct N1BRaw<f16_t, 8> { typedef uint4 type; };

template <typename io_t, int T> __device__ __forceinline__ void n1b_unpack(const typename N1BRaw<io_t, T>::type& r, float (&v)[T]) {
  if constexpr (sizeof(io_t) == 4) {
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  } else {
    io_t t[T];
    *(typename N1BRaw<io_t, T>::type*)t = r;
#pragma unroll
    for (int i = 0; i < T; ++i) v[i] = Io<io_t>::ld(t + i);
  }
}
template <typename io_t, int T> __device__ __forceinline__ void n1b_store(io_t* q, const float (&v)[T], bool vec) {
  if (vec) {
#pragma unroll
    for (int k = 0; k < T / 4; ++k) st4<io_t>(q + 4 * k, make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]));
  } else {
#pragma unroll
    for (int i = 0; i < T; ++i) Io<io_t>::st(q + i, v[i]);
  }
}

// inclusive scan of the affine maps (P, h) over the 64 lanes + exclusive shift (scan_fwd.hip scan_generic<64>): x enters as the
// carry in front of lane 0 (which has absorbed it into h already) and leaves as the value in front of every lane
__device__ inline void n1b_scan64(float& hl, float& P, float& x, int lane) {
  float pb, pa;
  pb = dpp<DPP_ROW_SHR(1)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(1)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_SHR(2)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(2)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_SHR(4)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(4)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_SHR(8)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(8)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_BCAST15, 0xa>(0.0f, hl); pa = dpp<DPP_ROW_BCAST15, 0xa>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_BCAST31, 0xc>(0.0f, hl); pa = dpp<DPP_ROW_BCAST31, 0xc>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  const float car = x;
  x = dpp<DPP_WAVE_SHR1>(car, hl);
  x = (lane == 0) ? car : x;
}
// sum over the 64 lanes; the total lands in lane 63
__device__ inline float n1b_sum_to_lane63(float v) {
  v += dpp<DPP_ROW_SHR(1)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(2)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(4)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(8)>(0.0f, v);
  v += dpp<DPP_ROW_BCAST15, 0xa>(0.0f, v);
  v += dpp<DPP_ROW_BCAST31, 0xc>(0.0f, v);
  return v;
}
__device__ __forceinline__ float n1b_mirror(float v, int lane) {      // value of lane 63 - lane
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((63 - lane) << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float n1b_lane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// OF32: dout is fp32 whatever the io dtype (oflex i16o32) -- a compile-time switch: as a run-time one both forms of the in-flight dout
// registers existed in every prefetch stage
template <typename io_t, int T, int NWAVES, bool OF32>
__global__ __launch_bounds__(NWAVES * 64) void scan_n1_bwd_kernel(const ScanBwdArgs p, const ScanN1BwdGeom gm) {
  typedef typename N1BRaw<io_t, T>::type raw_t;
  extern __shared__ __attribute__((aligned(16))) float n1b_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rw = gm.rw, L = p.L;
  float* sRed = n1b_smem;                                  // [NWAVES][2 T][64] dB / dC shares of the waves
  float* sG = sRed + NWAVES * 2 * T * 64 + wave * rw;      // [rw] adjoint leaving pass k to the left, per row of this wave
  const int64_t r0 = ((int64_t)blockIdx.x * NWAVES + wave) * rw;     // first row of this wave: (batch, channel) = (r0 / dim, r0 % dim)
  const int b = (int)(r0 / p.dim), d0 = (int)(r0 - (int64_t)b * p.dim);
  const int g = d0 / (p.dim / p.G);                        // 4 rw | dim / G: the workgroup's channels share one B / C group
  constexpr bool of32 = OF32;
  const int npass = (L + 64 * T - 1) / (64 * T);
  const io_t* Bp = (const io_t*)p.B + (int64_t)b * p.B_bs + (int64_t)g * p.B_gs;
  const io_t* Cp = (const io_t*)p.C + (int64_t)b * p.C_bs + (int64_t)g * p.C_gs;
  float* dBp = p.dB + (int64_t)b * p.dB_bs + (int64_t)g * p.dB_gs;
  float* dCp = p.dC + (int64_t)b * p.dC_bs + (int64_t)g * p.dC_gs;

  struct Req { raw_t u, dl; raw_t gy; float4 gf[OF32 ? T / 4 : 1]; float A2, Dv, bias, cF; };
  for (int k = npass - 1; k >= 0; --k) {
    const int t = k * (64 * T) + lane * T;
    const bool valid = t < L;                              // T | L: a lane is inside the row or past it as a whole
    const int tc = valid ? t : 0;                          // lanes past the end re-read step 0 (finite); delta and dout are zeroed
    float Bv[T], Cv[T];
    {
      const raw_t rb = *(const raw_t*)(Bp + tc), rc = *(const raw_t*)(Cp + tc);
      n1b_unpack<io_t, T>(rb, Bv);
      n1b_unpack<io_t, T>(rc, Cv);
    }
    float accB[T], accC[T];
#pragma unroll
    for (int i = 0; i < T; ++i) accB[i] = accC[i] = 0.0f;

    auto fetch = [&](int i, Req& r) {
      const int d = d0 + i;
      const int dr = delta_row(d, p.dl_ratio, p.dl_magic);
      r.u = *(const raw_t*)((const io_t*)p.u + (int64_t)b * p.u_bs + (int64_t)d * p.u_ds + tc);
      r.dl = *(const raw_t*)((const io_t*)p.delta + (int64_t)b * p.dl_bs + (int64_t)dr * p.dl_ds + tc);
      const int64_t go = (int64_t)b * p.do_bs + (int64_t)d * p.do_ds + tc;
      if constexpr (of32) {
        if (gm.do_vec) {
#pragma unroll
          for (int q = 0; q < T / 4; ++q) r.gf[q] = *(const float4*)((const float*)p.dout + go + 4 * q);
        } else {
#pragma unroll
          for (int q = 0; q < T / 4; ++q) {
            const float* s = (const float*)p.dout + go + 4 * q;
            r.gf[q] = make_float4(s[0], s[1], s[2], s[3]);
          }
        }
      } else if (gm.do_vec) {
        r.gy = *(const raw_t*)((const io_t*)p.dout + go);
      } else {
        io_t tmp[T];
#pragma unroll
        for (int q = 0; q < T; ++q) tmp[q] = ((const io_t*)p.dout)[go + q];
        r.gy = *(const raw_t*)tmp;
      }
      r.A2 = p.A[(int64_t)d * p.A_ds] * kLog2e;
      r.Dv = p.D ? p.D[d] : 0.0f;
      r.bias = p.bias ? p.bias[dr] : 0.0f;
      r.cF = k > 0 ? p.ckpt[((int64_t)b * p.dim + d) * p.n_ckpt + (k * (64 * T) >> 7)] : 0.0f;    // state entering step k * 64 T
    };
    // rows i + 1 .. i + 2 are in flight while row i is computed: with one row ahead a wave kept 1.5 - 3 KB outstanding and the
    // kernel ran at 2.2 - 2.5 TB/s whatever its arithmetic (profiles/r06_scan_n1_bench.txt)
    Req cur, nx1, nx2;
    fetch(0, cur);
    nx1 = cur;
    if (rw > 1) fetch(1, nx1);
    nx2 = nx1;
    for (int i = 0; i < rw; ++i) {
      if (i + 2 < rw) fetch(i + 2, nx2);
      const int d = d0 + i;
      float u[T], dl[T], dy[T], dsp[T], a[T], bb[T], h[T];
      n1b_unpack<io_t, T>(cur.u, u);
      n1b_unpack<io_t, T>(cur.dl, dl);
      if constexpr (of32) {
#pragma unroll
        for (int q = 0; q < T / 4; ++q) { dy[4 * q] = cur.gf[q].x; dy[4 * q + 1] = cur.gf[q].y; dy[4 * q + 2] = cur.gf[q].z; dy[4 * q + 3] = cur.gf[q].w; }
      } else {
        n1b_unpack<io_t, T>(cur.gy, dy);
      }
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const float xr = dl[q] + cur.bias;
        if (p.softplus) {                                   // softplus and its derivative from one exponential (mxvl_common.h softplus)
          const float w = fast_exp(xr), s = 1.0f + w, den = s - 1.0f;
          const float l = fast_log2(s) * 0.6931471805599453f;
          const float r = (den == 0.0f) ? w : l * w * fast_rcp(den);
          dl[q] = xr > 20.0f ? xr : r;
          dsp[q] = xr > 20.0f ? 1.0f : w * fast_rcp(s);
        } else {
          dl[q] = xr;
          dsp[q] = 1.0f;
        }
        if (!valid) { dl[q] = 0.0f; dy[q] = 0.0f; }
      }
#pragma unroll
      for (int q = 0; q < T; ++q) {
        a[q] = fast_exp2(dl[q] * cur.A2);
        bb[q] = dl[q] * u[q] * Bv[q];
      }
      if (t == 0) a[0] = 0.0f;                              // no state enters the row, no adjoint leaves it
      // ---- forward recompute ---------------------------------------------------------------------------------------
      float hl = bb[0], P = a[0];
#pragma unroll
      for (int q = 1; q < T; ++q) {
        hl = fmaf(a[q], hl, bb[q]);
        P *= a[q];
      }
      const float Plane = P;
      if (lane == 0) hl = fmaf(P, cur.cF, hl);
      float x = cur.cF;
      n1b_scan64(hl, P, x, lane);                           // x = state entering this lane's steps
      {
        float hh = x;
#pragma unroll
        for (int q = 0; q < T; ++q) {
          hh = fmaf(a[q], hh, bb[q]);
          h[q] = hh;
        }
      }
      // ---- adjoint: g_q = C_q dy_q + a_{q+1} g_{q+1}; a lane hands a_0 g_0 to its left neighbour -------------------------
      float ql = Cv[T - 1] * dy[T - 1];
#pragma unroll
      for (int q = T - 2; q >= 0; --q) ql = fmaf(a[q + 1], ql, Cv[q] * dy[q]);
      ql *= a[0];
      const float gin = (k == npass - 1) ? 0.0f : sG[i];   // a g entering the pass from the right (left by pass k + 1)
      float qm = n1b_mirror(ql, lane), Pm = n1b_mirror(Plane, lane);
      if (lane == 0) qm = fmaf(Pm, gin, qm);
      float gxm = gin;
      n1b_scan64(qm, Pm, gxm, lane);
      if (k > 0 && lane == 63) sG[i] = qm;                  // mirrored lane 63 = lane 0: what leaves the pass to the left
      float gg = n1b_mirror(gxm, lane);                     // a_{next} g_{next} entering this lane from the right
      // ---- per-step gradients ------------------------------------------------------------------------------------------
      float o_du[T], o_dd[T], dA_p = 0.0f, dD_p = 0.0f, db_p = 0.0f;
      const float Aln = cur.A2 * 0.6931471805599453f;       // A = A2 ln 2
#pragma unroll
      for (int q = T - 1; q >= 0; --q) {
        const float gi = fmaf(Cv[q], dy[q], gg);
        const float hprev = (q == 0) ? x : h[q - 1];
        const float ga = gi * a[q];
        const float gha = ga * hprev;
        accC[q] = fmaf(dy[q], h[q], accC[q]);
        const float gd = gi * dl[q];
        accB[q] = fmaf(gd, u[q], accB[q]);
        o_du[q] = fmaf(gd, Bv[q], dy[q] * cur.Dv);
        const float dd = fmaf(gi * u[q], Bv[q], gha * Aln) * dsp[q];
        o_dd[q] = dd;
        dA_p = fmaf(gha, dl[q], dA_p);
        dD_p = fmaf(dy[q], u[q], dD_p);
        db_p += dd;
        gg = ga;
      }
      if (valid) {
        n1b_store<io_t, T>((io_t*)p.du + (int64_t)b * p.du_bs + (int64_t)d * p.du_ds + t, o_du, gm.du_vec != 0);
        n1b_store<io_t, T>((io_t*)p.ddelta + (int64_t)b * p.dd_bs + (int64_t)d * p.dd_ds + t, o_dd, gm.dd_vec != 0);
      }
      dA_p = n1b_sum_to_lane63(dA_p);
      dD_p = n1b_sum_to_lane63(dD_p);
      db_p = n1b_sum_to_lane63(db_p);
      if (lane == 63) {
        unsafeAtomicAdd(p.dA + (int64_t)d, dA_p);           // dstate 1: dA is (dim, 1)
        if (p.dD) unsafeAtomicAdd(p.dD + d, dD_p);
        if (p.dbias) unsafeAtomicAdd(p.dbias + d, db_p);
      }
      cur = nx1;
      nx1 = nx2;
    }
    // ---- dB / dC of this pass: the four waves' register sums through LDS, one atomic per (step, workgroup) -------------------------
    __syncthreads();                                        // the previous pass's readers are done with sRed
#pragma unroll
    for (int q = 0; q < T; ++q) {
      sRed[(wave * 2 * T + q) * 64 + lane] = accB[q];
      sRed[(wave * 2 * T + T + q) * 64 + lane] = accC[q];
    }
    __syncthreads();
    // 2 T x 64 sums, NWAVES * 64 threads: thread e sums element e, e + 256, ...
    for (int e = threadIdx.x; e < 2 * T * 64; e += NWAVES * 64) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) s += sRed[w * 2 * T * 64 + e];
      const int q = e >> 6, ln = e & 63;                    // q < T: dB of step ln * T + q, else dC
      const int ts = k * (64 * T) + ln * T + (q < T ? q : q - T);
      if (ts < L) unsafeAtomicAdd((q < T ? dBp : dCp) + ts, s);
    }
  }
}

}  // namespace mxvl
