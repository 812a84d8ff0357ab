// scan_n1.h -- selective-scan forward for dstate = 1 (VMamba / R2GenCSR: SS2D "v3noz", d_state 1, K = 4 direction groups;
// R2GenCSR/VMamba/classification/models/vmamba.py:294-312, 318-427; vssm_base_224.yaml: scan dim K * d_inner = 1024 / 2048 / 4096 /
// 8192 at L = 3136 / 784 / 196 / 49).
//
// With ONE state per channel there is no state loop to amortise a chunk's set-up over and no B / C tile worth sharing through LDS:
// the N = 16 kernels spend a 128-step chunk's whole prologue (tile staging, barriers, tables) on 1/16 of the work, and a 196-token
// row pays for 256 steps.  Here the rows of the launch are ONE flat sequence of batch * dim * L elements and a wave owns `rw`
// consecutive rows of it: a pass is 64 lanes x T consecutive elements (T = 8 or 4: one 16- / 8-byte load of 16-bit rows, rows never
// straddle a lane because T | L), whatever row they belong to -- the first step of a row carries a = 0 (no state enters a row), so
// the wave-wide prefix scan of the lanes' affine maps (four DPP row shifts + row_bcast15 + row_bcast31) needs no segment logic,
// and the state that leaves lane 63 is the next pass's carry.  No LDS, no barrier; u / delta / B / C of pass p + 1 are requested
// before pass p is computed.  A lane derives (batch, channel, group, step) of its T elements from its flat offset with one
// multiply-high (magic reciprocal of L) and an add: parameters A, D, delta_bias and the B / C row pieces are per-lane loads that
// hit L1 / L2 (B / C rows are shared by dim / n_groups channels).
// Checkpoints keep the layout of the other kernels -- the state entering every 128-step chunk of every row,
// ckpt (batch, dim, ceil(L / 128), 1) -- so mxvl_scan_bwd needs no flag: a lane whose first step is a multiple of 128 stores the
// state that enters it.
// Algorithmic HBM bytes: elt * (2 B D L [u, delta] + out) + 2 elt B G L: no z in this path (v3noz has none).
#pragma once
#include "mxvl_common.h"

namespace mxvl {

struct ScanN1Geom {
  int rw;            // rows per wave
  int n_waves;       // waves of the launch
  uint32_t magL;     // floor(2^32 / L) + 1
  uint32_t magG;     // floor(2^32 / (dim / G)) + 1
  int out_vec;       // out rows allow T-element vector stores
};

template <typename io_t, int T> struct N1Raw;
template <> struct N1Raw<float, 4> { typedef float4 type; };
template <> struct N1Raw<bf16_t, 4> { typedef uint2 type; };
template <> struct N1Raw<f16_t, 4> { typedef uint2 type; };
template <> struct N1Raw<bf16_t, 8> { typedef uint4 type; };
template <> struct N1Raw<f16_t, 8> { typedef uint4 type; };

template <typename io_t, int T> __device__ __forceinline__ void n1_unpack(const typename N1Raw<io_t, T>::type& r, float (&v)[T]) {
  if constexpr (sizeof(io_t) == 4) {
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  } else {
    io_t t[T];
    *(typename N1Raw<io_t, T>::type*)t = r;
#pragma unroll
    for (int i = 0; i < T; ++i) v[i] = Io<io_t>::ld(t + i);
  }
}

template <typename io_t, int T, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void scan_n1_fwd_kernel(const ScanArgs p, const ScanN1Geom gm) {
  typedef typename N1Raw<io_t, T>::type raw_t;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NWAVES + (threadIdx.x >> 6)));
  if (wid >= gm.n_waves) return;
  const int64_t rows = (int64_t)p.batch * p.dim;
  const int64_t r0 = (int64_t)wid * gm.rw;
  const int nrows = (int)(rows - r0 < gm.rw ? rows - r0 : gm.rw);
  const int L = p.L, RWL = nrows * L;
  const int b0 = (int)(r0 / p.dim), d0 = (int)(r0 - (int64_t)b0 * p.dim);
  const int dpg = p.dim / p.G;
  const int npass = (RWL + 64 * T - 1) / (64 * T);
  const bool of32 = p.out_f32 != 0;

  struct Pos { int b, d, t; bool valid; };
  auto locate = [&](int pass) {
    const int f = pass * (64 * T) + lane * T;
    Pos q;
    q.valid = f < RWL;
    const int fc = q.valid ? f : 0;                        // lanes past the end re-read element 0 (finite); their delta is forced to 0
    const int rl = (int)__umulhi((uint32_t)fc, gm.magL);
    q.t = fc - rl * L;
    int d = d0 + rl, b = b0;
    while (d >= p.dim) { d -= p.dim; ++b; }                // rw <= dim is not required: a wave may span several batch elements
    q.b = b; q.d = d;
    return q;
  };
  struct Req { raw_t u, dl, B, C; float A2, Dv, bias; };
  auto fetch = [&](const Pos& q, Req& r) {
    const int g = (int)__umulhi((uint32_t)q.d, gm.magG);
    const int dr = delta_row(q.d, p.dl_ratio, p.dl_magic);
    r.u = *(const raw_t*)((const io_t*)p.u + (int64_t)q.b * p.u_bs + (int64_t)q.d * p.u_ds + q.t);
    r.dl = *(const raw_t*)((const io_t*)p.delta + (int64_t)q.b * p.dl_bs + (int64_t)dr * p.dl_ds + q.t);
    r.B = *(const raw_t*)((const io_t*)p.B + (int64_t)q.b * p.B_bs + (int64_t)g * p.B_gs + q.t);
    r.C = *(const raw_t*)((const io_t*)p.C + (int64_t)q.b * p.C_bs + (int64_t)g * p.C_gs + q.t);
    r.A2 = p.A[(int64_t)q.d * p.A_ds] * kLog2e;
    r.Dv = p.D ? p.D[q.d] : 0.0f;
    r.bias = p.bias ? p.bias[dr] : 0.0f;
  };

  float carry = 0.0f;
  Pos pos = locate(0), pos_n = pos;
  Req cur, nxt;
  fetch(pos, cur);
  nxt = cur;
  for (int pass = 0; pass < npass; ++pass) {
    if (pass + 1 < npass) {
      pos_n = locate(pass + 1);
      fetch(pos_n, nxt);
    }
    float u[T], dl[T], Bv[T], Cv[T], a[T], bb[T];
    n1_unpack<io_t, T>(cur.u, u);
    n1_unpack<io_t, T>(cur.dl, dl);
    n1_unpack<io_t, T>(cur.B, Bv);
    n1_unpack<io_t, T>(cur.C, Cv);
#pragma unroll
    for (int i = 0; i < T; ++i) dl[i] += cur.bias;
    if (p.softplus) {
#pragma unroll
      for (int i = 0; i < T; ++i) dl[i] = softplus(dl[i]);
    }
    if (!pos.valid) {
#pragma unroll
      for (int i = 0; i < T; ++i) dl[i] = 0.0f;           // identity map: a = 1, b = 0
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
      a[i] = fast_exp2(dl[i] * cur.A2);
      bb[i] = dl[i] * u[i] * Bv[i];
    }
    if (pos.t == 0) a[0] = 0.0f;                           // no state enters a row
    float hl = bb[0], P = a[0];
#pragma unroll
    for (int i = 1; i < T; ++i) {
      hl = fmaf(a[i], hl, bb[i]);
      P *= a[i];
    }
    if (lane == 0) hl = fmaf(P, carry, hl);                // lane 0 absorbs the state that left the previous pass
    float x = carry;
    scan_generic<64>(hl, P, x, lane);                      // x = state entering this lane's steps
    carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hl), 63));
    if (p.ckpt != nullptr && pos.valid && (pos.t & (kCkptLen - 1)) == 0)
      p.ckpt[((int64_t)pos.b * p.dim + pos.d) * p.n_ckpt + (pos.t >> 7)] = pos.t == 0 ? 0.0f : x;
    float h = x, y[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      h = fmaf(a[i], h, bb[i]);
      y[i] = fmaf(Cv[i], h, cur.Dv * u[i]);
    }
    if (pos.valid) {
      const int64_t o = (int64_t)pos.b * p.o_bs + (int64_t)pos.d * p.o_ds + pos.t;
      if (gm.out_vec) {
#pragma unroll
        for (int k = 0; k < T / 4; ++k) st4_out<io_t>(p.out, o + 4 * k, make_float4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]), of32);
      } else {
#pragma unroll
        for (int i = 0; i < T; ++i) st_out<io_t>(p.out, o + i, y[i], of32);
      }
      if (p.last_state != nullptr && pos.t + T == L) p.last_state[(int64_t)pos.b * p.dim + pos.d] = h;
    }
    cur = nxt;
    pos = pos_n;
  }
}

}  // namespace mxvl
