// scan_fwd_stream.h -- the streaming selective-scan forward kernel (headline path, dstate <= 16).
//
// Same math and lane mapping as scan_fwd_kernel (scan_fwd.hip: 16 lanes per row, 8 consecutive steps
// per lane, fused-DPP prefix scan) but organised so HBM traffic overlaps the VALU work:
//   * u / delta of chunk c+1 are loaded straight into registers (each lane owns 32 contiguous bytes
//     per row: two 16-byte loads when rows are 16-byte aligned, eight dword loads otherwise) while
//     chunk c is being scanned; z of chunk c is requested before the state loop and consumed after it;
//   * the B/C tile of chunk c+1 is fetched by the whole workgroup during chunk c and written into
//     the other half of a double-buffered LDS tile after the state loop: one barrier per chunk;
//   * out leaves as two 16-byte stores per lane (aligned rows) or through a wave-private LDS tile
//     that turns the lane-strided layout back into coalesced dword stores (unaligned rows).
// LDS per workgroup: 2 * 2*N*128*4 B (B/C) + DT*128*4 B (out tile) + DT*N*8 B  -> 40.5 KiB at N=16.
#pragma once
#include "mxvl_common.h"

namespace mxvl {

typedef float f2 __attribute__((ext_vector_type(2)));

template <typename io_t, int NWAVES, bool VEC, int MINW, int NU = 1>
__global__ __launch_bounds__(NWAVES * 64, MINW) void scan_fwd_stream_kernel(const ScanArgs p) {
  constexpr int T = 8, LPR = 16, RPW = 4, DT = NWAVES * RPW, CH = 128, NT = NWAVES * 64, NMAX = 16;
  constexpr int BCV = (NMAX * CH / 4 + NT - 1) / NT;  // float4 per thread per array (VEC)
  constexpr int BCS = (NMAX * CH + NT - 1) / NT;      // floats per thread per array (scalar)
  static_assert(NT >= CH / 4 && (NT % (CH / 4)) == 0 && (NT % CH == 0 || CH % NT == 0), "staging shape");
  using io = Io<io_t>;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = p.N, L = p.L;
  float* sBC = smem;                          // [2 buffers][B|C][N][CH]
  float* sO = sBC + 4 * N * CH;               // [DT][CH] out tile (unaligned rows / ragged tail)
  float2* sAC = (float2*)(sO + DT * CH);      // [DT + 1][N] {A*log2(e), running state h}; row DT stays zero

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane >> 4, j = lane & 15;
  const int row = wave * RPW + r;
  const int b = blockIdx.y;
  const int dpg = p.dim / p.G;
  const int tiles = (dpg + DT - 1) / DT;
  const int g = blockIdx.x / tiles;
  const int d0 = g * dpg + (blockIdx.x - g * tiles) * DT;
  const int d_end = (g + 1) * dpg;
  const int d = d0 + row;
  const bool row_ok = d < d_end;
  const int dc = row_ok ? d : d_end - 1;  // clamped row: loads stay in bounds, stores are masked

  const io_t* __restrict__ pu = (const io_t*)p.u + (int64_t)b * p.u_bs + (int64_t)dc * p.u_ds + j * T;
  const io_t* __restrict__ pd = (const io_t*)p.delta + (int64_t)b * p.dl_bs + (int64_t)dc * p.dl_ds + j * T;
  const io_t* __restrict__ pz = p.z ? (const io_t*)p.z + (int64_t)b * p.z_bs + (int64_t)dc * p.z_ds + j * T : nullptr;
  io_t* __restrict__ po = (io_t*)p.out + (int64_t)b * p.o_bs + (int64_t)dc * p.o_ds + j * T;
  const io_t* __restrict__ Bp = (const io_t*)p.B + (int64_t)b * p.B_bs + (int64_t)g * p.B_gs;
  const io_t* __restrict__ Cp = (const io_t*)p.C + (int64_t)b * p.C_bs + (int64_t)g * p.C_gs;
  const bool has_z = pz != nullptr;

  for (int i = tid; i < (DT + 1) * N; i += NT) {
    const int rr = i / N, n = i - rr * N;
    const int dd = d0 + rr;
    sAC[i] = make_float2((rr < DT && dd < d_end) ? p.A[(int64_t)dd * p.A_ds + (int64_t)n * p.A_ns] * kLog2e : 0.0f, 0.0f);
  }
  const float bias = p.bias ? p.bias[dc] : 0.0f;
  const float Dv = p.D ? p.D[dc] : 0.0f;

  // ---- B/C tile fetch (global -> registers) and commit (registers -> LDS buffer) ------------------
  float4 bq[VEC ? BCV : 1], cq[VEC ? BCV : 1];
  float bs[VEC ? 1 : BCS], cs[VEC ? 1 : BCS];
  auto bc_fetch = [&](int t0) {
    const bool full = t0 + CH <= L;
    if constexpr (VEC) {
      constexpr int CQ = CH / 4, RSTEP = NT / CQ;
      const int e4 = (tid % CQ) * 4;
#pragma unroll
      for (int k = 0; k < BCV; ++k) {
        // NU == 2: a thread fetches rows (2m, 2m+1) so the commit can interleave the state pair
        const int n = (NU == 2 && BCV == 2) ? 2 * (tid / CQ) + k : tid / CQ + k * RSTEP;
        bq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        cq[k] = bq[k];
        if (n < N) {
          if (full) {
            bq[k] = ld4<io_t>(Bp + (int64_t)n * p.B_ns + t0 + e4);
            cq[k] = ld4<io_t>(Cp + (int64_t)n * p.C_ns + t0 + e4);
          } else {
            float tb[4] = {0.f, 0.f, 0.f, 0.f}, tc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (t0 + e4 + q < L) {
                tb[q] = io::ld(Bp + (int64_t)n * p.B_ns + t0 + e4 + q);
                tc[q] = io::ld(Cp + (int64_t)n * p.C_ns + t0 + e4 + q);
              }
            bq[k] = make_float4(tb[0], tb[1], tb[2], tb[3]);
            cq[k] = make_float4(tc[0], tc[1], tc[2], tc[3]);
          }
        }
      }
    } else {
      constexpr int RSTEP = (NT >= CH) ? NT / CH : 1, CSTEP = (NT >= CH) ? CH : NT;
      static_assert(NT >= CH, "scalar staging assumes one column per thread");
      const int e = tid % CSTEP;
      const bool ok = t0 + e < L;
#pragma unroll
      for (int k = 0; k < BCS; ++k) {
        const int n = tid / CSTEP + k * RSTEP;
        bs[k] = 0.f;
        cs[k] = 0.f;
        if (n < N && ok) {
          bs[k] = io::ld(Bp + (int64_t)n * p.B_ns + t0 + e);
          cs[k] = io::ld(Cp + (int64_t)n * p.C_ns + t0 + e);
        }
      }
    }
  };
  auto bc_commit = [&](int buf) {
    float* dB = sBC + buf * 2 * N * CH;
    float* dC = dB + N * CH;
    if constexpr (VEC) {
      constexpr int CQ = CH / 4, RSTEP = NT / CQ;
      const int e4 = (tid % CQ) * 4;
      if constexpr (NU == 2) {
        static_assert(BCV == 2, "packed layout needs two rows per thread");
        const int m = tid / CQ;  // state pair
        if (2 * m < N) {         // layout [N/2][CH][2]
          float4* qB = (float4*)(dB + (m * CH + e4) * 2);
          float4* qC = (float4*)(dC + (m * CH + e4) * 2);
          qB[0] = make_float4(bq[0].x, bq[1].x, bq[0].y, bq[1].y);
          qB[1] = make_float4(bq[0].z, bq[1].z, bq[0].w, bq[1].w);
          qC[0] = make_float4(cq[0].x, cq[1].x, cq[0].y, cq[1].y);
          qC[1] = make_float4(cq[0].z, cq[1].z, cq[0].w, cq[1].w);
        }
      } else {
#pragma unroll
        for (int k = 0; k < BCV; ++k) {
          const int n = tid / CQ + k * RSTEP;
          // lane-major halves: [half][16 lanes][4] so a 16-lane ds_read_b128 touches 16 distinct slots
          const int pos = ((e4 >> 2) & 1) * 64 + (e4 >> 3) * 4;
          if (n < N) {
            *(float4*)(dB + n * CH + pos) = bq[k];
            *(float4*)(dC + n * CH + pos) = cq[k];
          }
        }
      }
    } else {
      constexpr int RSTEP = NT / CH;
      const int e = tid % CH;
#pragma unroll
      for (int k = 0; k < BCS; ++k) {
        const int n = tid / CH + k * RSTEP;
        if (n < N) {
          const int o = (NU == 2) ? ((n >> 1) * CH + e) * 2 + (n & 1)
                                  : n * CH + ((e >> 2) & 1) * 64 + (e >> 3) * 4 + (e & 3);
          dB[o] = bs[k];
          dC[o] = cs[k];
        }
      }
    }
  };
  // ---- a lane's 8 consecutive elements of one row ----------------------------------------------------
  auto row_fetch = [&](const io_t* q, int t0, float (&v)[T]) {
    if (t0 + CH <= L) {
      if constexpr (VEC) {
        const float4 a0 = ld4<io_t>(q + t0), a1 = ld4<io_t>(q + t0 + 4);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      } else {
#pragma unroll
        for (int i = 0; i < T; ++i) v[i] = io::ld(q + t0 + i);
      }
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) v[i] = (t0 + j * T + i < L) ? io::ld(q + t0 + i) : 0.0f;
    }
  };

  const int nchunks = (L + CH - 1) / CH;
  float un[T], dn[T];
  bc_fetch(0);
  row_fetch(pu, 0, un);
  row_fetch(pd, 0, dn);
  bc_commit(0);

  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * CH;
    const bool full = t0 + CH <= L;
    const bool more = c + 1 < nchunks;
    __syncthreads();  // B/C tile c is complete in buffer c&1; nobody still reads buffer (c+1)&1

    float dl[T], du[T], y[T], zz[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      float x = dn[i] + bias;
      if (p.softplus) x = softplus(x);
      if (!full) x = (t0 + j * T + i < L) ? x : 0.0f;  // padding steps are the identity map
      dl[i] = x;
      du[i] = x * un[i];
      y[i] = Dv * un[i];
    }
    // requests for the next chunk (and this chunk's z) go out before the long state loop
    if (more) {
      bc_fetch(t0 + CH);
      row_fetch(pu, t0 + CH, un);
      row_fetch(pd, t0 + CH, dn);
    }
    if (has_z) row_fetch(pz, t0, zz);

    float dsum = 0.0f;
#pragma unroll
    for (int i = 0; i < T; ++i) dsum += dl[i];

    if (p.ckpt != nullptr) {
      // checkpoint = state entering the chunk = the running state this wave left in LDS (CH == kCkptLen)
      for (int i = lane; i < RPW * N; i += 64) {
        const int rr = i / N, n = i - rr * N;
        const int dd = d0 + wave * RPW + rr;
        if (dd < d_end) p.ckpt[(((int64_t)b * p.dim + dd) * p.n_ckpt + c) * N + n] = sAC[(wave * RPW + rr) * N + n].y;
      }
    }
    float2* ac = sAC + row * N;
    const float2* ac_in = sAC + ((j == 0) ? row : DT) * N;  // only lane 0 sees the state entering the chunk
    const float* cB = sBC + (c & 1) * 2 * N * CH + j * 4;
    const float* cC = cB + N * CH;

    if constexpr (NU == 2) {
      // two states per iteration on packed fp32 (v_pk_mul/v_pk_fma): B/C tile is [N/2][CH][2]
      const float* cB2 = sBC + (c & 1) * 2 * N * CH + j * T * 2;
      const float* cC2 = cB2 + N * CH;
      for (int m = 0; m < ((p.ablate & 1) ? 0 : N / 2); ++m) {
        const float4 ac4 = *(const float4*)(ac + 2 * m);   // {A2_n, h_n, A2_n+1, h_n+1}
        const f2 A2 = f2{ac4.x, ac4.z};
        f2 a[T], bb[T], cv[T];
#pragma unroll
        for (int q = 0; q < T / 2; ++q) {
          const float4 b4 = *(const float4*)(cB2 + m * CH * 2 + q * 4);
          const float4 c4 = *(const float4*)(cC2 + m * CH * 2 + q * 4);
          bb[2 * q] = f2{b4.x, b4.y}; bb[2 * q + 1] = f2{b4.z, b4.w};
          cv[2 * q] = f2{c4.x, c4.y}; cv[2 * q + 1] = f2{c4.z, c4.w};
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
          const f2 e = dl[i] * A2;
          a[i] = f2{fast_exp2(e.x), fast_exp2(e.y)};
          bb[i] = du[i] * bb[i];
        }
        f2 h = bb[0];
#pragma unroll
        for (int i = 1; i < T; ++i) h = a[i] * h + bb[i];   // pass 1
        const f2 pe = dsum * A2;
        float P0 = fast_exp2(pe.x), P1 = fast_exp2(pe.y);
        float x0 = ac4.y, x1 = ac4.w;                        // states entering the chunk
        float h0 = fmaf(P0, (j == 0) ? x0 : 0.0f, h.x);
        float h1 = fmaf(P1, (j == 0) ? x1 : 0.0f, h.y);
        scan16_x2(h0, P0, x0, h1, P1, x1);
        if (j == LPR - 1) { ac[2 * m].y = h0; ac[2 * m + 1].y = h1; }
        h = f2{x0, x1};
        f2 y2[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {                        // pass 2
          h = a[i] * h + bb[i];
          y2[i] = cv[i] * h;
        }
#pragma unroll
        for (int i = 0; i < T; ++i) y[i] += y2[i].x + y2[i].y;
      }
    } else
#pragma unroll 4
    for (int n = 0; n < ((p.ablate & 1) ? 0 : N); ++n) {
      const float2 A2c = make_float2(ac[n].x, ac_in[n].y);
      float a[T], bb[T], cv[T];
      {
        const float4 b0 = *(const float4*)(cB + n * CH), b1 = *(const float4*)(cB + n * CH + 64);
        const float4 c0 = *(const float4*)(cC + n * CH), c1 = *(const float4*)(cC + n * CH + 64);
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
        cv[0] = c0.x; cv[1] = c0.y; cv[2] = c0.z; cv[3] = c0.w; cv[4] = c1.x; cv[5] = c1.y; cv[6] = c1.z; cv[7] = c1.w;
      }
#pragma unroll
      for (int i = 0; i < T; ++i) {
        a[i] = fast_exp2(dl[i] * A2c.x);
        bb[i] = du[i] * bb[i];
      }
      float h = bb[0];
#pragma unroll
      for (int i = 1; i < T; ++i) h = fmaf(a[i], h, bb[i]);   // pass 1: lane map h_out = P*h_in + h
      float P = fast_exp2(A2c.x * dsum);
      float x = A2c.y;                                        // state entering the chunk
      float hl = fmaf(P, A2c.y, h);                           // lane 0 absorbs it (others read 0)
      scan16_x1(hl, P, x);
      if (j == LPR - 1) ac[n].y = hl;                         // state leaving the chunk
      h = x;
#pragma unroll
      for (int i = 0; i < T; ++i) {                           // pass 2
        h = fmaf(a[i], h, bb[i]);
        y[i] = fmaf(cv[i], h, y[i]);
      }
    }

    if (more) bc_commit((c + 1) & 1);
    if (has_z) {
#pragma unroll
      for (int i = 0; i < T; ++i) y[i] *= silu(zz[i]);
    }
    if (p.ablate & 8) continue;
    if (VEC && full) {
      if (row_ok) {
        st4<io_t>(po + t0, make_float4(y[0], y[1], y[2], y[3]));
        st4<io_t>(po + t0 + 4, make_float4(y[4], y[5], y[6], y[7]));
      }
    } else {
      float4* so4 = (float4*)(sO + row * CH + j * T);
      so4[0] = make_float4(y[0], y[1], y[2], y[3]);
      so4[1] = make_float4(y[4], y[5], y[6], y[7]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int wrow = wave * RPW + rr;
        const int dd = d0 + wrow;
        io_t* q = (io_t*)p.out + (int64_t)b * p.o_bs + (int64_t)dd * p.o_ds + t0;
#pragma unroll
        for (int e = lane; e < CH; e += 64)
          if (dd < d_end && t0 + e < L) io::st(q + e, sO[wrow * CH + e]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  if (p.last_state) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < RPW * N; i += 64) {
      const int rr = i / N, n = i - rr * N;
      const int dd = d0 + wave * RPW + rr;
      if (dd < d_end) p.last_state[((int64_t)b * p.dim + dd) * N + n] = sAC[(wave * RPW + rr) * N + n].y;
    }
  }
}

}  // namespace mxvl
