// scan_fwd.hip -- selective-scan forward for gfx950 (MI355X, CDNA4, wave64).
//
// Replaces the reference's CUDA selective_scan_fwd_kernel
// (R2GenCSR/VMamba/kernels/selective_scan/csrc/selective_scan/cus/selective_scan_fwd_kernel.cuh:61-172)
// = mamba_ssm selective_scan_fn forward (call site
// CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:693-704).  Not a translation: the CUDA
// kernel gives one 32..128-thread block to ONE (b,d) row, re-reads the B/C rows for every row and
// leans on cub BlockLoad/BlockScan.  Here:
//
//   * a workgroup owns DT = NWAVES*64/LPR rows (consecutive d inside one B/C group of one batch
//     element) and walks the sequence in chunks of CH = LPR*T steps.  The B/C tile [N][CH] of the
//     chunk is staged in LDS ONCE and shared by all DT rows;
//   * each row is spread over LPR lanes, every lane owning T CONSECUTIVE time steps, so the
//     recurrence h_t = a_t h_{t-1} + b_t is: a serial fold over T steps in registers, a
//     log2(LPR)-step prefix scan of the per-lane affine maps (P,h) done with DPP row shifts fused
//     into v_fmac/v_mul (no LDS, no shuffles), and a second serial pass that applies the incoming
//     state and accumulates y_t += C_t h_t.  The per-lane product of the a's is
//     exp2(A * sum(delta)) -- one v_exp instead of T multiplies;
//   * u/delta/z/out travel HBM <-> LDS fully coalesced (wave-private rows, no barrier) whatever the
//     alignment of L (L = 197 / 4097 with the cls token is never a multiple of 4); 16-byte
//     accesses are used when every row start is 16-byte aligned;
//   * state, A, D, bias and every accumulator are fp32; io tensors fp32 / bf16 / fp16.
//
// Algorithmic HBM bytes per launch: elt*(4*B*D*L + 2*B*G*N*L) + 4*(D*N + 2*D).
#include "mxvl_common.h"
#include <type_traits>

namespace mxvl {

constexpr int kCkptLen = 128;  // checkpoint spacing in time steps (mxvl_scan_chunk_len)

struct ScanArgs {
  int batch, dim, L, N, G, n_ckpt;
  int softplus, vec_ok, ablate, dl_ratio, out_f32;
  uint32_t dl_magic, fold_magic;   // fold_magic != 0: the batch is folded into the sequence (scan_fwd_stream.h, FOLD)
  int fold_bpp, fold_cpp;          // batch elements / checkpoint slots per workgroup sequence
  int64_t u_bs, u_ds, dl_bs, dl_ds, z_bs, z_ds, o_bs, o_ds;
  int64_t B_bs, B_gs, B_ns, C_bs, C_gs, C_ns, A_ds, A_ns;
  const void *u, *delta, *B, *C, *z;
  const float *A, *D, *bias;
  void* out;
  float *last_state, *ckpt;
};

// ---- 4-wide global access (row starts 16-byte aligned for fp32, 8-byte for 16-bit types) --------
// ---- inclusive scan of the affine maps (P,h) over the 16 lanes of a DPP row, then the exclusive
// shift.  x enters holding the state that precedes lane 0 (the carry) and leaves holding the
// state that precedes each lane.  Lanes without a source lane are disabled by DPP bound_ctrl=0,
// which is exactly the identity element.  VALU-write -> DPP-read needs 2 wait states: the two
// interleaved chains provide them.
__device__ inline void scan16_x2(float& h0, float& P0, float& x0, float& h1, float& P1, float& x1) {
  asm volatile(
      "s_nop 1\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_mov_b32_dpp %5, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      : "+v"(h0), "+v"(P0), "+v"(x0), "+v"(h1), "+v"(P1), "+v"(x1));
}
__device__ inline void scan16_x1(float& h0, float& P0, float& x0) {
  asm volatile(
      "s_nop 1\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      : "+v"(h0), "+v"(P0), "+v"(x0));
}
// generic form for rows spread over 32 / 64 lanes (few-row problems); not on the headline path
template <int LPR>
__device__ inline void scan_generic(float& hl, float& P, float& x, int j) {
  float pb, pa;
  pb = dpp<DPP_ROW_SHR(1)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(1)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_SHR(2)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(2)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_SHR(4)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(4)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  pb = dpp<DPP_ROW_SHR(8)>(0.0f, hl); pa = dpp<DPP_ROW_SHR(8)>(1.0f, P); hl = fmaf(P, pb, hl); P *= pa;
  if constexpr (LPR >= 32) {
    pb = dpp<DPP_ROW_BCAST15, 0xa>(0.0f, hl); pa = dpp<DPP_ROW_BCAST15, 0xa>(1.0f, P);
    hl = fmaf(P, pb, hl); P *= pa;
  }
  if constexpr (LPR >= 64) {
    pb = dpp<DPP_ROW_BCAST31, 0xc>(0.0f, hl); pa = dpp<DPP_ROW_BCAST31, 0xc>(1.0f, P);
    hl = fmaf(P, pb, hl); P *= pa;
  }
  const float car = x;
  x = dpp<DPP_WAVE_SHR1>(car, hl);
  x = (j == 0) ? car : x;
}

template <typename io_t, int T, int LPR, int NWAVES, int NU, int MINW = 1>
__global__ __launch_bounds__(NWAVES * 64, MINW) void scan_fwd_kernel(const ScanArgs p) {
  constexpr int RPW = 64 / LPR;     // rows per wave
  constexpr int DT = NWAVES * RPW;  // rows per workgroup
  constexpr int CH = LPR * T;       // time steps per chunk
  constexpr int NT = NWAVES * 64;
  static_assert(CH % 64 == 0 && T % 4 == 0, "tile shape");
  static_assert(NT % CH == 0 || CH % NT == 0, "B/C staging shape");
  using io = Io<io_t>;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = p.N, L = p.L;
  float* sB = smem;
  float* sC = sB + N * CH;
  float* sU = sC + N * CH;    // u tile, later the out tile
  float* sD = sU + DT * CH;   // delta tile
  float* sZ = sD + DT * CH;   // z tile
  float2* sAC = (float2*)(sZ + DT * CH);  // [DT][N] {A*log2(e), running state h}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane / LPR, j = lane % LPR;
  const int row = wave * RPW + r;
  const int b = blockIdx.y;
  const int dpg = p.dim / p.G;
  const int tiles = (dpg + DT - 1) / DT;
  const int g = blockIdx.x / tiles;
  const int d0 = g * dpg + (blockIdx.x - g * tiles) * DT;
  const int d_end = (g + 1) * dpg;  // rows of this tile stay inside the B/C group
  const int d = d0 + row;
  const bool row_ok = d < d_end;

  const io_t* __restrict__ up = (const io_t*)p.u + (int64_t)b * p.u_bs;
  const io_t* __restrict__ dp = (const io_t*)p.delta + (int64_t)b * p.dl_bs;
  const io_t* __restrict__ zp = p.z ? (const io_t*)p.z + (int64_t)b * p.z_bs : nullptr;
  const int64_t op = (int64_t)b * p.o_bs;     // element offset into p.out (io dtype, or fp32 with MXVL_SCAN_OUT_F32)
  const bool of32 = p.out_f32 != 0;
  const io_t* __restrict__ Bp = (const io_t*)p.B + (int64_t)b * p.B_bs + (int64_t)g * p.B_gs;
  const io_t* __restrict__ Cp = (const io_t*)p.C + (int64_t)b * p.C_bs + (int64_t)g * p.C_gs;
  const bool has_z = zp != nullptr;
  const bool vec_ok = p.vec_ok != 0;

  for (int i = tid; i < DT * N; i += NT) {
    const int rr = i / N, n = i - rr * N;
    const int dd = d0 + rr;
    sAC[i] = make_float2(dd < d_end ? p.A[(int64_t)dd * p.A_ds + (int64_t)n * p.A_ns] * kLog2e : 0.0f, 0.0f);
  }
  const float bias = (p.bias && row_ok) ? p.bias[delta_row(d, p.dl_ratio, p.dl_magic)] : 0.0f;
  const float Dv = (p.D && row_ok) ? p.D[d] : 0.0f;

  const int nchunks = (L + CH - 1) / CH;
  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * CH;
    const bool full = t0 + CH <= L;
    __syncthreads();  // every wave is done with the previous B/C tile (first pass: sAC visible)
    // ---- stage the shared B/C tile -------------------------------------------------------------
    if (MXVL_ABL(p.ablate & 2)) {
    } else if (vec_ok && full) {
      constexpr int CQ = CH / 4;                 // float4 columns per row
      constexpr int RSTEP = NT / CQ;             // rows covered per pass
      const int e4 = (tid % CQ) * 4;
      const io_t* pb = Bp + t0 + e4 + (int64_t)(tid / CQ) * p.B_ns;
      const io_t* pc = Cp + t0 + e4 + (int64_t)(tid / CQ) * p.C_ns;
      for (int n = tid / CQ; n < N; n += RSTEP) {
        *(float4*)(sB + n * CH + e4) = ld4<io_t>(pb);
        *(float4*)(sC + n * CH + e4) = ld4<io_t>(pc);
        pb += (int64_t)RSTEP * p.B_ns;
        pc += (int64_t)RSTEP * p.C_ns;
      }
    } else {
      constexpr int RSTEP = (NT >= CH) ? NT / CH : 1;
      constexpr int CSTEP = (NT >= CH) ? CH : NT;
      for (int e = tid % CSTEP; e < CH; e += CSTEP) {
        const bool ok = t0 + e < L;
        const io_t* pb = Bp + t0 + e + (int64_t)(tid / CSTEP) * p.B_ns;
        const io_t* pc = Cp + t0 + e + (int64_t)(tid / CSTEP) * p.C_ns;
        for (int n = tid / CSTEP; n < N; n += RSTEP) {
          sB[n * CH + e] = ok ? io::ld(pb) : 0.0f;
          sC[n * CH + e] = ok ? io::ld(pc) : 0.0f;
          pb += (int64_t)RSTEP * p.B_ns;
          pc += (int64_t)RSTEP * p.C_ns;
        }
      }
    }
    // ---- stage this wave's own rows of u / delta / z -------------------------------------------
    if (MXVL_ABL(p.ablate & 4)) {
    } else if (vec_ok && full) {
      constexpr int CQ = CH / 4;  // float4 columns per row; a wave owns RPW*CQ = 16*T of them
#pragma unroll
      for (int q = lane; q < RPW * CQ; q += 64) {
        const int rr = q / CQ, e4 = (q % CQ) * 4;
        const int wrow = wave * RPW + rr;
        const int dd = d0 + wrow;
        if (dd < d_end) {
          *(float4*)(sU + wrow * CH + e4) = ld4<io_t>(up + (int64_t)dd * p.u_ds + t0 + e4);
          *(float4*)(sD + wrow * CH + e4) = ld4<io_t>(dp + (int64_t)delta_row(dd, p.dl_ratio, p.dl_magic) * p.dl_ds + t0 + e4);
          if (has_z) *(float4*)(sZ + wrow * CH + e4) = ld4<io_t>(zp + (int64_t)dd * p.z_ds + t0 + e4);
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int wrow = wave * RPW + rr;
        const int dd = d0 + wrow;
        const io_t* pu = up + (int64_t)dd * p.u_ds + t0;
        const io_t* pd = dp + (int64_t)delta_row(dd, p.dl_ratio, p.dl_magic) * p.dl_ds + t0;
        const io_t* pz = has_z ? zp + (int64_t)dd * p.z_ds + t0 : nullptr;
#pragma unroll
        for (int e = lane; e < CH; e += 64) {
          const bool ok = dd < d_end && t0 + e < L;
          sU[wrow * CH + e] = ok ? io::ld(pu + e) : 0.0f;
          sD[wrow * CH + e] = ok ? io::ld(pd + e) : 0.0f;
          if (has_z) sZ[wrow * CH + e] = ok ? io::ld(pz + e) : 0.0f;
        }
      }
    }
    __syncthreads();

    float dl[T], du[T], y[T];
    {
      const float4* su4 = (const float4*)(sU + row * CH + j * T);
      const float4* sd4 = (const float4*)(sD + row * CH + j * T);
#pragma unroll
      for (int q = 0; q < T / 4; ++q) {
        const float4 uu = su4[q], dd4 = sd4[q];
        const float uv[4] = {uu.x, uu.y, uu.z, uu.w};
        const float dv[4] = {dd4.x, dd4.y, dd4.z, dd4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = q * 4 + k;
          float x = dv[k] + bias;
          if (p.softplus) x = softplus(x);
          if (!full) x = (t0 + j * T + i < L) ? x : 0.0f;  // padding steps are the identity map
          dl[i] = x;
          du[i] = x * uv[k];
          y[i] = Dv * uv[k];
        }
      }
    }
    float dsum = 0.0f;
#pragma unroll
    for (int i = 0; i < T; ++i) dsum += dl[i];

    const bool ckpt_here = p.ckpt != nullptr && row_ok && ((t0 + j * T) % kCkptLen == 0) && (t0 + j * T < L);
    float* ckpt_row = p.ckpt ? p.ckpt + (((int64_t)b * p.dim + d) * p.n_ckpt + (t0 + j * T) / kCkptLen) * N : nullptr;
    float2* ac = sAC + row * N;

    for (int n0 = 0; n0 < (MXVL_ABL(p.ablate & 1) ? 0 : N); n0 += NU) {
      float a[NU][T], bb[NU][T], cv[NU][T], hl[NU], P[NU], x[NU];
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        const int n = n0 + k;
        const float2 A2c = ac[n];
        const float4* sb4 = (const float4*)(sB + n * CH + j * T);
        const float4* sc4 = (const float4*)(sC + n * CH + j * T);
#pragma unroll
        for (int q = 0; q < T / 4; ++q) {
          const float4 b4 = sb4[q], c4 = sc4[q];
          bb[k][q * 4 + 0] = b4.x; bb[k][q * 4 + 1] = b4.y; bb[k][q * 4 + 2] = b4.z; bb[k][q * 4 + 3] = b4.w;
          cv[k][q * 4 + 0] = c4.x; cv[k][q * 4 + 1] = c4.y; cv[k][q * 4 + 2] = c4.z; cv[k][q * 4 + 3] = c4.w;
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
          a[k][i] = fast_exp2(dl[i] * A2c.x);
          bb[k][i] = du[i] * bb[k][i];
        }
        // pass 1: the lane's affine map h_out = P * h_in + hl
        float h = bb[k][0];
#pragma unroll
        for (int i = 1; i < T; ++i) h = fmaf(a[k][i], h, bb[k][i]);
        P[k] = fast_exp2(A2c.x * dsum);
        x[k] = A2c.y;                                       // state entering the chunk
        hl[k] = fmaf(P[k], (j == 0) ? A2c.y : 0.0f, h);     // lane 0 absorbs it
      }
      if constexpr (LPR == 16 && NU == 2) {
        scan16_x2(hl[0], P[0], x[0], hl[1], P[1], x[1]);
      } else if constexpr (LPR == 16) {
#pragma unroll
        for (int k = 0; k < NU; ++k) scan16_x1(hl[k], P[k], x[k]);
      } else {
#pragma unroll
        for (int k = 0; k < NU; ++k) scan_generic<LPR>(hl[k], P[k], x[k], j);
      }
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        const int n = n0 + k;
        if (j == LPR - 1) ac[n].y = hl[k];  // state leaving the chunk
        if (ckpt_here) ckpt_row[n] = x[k];
        float h = x[k];
#pragma unroll
        for (int i = 0; i < T; ++i) {  // pass 2
          h = fmaf(a[k][i], h, bb[k][i]);
          y[i] = fmaf(cv[k][i], h, y[i]);
        }
      }
    }

    if (has_z) {
      const float4* sz4 = (const float4*)(sZ + row * CH + j * T);
#pragma unroll
      for (int q = 0; q < T / 4; ++q) {
        const float4 z4 = sz4[q];
        y[q * 4 + 0] *= silu(z4.x); y[q * 4 + 1] *= silu(z4.y);
        y[q * 4 + 2] *= silu(z4.z); y[q * 4 + 3] *= silu(z4.w);
      }
    }
    // out tile through the (wave-private) u tile, then coalesced to HBM
    {
      float4* so4 = (float4*)(sU + row * CH + j * T);
#pragma unroll
      for (int q = 0; q < T / 4; ++q) so4[q] = make_float4(y[q * 4], y[q * 4 + 1], y[q * 4 + 2], y[q * 4 + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (MXVL_ABL(p.ablate & 8)) {
    } else if (vec_ok && full) {
      constexpr int CQ = CH / 4;
#pragma unroll
      for (int q = lane; q < RPW * CQ; q += 64) {
        const int rr = q / CQ, e4 = (q % CQ) * 4;
        const int wrow = wave * RPW + rr;
        const int dd = d0 + wrow;
        if (dd < d_end) st4_out<io_t>(p.out, op + (int64_t)dd * p.o_ds + t0 + e4, *(const float4*)(sU + wrow * CH + e4), of32);
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int wrow = wave * RPW + rr;
        const int dd = d0 + wrow;
        const int64_t po = op + (int64_t)dd * p.o_ds + t0;
#pragma unroll
        for (int e = lane; e < CH; e += 64)
          if (dd < d_end && t0 + e < L) st_out<io_t>(p.out, po + e, sU[wrow * CH + e], of32);
      }
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
#include "scan_fwd_stream.h"
#include "scan_n1.h"
#include "scan_n1_short.h"
namespace mxvl {

// ---------------------------------------------------------------------------------------------
static thread_local int g_last_hip_error = 0;
static thread_local const char* g_last_kernel = "none";
static thread_local int g_variant = 0;   // per calling thread: a test / bench hook, not process-global state

template <typename io_t, int T, int LPR, int NWAVES, int NU, int MINW = 1>
static int launch_fwd(const ScanArgs& a, hipStream_t stream, const char* name) {
  constexpr int RPW = 64 / LPR, DT = NWAVES * RPW, CH = LPR * T;
  const size_t lds = sizeof(float) * ((size_t)2 * a.N * CH + (size_t)3 * DT * CH + (size_t)2 * DT * a.N);
  if (lds > 160 * 1024) return MXVL_ERR_UNSUPPORTED;
  auto kern = scan_fwd_kernel<io_t, T, LPR, NWAVES, NU, MINW>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  }
  const int dpg = a.dim / a.G;
  dim3 grid(a.G * ((dpg + DT - 1) / DT), a.batch), block(NWAVES * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  hipError_t e = hipGetLastError();
  g_last_kernel = name;
  if (e != hipSuccess) { g_last_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}

template <typename io_t, int NWAVES, bool VEC, int MINW, int T, int NS, bool FOLD = false, bool PK = false>
static int launch_stream1(const ScanArgs& a, hipStream_t stream, const char* name) {
  constexpr int CH = 128, DT = NWAVES * (64 / (CH / T));
  // PK: sA [DT][18] + sH [DT + 1][18] + a dump pair per thread in the place of the float2 table (scan_fwd_stream.h)
  const size_t tail = PK ? (size_t)(2 * DT + 1) * 18 + 2 * NWAVES * 64 + a.N : (size_t)2 * (DT + 1) * (a.N + 1) + NWAVES * 64 + 2 * a.N;
  const size_t lds = sizeof(float) * ((size_t)4 * a.N * CH + (size_t)DT * CH + tail);
  auto kern = scan_fwd_stream_kernel<io_t, NWAVES, VEC, MINW, T, NS, FOLD, PK>;
  const int dpg = a.dim / a.G;
  dim3 grid(a.G * ((dpg + DT - 1) / DT), FOLD ? (a.batch + a.fold_bpp - 1) / a.fold_bpp : a.batch), block(NWAVES * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  hipError_t e = hipGetLastError();
  g_last_kernel = name;
  if (e != hipSuccess) { g_last_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}
// dstate 16 (every Mamba block of the reference) takes the instantiation with the compile-time state count
template <typename io_t, int NWAVES, bool VEC, int MINW, int T = 8>
static int launch_stream(const ScanArgs& a, hipStream_t stream, const char* name) {
  return a.N == 16 ? launch_stream1<io_t, NWAVES, VEC, MINW, T, 16>(a, stream, name)
                   : launch_stream1<io_t, NWAVES, VEC, MINW, T, 0>(a, stream, name);
}
#define MXVL_STREAM_CASE(NW, MW) \
  (a.vec_ok ? launch_stream<io_t, NW, true, MW>(a, stream, "scan_fwd_stream<W" #NW ",vec,occ" #MW ">") \
            : launch_stream<io_t, NW, false, MW>(a, stream, "scan_fwd_stream<W" #NW ",scalar,occ" #MW ">"))
// state pairs in packed fp32 (scan_fwd_stream.h, PK): dstate 16, aligned rows
#define MXVL_STREAM_PK(NW, MW) \
  launch_stream1<io_t, NW, true, MW, 8, 16, false, true>(a, stream, "scan_fwd_stream<W" #NW ",vec,occ" #MW ",pk>")
#define MXVL_FWD_CASE(T, LPR, NW, NU) \
  launch_fwd<io_t, T, LPR, NW, NU>(a, stream, "scan_fwd<T" #T ",LPR" #LPR ",W" #NW ",NU" #NU ">")
#define MXVL_FWD_CASE_OCC(T, LPR, NW, NU, MW) \
  launch_fwd<io_t, T, LPR, NW, NU, MW>(a, stream, "scan_fwd<T" #T ",LPR" #LPR ",W" #NW ",NU" #NU ",occ" #MW ">")

// dstate 1 without z (VMamba's SS2D): the flat-row kernel of scan_n1.h.  Rows need T-element alignment (T = 8 for 16-bit rows with
// L % 8 == 0, else 4); a wave's flat range (rw * L elements) and the magic divisions stay inside 32 bits.
template <typename io_t>
static int try_n1_fwd(const ScanArgs& a, hipStream_t stream, bool& taken) {
  taken = false;
  if (a.N != 1 || a.z || a.fold_magic || (g_variant & 0xff) == 30) return MXVL_OK;      // variant 30: A/B hook, the general kernels
  const int64_t rows = (int64_t)a.batch * a.dim;
  const int dpg = a.dim / a.G;
  if (a.L < 8 || a.L % 4 != 0 || a.L > (1 << 20) || (int64_t)a.dim * dpg >= (1ll << 32)) return MXVL_OK;
  constexpr int esz = (int)sizeof(io_t);
  auto aligned = [&](int T) {
    if (a.L % T != 0) return false;
    for (int64_t s : {a.u_bs, a.u_ds, a.dl_bs, a.dl_ds, a.B_bs, a.B_gs, a.C_bs, a.C_gs})
      if (s % T != 0) return false;
    for (const void* q : {a.u, a.delta, a.B, a.C})
      if (((uintptr_t)q) % (size_t)(T * esz) != 0) return false;
    return true;
  };
  const int T = (esz == 2 && aligned(8)) ? 8 : (aligned(4) ? 4 : 0);
  if (T == 0) return MXVL_OK;
  ScanN1Geom gm;
  // ~8 waves per SIMD of work items, at least 8 passes per wave where the problem is large enough; rw * L * L < 2^32 (magL exact)
  int64_t rw = (rows + 8191) / 8192;
  const int64_t min_rw = (8ll * 64 * T + a.L - 1) / a.L;
  if (rw < min_rw) rw = min_rw;
  while (rw > 1 && rw * a.L * (int64_t)a.L >= (1ll << 32)) --rw;
  if (rw * a.L * (int64_t)a.L >= (1ll << 32) || rw * a.L >= (1ll << 30)) return MXVL_OK;
  if (rw > rows) rw = rows;
  gm.rw = (int)rw;
  gm.n_waves = (int)((rows + rw - 1) / rw);
  gm.magL = (uint32_t)((1ull << 32) / (uint64_t)a.L + 1ull);
  gm.magG = (uint32_t)((1ull << 32) / (uint64_t)dpg + 1ull);
  const int oesz = a.out_f32 ? 4 : esz;
  gm.out_vec = (a.o_bs % 4 == 0 && a.o_ds % 4 == 0 && ((uintptr_t)a.out) % (size_t)(4 * oesz) == 0) ? 1 : 0;
  constexpr int NW = 4;
  const dim3 grid((gm.n_waves + NW - 1) / NW), block(NW * 64);
  if constexpr (esz == 2) {
    if (T == 8) hipLaunchKernelGGL((scan_n1_fwd_kernel<io_t, 8, NW>), grid, block, 0, stream, a, gm);
    else hipLaunchKernelGGL((scan_n1_fwd_kernel<io_t, 4, NW>), grid, block, 0, stream, a, gm);
    g_last_kernel = T == 8 ? "scan_n1_fwd<T8>" : "scan_n1_fwd<T4>";
  } else {
    hipLaunchKernelGGL((scan_n1_fwd_kernel<io_t, 4, NW>), grid, block, 0, stream, a, gm);
    g_last_kernel = "scan_n1_fwd<T4>";
  }
  taken = true;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_last_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}

// dstate 1, rows of at most 128 steps that scan_n1.h cannot take (L % 4 != 0 -- VMamba's 7 x 7 stage -- or unaligned rows): a lane per
// row (scan_n1_short.h).  u / delta / out must be dense (batch, dim, L) arrays, 64 | dim / n_groups.
template <typename io_t>
static int try_n1_short_fwd(const ScanArgs& a, hipStream_t stream, bool& taken) {
  taken = false;
  if (a.N != 1 || a.z || a.fold_magic || (g_variant & 0xff) == 30) return MXVL_OK;
  const int dpg = a.dim / a.G;
  const int64_t DL = (int64_t)a.dim * a.L;
  if (a.L < 1 || a.L > 128 || dpg % 64 != 0 || a.dl_ratio > 1) return MXVL_OK;
  if (a.u_ds != a.L || a.u_bs != DL || a.dl_ds != a.L || a.dl_bs != DL || a.o_ds != a.L || a.o_bs != DL) return MXVL_OK;
  for (const void* q : {a.u, a.delta, (const void*)a.out})
    if (((uintptr_t)q) % 16 != 0) return MXVL_OK;
  constexpr int esz = (int)sizeof(io_t);
  const int oesz = a.out_f32 ? 4 : esz;
  const size_t per_wave = (size_t)64 * a.L * (2 * esz + oesz) + (size_t)2 * ((a.L + 3) & ~3) * sizeof(float);
  if (per_wave > 64 * 1024) return MXVL_OK;
  const int64_t waves = (int64_t)a.batch * a.dim / 64;
  const bool two = 2 * per_wave <= 64 * 1024;
  const dim3 grid((unsigned)(two ? (waves + 1) / 2 : waves)), block(two ? 128 : 64);
  const size_t lds = per_wave * (two ? 2 : 1);
  if (a.out_f32 && esz == 2) {
    if (two) hipLaunchKernelGGL((scan_n1_short_fwd_kernel<io_t, 2, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((scan_n1_short_fwd_kernel<io_t, 1, true>), grid, block, lds, stream, a);
  } else {
    if (two) hipLaunchKernelGGL((scan_n1_short_fwd_kernel<io_t, 2, false>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((scan_n1_short_fwd_kernel<io_t, 1, false>), grid, block, lds, stream, a);
  }
  g_last_kernel = "scan_n1_short_fwd";
  taken = true;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_last_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}

template <typename io_t>
static int dispatch_fwd(const ScanArgs& a, hipStream_t stream) {
  {
    bool taken = false;
    int rc = try_n1_fwd<io_t>(a, stream, taken);
    if (rc != MXVL_OK || taken) return rc;
    rc = try_n1_short_fwd<io_t>(a, stream, taken);
    if (rc != MXVL_OK || taken) return rc;
  }
  if (a.fold_magic) {   // batch folded into the sequence: aligned rows, dstate 16 (the caller asked with MXVL_SCAN_FOLD_BATCH)
    if (!a.vec_ok || a.N != 16) return MXVL_ERR_UNSUPPORTED;
    for (int64_t bs : {a.u_bs, a.dl_bs, a.z_bs, a.o_bs, a.B_bs, a.C_bs})
      if (bs >= (1ll << 32)) return MXVL_ERR_UNSUPPORTED;    // seg_off: 32-bit batch strides
    // the same workgroup-shape rule as the plain launch, with (parts of the batch) in the place of (batch elements)
    const int64_t tiles8 = (int64_t)((a.batch + a.fold_bpp - 1) / a.fold_bpp) * a.G * ((a.dim / a.G + 31) / 32);
    const int fv = g_variant & 0xff;
    // packed state pairs (scan_fwd_stream.h PK): the automatic pick for 16-bit rows -- B64 x D4096 x L200 bf16 (the 4-direction encoder):
    // 279 -> 247 us (profiles/r06_scan_fwd_exp_ab.txt), arm_encoder_large_224 823-826 -> 828-832 images/s, stage-3 step 54.2-54.9 ->
    // 55.0-55.1 studies/s (profiles/r06_fold_pk_step_ab.txt); variant 14 forces the unpacked walk
    if ((fv >= 15 && fv <= 18) || (fv == 0 && sizeof(io_t) == 2)) {
      if (tiles8 >= 512) return launch_stream1<io_t, 8, true, 2, 8, 16, true, true>(a, stream, "scan_fwd_stream<W8,vec,occ2,fold,pk>");
      return launch_stream1<io_t, 4, true, 2, 8, 16, true, true>(a, stream, "scan_fwd_stream<W4,vec,occ2,fold,pk>");
    }
    if (tiles8 >= 512) return launch_stream1<io_t, 8, true, 3, 8, 16, true>(a, stream, "scan_fwd_stream<W8,vec,occ3,fold>");
    return launch_stream1<io_t, 4, true, 2, 8, 16, true>(a, stream, "scan_fwd_stream<W4,vec,occ2,fold>");
  }
  const bool even = (a.N % 2) == 0;
  const int64_t rows = (int64_t)a.batch * a.dim;
  int v = g_variant & 0xff;
  if (v == 0) {
    // 16 lanes per row needs rows/4 waves: use wider rows when that cannot fill the 1024 SIMDs
    if (rows >= 4096 || a.L <= 128) {
      if (a.N <= 16) {
        // measured (profiles/r03_scan_variants.txt): 32-row / 8-wave workgroups win when they still give every CU two rounds of
        // work on long rows (pre-training shape 342 vs 414 us) or many rounds on short ones (197-token encoder 322 vs 382 us);
        // otherwise 16-row workgroups at 2 waves per SIMD (fp32 roofline shape 260 vs 286 us at 3 waves per SIMD)
        const int64_t tiles8 = (int64_t)a.batch * a.G * ((a.dim / a.G + 31) / 32);
        v = (tiles8 >= 2048 || (tiles8 >= 512 && a.L >= 1024)) ? 12 : 14;
      } else {
        v = 1;
      }
    } else if (rows >= 1024 || a.L <= 256) v = 3;
    else v = 4;
  }
  if (a.N == 16 && a.vec_ok) {
    switch (v) {
      case 15: return MXVL_STREAM_PK(4, 2);
      case 16: return MXVL_STREAM_PK(8, 3);
      case 17: return MXVL_STREAM_PK(4, 3);
      case 18: return MXVL_STREAM_PK(8, 2);
      default: break;
    }
  }
  if (a.N <= 16) {
    switch (v) {
      case 10: return MXVL_STREAM_CASE(4, 3);
      case 12: return MXVL_STREAM_CASE(8, 3);
      case 14: return MXVL_STREAM_CASE(4, 2);
      case 20: return MXVL_STREAM_CASE(8, 2);
      default: break;
    }
  }
  switch (v) {
    case 1: return even ? MXVL_FWD_CASE(8, 16, 4, 2) : MXVL_FWD_CASE(8, 16, 4, 1);   // CH=128, 16 rows
    case 2: return even ? MXVL_FWD_CASE(8, 16, 4, 1) : MXVL_FWD_CASE(8, 16, 4, 1);   // same, one state at a time
    case 3: return even ? MXVL_FWD_CASE(8, 32, 4, 2) : MXVL_FWD_CASE(8, 32, 4, 1);   // CH=256, 8 rows
    case 4: return even ? MXVL_FWD_CASE(8, 64, 4, 2) : MXVL_FWD_CASE(8, 64, 4, 1);   // CH=512, 4 rows
    case 5: return even ? MXVL_FWD_CASE(8, 16, 2, 2) : MXVL_FWD_CASE(8, 16, 2, 1);   // CH=128, 8 rows
    case 6: return even ? MXVL_FWD_CASE(8, 16, 8, 2) : MXVL_FWD_CASE(8, 16, 8, 1);   // CH=128, 32 rows
    case 8: return MXVL_FWD_CASE_OCC(8, 16, 4, 1, 4);                                  // <=128 VGPR
    case 9: return even ? MXVL_FWD_CASE_OCC(8, 16, 4, 2, 3) : MXVL_FWD_CASE_OCC(8, 16, 4, 1, 4);
    case 7: return even ? MXVL_FWD_CASE(16, 16, 4, 2) : MXVL_FWD_CASE(16, 16, 4, 1); // CH=256, 16 rows
    default: break;
  }
  return even ? MXVL_FWD_CASE(8, 16, 1, 2) : MXVL_FWD_CASE(8, 16, 1, 1);
}

}  // namespace mxvl

using namespace mxvl;

extern "C" {

int mxvl_abi_version(void) { return MXVL_ABI_VERSION; }
int mxvl_scan_chunk_len(int, int) { return kCkptLen; }
int mxvl_scan_n_chunks(int seqlen, int) { return (seqlen + kCkptLen - 1) / kCkptLen; }
// 1: rows this short leave enough of their last 128-step chunk empty that MXVL_SCAN_FOLD_BATCH pays (and the kernels can do it)
int mxvl_scan_fold_ok(int batch, int seqlen, int dstate) {
  if (batch < 2 || seqlen < 8 || seqlen % 8 != 0 || dstate != 16) return 0;
  if ((uint64_t)batch * (uint64_t)seqlen * (uint64_t)seqlen >= (1ull << 32)) return 0;   // multiply-high division by seqlen stays exact
  const int padded = (seqlen + kCkptLen - 1) / kCkptLen * kCkptLen;
  // >= 12.5 % of the computed steps are padding, or one-chunk rows (a workgroup per row is all prologue: at 128 steps the folded
  // backward is 16 % faster and the folded forward 25 % slower, -7 % for the pair; from 256 steps on the pair does not gain,
  // profiles/r03_scan_fold.txt)
  return (padded - seqlen) * 8 >= padded || seqlen <= 128;
}
// checkpoint slots per channel of a folded call: ckpt = (dim, slots, dstate) fp32
int mxvl_scan_fold_slots(int batch, int seqlen, int dim, int n_groups) {
  if (batch <= 0 || seqlen <= 0 || dim <= 0 || n_groups <= 0 || dim % n_groups) return 0;
  const int bpp = scan_fold_bpp(batch, seqlen, dim, n_groups);
  return ((batch + bpp - 1) / bpp) * ((bpp * seqlen + kCkptLen - 1) / kCkptLen);
}
int mxvl_last_hip_error(void) { return g_last_hip_error; }
void mxvl_set_scan_variant(int v) { g_variant = MXVL_ABL(true) ? v : (v & 0xffff); }
int mxvl_scan_bwd_variant(void) { return (g_variant >> 8) & 0xff; }
const char* mxvl_last_scan_kernel(void) { return g_last_kernel; }

int mxvl_scan_check(const mxvl_scan_desc* d) {
  if (!d) return MXVL_ERR_NULL;
  if (!d->u || !d->delta || !d->A || !d->B || !d->C) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_F32 && d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->batch <= 0 || d->dim <= 0 || d->seqlen <= 0 || d->dstate <= 0 || d->n_groups <= 0) return MXVL_ERR_SHAPE;
  if (d->dim % d->n_groups != 0) return MXVL_ERR_SHAPE;
  if (d->delta_group_ratio < 0 || (d->delta_group_ratio > 1 && d->dim % d->delta_group_ratio != 0)) return MXVL_ERR_SHAPE;
  if (d->delta_group_ratio > 1 && (uint64_t)d->dim * (uint64_t)d->delta_group_ratio >= (1ull << 32)) return MXVL_ERR_UNSUPPORTED;
  if (d->dstate > MXVL_MAX_DSTATE) return MXVL_ERR_DSTATE;
  const int64_t s[] = {d->u_bs, d->u_ds, d->delta_bs, d->delta_ds, d->B_bs, d->B_gs, d->B_ns,
                       d->C_bs, d->C_gs, d->C_ns, d->A_ds, d->A_ns};
  for (int64_t v : s) if (v < 0) return MXVL_ERR_STRIDE;
  if (d->z && (d->z_bs < 0 || d->z_ds < 0)) return MXVL_ERR_STRIDE;
  return MXVL_OK;
}

int mxvl_scan_fwd(const mxvl_scan_desc* d, void* hip_stream) {
  int rc = mxvl_scan_check(d);
  if (rc != MXVL_OK) return rc;
  if (!d->out) return MXVL_ERR_NULL;
  if (d->out_bs < 0 || d->out_ds < 0) return MXVL_ERR_STRIDE;
  ScanArgs a;
  a.batch = d->batch; a.dim = d->dim; a.L = d->seqlen; a.N = d->dstate; a.G = d->n_groups;
  a.n_ckpt = (d->seqlen + kCkptLen - 1) / kCkptLen;
  a.softplus = (d->flags & MXVL_SCAN_DELTA_SOFTPLUS) ? 1 : 0;
  a.out_f32 = (d->flags & MXVL_SCAN_OUT_F32) ? 1 : 0;
  a.u_bs = d->u_bs; a.u_ds = d->u_ds; a.dl_bs = d->delta_bs; a.dl_ds = d->delta_ds;
  a.z_bs = d->z_bs; a.z_ds = d->z_ds; a.o_bs = d->out_bs; a.o_ds = d->out_ds;
  a.B_bs = d->B_bs; a.B_gs = d->B_gs; a.B_ns = d->B_ns;
  a.C_bs = d->C_bs; a.C_gs = d->C_gs; a.C_ns = d->C_ns; a.A_ds = d->A_ds; a.A_ns = d->A_ns;
  a.u = d->u; a.delta = d->delta; a.B = d->B; a.C = d->C; a.z = d->z;
  a.A = (const float*)d->A; a.D = (const float*)d->D; a.bias = (const float*)d->delta_bias;
  a.out = d->out; a.last_state = (float*)d->last_state; a.ckpt = (float*)d->ckpt;
  a.dl_ratio = d->delta_group_ratio > 1 ? d->delta_group_ratio : 1;
  a.dl_magic = delta_magic(a.dl_ratio);
  a.ablate = MXVL_ABL(true) ? (g_variant >> 16) & 0xfff : 0;
  a.fold_magic = 0; a.fold_bpp = 0; a.fold_cpp = 0;
  if (d->flags & MXVL_SCAN_FOLD_BATCH) {
    if (!mxvl_scan_fold_ok(d->batch, d->seqlen, d->dstate) || d->last_state) return MXVL_ERR_UNSUPPORTED;
    a.fold_magic = scan_fold_magic(d->seqlen);
    a.fold_bpp = scan_fold_bpp(d->batch, d->seqlen, d->dim, d->n_groups);
    a.fold_cpp = (a.fold_bpp * d->seqlen + kCkptLen - 1) / kCkptLen;
  }
  // 4-element vector access is legal when every row of every io tensor starts on a 4-element boundary
  {
    const int64_t esz = d->io_dtype == MXVL_F32 ? 4 : 2;
    const int64_t strides[] = {d->u_bs, d->u_ds, d->delta_bs, d->delta_ds, d->out_bs, d->out_ds,
                               d->B_bs, d->B_gs, d->B_ns, d->C_bs, d->C_gs, d->C_ns,
                               d->z ? d->z_bs : 0, d->z ? d->z_ds : 0};
    bool ok = true;
    for (int64_t s : strides) ok = ok && (s % 4 == 0);
    const void* ptrs[] = {d->u, d->delta, d->B, d->C, d->z};
    for (const void* q : ptrs) ok = ok && (((uintptr_t)q) % (4 * esz) == 0);
    ok = ok && (((uintptr_t)d->out) % (4 * (a.out_f32 ? 4 : esz)) == 0);
    // 16-bit rows: the vector kernels read u / delta / z eight steps (16 bytes) at a time (scan_fwd_stream.h raw_row, and every
    // folded walk): those rows must start on 16-byte boundaries -- the same rule mxvl_scan_bwd applies.  A view that is only
    // 8-byte aligned takes the element-wise kernels (and MXVL_SCAN_FOLD_BATCH is refused for it, as mxvl.h documents).
    if (ok && esz == 2) {
      const int64_t s8[] = {d->u_bs, d->u_ds, d->delta_bs, d->delta_ds, d->z ? d->z_bs : 0, d->z ? d->z_ds : 0};
      for (int64_t s : s8) ok = ok && (s % 8 == 0);
      const void* p16[] = {d->u, d->delta, d->z};
      for (const void* q : p16) ok = ok && (((uintptr_t)q) % 16 == 0);
    }
    a.vec_ok = ok ? 1 : 0;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  switch (d->io_dtype) {
    case MXVL_F32: return dispatch_fwd<float>(a, stream);
    case MXVL_BF16: return dispatch_fwd<bf16_t>(a, stream);
    default: return dispatch_fwd<f16_t>(a, stream);
  }
}

}  // extern "C"
