// scan_fwd_stream.h -- the streaming selective-scan forward kernel (headline path, dstate <= 16).
//
// Same math as scan_fwd_kernel (scan_fwd.hip) but organised so HBM traffic overlaps the VALU work.  A row is spread
// over LPR lanes of a DPP row, every lane owns T = 128/LPR consecutive steps; (T, LPR) = (8, 16) or (16, 8):
//   * u / delta of chunk c+1 are loaded straight into registers (each lane owns T contiguous elements per row:
//     16-byte loads when rows are 16-byte aligned, dword loads otherwise) while chunk c is being scanned; z of
//     chunk c is requested before the state loop and consumed after it;
//   * the B/C tile of chunk c+1 is fetched by the whole workgroup during chunk c and written into the other half
//     of a double-buffered LDS tile after the state loop: one barrier per chunk.  The tile is stored
//     "quarter-major" ([T/4][LPR][4] per row) so the per-lane 16-byte reads of a DPP row are conflict-free, with the odd
//     quarter rotated by 16 words so the staging writes are too (see qpos);
//   * the prefix scan over the LPR lanes is fused-DPP (v_fmac/v_mul with row_shr).  For LPR = 8 two rows share a
//     16-lane DPP row: the first lane of every row scans with P = 0 (after absorbing the incoming state), which
//     cuts every contribution that would cross the row boundary -- no exec masking needed;
//   * out leaves as 16-byte stores (aligned rows) or through a wave-private LDS tile that turns the lane-strided
//     layout back into coalesced dword stores (unaligned rows).
#pragma once
#include "mxvl_common.h"

namespace mxvl {

// 3-step scan for rows of 8 lanes (two rows per DPP row); x leaves holding the inclusive value of the lane to the left
__device__ inline void scan8_x1(float& h0, float& P0, float& x0) {
  asm volatile(
      "s_nop 1\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      : "+v"(h0), "+v"(P0), "+v"(x0));
}

// NS > 0: dstate is the compile-time constant NS (16 = every Mamba block of the reference): the state loop unrolls with
// immediate LDS offsets and the per-chunk index arithmetic (divisions by N) folds; NS == 0: any dstate <= 16 at run time.
//
// FOLD: the batch is folded into the sequence.  Rows of L <= a few hundred steps leave the second 128-step chunk mostly empty
// (197-token encoders, padded to 200: 56 of 256 computed steps are padding; 144-token rows: 112 of 256), and a workgroup per
// (batch element, channel tile) is all prologue.  With FOLD a workgroup owns a channel tile and walks the VIRTUAL sequence
// tv = b * L + l of all batch elements: chunks are always full (but the last one), the per-chunk tables and A are set up once.
// A recurrence must not run across a segment boundary: the first step of every segment gets a_t = 0 (exp2(-inf)) -- in the
// lane product P as well -- which is exactly "state 0 enters the segment" (h = 0 * h_in + b).  L % 8 == 0, so a lane's 8 steps
// and a staging quarter never straddle two segments; (b, l) of a step come from one multiply-high (fold_magic = 2^32 / L + 1).
// Checkpoints are indexed by virtual chunk in the same buffer (batch * chunks(L) >= chunks(batch * L) entries per channel).
//
// PK (round 6): the state loop walks state PAIRS (n, n + 1) in the two halves of 64-bit register pairs, so every fp32 multiply / FMA of
// the recurrence -- delta_i * A, delta_i u_i * B_i, both Horner passes, the C_i h_i accumulation -- is ONE v_pk_mul_f32 / v_pk_fma_f32 for
// two states (measured on this chip at 2 waves per SIMD: 4.5 cycles against 2 x 2.8, profiles/r01_ubench_valu_rates.txt); the v_exp_f32
// and the DPP steps stay per state (no packed forms exist).  The same IEEE operations per state in the same order: h_t is bit-identical
// to the unpacked kernel's; y sums the even and the odd states separately and adds the two halves at the end of the chunk.  For that the
// B / C tile is stored pair-interleaved -- tile row p holds states (2p, 2p + 1): [4 blocks of 2 steps][LPR lanes][step s, state] so that
// one ds_read_b128 yields (B_n[i], B_n+1[i], B_n[i+1], B_n+1[i+1]) -- and A * log2(e) and the running state live in two separate
// [rows][N + 2] arrays (an (n, n + 1) pair is one aligned ds_read_b64 / ds_write_b64).

template <typename io_t, int NWAVES, bool VEC, int MINW, int T = 8, int NS = 0, bool FOLD = false, bool PK = false>
__global__ __launch_bounds__(NWAVES * 64, MINW) void scan_fwd_stream_kernel(const ScanArgs p) {
  constexpr int CH = 128, LPR = CH / T, RPW = 64 / LPR, DT = NWAVES * RPW, NT = NWAVES * 64, NMAX = 16;
  static_assert(!PK || (VEC && T == 8 && NS > 0 && NS % 2 == 0 && (NS * 32) % NT == 0), "PK: aligned rows, 16 lanes per row, even compile-time dstate");
  constexpr int TQ = T / 4;                           // 16-byte quarters per lane
  constexpr int BCV = (NMAX * CH / 4 + NT - 1) / NT;  // float4 per thread per array (VEC)
  constexpr int BCS = (NMAX * CH + NT - 1) / NT;      // floats per thread per array (scalar)
  static_assert(T == 8 || T == 16, "lane span");
  static_assert(NT >= CH && NT % CH == 0 && NT % (CH / 4) == 0, "staging shape");
  using io = Io<io_t>;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  // SL: segment (row) length; FOLD: this workgroup walks batch elements [b0, b0 + nb) as ONE sequence of L = nb * SL steps
  const int N = NS > 0 ? NS : p.N, SL = p.L;
  const int b0 = FOLD ? (int)blockIdx.y * p.fold_bpp : 0;
  const int L = FOLD ? (p.batch - b0 < p.fold_bpp ? p.batch - b0 : p.fold_bpp) * SL : SL;
  float* sBC = smem;                          // [2 buffers][B|C][N][CH]
  float* sO = sBC + 4 * N * CH;               // [DT][CH] out tile (unaligned rows / ragged tail)
  float2* sAC = (float2*)(sO + DT * CH);      // [DT + 2][NP] {A*log2(e), running state h}; row DT stays zero, row DT + 1.. = dump
  // rows are padded by one float2: with a stride of 2N = 32 words every row's state n sat on the same bank, and the per-state
  // b32 accesses of the 4 rows of a wave (+ the zero row) were 2-3-way conflicts (half of SQ_LDS_BANK_CONFLICT)
  const int NP = N + 1;
  float* sDump = (float*)(sAC + (DT + 1) * NP);   // [NT + 2 N] write-only words of the lanes that do not own a state entry
  // PK: the same region as sA [DT][SA] (A * log2 e), sH [DT + 1][SA] (running state; row DT stays zero), sDump2 [2 NT + N]
  constexpr int SA = NMAX + 2;
  float* sA = (float*)sAC;
  float* sH = sA + DT * SA;
  float* sDump2 = sH + (DT + 1) * SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane / LPR, j = lane % LPR;
  const int row = wave * RPW + r;
  const int b = FOLD ? b0 : blockIdx.y;           // FOLD: the first batch element of the part; the arrays' base pointers start there
  const int dpg = p.dim / p.G;
  const int tiles = (dpg + DT - 1) / DT;
  const int g = blockIdx.x / tiles;
  const int d0 = g * dpg + (blockIdx.x - g * tiles) * DT;
  const int d_end = (g + 1) * dpg;
  const int d = d0 + row;
  const bool row_ok = d < d_end;
  const int dc = row_ok ? d : d_end - 1;  // clamped row: loads stay in bounds, stores are masked

  const int dr = delta_row(dc, p.dl_ratio, p.dl_magic);   // delta / delta_bias row of this channel
  const io_t* __restrict__ pu = (const io_t*)p.u + (int64_t)b * p.u_bs + (int64_t)dc * p.u_ds + j * T;
  const io_t* __restrict__ pd = (const io_t*)p.delta + (int64_t)b * p.dl_bs + (int64_t)dr * p.dl_ds + j * T;
  const io_t* __restrict__ pz = p.z ? (const io_t*)p.z + (int64_t)b * p.z_bs + (int64_t)dc * p.z_ds + j * T : nullptr;
  const int64_t po = (int64_t)b * p.o_bs + (int64_t)dc * p.o_ds + j * T;   // element offset into p.out (io dtype, or fp32)
  const bool of32 = p.out_f32 != 0;
  const io_t* __restrict__ Bp = (const io_t*)p.B + (int64_t)b * p.B_bs + (int64_t)g * p.B_gs;
  const io_t* __restrict__ Cp = (const io_t*)p.C + (int64_t)b * p.C_bs + (int64_t)g * p.C_gs;
  const bool has_z = pz != nullptr;
  // FOLD: virtual step tv -> element offset of (batch tv / SL, step tv % SL) for an array with batch stride bs
  auto seg_of = [&](int tv) { return (int)__umulhi((unsigned)tv, p.fold_magic); };
  auto fold_off = [&](int tv, int64_t bs) {
    const int sb = seg_of(tv);
    return seg_off(sb, bs, tv - sb * SL);
  };

  for (int i = tid; i < (DT + 1) * N; i += NT) {
    const int rr = i / N, n = i - rr * N;
    const int dd = d0 + rr;
    const float a2 = (rr < DT && dd < d_end) ? p.A[(int64_t)dd * p.A_ds + (int64_t)n * p.A_ns] * kLog2e : 0.0f;
    if constexpr (PK) {
      if (rr < DT) sA[rr * SA + n] = a2;
      sH[rr * SA + n] = 0.0f;
    } else {
      sAC[rr * NP + n] = make_float2(a2, 0.0f);
    }
  }
  const float bias = p.bias ? p.bias[dr] : 0.0f;
  const float Dv = p.D ? p.D[dc] : 0.0f;

  // element e = jj*T + i of a tile row lives at quarter-major position (i/4)*(LPR*4) + jj*4 + i%4.  For 16-lane rows the odd
  // quarter is stored ROTATED by 16 words (SWZ): ds_write_b128 is serviced in groups of 8 consecutive lanes against 32 banks
  // (MI355X_MICROARCH.md, LDS table), and the 8 staging lanes of a group hold 4 even + 4 odd quarters -- unrotated, both sets sit
  // on banks 0-15 (the quarters are 64 words apart): a 2-way conflict on every staging write (SQ_LDS_BANK_CONFLICT was 29 % of
  // SQ_LDS_IDX_ACTIVE, profiles/r01_scan_sq.txt).  The reads stay one 256-byte run per quarter, rotated: still conflict-free.
  constexpr int SWZ = (LPR == 16) ? 16 : 0;
  auto qpos = [](int e) { const int q = (e % T) >> 2; return q * (LPR * 4) + ((((e / T) * 4) + (q & 1) * SWZ) & (LPR * 4 - 1)) + (e & 3); };

  // ---- B/C tile fetch (global -> registers) and commit (registers -> LDS buffer) ------------------
  // 16-bit aligned rows: the quarters stay PACKED (8 bytes) until the commit after the state loop -- converted at the load,
  // hipcc waits for the whole tile in front of the loop
  constexpr bool RAWBC = VEC && sizeof(io_t) == 2;
  float4 bq[(VEC && !RAWBC) ? BCV : 1], cq[(VEC && !RAWBC) ? BCV : 1];
  uint2 bqr[RAWBC ? BCV : 1], cqr[RAWBC ? BCV : 1];
  float bs[VEC ? 1 : BCS], cs[VEC ? 1 : BCS];
  auto q_unpack = [](uint2 r) {
    io_t t[4];
    *(uint2*)t = r;
    return make_float4(io::ld(t), io::ld(t + 1), io::ld(t + 2), io::ld(t + 3));
  };
  // PK: work item = (array B | C, state pair, quarter of 4 steps): two row pieces (states 2p, 2p + 1) per item, NPAIR * 32 items per array
  constexpr int NPAIR = PK ? NS / 2 : 1, IPT = PK ? (2 * NPAIR * 32) / NT : 1;
  float4 pl[(PK && !RAWBC) ? IPT : 1], ph[(PK && !RAWBC) ? IPT : 1];
  uint2 plr[(PK && RAWBC) ? IPT : 1], phr[(PK && RAWBC) ? IPT : 1];
  auto bc_fetch_pk = [&](int t0) {
    const bool full = t0 + CH <= L;
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
      const int id = tid + it * NT, arr = id / (NPAIR * 32), pr = (id >> 5) % NPAIR, e4 = (id & 31) * 4;
      const io_t* q0 = (arr ? Cp : Bp) + (int64_t)(2 * pr) * (arr ? p.C_ns : p.B_ns);
      const io_t* q1 = q0 + (arr ? p.C_ns : p.B_ns);
      int64_t off = t0 + e4;
      if constexpr (FOLD) {        // a quarter past the end reads quarter 0 (see bc_fetch): unconditional loads
        const int tq = t0 + e4 < L ? t0 + e4 : 0;
        const int sb = seg_of(tq);
        off = seg_off(sb, arr ? p.C_bs : p.B_bs, tq - sb * SL);
      }
      if (FOLD || full) {
        if constexpr (RAWBC) { plr[it] = *(const uint2*)(q0 + off); phr[it] = *(const uint2*)(q1 + off); }
        else { pl[it] = ld4<io_t>(q0 + off); ph[it] = ld4<io_t>(q1 + off); }
      } else {
        float tl[4] = {0.f, 0.f, 0.f, 0.f}, th[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (t0 + e4 + q < L) { tl[q] = io::ld(q0 + off + q); th[q] = io::ld(q1 + off + q); }
        if constexpr (RAWBC) {
          io_t a[4], b[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { io::st(a + q, tl[q]); io::st(b + q, th[q]); }    // exact: the values came from io_t
          plr[it] = *(uint2*)a; phr[it] = *(uint2*)b;
        } else {
          pl[it] = make_float4(tl[0], tl[1], tl[2], tl[3]); ph[it] = make_float4(th[0], th[1], th[2], th[3]);
        }
      }
    }
  };
  // block blk (2 steps: i = 2 blk, 2 blk + 1) of lane jj sits at word (blk * LPR + ((jj + 4 (blk / 2)) % LPR)) * 4: the rotation of the odd
  // quarter's blocks keeps the 8-lane groups of the staging ds_write_b128 (4 lanes x 2 quarters) on 32 different banks; the reads of a
  // DPP row stay one 256-byte run per block
  auto pk_pos = [](int blk, int jj) { return (blk * LPR + ((jj + 4 * (blk >> 1)) & (LPR - 1))) * 4; };
  auto bc_commit_pk = [&](int buf) {
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
      const int id = tid + it * NT, arr = id / (NPAIR * 32), pr = (id >> 5) % NPAIR, Q = id & 31;
      float* dst = sBC + buf * 2 * N * CH + arr * N * CH + pr * 2 * CH;
      float4 lo, hi;
      if constexpr (RAWBC) { lo = q_unpack(plr[it]); hi = q_unpack(phr[it]); } else { lo = pl[it]; hi = ph[it]; }
      *(float4*)(dst + pk_pos(2 * (Q & 1), Q >> 1)) = make_float4(lo.x, hi.x, lo.y, hi.y);
      *(float4*)(dst + pk_pos(2 * (Q & 1) + 1, Q >> 1)) = make_float4(lo.z, hi.z, lo.w, hi.w);
    }
  };
  auto bc_fetch = [&](int t0) {
    if constexpr (PK) { bc_fetch_pk(t0); return; }
    const bool full = t0 + CH <= L;
    if constexpr (VEC) {
      constexpr int CQ = CH / 4, RSTEP = NT / CQ;
      const int e4 = (tid % CQ) * 4;
      int64_t fB = 0, fC = 0;          // FOLD: this thread's quarter lies in ONE segment: its (segment, step) offset, once per chunk
      if constexpr (FOLD) {
        // a quarter past the end reads quarter 0 instead (finite values under delta = 0): the loads stay UNCONDITIONAL -- inside
        // a per-lane `if` the compiler waits for them at once (the else side writes the same registers), and the prefetch of
        // chunk c + 1 no longer overlaps the state loop of chunk c (+25 % on the folded forward)
        const int tq = t0 + e4 < L ? t0 + e4 : 0;
        const int sb = seg_of(tq), sl = tq - sb * SL;
        fB = seg_off(sb, p.B_bs, sl);
        fC = seg_off(sb, p.C_bs, sl);
      }
#pragma unroll
      for (int k = 0; k < BCV; ++k) {
        const int n = tid / CQ + k * RSTEP;
        if constexpr (RAWBC) {
          bqr[k] = make_uint2(0, 0);
          cqr[k] = bqr[k];
          if (n < N) {
            if constexpr (FOLD) {
              bqr[k] = *(const uint2*)(Bp + (int64_t)n * p.B_ns + fB);
              cqr[k] = *(const uint2*)(Cp + (int64_t)n * p.C_ns + fC);
            } else if (full) {
              bqr[k] = *(const uint2*)(Bp + (int64_t)n * p.B_ns + t0 + e4);
              cqr[k] = *(const uint2*)(Cp + (int64_t)n * p.C_ns + t0 + e4);
            } else {
              uint16_t tb[4] = {0, 0, 0, 0}, tc[4] = {0, 0, 0, 0};
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (t0 + e4 + q < L) {
                  tb[q] = *(const uint16_t*)(Bp + (int64_t)n * p.B_ns + t0 + e4 + q);
                  tc[q] = *(const uint16_t*)(Cp + (int64_t)n * p.C_ns + t0 + e4 + q);
                }
              bqr[k] = make_uint2((uint32_t)tb[0] | ((uint32_t)tb[1] << 16), (uint32_t)tb[2] | ((uint32_t)tb[3] << 16));
              cqr[k] = make_uint2((uint32_t)tc[0] | ((uint32_t)tc[1] << 16), (uint32_t)tc[2] | ((uint32_t)tc[3] << 16));
            }
          }
          continue;
        }
        bq[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        cq[k] = bq[k];
        if (n < N) {
          if constexpr (FOLD) {     // a quarter lies inside one segment (SL % 4 == 0)
            bq[k] = ld4<io_t>(Bp + (int64_t)n * p.B_ns + fB);
            cq[k] = ld4<io_t>(Cp + (int64_t)n * p.C_ns + fC);
          } else if (full) {
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
      constexpr int RSTEP = NT / CH;
      const int e = tid % CH;
      const bool ok = t0 + e < L;
#pragma unroll
      for (int k = 0; k < BCS; ++k) {
        const int n = tid / CH + k * RSTEP;
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
    if constexpr (PK) { bc_commit_pk(buf); return; }
    float* dB = sBC + buf * 2 * N * CH;
    float* dC = dB + N * CH;
    if constexpr (VEC) {
      constexpr int CQ = CH / 4, RSTEP = NT / CQ;
      const int pos = qpos((tid % CQ) * 4);
#pragma unroll
      for (int k = 0; k < BCV; ++k) {
        const int n = tid / CQ + k * RSTEP;
        if (n < N) {
          if constexpr (RAWBC) {
            *(float4*)(dB + n * CH + pos) = q_unpack(bqr[k]);
            *(float4*)(dC + n * CH + pos) = q_unpack(cqr[k]);
          } else {
            *(float4*)(dB + n * CH + pos) = bq[k];
            *(float4*)(dC + n * CH + pos) = cq[k];
          }
        }
      }
    } else {
      constexpr int RSTEP = NT / CH;
      const int pos = qpos(tid % CH);
#pragma unroll
      for (int k = 0; k < BCS; ++k) {
        const int n = tid / CH + k * RSTEP;
        if (n < N) {
          dB[n * CH + pos] = bs[k];
          dC[n * CH + pos] = cs[k];
        }
      }
    }
  };
  // ---- a lane's T consecutive elements of one row ----------------------------------------------------
  // u and delta of a chunk: ONE whole / partial branch for both arrays, so both requests are in flight before either is used
  // (two independent fetch calls serialised their HBM round trips on ragged chunks -- every second chunk of a 197-token row)
  auto ldT = [&](const io_t* q, float (&v)[T]) {
    if constexpr (VEC) {
#pragma unroll
      for (int k = 0; k < TQ; ++k) {
        const float4 a = ld4<io_t>(q + 4 * k);
        v[4 * k] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) v[i] = io::ld(q + i);
    }
  };
  auto row_fetch = [&](const io_t* q, int t0, float (&v)[T], int64_t bs) {
    if constexpr (FOLD) {
      ldT(q - j * T + bs, v);            // FOLD: `bs` carries the lane's (segment, step) element offset of this chunk (clamped)
    } else if (t0 + CH <= L) {           // wave-uniform (see ud_fetch)
      ldT(q + t0, v);
    } else if (VEC && t0 + j * T + T <= L) {
      ldT(q + t0, v);
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) v[i] = (t0 + j * T + i < L) ? io::ld(q + t0 + i) : 0.0f;
    }
  };
  auto ud_fetch = [&](int t0, float (&vu)[T], float (&vd)[T]) {
    if constexpr (FOLD) {   // lanes past the end read step 0 (see bc_fetch): delta is forced to 0 for them below, nothing is stored
      const int tv = t0 + j * T < L ? t0 + j * T : 0, sb = seg_of(tv), sl = tv - sb * SL;
      const io_t* qu = pu - j * T + seg_off(sb, p.u_bs, sl);
      const io_t* qd = pd - j * T + seg_off(sb, p.dl_bs, sl);
      if constexpr (T == 8 && sizeof(io_t) == 2) {
        // both rows requested before either is converted: with two ldT calls hipcc waited for u before it asked for delta
        // (an extra HBM round trip per chunk, +25 % on the folded forward)
        uint4 ra = *(const uint4*)qu, rb = *(const uint4*)qd;
        // (an empty asm that "uses" all eight registers: the scheduler otherwise converts u before it issues the load of delta)
        asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w), "+v"(rb.x), "+v"(rb.y), "+v"(rb.z), "+v"(rb.w));
        io_t ta[T], tb[T];
        *(uint4*)ta = ra;
        *(uint4*)tb = rb;
#pragma unroll
        for (int i = 0; i < T; ++i) { vu[i] = io::ld(ta + i); vd[i] = io::ld(tb + i); }
      } else {
        ldT(qu, vu);
        ldT(qd, vd);
      }
    } else if (t0 + CH <= L) {
      // the full-chunk test alone is wave-uniform -- a scalar branch.  Merged with the per-lane "whole lane inside the row" test
      // it became an exec-masked region whose other side writes the same registers, and hipcc then waits for the loads right
      // there: the prefetch of chunk c + 1 never overlapped the state loop of chunk c
      ldT(pu + t0, vu);
      ldT(pd + t0, vd);
    } else if (VEC && t0 + j * T + T <= L) {
      ldT(pu + t0, vu);
      ldT(pd + t0, vd);
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const bool ok = t0 + j * T + i < L;
        vu[i] = ok ? io::ld(pu + t0 + i) : 0.0f;
        vd[i] = ok ? io::ld(pd + t0 + i) : 0.0f;
      }
    }
  };

  const int nchunks = (L + CH - 1) / CH;
#ifdef MXVL_ABLATE
  const long long clk0 = clock64(), wall0 = wall_clock64();   // (ablate & 16) measurement: shader clock vs 100 MHz wall
#else
  const long long clk0 = 0, wall0 = 0;
#endif
  float un[T], dn[T];
  // 16-bit aligned rows: u / delta of the next chunk stay PACKED (2 x 16 bytes) across the state loop and are converted at the
  // top of the next iteration, z of this chunk until the loop is over -- converted where they are loaded, hipcc waits for them in
  // front of the loop: no overlap at all (folded forward 714 -> 688 us, profiles/r03_scan_fold.txt)
  constexpr bool RAWPF = VEC && T == 8 && sizeof(io_t) == 2;
  uint4 ru = make_uint4(0, 0, 0, 0), rd = ru;
  auto raw_row = [&](const io_t* q, int t0) {          // 8 steps of a plain (not folded) row; beyond L: zeros
    if (t0 + CH <= L) return *(const uint4*)(q + t0);   // wave-uniform: every full chunk takes this path, nothing per lane
    uint16_t e[T];
#pragma unroll
    for (int i = 0; i < T; ++i) e[i] = (t0 + j * T + i < L) ? *(const uint16_t*)(q + t0 + i) : (uint16_t)0;
    return make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                      (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
  };
  auto raw_ud = [&](int t0) {
    if constexpr (FOLD) {
      const int tv = t0 + j * T < L ? t0 + j * T : 0, sb = seg_of(tv), sl = tv - sb * SL;
      ru = *(const uint4*)(pu - j * T + seg_off(sb, p.u_bs, sl));
      rd = *(const uint4*)(pd - j * T + seg_off(sb, p.dl_bs, sl));
    } else {
      ru = raw_row(pu, t0);
      rd = raw_row(pd, t0);
    }
  };
  bc_fetch(0);
  if constexpr (RAWPF) raw_ud(0); else ud_fetch(0, un, dn);
  bc_commit(0);

  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * CH;
    const bool full = t0 + CH <= L;
    const bool more = c + 1 < nchunks;
    __syncthreads();  // B/C tile c is complete in buffer c&1; nobody still reads buffer (c+1)&1

    float dl[T], du[T], y[T], zz[T];
    // flag tests stay OUTSIDE the per-step loops: a uniform branch per step serialises the 8 exp -> log -> rcp chains
#pragma unroll
    for (int i = 0; i < T; ++i) dl[i] = dn[i] + bias;
    if constexpr (RAWPF) {
      io_t ta[T], tb[T];
      *(uint4*)ta = ru;
      *(uint4*)tb = rd;
#pragma unroll
      for (int i = 0; i < T; ++i) { un[i] = io::ld(ta + i); dl[i] = io::ld(tb + i) + bias; }
    }
    if (p.softplus) {
#pragma unroll
      for (int i = 0; i < T; ++i) dl[i] = softplus(dl[i]);
    }
    if (!full) {
#pragma unroll
      for (int i = 0; i < T; ++i) dl[i] = (t0 + j * T + i < L) ? dl[i] : 0.0f;  // padding steps are the identity map
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
      du[i] = dl[i] * un[i];
      y[i] = Dv * un[i];
    }
    // requests for the next chunk (and this chunk's z) go out before the long state loop
    // (measurement build: ablate & 32 = NO global loads after chunk 0 -- every chunk re-uses stale rows and the B / C tile of chunk 0;
    //  & 64 / 128 / 256 drop one class only: the B / C tile, z, u / delta.  With & 8 no stores either: all of the VALU / LDS work,
    //  none of the HBM traffic)
    const int abl = MXVL_ABL(p.ablate);
    if (more) {
      if (!(abl & (32 | 64))) bc_fetch(t0 + CH);
      if (!(abl & (32 | 256))) {
        if constexpr (RAWPF) raw_ud(t0 + CH); else ud_fetch(t0 + CH, un, dn);
      }
    }
    // FOLD: (segment, step) of this lane's 8 steps in chunk c, once for z, the reset test and the output store
    int csb = 0, csl = 0;
    if constexpr (FOLD) {
      const int tv = t0 + j * T < L ? t0 + j * T : 0;
      csb = seg_of(tv);
      csl = tv - csb * SL;
    }
    uint4 rz = make_uint4(0, 0, 0, 0);       // RAWPF: z of this chunk stays packed until the state loop is over
    if (has_z && !((abl & (32 | 128)) && c > 0)) {
      if constexpr (RAWPF && FOLD) rz = *(const uint4*)(pz - j * T + seg_off(csb, p.z_bs, csl));
      else if constexpr (RAWPF) rz = raw_row(pz, t0);
      else row_fetch(pz, t0, zz, FOLD ? seg_off(csb, p.z_bs, csl) : 0);
    }

    float dsum = 0.0f;
#pragma unroll
    for (int i = 0; i < T; ++i) dsum += dl[i];

    if (p.ckpt != nullptr) {
      // checkpoint = state entering the chunk = the running state this wave left in LDS (CH == kCkptLen)
      for (int i = lane; i < RPW * N; i += 64) {
        const int rr = i / N, n = i - rr * N;
        const int dd = d0 + wave * RPW + rr;
        if (dd < d_end) {
          // FOLD: (channel, part, virtual chunk); a part owns fold_cpp = chunks(fold_bpp * SL) slots
          const int64_t slot = FOLD ? ((int64_t)dd * gridDim.y + blockIdx.y) * p.fold_cpp + c : ((int64_t)b * p.dim + dd) * p.n_ckpt + c;
          p.ckpt[slot * N + n] = PK ? sH[(wave * RPW + rr) * SA + n] : sAC[(wave * RPW + rr) * NP + n].y;
        }
      }
    }
    // FOLD: a lane whose first step opens a segment adds -inf to the exponent of a_0 and of the lane product: a_0 = P = 0
    float rbias = 0.0f;
    if constexpr (FOLD) rbias = (csl == 0) ? -__builtin_inff() : 0.0f;
    float2* ac = sAC + row * NP;
    const float2* ac_in = sAC + ((j == 0) ? row : DT) * NP;  // only lane 0 sees the state entering the chunk
    // the last lane of a row stores the state leaving the chunk; the others store into a private dump word (an exec-masked
    // store cost s_and_saveexec / s_cbranch / s_or per state)
    float* ac_out = (j == LPR - 1) ? &ac[0].y : sDump + tid;
    const float* cB = sBC + (c & 1) * 2 * N * CH;
    const float* cC = cB + N * CH;
    int rq[TQ];                                               // this lane's word offset inside a tile row, per quarter
#pragma unroll
    for (int k = 0; k < TQ; ++k) rq[k] = k * (LPR * 4) + ((j * 4 + (k & 1) * SWZ) & (LPR * 4 - 1));

    // one state up to the lane map (P, h): B/C tile rows, a_i = exp2(delta_i A), b_i = delta_i u_i B_i, Horner fold.
    // (Requesting the LDS operands one state ahead -- a software pipeline over Pre{A2, car, b[], c[]} -- measured SLOWER at every
    // shape, 260 -> 307 us at the roofline shape, profiles/r03b_scan_variants_prefetch.txt: the compiler's own placement of the
    // next state's ds_reads behind the DPP scan is kept.  Likewise rejected: filling the 7 wait-state slots of the DPP scan with the
    // next state's 8 `delta_i * A2` products (one asm block, bit-identical results) -- 288 -> 295 us here, 363 -> 482 us on the
    // 8-wave instantiation, profiles/r03_fwd_dpp_fill_ab.txt: the longer opaque block costs the compiler more than the nops.)
    auto fold = [&](int n, float (&a)[T], float (&bb)[T], float (&cv)[T], float& P, float& hl, float& car) {
      const float A2 = ac[n].x;
      car = ac_in[n].y;                                       // state entering the chunk (lane 0), 0 elsewhere
#pragma unroll
      for (int k = 0; k < TQ; ++k) {
        const float4 b4 = *(const float4*)(cB + n * CH + rq[k]);
        const float4 c4 = *(const float4*)(cC + n * CH + rq[k]);
        bb[4 * k] = b4.x; bb[4 * k + 1] = b4.y; bb[4 * k + 2] = b4.z; bb[4 * k + 3] = b4.w;
        cv[4 * k] = c4.x; cv[4 * k + 1] = c4.y; cv[4 * k + 2] = c4.z; cv[4 * k + 3] = c4.w;
      }
      // steps (i, i + 1) of ONE state in the halves of a register pair: the two element-wise products are v_pk_mul_f32 (delta and
      // delta u pairs are neighbours already, the B quarter came as a float4) -- 16 packed multiplies in the place of 32 per two
      // states, the same IEEE products, no layout or register change: 277.8 -> 265.6 us at the roofline shape, one box, alternating
      // (profiles/r06_scan_fwd_exp_ab.txt)
#pragma unroll
      for (int i = 0; i < T; i += 2) {
        const v2f t = v2f{dl[i], dl[i + 1]} * A2;
        const v2f w = v2f{du[i], du[i + 1]} * v2f{bb[i], bb[i + 1]};
        a[i] = t.x; a[i + 1] = t.y;
        bb[i] = w.x; bb[i + 1] = w.y;
      }
      P = A2 * dsum;
      if constexpr (FOLD) {
        a[0] += rbias;
        P += rbias;
      }
      // the T + 1 v_exp_f32 of a state back to back: 8 cycles each alone, 10-16 when interleaved with FMAs
      // (profiles/r01_ubench_valu_mix.txt)
      if constexpr (T == 8) {
        asm volatile(
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n"
            "v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n v_exp_f32 %8, %8\n"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(P));
      } else {
        asm volatile(
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n"
            "v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n v_exp_f32 %8, %8\n"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(P));
        asm volatile(
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n"
            "v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
      }
      float h = bb[0];
#pragma unroll
      for (int i = 1; i < T; ++i) h = fmaf(a[i], h, bb[i]);   // pass 1: lane map h_out = P*h_in + h
      hl = fmaf(P, car, h);                                   // lane 0 absorbs the incoming state
    };
    auto pass2 = [&](int n, const float (&a)[T], const float (&bb)[T], const float (&cv)[T], float hl, float x) {
      ac_out[2 * n] = hl;                                     // state leaving the chunk (last lane of the row)
      float h = x;
#pragma unroll
      for (int i = 0; i < T; ++i) {
        h = fmaf(a[i], h, bb[i]);
        y[i] = fmaf(cv[i], h, y[i]);
      }
    };
    const int n_states = MXVL_ABL(p.ablate & 1) ? 0 : MXVL_ABL(p.ablate & 2) ? N / 2 : N;
    if constexpr (PK) {
      const float* hin = sH + ((j == 0) ? row : DT) * SA;            // only lane 0 sees the state entering the chunk
      float* hout = (j == LPR - 1) ? sH + row * SA : sDump2 + 2 * tid;  // the last lane stores the state leaving it; the others a dump pair
      const float* a2row = sA + row * SA;
      int rp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) rp[k] = pk_pos(k, j);
      v2f y2[T];
#pragma unroll
      for (int i = 0; i < T; ++i) y2[i] = v2f{y[i], 0.0f};
      const v2f rb2 = v2f{rbias, rbias};
      // one pair per trip = the two states in flight of the unpacked loop's `unroll 2`; two pairs per trip keep 96 registers of
      // a / b / c live and cost the third workgroup per CU (218 VGPRs: 274.8 vs 252.8 us at the roofline shape)
#pragma unroll 1
      for (int n = 0; n < n_states; n += 2) {
        v2f a[T], bb[T], cv[T];
        // delta_i and delta_i u_i multiply BOTH halves: left alone, the compiler hoists 16 (x, x) register pairs out of the loop (32
        // VGPRs -- the kernel lands at 199 and loses the third workgroup per CU); redefined here, the splat is built at the use and
        // folds into the instruction's op_sel
        asm volatile("" : "+v"(dl[0]), "+v"(dl[1]), "+v"(dl[2]), "+v"(dl[3]), "+v"(dl[4]), "+v"(dl[5]), "+v"(dl[6]), "+v"(dl[7]),
                          "+v"(du[0]), "+v"(du[1]), "+v"(du[2]), "+v"(du[3]), "+v"(du[4]), "+v"(du[5]), "+v"(du[6]), "+v"(du[7]));
        const v2f A2 = *(const v2f*)(a2row + n);
        const v2f car = *(const v2f*)(hin + n);
        const float* tB = cB + (n >> 1) * 2 * CH;
        const float* tC = cC + (n >> 1) * 2 * CH;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 b4 = *(const float4*)(tB + rp[k]);
          const float4 c4 = *(const float4*)(tC + rp[k]);
          bb[2 * k] = v2f{b4.x, b4.y}; bb[2 * k + 1] = v2f{b4.z, b4.w};
          cv[2 * k] = v2f{c4.x, c4.y}; cv[2 * k + 1] = v2f{c4.z, c4.w};
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
          a[i] = A2 * dl[i];
          bb[i] = bb[i] * du[i];
        }
        v2f P = A2 * dsum;
        if constexpr (FOLD) {
          a[0] += rb2;
          P += rb2;
        }
        // the 18 v_exp_f32 of a pair back to back (see fold below).  Outputs are NOT tied to the inputs: a tied ("+v") half of a
        // 64-bit pair made the register allocator copy the other half around it (26 v_mov_b32 per two pairs in the first build)
        float P0, P1;
        {
          float e0, e1, e2, e3, e4, e5, e6, e7, f0, f1, f2, f3, f4, f5, f6, f7;
          asm volatile(
              "v_exp_f32 %0, %9\n v_exp_f32 %1, %10\n v_exp_f32 %2, %11\n v_exp_f32 %3, %12\n v_exp_f32 %4, %13\n"
              "v_exp_f32 %5, %14\n v_exp_f32 %6, %15\n v_exp_f32 %7, %16\n v_exp_f32 %8, %17\n"
              : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7), "=&v"(P0)
              : "v"(a[0].x), "v"(a[0].y), "v"(a[1].x), "v"(a[1].y), "v"(a[2].x), "v"(a[2].y), "v"(a[3].x), "v"(a[3].y), "v"(P.x));
          asm volatile(
              "v_exp_f32 %0, %9\n v_exp_f32 %1, %10\n v_exp_f32 %2, %11\n v_exp_f32 %3, %12\n v_exp_f32 %4, %13\n"
              "v_exp_f32 %5, %14\n v_exp_f32 %6, %15\n v_exp_f32 %7, %16\n v_exp_f32 %8, %17\n"
              : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(f4), "=&v"(f5), "=&v"(f6), "=&v"(f7), "=&v"(P1)
              : "v"(a[4].x), "v"(a[4].y), "v"(a[5].x), "v"(a[5].y), "v"(a[6].x), "v"(a[6].y), "v"(a[7].x), "v"(a[7].y), "v"(P.y));
          a[0] = v2f{e0, e1}; a[1] = v2f{e2, e3}; a[2] = v2f{e4, e5}; a[3] = v2f{e6, e7};
          a[4] = v2f{f0, f1}; a[5] = v2f{f2, f3}; a[6] = v2f{f4, f5}; a[7] = v2f{f6, f7};
        }
        v2f h = bb[0];
#pragma unroll
        for (int i = 1; i < T; ++i) h = __builtin_elementwise_fma(a[i], h, bb[i]);   // pass 1: lane map h_out = P * h_in + h
        // from here to the first step of pass 2 the two states are scalars: the DPP block ties its operands, and a pair rebuilt from
        // tied scalars costs copies
        float h0 = fmaf(P0, car.x, h.x), h1 = fmaf(P1, car.y, h.y);                     // lane 0 absorbs the incoming state
        float x0 = car.x, x1 = car.y;
        scan16_x2(h0, P0, x0, h1, P1, x1);
        *(v2f*)(hout + n) = v2f{h0, h1};
        h = v2f{fmaf(a[0].x, x0, bb[0].x), fmaf(a[0].y, x1, bb[0].y)};
        y2[0] = __builtin_elementwise_fma(cv[0], h, y2[0]);
#pragma unroll
        for (int i = 1; i < T; ++i) {
          h = __builtin_elementwise_fma(a[i], h, bb[i]);
          y2[i] = __builtin_elementwise_fma(cv[i], h, y2[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < T; ++i) y[i] = y2[i].x + y2[i].y;
    } else {
    // (measured and removed, profiles/r06_scan_fwd_exp_ab.txt: two states per trip through ONE interleaved DPP block, scan16_x2, instead of
    // two scan16_x1 blocks with their s_nops -- 171 VGPRs, the third workgroup per CU is lost: 318.9 vs 259.9 us; with the B / C tile
    // by LDS-DMA to win the registers back, 167 VGPRs: 275.2 us; the tile by LDS-DMA alone: 271.4 us)
    // (unroll 4: the same instruction count per state, 157 VGPRs; the whole loop unrolled: 256 VGPRs + 1 KB of scratch)
#pragma unroll 2
    for (int n = 0; n < n_states; ++n) {
      float a[T], bb[T], cv[T], P, hl, x;
      fold(n, a, bb, cv, P, hl, x);
      if constexpr (LPR == 16) {
        scan16_x1(hl, P, x);
      } else {
        const float car = x;
        P = (j == 0) ? 0.0f : P;                              // nothing may flow in from the neighbouring row
        scan8_x1(hl, P, x);
        x = (j == 0) ? car : x;
      }
      pass2(n, a, bb, cv, hl, x);
    }
    }

    if (more) bc_commit((c + 1) & 1);
    if (has_z) {
      if constexpr (RAWPF) {
        io_t tz[T];
        *(uint4*)tz = rz;
#pragma unroll
        for (int i = 0; i < T; ++i) zz[i] = io::ld(tz + i);
      }
#pragma unroll
      for (int i = 0; i < T; ++i) y[i] *= silu(zz[i]);
    }
    if (MXVL_ABL(p.ablate & 8)) continue;
    if constexpr (FOLD) {
      if (row_ok && t0 + j * T < L) {
        const int64_t o = po - j * T + seg_off(csb, p.o_bs, csl);
#pragma unroll
        for (int k = 0; k < TQ; ++k)
          st4_out<io_t>(p.out, o + 4 * k, make_float4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]), of32);
      }
    } else if (VEC && full) {
      if (row_ok) {
#pragma unroll
        for (int k = 0; k < TQ; ++k)
          st4_out<io_t>(p.out, po + t0 + 4 * k, make_float4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]), of32);
      }
    } else if (VEC) {   // ragged last chunk of aligned rows: whole lane groups as vectors, the boundary lane element-wise
      if (row_ok) {
        if (t0 + j * T + T <= L) {
#pragma unroll
          for (int k = 0; k < TQ; ++k)
            st4_out<io_t>(p.out, po + t0 + 4 * k, make_float4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]), of32);
        } else {
#pragma unroll
          for (int i = 0; i < T; ++i)
            if (t0 + j * T + i < L) st_out<io_t>(p.out, po + t0 + i, y[i], of32);
        }
      }
    } else {
      float4* so4 = (float4*)(sO + row * CH + j * T);
#pragma unroll
      for (int k = 0; k < TQ; ++k) so4[k] = make_float4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int wrow = wave * RPW + rr;
        const int dd = d0 + wrow;
        const int64_t q = (int64_t)b * p.o_bs + (int64_t)dd * p.o_ds + t0;
#pragma unroll
        for (int e = lane; e < CH; e += 64)
          if (dd < d_end && t0 + e < L) st_out<io_t>(p.out, q + e, sO[wrow * CH + e], of32);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  if (MXVL_ABL(p.ablate & 16) && tid == 0) {   // per-workgroup start / end wall ticks (use with ablate & 8: no out stores)
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    ((float*)p.out)[4 * wg] = (float)(wall0 & 0xffffff);
    ((float*)p.out)[4 * wg + 1] = (float)(wall_clock64() & 0xffffff);
    ((float*)p.out)[4 * wg + 2] = (float)(clock64() - clk0);
    ((float*)p.out)[4 * wg + 3] = (float)((__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffff) | ((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf) << 16));
  }
  if (p.last_state) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < RPW * N; i += 64) {
      const int rr = i / N, n = i - rr * N;
      const int dd = d0 + wave * RPW + rr;
      if (dd < d_end) p.last_state[((int64_t)b * p.dim + dd) * N + n] = PK ? sH[(wave * RPW + rr) * SA + n] : sAC[(wave * RPW + rr) * NP + n].y;
    }
  }
}

}  // namespace mxvl
