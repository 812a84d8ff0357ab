// scan_bwd.hip -- selective-scan backward for gfx950 (MI355X, CDNA4, wave64).
//
// Replaces the reference's CUDA selective_scan_bwd_kernel
// (R2GenCSR/VMamba/kernels/selective_scan/csrc/selective_scan/cus/selective_scan_bwd_kernel.cuh:66-273,
// host selective_scan.cpp:241-349).  Same lane mapping as the forward streaming kernel (workgroup = 16
// rows of one batch element, a row over the 16 lanes of a DPP row, 8 consecutive steps per lane, 128-step
// chunks) walked from the LAST chunk to the first:
//   * the forward states of the chunk are recomputed from the checkpoint the forward pass wrote (state
//     entering every 128-step chunk) -- fold, forward DPP scan, second pass keeping h_t in registers;
//   * the adjoint recurrence g_t = C_t dy_t + a_{t+1} g_{t+1} is the mirror image: a Horner fold from the
//     lane's last step, a REVERSE prefix scan over the lanes (DPP row_shl fused into v_fmac/v_mul), and a
//     second pass that produces every gradient contribution of the step;
//   * dB/dC (summed over the rows of a B/C group): a wave sums the shares of its 4 rows in registers
//     (v_permlane32_swap / v_permlane16_swap), the workgroup's waves meet in an LDS tile flushed in groups of 4 states one
//     group behind the computation, and each (n,t) leaves as ONE fp32 global atomic per workgroup (or a plain store into the
//     caller's workspace + a reduce kernel) -- the CUDA kernel issues one global atomic per (row,n,t)
//     (bwd_kernel.cuh:215-221);
//   * dA, dD, ddelta_bias are reduced over lanes with DPP and over chunks in LDS/registers: one global
//     atomic per (row,n) / row per workgroup.
// du, ddelta, dz are fully written; dA, dB, dC, dD, ddelta_bias are accumulated into caller-zeroed fp32.

#include "mxvl_common.h"
#include <type_traits>

namespace mxvl {

constexpr int kCkptLenB = 128;

struct ScanBwdArgs {
  int batch, dim, L, N, G, n_ckpt;
  int softplus, vec_ok, ablate, dl_ratio, out_f32;
  uint32_t dl_magic, fold_magic;   // fold_magic != 0: batch folded into the sequence (FOLD below)
  int fold_bpp, fold_cpp;
  int64_t u_bs, u_ds, dl_bs, dl_ds, z_bs, z_ds, do_bs, do_ds;
  int64_t du_bs, du_ds, dd_bs, dd_ds, dz_bs, dz_ds;
  int64_t B_bs, B_gs, B_ns, C_bs, C_gs, C_ns, A_ds, A_ns;
  int64_t dB_bs, dB_gs, dB_ns, dC_bs, dC_gs, dC_ns;
  const void *u, *delta, *B, *C, *z, *dout;
  const float *A, *D, *bias, *ckpt;
  void *du, *ddelta, *dz;
  float *dA, *dB, *dC, *dD, *dbias;
};

// forward inclusive scan + exclusive shift over a 16-lane DPP row (see scan_fwd.hip)
__device__ inline void scan16_fwd(float& h0, float& P0, float& x0) {
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
// mirror image: lane j combines with lanes j+1.. (row_shl); x leaves holding the value entering from the right
__device__ inline void scan16_rev(float& q0, float& P0, float& x0) {
  asm volatile(
      "s_nop 1\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:2 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shl:2 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:4 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:8 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_mov_b32_dpp %2, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n"
      : "+v"(q0), "+v"(P0), "+v"(x0));
}
// Both scans of a state in one instruction stream: the forward scan of the recomputed states (row_shr) and the reverse scan of
// the adjoints (row_shl) are independent, so interleaving them puts >= 2 instructions between every VALU write and the DPP read
// of the same register -- the s_nop padding of the two separate blocks (10 idle issue slots per state) disappears.
__device__ __forceinline__ void scan16_fwd_rev(float& h0, float& Pf, float& x0, float& q0, float& Pr, float& g0) {
  asm volatile(
      "s_nop 1\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %4, %4, %4 row_shl:1 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shl:2 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %4, %4, %4 row_shl:2 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shl:4 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "v_mul_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n"
      "v_fmac_f32_dpp %3, %3, %4 row_shl:8 row_mask:0xf bank_mask:0xf\n"
      "s_nop 0\n"
      "v_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "v_mov_b32_dpp %5, %3 row_shl:1 row_mask:0xf bank_mask:0xf\n"
      : "+v"(h0), "+v"(Pf), "+v"(x0), "+v"(q0), "+v"(Pr), "+v"(g0));
}
// sum over the 16 lanes of a DPP row; the total lands in lane 15 of the row
__device__ inline float row_sum_to_lane15(float v) {
  v += dpp<DPP_ROW_SHR(1)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(2)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(4)>(0.0f, v);
  v += dpp<DPP_ROW_SHR(8)>(0.0f, v);
  return v;
}

// lanes l <-> l+32 / 16-lane rows 1 <-> 0, 3 <-> 2 exchanged between two registers: the building blocks of the in-register
// sum over the 4 rows of a wave.  Inline asm: this toolchain's __builtin_amdgcn_permlane32_swap returns its first
// result twice.  A swap needs two wait states after a VALU write of its operands: one s_nop 1 opens each block, the swaps
// that follow (independent registers) cover each other.
__device__ __forceinline__ void lane32_swap_x8(float (&x)[8], float (&y)[8]) {   // (x[i], x[i+4]) and (y[i], y[i+4]), i < 4
  asm volatile(
      "s_nop 1\n"
      "v_permlane32_swap_b32 %0, %4\n v_permlane32_swap_b32 %8, %12\n"
      "v_permlane32_swap_b32 %1, %5\n v_permlane32_swap_b32 %9, %13\n"
      "v_permlane32_swap_b32 %2, %6\n v_permlane32_swap_b32 %10, %14\n"
      "v_permlane32_swap_b32 %3, %7\n v_permlane32_swap_b32 %11, %15\n"
      : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
        "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
}
__device__ __forceinline__ void lane16_swap_x4(float (&s)[8]) {                   // (s[0], s[1]) (s[2], s[3]) (s[4], s[5]) (s[6], s[7])
  asm volatile(
      "s_nop 1\n"
      "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n"
      "v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
      : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]));
}

// Registers are held to 2 waves per SIMD (<= 256 VGPR).  A 3-waves/SIMD build (<= 168 VGPR, the per-chunk row state
// spilled once per chunk) measured 1.80 ms against 1.34 ms at the pre-training shape (profiles/r02_bwd_variants.txt) and was
// dropped.
//
// NS > 0: dstate is the compile-time constant NS (16 = every Mamba block of the reference, mamba_simple.py:42), so the state loop
// is unrolled in flush groups of FG = 4 states with immediate LDS offsets; NS == 0: any dstate <= 64 at run time, same code.
// Round-3 restructuring of the state loop (profiles/r03_bwd_isa.txt: 226 VALU + 69 SALU + 21 LDS instructions per state before):
//   * no exec-masked LDS stores: the lanes that do not own a table entry (adjoint hand-off: lane 0; dA: lane 15) write a private
//     dump word instead -- a masked store cost s_and_saveexec / s_cbranch / s_or per state, and for dA an exposed LDS round trip;
//   * the dB/dC flush addresses are wave-uniform (SGPR base + lane): no 64-bit VALU multiplies per atomic;
//   * the 9 v_exp_f32 of a state issue back to back (one asm block): v_exp costs 8 cycles alone and 10-16 when interleaved with
//     FMAs (profiles/r01_ubench_valu_mix.txt).
//
// FOLD (MXVL_SCAN_FOLD_BATCH, see scan_fwd_stream.h): the workgroup walks batch elements [b0, b0 + nb) of its channels as ONE
// sequence of nb * SL steps, last chunk first.  The first step of every segment has a_t = 0 (and the lane product P = 0): no
// state enters a segment in the recomputation, no adjoint leaves it towards the previous one, and dA / ddelta see h_{-1} = 0.
// Addresses of a lane's 8 steps, of a staging quarter and of a flush lane's step come from one multiply-high each.
template <typename io_t, int NWAVES, bool VEC, int NS, bool FOLD = false, bool DMAR = false>
__global__ __launch_bounds__(NWAVES * 64, 2) void scan_bwd_kernel(const ScanBwdArgs p) {
  constexpr int T = 8, LPR = 16, RPW = 4, DT = NWAVES * RPW, CH = 128, NT = NWAVES * 64;
  constexpr int FG = 4;                    // states per dB/dC flush group
  constexpr bool PKB = FOLD;               // the packed state body (below): the instantiations it was measured faster on
  static_assert(CH == kCkptLenB, "one checkpoint per chunk");
  static_assert(NS % FG == 0, "compile-time dstate is a whole number of flush groups");
  using io = Io<io_t>;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = NS > 0 ? NS : p.N, SL = p.L;
  const int b0 = FOLD ? (int)blockIdx.y * p.fold_bpp : 0;
  const int L = FOLD ? (p.batch - b0 < p.fold_bpp ? p.batch - b0 : p.fold_bpp) * SL : SL;   // steps this workgroup walks
  float* sB = smem;                        // [N][CH]  quarter-major, odd quarter rotated by 16 words (see scan_fwd_stream.h qpos)
  float* sC = sB + N * CH;                 // [N][CH]
  // dB/dC shares of a group of FG states, one tile per WAVE (its 4 rows summed in registers first), double-buffered:
  // plain LDS stores, then a NWAVES-way tree sum one group behind.  (ds_add_f32 measured 7x the whole rest of the kernel,
  // with or without same-address conflicts: LDS float atomics are not usable here.)
  float* sAcc = sC + N * CH;               // [2 buffers][FG][NWAVES][2 (dB,dC)][CH]
  float* sO = sAcc;                        // [DT][CH] store transpose tile of unaligned rows: aliases sAcc (idle between the
                                           // last flush barrier of a chunk and the barrier that opens the next one)
  // The three per-(row, state) tables are padded to NP = N + 1 entries per row: at a stride of N (= 16 or 32 words) the same
  // state of every row sat on one bank and the per-state b32 accesses of a wave's 4 rows (+ the zero row) conflicted 2-3-way.
  const int NP = N + 1;
  float2* sAC = (float2*)(sAcc + 2 * FG * NWAVES * 2 * CH);   // [DT+1][NP] {A*log2e, state entering the chunk}; row DT = 0
  float* sG = (float*)(sAC + (DT + 1) * NP);  // [DT+1][NP] adjoint entering the chunk from the right; row DT = 0
  float* sdA = sG + (DT + 1) * NP;         // [DT][NP] dA accumulated over the chunks
  float* sDump = sdA + DT * NP;            // [NT + N] write-only / garbage words of the lanes that own no table entry
  // u, delta, z, dout of the chunk as loaded, parked here across the state loop.  They are only needed again for the
  // per-step outputs; in registers (32 VGPRs) they push the loop to the 256-VGPR limit, where the compiler re-computes the 8
  // v_exp_f32 of a_t in the second pass instead of keeping them (17 instead of 9 transcendentals per state).
  io_t* sPark = (io_t*)(smem + (((sDump + NT + N) - smem + 3) & ~3));     // [4][NT][T], 16-byte aligned
  // DMABC: the B/C tile of the NEXT chunk to be processed (c - 1) arrives by LDS-DMA, raw io dtype [B|C][N][CH], while chunk c is
  // in its state loop; at the top of a chunk the staging pass converts it LDS -> LDS.  Staged from global memory at the top of the
  // chunk (as before, and still for run-time dstate / unaligned rows) the tile cost one fully exposed HBM round trip per chunk:
  // the one workgroup of a CU has nothing else to run meanwhile.  The checkpoint of chunk c - 1 is prefetched into a register.
  // (A/B in one process, profiles/r03_scan_fold.txt: 1157.7 -> 1128.0 us at the pre-training shape; the folded walk got SLOWER,
  //  895.7 -> 943.3 us at B64 x L200, and keeps the global staging)
  constexpr bool DMABC = VEC && NS == 16 && sizeof(io_t) == 2 && (!FOLD || DMAR);     // (fp32 rows: the parked rows already fill the LDS)
  // DMAR (16-bit rows, L % 8 == 0, io-dtype dout; the launcher decides): the ROWS of the next chunk (u, delta, z, dout) and its
  // checkpoint arrive by LDS-DMA as well, into the park buffer -- a wave requests exactly the 16 bytes per lane it will read
  // back, so only its own vmcnt orders them -- while the rows of the CURRENT chunk stay packed in 16 VGPRs (read from the park
  // buffer at the top of the chunk, unpacked there and again after the state loop; a second park buffer does not fit: 165 KB).
  // No compiler-visible load is left in the chunk loop: hipcc waited for the register prefetch (`s_waitcnt vmcnt(1)` right
  // behind the loads: they sit under a per-lane condition) and again for everything outstanding (`vmcnt(0)`) before the state
  // loop, so nothing of the next chunk was in flight while it ran.  The one wait of a chunk sits AFTER the state loop, where the
  // requests are ~30 us old, and no longer at the chunk top, where it also waited for the acknowledgements of the previous
  // chunk's stores and atomics.
  static_assert(!DMAR || DMABC, "row DMA rides on the B/C DMA path");
  constexpr int PARKN = 4 * NT * T;                                        // elements of one park buffer
  io_t* sRawBC = sPark + (size_t)PARKN;                   // [2][N][CH] io dtype (DMABC)
  float* sCk = (float*)(sRawBC + 2 * 16 * CH);                             // [NT] checkpoint entries of the next chunk (DMAR)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = __builtin_amdgcn_readfirstlane(wave);     // the wave index as a scalar: flush addresses are SGPR + lane
  const int r = lane >> 4, j = lane & 15;
  const int row = wave * RPW + r;
  const int b = FOLD ? b0 : blockIdx.y;            // FOLD: first batch element of the part (base pointers start there)
  const int dpg = p.dim / p.G;
  const int tiles = (dpg + DT - 1) / DT;
  const int g = blockIdx.x / tiles;
  const int d0 = g * dpg + (blockIdx.x - g * tiles) * DT;
  const int d_end = (g + 1) * dpg;
  const int d = d0 + row;
  const bool row_ok = d < d_end;
  const int dc = row_ok ? d : d_end - 1;

  const int dr = delta_row(dc, p.dl_ratio, p.dl_magic);   // delta / delta_bias row read by this channel (gradients stay per channel)
  const io_t* __restrict__ pu = (const io_t*)p.u + (int64_t)b * p.u_bs + (int64_t)dc * p.u_ds + j * T;
  const io_t* __restrict__ pd = (const io_t*)p.delta + (int64_t)b * p.dl_bs + (int64_t)dr * p.dl_ds + j * T;
  const io_t* __restrict__ pz = p.z ? (const io_t*)p.z + (int64_t)b * p.z_bs + (int64_t)dc * p.z_ds + j * T : nullptr;
  const int64_t pg_off = (int64_t)b * p.do_bs + (int64_t)dc * p.do_ds + j * T;   // element offset into dout (io dtype, or fp32)
  const io_t* __restrict__ pg = (const io_t*)p.dout + pg_off;                       // io-dtype view (not used when dout is fp32)
  // An fp32 dout (i16o32) is loaded by ordinary register loads around the state loop.  The folded and the row-DMA
  // instantiations exclude it at COMPILE time (their launchers refuse such calls): s_waitcnt insertion is static, so the mere
  // presence of that branch put a `vmcnt(0)` at the first use of its registers after the join -- on every call, where it waited
  // for the LDS-DMA of the next chunk before the state loop had even started.
  const bool of32 = (FOLD || DMAR) ? false : p.out_f32 != 0;
  io_t* __restrict__ qdu = (io_t*)p.du + (int64_t)b * p.du_bs + (int64_t)dc * p.du_ds + j * T;
  io_t* __restrict__ qdd = (io_t*)p.ddelta + (int64_t)b * p.dd_bs + (int64_t)dc * p.dd_ds + j * T;
  io_t* __restrict__ qdz = p.dz ? (io_t*)p.dz + (int64_t)b * p.dz_bs + (int64_t)dc * p.dz_ds + j * T : nullptr;
  const io_t* __restrict__ Bp = (const io_t*)p.B + (int64_t)b * p.B_bs + (int64_t)g * p.B_gs;
  const io_t* __restrict__ Cp = (const io_t*)p.C + (int64_t)b * p.C_bs + (int64_t)g * p.C_gs;
  float* __restrict__ dBp = p.dB + (int64_t)b * p.dB_bs + (int64_t)g * p.dB_gs;
  float* __restrict__ dCp = p.dC + (int64_t)b * p.dC_bs + (int64_t)g * p.dC_gs;
  const bool has_z = pz != nullptr;
  auto seg_of = [&](int tv) { return (int)__umulhi((unsigned)tv, p.fold_magic); };
  auto fold_off = [&](int tv, int64_t bs) {        // element offset of virtual step tv: (segment tv / SL) * bs + tv % SL
    const int sb = seg_of(tv);
    return seg_off(sb, bs, tv - sb * SL);
  };

  const float bias = p.bias ? p.bias[dr] : 0.0f;
  const float Dv = p.D ? p.D[dc] : 0.0f;
  float dD_acc = 0.0f, dbias_acc = 0.0f;

  // u, delta, dout, z of one chunk, all requests issued before the first use: ONE whole/partial branch for the four arrays.
  // (Four independent fetch calls, each with its own full / whole-lane / element-wise paths, made the compiler wait for every
  // array before requesting the next -- four serialised HBM round trips on the ragged chunk, which is HALF of the chunks of the
  // 197-token encoders.)  dout is the io dtype, or fp32 (MXVL_SCAN_OUT_F32: oflex i16o32).
  auto ld8 = [&](const io_t* q, float (&v)[T]) {
    const float4 a0 = ld4<io_t>(q), a1 = ld4<io_t>(q + 4);
    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
  };
  auto dout_fetch = [&](int t0, float (&v)[T]) {      // the fp32 dout of an i16o32 call (never prefetched / parked)
    if (VEC && (t0 + CH <= L || t0 + j * T + T <= L)) {
      const float4 a0 = ld4_out<io_t>(p.dout, pg_off + t0, true), a1 = ld4_out<io_t>(p.dout, pg_off + t0 + 4, true);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) v[i] = (t0 + j * T + i < L) ? ld_out<io_t>(p.dout, pg_off + t0 + i, true) : 0.0f;
    }
  };
  auto rows_fetch = [&](int t0, float (&vu)[T], float (&vd)[T], float (&vg)[T], float (&vz)[T]) {
    if constexpr (FOLD) {
      // lanes past the end read step 0 (finite values; delta, d softplus and dout are zeroed for them below): unconditional loads
      const int tv = t0 + j * T < L ? t0 + j * T : 0;
      const int sb = seg_of(tv), sl = tv - sb * SL;       // (the row pointers already carry + j * T)
      ld8(pu - j * T + seg_off(sb, p.u_bs, sl), vu);
      ld8(pd - j * T + seg_off(sb, p.dl_bs, sl), vd);
      if (!of32) ld8(pg - j * T + seg_off(sb, p.do_bs, sl), vg);
      if (has_z) ld8(pz - j * T + seg_off(sb, p.z_bs, sl), vz);
    } else if (VEC && (t0 + CH <= L || t0 + j * T + T <= L)) {     // every step of this lane is valid: 16-byte loads
      ld8(pu + t0, vu);
      ld8(pd + t0, vd);
      if (!of32) ld8(pg + t0, vg);
      if (has_z) ld8(pz + t0, vz);
    } else {
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const bool ok = t0 + j * T + i < L;
        vu[i] = ok ? io::ld(pu + t0 + i) : 0.0f;
        vd[i] = ok ? io::ld(pd + t0 + i) : 0.0f;
        if (!of32) vg[i] = ok ? io::ld(pg + t0 + i) : 0.0f;
        if (has_z) vz[i] = ok ? io::ld(pz + t0 + i) : 0.0f;
      }
    }
    if (of32) dout_fetch(t0, vg);
  };
  auto row_store = [&](io_t* q, const void* base, int64_t bs, int64_t ds, int t0, const float (&v)[T]) {
    if constexpr (FOLD) {
      const int tv = t0 + j * T;
      if (row_ok && tv < L) {
        io_t* w = q + fold_off(tv, bs) - j * T;
        st4<io_t>(w, make_float4(v[0], v[1], v[2], v[3]));
        st4<io_t>(w + 4, make_float4(v[4], v[5], v[6], v[7]));
      }
    } else if (VEC && t0 + CH <= L) {
      if (row_ok) {
        st4<io_t>(q + t0, make_float4(v[0], v[1], v[2], v[3]));
        st4<io_t>(q + t0 + 4, make_float4(v[4], v[5], v[6], v[7]));
      }
    } else if (VEC) {   // ragged last chunk of aligned rows
      if (row_ok) {
        if (t0 + j * T + T <= L) {
          st4<io_t>(q + t0, make_float4(v[0], v[1], v[2], v[3]));
          st4<io_t>(q + t0 + 4, make_float4(v[4], v[5], v[6], v[7]));
        } else {
#pragma unroll
          for (int i = 0; i < T; ++i)
            if (t0 + j * T + i < L) io::st(q + t0 + i, v[i]);
        }
      }
    } else {
      float4* so4 = (float4*)(sO + row * CH + j * T);
      so4[0] = make_float4(v[0], v[1], v[2], v[3]);
      so4[1] = make_float4(v[4], v[5], v[6], v[7]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int wrow = wave * RPW + rr;
        const int dd = d0 + wrow;
        io_t* w = (io_t*)base + (int64_t)b * bs + (int64_t)dd * ds + t0;
#pragma unroll
        for (int e = lane; e < CH; e += 64)
          if (dd < d_end && t0 + e < L) io::st(w + e, sO[wrow * CH + e]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  };

  // Half-precision io: the NEXT chunk's rows (u, delta, dout, z: 8 elements = one 16-byte load each) are requested before the
  // state loop of the current chunk and sit packed in 16 VGPRs while it runs, so their HBM latency is hidden behind it
  // (at 2 waves/SIMD there is little else to hide it); the first chunk's are requested at kernel entry, ahead of the table
  // initialisation.  fp32 rows would need 32 VGPRs: they take the direct path.
  constexpr bool PF = VEC && sizeof(io_t) == 2;
  uint4 ru = make_uint4(0, 0, 0, 0), rd = ru, rg = ru, rz = ru;
  auto unpack = [&](const uint4& r, float (&v)[T]) {
    io_t tmp[T];
    *(uint4*)tmp = r;
#pragma unroll
    for (int i = 0; i < T; ++i) v[i] = io::ld(tmp + i);
  };
  auto raw_prefetch = [&](int tn) {
    if constexpr (FOLD) {     // unconditional (a per-lane `if` makes the compiler wait for the loads at once); lanes past the end: step 0
      const int tv = tn + j * T < L ? tn + j * T : 0;
      const int sb = seg_of(tv), sl = tv - sb * SL;
      ru = *(const uint4*)(pu - j * T + seg_off(sb, p.u_bs, sl));
      rd = *(const uint4*)(pd - j * T + seg_off(sb, p.dl_bs, sl));
      rg = *(const uint4*)(pg - j * T + seg_off(sb, p.do_bs, sl));
      if (has_z) rz = *(const uint4*)(pz - j * T + seg_off(sb, p.z_bs, sl));
    } else if (tn + CH <= L || tn + j * T + T <= L) {
      ru = *(const uint4*)(pu + tn);
      rd = *(const uint4*)(pd + tn);
      if (!of32) rg = *(const uint4*)(pg + tn);
      if (has_z) rz = *(const uint4*)(pz + tn);
    } else {                    // the lane that straddles the end of the row: element loads, zero beyond L
      uint16_t eu[T], ed[T], eg[T], ez[T];
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const bool ok = tn + j * T + i < L;
        eu[i] = ok ? *(const uint16_t*)(pu + tn + i) : (uint16_t)0;
        ed[i] = ok ? *(const uint16_t*)(pd + tn + i) : (uint16_t)0;
        eg[i] = (ok && !of32) ? *(const uint16_t*)(pg + tn + i) : (uint16_t)0;
        ez[i] = (ok && has_z) ? *(const uint16_t*)(pz + tn + i) : (uint16_t)0;
      }
      auto pk = [](const uint16_t (&e)[T]) {
        return make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                          (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
      };
      ru = pk(eu); rd = pk(ed); rg = pk(eg); rz = pk(ez);
    }
  };

  const int nchunks = (L + CH - 1) / CH;
  // one DMA instruction = 64 lanes x 16 bytes = 1 KB of the raw tile; lane -> (array, state row, first step) of its 16 bytes
  constexpr int EPL = 16 / (int)sizeof(io_t);                   // elements per lane and instruction
  constexpr int NDMA = 2 * 16 * CH * (int)sizeof(io_t) / 1024;  // instructions per tile (dstate 16): 8 (16-bit) / 16 (fp32)
  constexpr int DPW = (NDMA + NWAVES - 1) / NWAVES;             // per wave
  // (array, state row, first step) of a lane's 16 bytes do not depend on the chunk: the row pointer and the batch stride are set
  // up once (read inside the loop, two of these kernel arguments came back as VECTOR loads from the kernarg segment, each with
  // a vmcnt wait that also waited for the DMA requests issued just before)
  const io_t* bc_g[DPW];
  int64_t bc_bs[DPW];
#pragma unroll
  for (int k = 0; k < DPW; ++k) {
    const int q = wq * DPW + k;                                  // wave-uniform 1 KB block of the raw tile
    const int e0 = q * (1024 / (int)sizeof(io_t)) + lane * EPL;  // element index inside [B|C][16][CH]
    const int arr = e0 / (16 * CH), n = (e0 / CH) & 15;
    bc_g[k] = arr ? Cp + (int64_t)n * p.C_ns : Bp + (int64_t)n * p.B_ns;
    bc_bs[k] = arr ? p.C_bs : p.B_bs;
  }
  auto bc_dma = [&](int t0) {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)((char*)sRawBC);
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      const int q = wq * DPW + k;
      if (q < NDMA) {
        int tv = t0 + ((q * (1024 / (int)sizeof(io_t)) + lane * EPL) & (CH - 1));
        tv = tv < L ? tv : 0;                                    // past the end: step 0 (finite values under delta = 0)
        const io_t* g = bc_g[k];
        if constexpr (FOLD) g += fold_off(tv, bc_bs[k]); else g += tv;
        const unsigned dst = lds0 + (unsigned)q * 1024u;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g], off\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [dst] "s"(dst), [g] "v"(g)
            : "memory", "scc");
      }
    }
  };
  float ck_next = 0.0f;                                          // checkpoint (row, state) = lane of this wave, for the next chunk
  auto ckpt_prefetch = [&](int c) {
    const int rr = lane / N, n = lane - rr * N;                  // RPW * N == 64: one entry per lane
    const int dd = d0 + wave * RPW + rr;
    ck_next = 0.0f;
    if (c > 0 && dd < d_end) {
      const int64_t slot = FOLD ? ((int64_t)dd * gridDim.y + blockIdx.y) * p.fold_cpp + c : ((int64_t)b * p.dim + dd) * p.n_ckpt + c;
      ck_next = p.ckpt[slot * N + n];
    }
  };
  auto dma_to = [&](unsigned dst, const void* g, auto wide) {     // one LDS-DMA instruction: lane i -> dst + i * (16 | 4) bytes
    unsigned keep;
    if constexpr (decltype(wide)::value) {
      asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g], off\n\ts_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep) : [dst] "s"(dst), [g] "v"(g) : "memory", "scc");
    } else {
      asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dword %[g], off\n\ts_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep) : [dst] "s"(dst), [g] "v"(g) : "memory", "scc");
    }
  };
  auto lds_addr = [](const void* q) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)q; };
  // rows of chunk [tn, tn + CH) into park buffer `buf`, layout [array][thread][T]: a wave's lanes are 16 bytes apart
  auto rows_dma = [&](int tn) {
    // L % 8 == 0: a lane is inside the row or past it as a whole; past the end it reads step 0 (finite; delta, d softplus and
    // dout are zeroed at use)
    const unsigned w0 = lds_addr(sPark) + (unsigned)wq * 1024u;
    if constexpr (FOLD) {
      const int tv = tn + j * T < L ? tn + j * T : 0;
      const int sb = seg_of(tv), sl = tv - sb * SL;       // (the row pointers already carry + j * T)
      dma_to(w0, pu - j * T + seg_off(sb, p.u_bs, sl), std::true_type{});
      dma_to(w0 + NT * 16u, pd - j * T + seg_off(sb, p.dl_bs, sl), std::true_type{});
      if (has_z) dma_to(w0 + 2u * NT * 16u, pz - j * T + seg_off(sb, p.z_bs, sl), std::true_type{});
      dma_to(w0 + 3u * NT * 16u, pg - j * T + seg_off(sb, p.do_bs, sl), std::true_type{});
    } else {
      const int off = (tn + j * T + T <= L) ? tn : -j * T;
      dma_to(w0, pu + off, std::true_type{});
      dma_to(w0 + NT * 16u, pd + off, std::true_type{});
      if (has_z) dma_to(w0 + 2u * NT * 16u, pz + off, std::true_type{});
      dma_to(w0 + 3u * NT * 16u, pg + off, std::true_type{});
    }
  };
  auto ckpt_dma = [&](int c) {                              // checkpoint entering chunk c > 0: (row, state) = lane of this wave
    const int rr = lane / N, n = lane - rr * N;
    int dd = d0 + wave * RPW + rr;
    dd = dd < d_end ? dd : d_end - 1;
    const int64_t slot = FOLD ? ((int64_t)dd * gridDim.y + blockIdx.y) * p.fold_cpp + c : ((int64_t)b * p.dim + dd) * p.n_ckpt + c;
    dma_to(lds_addr(sCk) + (unsigned)wq * 256u, p.ckpt + slot * N + n, std::false_type{});
  };
  if constexpr (DMAR) {
    rows_dma((nchunks - 1) * CH);
    bc_dma((nchunks - 1) * CH);
    if (nchunks > 1) ckpt_dma(nchunks - 1);
  } else {
    if constexpr (DMABC) {
      bc_dma((nchunks - 1) * CH);
      ckpt_prefetch(nchunks - 1);
    }
    if constexpr (PF) raw_prefetch((nchunks - 1) * CH);
  }
  for (int i = tid; i < (DT + 1) * N; i += NT) {
    const int rr = i / N, n = i - rr * N;
    const int dd = d0 + rr;
    sAC[rr * NP + n] = make_float2((rr < DT && dd < d_end) ? p.A[(int64_t)dd * p.A_ds + (int64_t)n * p.A_ns] * kLog2e : 0.0f, 0.0f);
    sG[rr * NP + n] = 0.0f;
    if (rr < DT) sdA[rr * NP + n] = 0.0f;
  }
  for (int i = tid; i < NT + N; i += NT) sDump[i] = 0.0f;

  // ---- chunk-invariant addresses of the state loop ------------------------------------------------------------------
  float2* const ac = sAC + row * NP;                                  // {A2, state entering the chunk} of this row
  const float2* const ac_in = sAC + ((j == 0) ? row : DT) * NP;       // only lane 0 sees the state entering the chunk
  const float* const gq_in = sG + ((j == LPR - 1) ? row : DT) * NP;   // only lane 15 sees the adjoint entering from chunk c+1
  float* const gq_w = (j == 0) ? sG + row * NP : sDump + tid;         // lane 0 hands the adjoint to chunk c-1; the others: dump
  float* const dA_w = (j == LPR - 1) ? sdA + row * NP : sDump + tid;  // lane 15 owns dA of (row, n); the others: dump
  const float* const cB = sB + j * 4;      // even quarter (steps 8j..8j+3) at word 4j, odd quarter at 64 + ((4j + 16) & 63)
  const float* const cC = sC + j * 4;
  const int q1 = 64 + ((j * 4 + 16) & 63) - j * 4;     // word offset of the odd quarter relative to cB / cC
  // A wave first sums the dB / dC shares of its 4 rows in registers (v_permlane{32,16}_swap), so only ONE
  // share per wave goes to LDS; states are flushed in groups of FG, one group behind -- sAcc is two buffers of
  // [FG][NWAVES][2][CH] -- so the sums of group g run while group g+1 is computed and one barrier per FG states separates a
  // buffer's writers from its readers.  (History, profiles/r02_bwd_ablation.txt: per-row tiles + a two-barrier flush every 2
  // states cost 30 % of the kernel in LDS stores, flush reads and barriers; per-wave global atomics without LDS, 8x the
  // atomics, ran 4x slower.)
  constexpr int FK = FG * 2 * CH / NT;          // elements of a group per thread: 2 (8 waves) / 4 (4 waves)
  constexpr int GBUF = FG * NWAVES * 2 * CH;    // floats per flush buffer
  // lane (r, j) holds step e = 8 j + i, i = {0, 1, 4, 5}[r] (and i + 2).  Inside its 8-word block a step sits rotated by
  // 2 (j >> 2): the 32 lanes of one ds_write_b32 group (two rows x 16 j) then hit 32 different banks -- at position e the four
  // j that are 4 apart collided 4-way, 50 M of the kernel's 55 M conflict cycles -- and the flush's consecutive-e reads stay
  // conflict-free (a 32-step run keeps j >> 2 fixed).
  const int i0 = (r & 1) + (r >> 1) * 4, rot = 2 * (j >> 2);
  float* const wAcc = sAcc + wave * (2 * CH) + j * T;
  float* const wA0 = wAcc + ((i0 + rot) & 7);
  float* const wA2 = wAcc + ((i0 + 2 + rot) & 7);
  // element x = wq * 64 + lane + NT * k of a flush group: state slot x / (2 CH), dB|dC (x / CH) & 1, step x % CH.  All but the
  // lane are wave-uniform (64 divides CH): slot, array and the first step of the wave's 64-step run stay in SGPRs.
  int fo[FK];                                   // LDS word offset (inside a flush buffer) of this thread's element k
#pragma unroll
  for (int k = 0; k < FK; ++k) {
    const int xw = wq * 64 + NT * k;
    const int xe = (xw & (2 * CH - 1)) + lane;                        // dB|dC * CH + step; the step's slot is rotated (see wAcc)
    fo[k] = (xw / (2 * CH)) * (NWAVES * 2 * CH) + (xe & ~7) + ((xe + 2 * ((xe % CH) >> 5)) & 7);
  }

  const int NGRP = (N + FG - 1) / FG;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int t0 = c * CH;
    const bool full = t0 + CH <= L;
    if constexpr (DMABC) {     // this wave's share of the B/C DMA has landed ... (DMAR: waited for after the previous state loop)
      if (!DMAR || c == nchunks - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // previous chunk: accumulators flushed, B/C tile free (first pass: init visible); ... and everybody's share
    // row data first: their HBM latency overlaps the B/C staging below (one exposed round trip per chunk, not two)
    float uu[T], dl[T], zz[T], go[T];
    auto unpark = [&](int arr, float (&v)[T]) {
      const io_t* q = sPark + ((size_t)arr * NT + tid) * T;
      const float4 a0 = ld4<io_t>(q), a1 = ld4<io_t>(q + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    };
    if constexpr (DMAR) {        // the DMA of the previous chunk's state loop landed here; from now on the rows live in ru .. rz
      ru = *(const uint4*)(sPark + ((size_t)0 * NT + tid) * T);
      rd = *(const uint4*)(sPark + ((size_t)1 * NT + tid) * T);
      rg = *(const uint4*)(sPark + ((size_t)3 * NT + tid) * T);
      if (has_z) rz = *(const uint4*)(sPark + ((size_t)2 * NT + tid) * T);
    }
    if constexpr (PF) {
      unpack(ru, uu);
      unpack(rd, dl);
      if (of32) dout_fetch(t0, go); else unpack(rg, go);
      if (has_z) unpack(rz, zz);
    } else {
      rows_fetch(t0, uu, dl, go, zz);
    }
    // ---- B/C tile of this chunk + state entering the chunk ---------------------------------------------
    if constexpr (DMABC) {       // the raw tile landed before the barrier that opened this chunk
      for (int i = tid; i < N * (CH / 4); i += NT) {
        const int n = i / (CH / 4), e = (i % (CH / 4)) * 4;
        const float4 bv = ld4<io_t>(sRawBC + n * CH + e);
        const float4 cv = ld4<io_t>(sRawBC + 16 * CH + n * CH + e);
        const int pos = n * CH + ((e >> 2) & 1) * 64 + (((e >> 3) * 4 + ((e >> 2) & 1) * 16) & 63);
        *(float4*)(sB + pos) = bv;
        *(float4*)(sC + pos) = cv;
      }
      if constexpr (DMAR) sAC[(wave * RPW + lane / N) * NP + (lane % N)].y = c > 0 ? sCk[tid] : 0.0f;
      else sAC[(wave * RPW + lane / N) * NP + (lane % N)].y = ck_next;
    } else if (FOLD || (VEC && full)) {
      for (int i = tid; i < N * (CH / 4); i += NT) {   // 16-byte loads, 4 consecutive steps per thread
        const int n = i / (CH / 4), e = (i % (CH / 4)) * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cv = bv;
        if constexpr (FOLD) {
          if (t0 + e < L) {       // a quarter lies inside one segment (SL % 4 == 0)
            bv = ld4<io_t>(Bp + (int64_t)n * p.B_ns + fold_off(t0 + e, p.B_bs));
            cv = ld4<io_t>(Cp + (int64_t)n * p.C_ns + fold_off(t0 + e, p.C_bs));
          }
        } else {
          bv = ld4<io_t>(Bp + (int64_t)n * p.B_ns + t0 + e);
          cv = ld4<io_t>(Cp + (int64_t)n * p.C_ns + t0 + e);
        }
        const int pos = n * CH + ((e >> 2) & 1) * 64 + (((e >> 3) * 4 + ((e >> 2) & 1) * 16) & 63);   // odd quarter rotated by 16 words: conflict-free staging writes (scan_fwd_stream.h qpos)
        *(float4*)(sB + pos) = bv;
        *(float4*)(sC + pos) = cv;
      }
    } else {
      for (int i = tid; i < N * CH; i += NT) {
        const int n = i / CH, e = i - n * CH;
        const int t = t0 + e;
        float bv = 0.0f, cv = 0.0f;
        if (t < L) {
          bv = io::ld(Bp + (int64_t)n * p.B_ns + t);
          cv = io::ld(Cp + (int64_t)n * p.C_ns + t);
        }
        const int pos = n * CH + ((e >> 2) & 1) * 64 + (((e >> 3) * 4 + ((e >> 2) & 1) * 16) & 63) + (e & 3);
        sB[pos] = bv;
        sC[pos] = cv;
      }
    }
    if constexpr (!DMABC) {
      for (int i = lane; i < RPW * N; i += 64) {
        const int rr = i / N, n = i - rr * N;
        const int dd = d0 + wave * RPW + rr;
        float h0 = 0.0f;
        if (c > 0 && dd < d_end) {
          const int64_t slot = FOLD ? ((int64_t)dd * gridDim.y + blockIdx.y) * p.fold_cpp + c : ((int64_t)b * p.dim + dd) * p.n_ckpt + c;
          h0 = p.ckpt[slot * N + n];
        }
        sAC[(wave * RPW + rr) * NP + n].y = h0;
      }
    }
    __syncthreads();
    if constexpr (DMAR) {
      if (c > 0) {
        rows_dma(t0 - CH);
        bc_dma(t0 - CH);
        if (c > 1) ckpt_dma(c - 1);
      }
    } else {
      if constexpr (PF) {
        if (c > 0) raw_prefetch(t0 - CH);     // chunk c-1 is always a full chunk
      }
      if constexpr (DMABC) {                  // the raw tile has been read: the next one may overwrite it while this chunk computes
        if (c > 0) {
          bc_dma(t0 - CH);
          ckpt_prefetch(c - 1);
        }
      }
    }

    if constexpr (!DMAR) {
      auto park = [&](int arr, const float (&v)[T]) {
        io_t* q = sPark + ((size_t)arr * NT + tid) * T;
        st4<io_t>(q, make_float4(v[0], v[1], v[2], v[3]));
        st4<io_t>(q + 4, make_float4(v[4], v[5], v[6], v[7]));
      };
      park(0, uu);
      park(1, dl);
      if (!of32) park(3, go);       // an fp32 dout is re-read after the loop instead (parking would round it to the io dtype)
      if (has_z) park(2, zz);
    }
    float du[T], dy[T], y[T], dsp[T], sgB[T], sAh[T];
    // uniform flag tests outside the per-step loops (a branch per step serialises the transcendental chains)
#pragma unroll
    for (int i = 0; i < T; ++i) { dl[i] += bias; dsp[i] = 1.0f; }
    if (p.softplus) {
#pragma unroll
      for (int i = 0; i < T; ++i) {
        dsp[i] = (dl[i] > 20.0f) ? 1.0f : sigmoid(dl[i]);  // d softplus / dx
        dl[i] = softplus(dl[i]);
      }
    }
    if (!full) {
#pragma unroll
      for (int i = 0; i < T; ++i)
        if (!(t0 + j * T + i < L)) { dl[i] = 0.0f; dsp[i] = 0.0f; if (FOLD || DMAR) go[i] = 0.0f; }   // FOLD / DMAR: those lanes loaded step 0
    }
    if (!row_ok) {  // clamped duplicate rows must not add into the shared dB/dC tile
#pragma unroll
      for (int i = 0; i < T; ++i) go[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < T; ++i) dy[i] = go[i];
    if (has_z) {
#pragma unroll
      for (int i = 0; i < T; ++i) dy[i] = go[i] * silu(zz[i]);
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
      du[i] = dl[i] * uu[i];
      y[i] = 0.0f;                       // D * u joins after the state loop (u is parked)
      sgB[i] = 0.0f;
      sAh[i] = 0.0f;
    }
    float dsum = 0.0f;
#pragma unroll
    for (int i = 0; i < T; ++i) dsum += dl[i];
    // FOLD: a lane whose first step opens a segment: a_0 = P = 0 (exp2(-inf)); flush lanes know the (segment, step) of their element
    float rbias = 0.0f;
    int64_t f_off = 0;
    bool f_ok = true;
    if constexpr (FOLD) {
      const int tv = t0 + j * T;
      rbias = (tv - seg_of(tv) * SL == 0) ? -__builtin_inff() : 0.0f;
      const int tf = t0 + ((wq * 64) & (CH - 1)) + lane;       // the step of this thread's flush elements (the same for every k)
      f_ok = tf < L;
      const int sb = seg_of(tf);
      f_off = (int64_t)sb * p.dB_bs + (tf - sb * SL);          // dB and dC share their strides (checked by the launcher)
    }

    // pair views of the per-chunk arrays for the packed body (PKB)
    v2f dl2[4], du2[4], dy2[4], y2[4], sgB2[4], sAh2[4];
    if constexpr (PKB) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dl2[q] = v2f{dl[2 * q], dl[2 * q + 1]};
        du2[q] = v2f{du[2 * q], du[2 * q + 1]};
        dy2[q] = v2f{dy[2 * q], dy[2 * q + 1]};
        y2[q] = sgB2[q] = sAh2[q] = v2f{0.0f, 0.0f};
      }
    }
    auto flush_load = [&](int grp, float (&part)[FK * NWAVES]) {
      const float* src = sAcc + (grp & 1) * GBUF;
#pragma unroll
      for (int k = 0; k < FK; ++k) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) part[k * NWAVES + w] = src[fo[k] + w * 2 * CH];
      }
    };
    auto flush_add = [&](int grp, float (&part)[FK * NWAVES]) {
#pragma unroll
      for (int k = 0; k < FK; ++k) {
#pragma unroll
        for (int w = NWAVES / 2; w > 0; w >>= 1) {
#pragma unroll
          for (int rr = 0; rr < w; ++rr) part[k * NWAVES + rr] += part[k * NWAVES + rr + w];
        }
        const int xw = wq * 64 + NT * k;                              // wave-uniform part of the element index
        const int nn = grp * FG + xw / (2 * CH), eb = xw & (CH - 1);
        if (nn < N) {
          float* dst = ((xw / CH) & 1) ? dCp + (int64_t)nn * p.dC_ns : dBp + (int64_t)nn * p.dB_ns;   // SGPRs
          if constexpr (FOLD) {
            if (f_ok) unsafeAtomicAdd(dst + f_off, part[k * NWAVES]);
          } else {
            dst += t0 + eb;
            if (full || t0 + eb + lane < L) unsafeAtomicAdd(dst + lane, part[k * NWAVES]);
          }
        }
      }
    };

    for (int ng = 0; ng < NGRP; ++ng) {
      float fpart[FK * NWAVES];
      if (ng > 0) flush_load(ng - 1, fpart);               // the previous group's shares are summed during this group's first state
      float* const wg0 = wA0 + (ng & 1) * GBUF;
      float* const wg2 = wA2 + (ng & 1) * GBUF;
#pragma unroll
      for (int k = 0; k < FG; ++k) {
        const int n = ng * FG + k;
        if (NS > 0 || n < N) {
          if constexpr (PKB) {
            // Steps (2k, 2k + 1) of the state in the halves of an aligned register pair: every element-wise product / FMA of the
            // recompute and of the adjoint pass is ONE v_pk_*_f32 over the pair (12 of the 16 operations per step; the four
            // recurrences -- fold, states, adjoint fold, adjoints -- stay scalar chains over the halves).  The adjoint itself is
            // carried, g_i = a_{i+1} g_{i+1} + C_i dy_i, so a_i g_i is element-wise too; h_{i-1} next to h_i takes one pair move per
            // two steps.  690 instead of 811 instructions per four states.  Measured per instantiation, interleaved in one process
            // (profiles/r06_scan_bwd_pk_ab.txt): the batch-folded walk 799 -> 751 us (B64 x D4096 x L200, the 4-direction encoders),
            // fp32 rows 1306 -> 1289 -- but the row-DMA kernel of the pre-training shape 1003 -> 1036 (the instruction mix is not what
            // bounds it: 4.3), so only the folded instantiations take this body.  Gradients equal to rounding (<= 2e-5 relative L2 in
            // bf16 outputs, <= 3e-7 in the fp32 accumulators).
            const float A2 = ac[n].x;
            const float hin = ac_in[n].y;
            const float gin = gq_in[n];
            const float dA_old = dA_w[n];
            v2f a2[4], b2[4], c2[4], h2[4], g2[4];
            {
              const float4 b0 = *(const float4*)(cB + n * CH), b1 = *(const float4*)(cB + n * CH + q1);
              const float4 c0 = *(const float4*)(cC + n * CH), c1 = *(const float4*)(cC + n * CH + q1);
              b2[0] = v2f{b0.x, b0.y}; b2[1] = v2f{b0.z, b0.w}; b2[2] = v2f{b1.x, b1.y}; b2[3] = v2f{b1.z, b1.w};
              c2[0] = v2f{c0.x, c0.y}; c2[1] = v2f{c0.z, c0.w}; c2[2] = v2f{c1.x, c1.y}; c2[3] = v2f{c1.z, c1.w};
            }
            float P = A2 * dsum;
            {
              v2f t[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) t[q] = dl2[q] * A2;
              if constexpr (FOLD) {
                t[0].x += rbias;
                P += rbias;
              }
              float e0, e1, e2, e3, e4, e5, e6, e7, e8;       // outputs not tied to the inputs (scan_fwd_stream.h, PK)
              asm volatile(
                  "v_exp_f32 %0, %9\n v_exp_f32 %1, %10\n v_exp_f32 %2, %11\n v_exp_f32 %3, %12\n v_exp_f32 %4, %13\n"
                  "v_exp_f32 %5, %14\n v_exp_f32 %6, %15\n v_exp_f32 %7, %16\n v_exp_f32 %8, %17\n"
                  : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7), "=&v"(e8)
                  : "v"(t[0].x), "v"(t[0].y), "v"(t[1].x), "v"(t[1].y), "v"(t[2].x), "v"(t[2].y), "v"(t[3].x), "v"(t[3].y), "v"(P));
              a2[0] = v2f{e0, e1}; a2[1] = v2f{e2, e3}; a2[2] = v2f{e4, e5}; a2[3] = v2f{e6, e7};
              P = e8;
            }
            v2f ub[4], cd[4];                     // delta_i u_i B_i and C_i dy_i
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              ub[q] = du2[q] * b2[q];
              cd[q] = c2[q] * dy2[q];
            }
            float hl = ub[0].x;
            hl = fmaf(a2[0].y, hl, ub[0].y);
#pragma unroll
            for (int q = 1; q < 4; ++q) {
              hl = fmaf(a2[q].x, hl, ub[q].x);
              hl = fmaf(a2[q].y, hl, ub[q].y);
            }
            float ql = cd[3].y;
            ql = fmaf(a2[3].y, ql, cd[3].x);
#pragma unroll
            for (int q = 2; q >= 0; --q) {
              ql = fmaf(a2[q + 1].x, ql, cd[q].y);
              ql = fmaf(a2[q].y, ql, cd[q].x);
            }
            ql *= a2[0].x;                   // what this lane hands to its left neighbour for gamma_in = 0
            float Pf = P, x = hin, Pr = P, gx = gin;
            hl = fmaf(P, hin, hl);
            ql = fmaf(P, gin, ql);
            scan16_fwd_rev(hl, Pf, x, ql, Pr, gx);   // x = state entering this lane's steps, gx = a_{next} g_{next} entering from the right
            gq_w[n] = ql;                    // lane 0: leaves the chunk towards chunk c-1
            {
              float hh = x;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                hh = fmaf(a2[q].x, hh, ub[q].x); h2[q].x = hh;
                hh = fmaf(a2[q].y, hh, ub[q].y); h2[q].y = hh;
              }
              float gg = cd[3].y + gx;
              g2[3].y = gg;
              gg = fmaf(a2[3].y, gg, cd[3].x); g2[3].x = gg;
#pragma unroll
              for (int q = 2; q >= 0; --q) {
                gg = fmaf(a2[q + 1].x, gg, cd[q].y); g2[q].y = gg;
                gg = fmaf(a2[q].y, gg, cd[q].x); g2[q].x = gg;
              }
            }
            v2f vB2[4], vC2[4], dA2 = v2f{0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const v2f hp = (q == 0) ? v2f{x, h2[0].x} : __builtin_shufflevector(h2[q > 0 ? q - 1 : 0], h2[q], 1, 2);   // (h_{i-1}, h_i)
              const v2f gha = g2[q] * a2[q] * hp;                      // g_i a_i h_{i-1}
              y2[q] = __builtin_elementwise_fma(c2[q], h2[q], y2[q]);
              vC2[q] = dy2[q] * h2[q];     // dC_{n,t} share of this row
              vB2[q] = g2[q] * du2[q];     // dB_{n,t} share of this row
              sgB2[q] = __builtin_elementwise_fma(g2[q], b2[q], sgB2[q]);
              sAh2[q] = __builtin_elementwise_fma(gha, v2f{A2, A2}, sAh2[q]);
              dA2 = __builtin_elementwise_fma(gha, dl2[q], dA2);
            }
            const float dA_part = row_sum_to_lane15(dA2.x + dA2.y);
            dA_w[n] = dA_old + dA_part;      // lane 15: dA of (row, n) over the chunks
            {
              float vB[8] = {vB2[0].x, vB2[0].y, vB2[1].x, vB2[1].y, vB2[2].x, vB2[2].y, vB2[3].x, vB2[3].y};
              float vC[8] = {vC2[0].x, vC2[0].y, vC2[1].x, vC2[1].y, vC2[2].x, vC2[2].y, vC2[3].x, vC2[3].y};
              lane32_swap_x8(vB, vC);          // (see the unpacked body below)
              const v2f s01 = v2f{vB[0], vB[1]} + v2f{vB[4], vB[5]}, s23 = v2f{vB[2], vB[3]} + v2f{vB[6], vB[7]};
              const v2f s45 = v2f{vC[0], vC[1]} + v2f{vC[4], vC[5]}, s67 = v2f{vC[2], vC[3]} + v2f{vC[6], vC[7]};
              float sr[8] = {s01.x, s01.y, s23.x, s23.y, s45.x, s45.y, s67.x, s67.y};
              lane16_swap_x4(sr);
              float* w0 = wg0 + k * (NWAVES * 2 * CH);
              float* w2 = wg2 + k * (NWAVES * 2 * CH);
              w0[0] = sr[0] + sr[1]; w2[0] = sr[2] + sr[3];
              w0[CH] = sr[4] + sr[5]; w2[CH] = sr[6] + sr[7];
            }
          } else {
            const float A2 = ac[n].x;
            const float hin = ac_in[n].y;
            const float gin = gq_in[n];
            const float dA_old = dA_w[n];
            float a[T], bb[T], cv[T], h[T];
            {
              const float4 b0 = *(const float4*)(cB + n * CH), b1 = *(const float4*)(cB + n * CH + q1);
              const float4 c0 = *(const float4*)(cC + n * CH), c1 = *(const float4*)(cC + n * CH + q1);
              bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
              cv[0] = c0.x; cv[1] = c0.y; cv[2] = c0.z; cv[3] = c0.w; cv[4] = c1.x; cv[5] = c1.y; cv[6] = c1.z; cv[7] = c1.w;
            }
            // ---- forward recompute -----------------------------------------------------------------------
  #pragma unroll
            for (int i = 0; i < T; ++i) a[i] = dl[i] * A2;
            float P = A2 * dsum;
            if constexpr (FOLD) {
              a[0] += rbias;
              P += rbias;
            }
            asm volatile(
                "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n"
                "v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n v_exp_f32 %8, %8\n"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(P));
            float hl = du[0] * bb[0];
  #pragma unroll
            for (int i = 1; i < T; ++i) hl = fmaf(a[i], hl, du[i] * bb[i]);
            // ---- adjoint fold: g_i = C_i dy_i + a_{i+1} g_{i+1} ----------------------------------------------
            float ql = cv[T - 1] * dy[T - 1];
  #pragma unroll
            for (int i = T - 2; i >= 0; --i) ql = fmaf(a[i + 1], ql, cv[i] * dy[i]);
            ql *= a[0];                      // what this lane hands to its left neighbour for gamma_in = 0
            float Pf = P, x = hin, Pr = P, gx = gin;
            hl = fmaf(P, hin, hl);
            ql = fmaf(P, gin, ql);
            scan16_fwd_rev(hl, Pf, x, ql, Pr, gx);   // x = state entering this lane's steps, gx = a_{next} g_{next} entering from the right
            {
              float hh = x;
  #pragma unroll
              for (int i = 0; i < T; ++i) {
                hh = fmaf(a[i], hh, du[i] * bb[i]);
                h[i] = hh;
                y[i] = fmaf(cv[i], hh, y[i]);
              }
            }
            gq_w[n] = ql;                    // lane 0: leaves the chunk towards chunk c-1
            float gg = gx;                   // = a_{i+1} g_{i+1} for i = T-1
            float dA_part = 0.0f;
            float vB[T], vC[T];
  #pragma unroll
            for (int i = T - 1; i >= 0; --i) {
              const float gi = fmaf(cv[i], dy[i], gg);          // g_i
              const float hprev = (i == 0) ? x : h[i - 1];
              const float ga = gi * a[i];                       // a_i g_i
              const float gha = ga * hprev;                     // g_i h_{i-1} a_i
              vC[i] = dy[i] * h[i];   // dC_{n,t} share of this row
              vB[i] = gi * du[i];     // dB_{n,t} share of this row
              sgB[i] = fmaf(gi, bb[i], sgB[i]);
              sAh[i] = fmaf(gha, A2, sAh[i]);
              dA_part = fmaf(gha, dl[i], dA_part);
              gg = ga;
            }
            dA_part = row_sum_to_lane15(dA_part);
            dA_w[n] = dA_old + dA_part;      // lane 15: dA of (row, n) over the chunks
            {
              // rows r and r+2 (lanes l, l+32): register pair (v[i], v[i+4]) -> one register holding v[i] summed in lanes 0-31 and
              // v[i+4] summed in lanes 32-63; then rows r and r+1: pair (s[i], s[i+1]) -> 16-lane rows holding the 4-row sums of
              // steps {i, i+1, i+4, i+5}
              float sr[8];
              lane32_swap_x8(vB, vC);
  #pragma unroll
              for (int i = 0; i < 4; ++i) { sr[i] = vB[i] + vB[i + 4]; sr[4 + i] = vC[i] + vC[i + 4]; }
              lane16_swap_x4(sr);
              float* w0 = wg0 + k * (NWAVES * 2 * CH);
              float* w2 = wg2 + k * (NWAVES * 2 * CH);
              w0[0] = sr[0] + sr[1]; w2[0] = sr[2] + sr[3];
              w0[CH] = sr[4] + sr[5]; w2[CH] = sr[6] + sr[7];
            }
          }
          if (k == 0 && ng > 0) flush_add(ng - 1, fpart);
          if (k == FG - 1 || n == N - 1) __syncthreads();
        }
      }
    }

    if constexpr (PKB) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        y[2 * q] = y2[q].x; y[2 * q + 1] = y2[q].y;
        sgB[2 * q] = sgB2[q].x; sgB[2 * q + 1] = sgB2[q].y;
        sAh[2 * q] = sAh2[q].x; sAh[2 * q + 1] = sAh2[q].y;
      }
    }
    if constexpr (DMAR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next chunk's tiles, requested before the state loop
    {
      {     // the last group's shares
        float fpart[FK * NWAVES];
        flush_load(NGRP - 1, fpart);
        flush_add(NGRP - 1, fpart);
        if (!VEC) __syncthreads();       // unaligned rows: the store transpose tile sO aliases buffer 0
      }
      float raw[T];
      if constexpr (DMAR) {
        unpack(ru, uu);
        unpack(rd, raw);
        unpack(rg, go);
        if (has_z) unpack(rz, zz);
      } else {
        unpark(0, uu);
        unpark(1, raw);
        if (of32) dout_fetch(t0, go); else unpark(3, go);
        if (has_z) unpark(2, zz);
      }
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const float rb = raw[i] + bias;
        dsp[i] = (p.softplus && !(rb > 20.0f)) ? sigmoid(rb) : 1.0f;
        if (!full && !(t0 + j * T + i < L)) dsp[i] = 0.0f;
        y[i] = fmaf(Dv, uu[i], y[i]);
      }
    }
    // ---- per-step outputs --------------------------------------------------------------------------------
    float o_du[T], o_dd[T], o_dz[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      o_du[i] = fmaf(dl[i], sgB[i], dy[i] * Dv);
      const float dd = fmaf(uu[i], sgB[i], sAh[i] * 0.6931471805599453f) * dsp[i];  // A = A2 * ln2
      o_dd[i] = dd;
      dD_acc = fmaf(dy[i], uu[i], dD_acc);
      dbias_acc += dd;
    }
    if (has_z) {
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const float s = sigmoid(zz[i]);
        o_dz[i] = go[i] * y[i] * s * fmaf(zz[i], 1.0f - s, 1.0f);
      }
    }
    row_store(qdu, p.du, p.du_bs, p.du_ds, t0, o_du);
    row_store(qdd, p.ddelta, p.dd_bs, p.dd_ds, t0, o_dd);
    if (has_z) row_store(qdz, p.dz, p.dz_bs, p.dz_ds, t0, o_dz);

  }

  // ---- per-row reductions: dA (LDS, lane 15 wrote), dD, ddelta_bias (registers -> row sum) -----------------
  dD_acc = row_sum_to_lane15(dD_acc);
  dbias_acc = row_sum_to_lane15(dbias_acc);
  if (j == LPR - 1 && row_ok) {
    if (p.dD) unsafeAtomicAdd(p.dD + d, dD_acc);
    if (p.dbias) unsafeAtomicAdd(p.dbias + d, dbias_acc);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // (lane / wave re-derived from the work-item id behind an opaque move: as loop-invariant values computed at the top they were
  //  the two VGPRs the register allocator spilled around the chunk loop -- the kernel's 12 bytes of scratch)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, wave_e = tid_e >> 6;
  for (int i = lane_e; i < RPW * N; i += 64) {
    const int rr = i / N, n = i - rr * N;
    const int dd = d0 + wave_e * RPW + rr;
    // d a / d A = delta * a and A2 = A*log2e only rescales the exponent argument: sdA already holds dA
    if (dd < d_end) unsafeAtomicAdd(p.dA + (int64_t)dd * N + n, sdA[(wave_e * RPW + rr) * NP + n]);
  }
}

static thread_local int g_bwd_hip_error = 0;
extern "C" int mxvl_scan_bwd_variant(void);   // scan_fwd.hip: bits 8..15 of mxvl_set_scan_variant

}  // namespace mxvl
#include "scan_n1_bwd.h"
#define MXVL_N1_SHORT_BWD
#include "scan_n1_short.h"
namespace mxvl {

// dstate 1 without z (VMamba's SS2D): the pass-major kernel of scan_n1_bwd.h.  u / delta / B / C rows need T-element alignment (T = 8
// for 16-bit rows with L % 8 == 0, else 4), a workgroup's 4 rw channels one B / C group; dout / du / ddelta fall back to element
// accesses inside the kernel when their rows are not aligned.  Reads the forward's 128-step checkpoints (any forward kernel's).
template <typename io_t>
static int try_n1_bwd(const ScanBwdArgs& a, hipStream_t stream, bool& taken) {
  taken = false;
  if (a.N != 1 || a.z || a.fold_magic || mxvl_scan_bwd_variant() == 3) return MXVL_OK;     // variant 3: A/B hook, the general kernel
  const int64_t rows = (int64_t)a.batch * a.dim;
  const int dpg = a.dim / a.G;
  if (a.L < 8 || a.L % 4 != 0 || dpg % 4 != 0) return MXVL_OK;
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
  if (a.L > 64 * T && !a.ckpt) return MXVL_OK;             // the state entering a later pass comes from the forward's checkpoints
  constexpr int NW = 4;
  // ~2048 workgroups where the problem allows, at least 8 rows per wave to amortise the dB / dC flush of a pass
  int64_t rw = rows / ((int64_t)NW * 2048);
  if (rw < 8) rw = 8;
  if (rw > 64) rw = 64;
  while (rw > 1 && dpg % (NW * rw) != 0) --rw;
  if (dpg % (NW * rw) != 0) return MXVL_OK;
  ScanN1BwdGeom gm;
  gm.rw = (int)rw;
  const int gesz = a.out_f32 ? 4 : esz;
  auto vec = [&](const void* q, int64_t bs, int64_t ds, int e, int gran) {
    return (bs % gran == 0 && ds % gran == 0 && ((uintptr_t)q) % (size_t)(gran * e) == 0) ? 1 : 0;
  };
  gm.do_vec = vec(a.dout, a.do_bs, a.do_ds, gesz, a.out_f32 ? 4 : T);
  gm.du_vec = vec(a.du, a.du_bs, a.du_ds, esz, 4);
  gm.dd_vec = vec(a.ddelta, a.dd_bs, a.dd_ds, esz, 4);
  const dim3 grid((unsigned)(rows / (NW * rw))), block(NW * 64);
  const size_t lds = sizeof(float) * ((size_t)NW * 2 * T * 64 + (size_t)NW * rw);
  if constexpr (esz == 2) {
    if (a.out_f32) {
      if (T == 8) hipLaunchKernelGGL((scan_n1_bwd_kernel<io_t, 8, NW, true>), grid, block, lds, stream, a, gm);
      else hipLaunchKernelGGL((scan_n1_bwd_kernel<io_t, 4, NW, true>), grid, block, lds, stream, a, gm);
    } else {
      if (T == 8) hipLaunchKernelGGL((scan_n1_bwd_kernel<io_t, 8, NW, false>), grid, block, lds, stream, a, gm);
      else hipLaunchKernelGGL((scan_n1_bwd_kernel<io_t, 4, NW, false>), grid, block, lds, stream, a, gm);
    }
  } else {
    hipLaunchKernelGGL((scan_n1_bwd_kernel<io_t, 4, NW, false>), grid, block, lds, stream, a, gm);
  }
  taken = true;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_bwd_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}

// rows of at most 128 steps that scan_n1_bwd.h cannot take (L % 4 != 0 or unaligned rows): a lane per row (scan_n1_short.h).  Dense
// (batch, dim, L) u / delta / dout / du / ddelta, 64 | dim / n_groups.
template <typename io_t>
static int try_n1_short_bwd(const ScanBwdArgs& a, hipStream_t stream, bool& taken) {
  taken = false;
  if (a.N != 1 || a.z || a.fold_magic || mxvl_scan_bwd_variant() == 3) return MXVL_OK;
  const int dpg = a.dim / a.G;
  const int64_t DL = (int64_t)a.dim * a.L;
  if (a.L < 1 || a.L > 128 || dpg % 64 != 0 || a.dl_ratio > 1) return MXVL_OK;
  if (a.u_ds != a.L || a.u_bs != DL || a.dl_ds != a.L || a.dl_bs != DL || a.do_ds != a.L || a.do_bs != DL ||
      a.du_ds != a.L || a.du_bs != DL || a.dd_ds != a.L || a.dd_bs != DL) return MXVL_OK;
  for (const void* q : {a.u, a.delta, a.dout, (const void*)a.du, (const void*)a.ddelta})
    if (((uintptr_t)q) % 16 != 0) return MXVL_OK;
  constexpr int esz = (int)sizeof(io_t);
  const int gesz = a.out_f32 ? 4 : esz;
  const size_t per_wave = (size_t)64 * a.L * (2 * esz + gesz + sizeof(float)) + (size_t)2 * ((a.L + 3) & ~3) * sizeof(float);
  if (per_wave > 64 * 1024) return MXVL_OK;
  const int64_t waves = (int64_t)a.batch * a.dim / 64;
  const bool two = 2 * per_wave <= 64 * 1024;
  const dim3 grid((unsigned)(two ? (waves + 1) / 2 : waves)), block(two ? 128 : 64);
  const size_t lds = per_wave * (two ? 2 : 1);
  if (a.out_f32 && esz == 2) {
    if (two) hipLaunchKernelGGL((scan_n1_short_bwd_kernel<io_t, 2, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((scan_n1_short_bwd_kernel<io_t, 1, true>), grid, block, lds, stream, a);
  } else {
    if (two) hipLaunchKernelGGL((scan_n1_short_bwd_kernel<io_t, 2, false>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((scan_n1_short_bwd_kernel<io_t, 1, false>), grid, block, lds, stream, a);
  }
  taken = true;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_bwd_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}

template <typename io_t, int NWAVES, bool VEC, int NS, bool FOLD = false, bool DMAR = false>
static int launch_bwd1(const ScanBwdArgs& a, hipStream_t stream) {
  constexpr int DT = NWAVES * 4, CH = 128, NT = NWAVES * 64;
  const size_t lds = sizeof(float) * ((size_t)2 * a.N * CH + (size_t)DT * 2 * 2 * CH + (size_t)3 * (DT + 1) * (a.N + 1) + (size_t)DT * (a.N + 1) +
                                      (size_t)NT + a.N) + 16 + (size_t)4 * NT * 8 * sizeof(io_t) +
                     ((VEC && NS == 16 && sizeof(io_t) == 2 && (!FOLD || DMAR)) ? (size_t)2 * 16 * CH * sizeof(io_t) : 0) +      // + the raw B/C tile of the LDS-DMA prefetch
                     (DMAR ? (size_t)NT * sizeof(float) : 0);                                                           // + its checkpoint entries
  if (lds > 160 * 1024) return MXVL_ERR_UNSUPPORTED;
  auto kern = scan_bwd_kernel<io_t, NWAVES, VEC, NS, FOLD, DMAR>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_bwd_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  }
  const int dpg = a.dim / a.G;
  dim3 grid(a.G * ((dpg + DT - 1) / DT), FOLD ? (a.batch + a.fold_bpp - 1) / a.fold_bpp : a.batch), block(NT);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_bwd_hip_error = (int)e; return MXVL_ERR_LAUNCH; }
  return MXVL_OK;
}
// dstate 16 (every Mamba block of the reference) takes the instantiation with the unrolled state loop
template <typename io_t, int NWAVES, bool VEC>
static int launch_bwd(const ScanBwdArgs& a, hipStream_t stream) {
  if constexpr (VEC && sizeof(io_t) == 2 && NWAVES == 8) {     // rows by LDS-DMA: whole 8-step lanes only.  (The 4-wave kernel, 4 flush
    // elements per thread, spills row pointers at 256 VGPRs and reloads them between the DMA requests: every reload waits for the
    // requests before it.)
    if (a.N == 16 && a.L % 8 == 0 && !a.out_f32) return launch_bwd1<io_t, NWAVES, VEC, 16, false, true>(a, stream);
  }
  return a.N == 16 ? launch_bwd1<io_t, NWAVES, VEC, 16>(a, stream) : launch_bwd1<io_t, NWAVES, VEC, 0>(a, stream);
}

// 8-wave workgroups own 32 rows: the dB/dC tile is pre-summed over twice as many rows before it leaves the workgroup
// at the same 8 waves per CU.  Used when 32-row tiles still give every CU a workgroup.

static bool bwd_wide(int batch, int dim, int G, int /*L*/) {
  static const int forced = MXVL_ABL_ENV("MXVL_BWD_WAVES");
  const long tiles32 = (long)batch * G * ((dim / G + 31) / 32);
  // round 3 (profiles/r03_scan_variants.txt): with the restructured state loop the 32-row workgroups are never slower where
  // they still give every CU a workgroup -- B32 x D768 x L196 fp32: 155 vs 213 us, the 197-token encoder: 1296 vs 1281 us
  return forced ? forced == 8 : ((dim / G) % 32 == 0 && tiles32 >= 256);
}

template <typename io_t>
static int dispatch_bwd(const ScanBwdArgs& a, hipStream_t stream) {
  {
    bool taken = false;
    int rc = try_n1_bwd<io_t>(a, stream, taken);
    if (rc != MXVL_OK || taken) return rc;
    rc = try_n1_short_bwd<io_t>(a, stream, taken);
    if (rc != MXVL_OK || taken) return rc;
  }
  if (a.fold_magic) {   // batch folded into the sequence (MXVL_SCAN_FOLD_BATCH): aligned rows, dstate 16, io-dtype dout
    if (!a.vec_ok || a.N != 16 || a.out_f32 || a.dB_bs != a.dC_bs) return MXVL_ERR_UNSUPPORTED;
    for (int64_t bs : {a.u_bs, a.dl_bs, a.z_bs, a.do_bs, a.du_bs, a.dd_bs, a.dz_bs, a.B_bs, a.C_bs, a.dB_bs})
      if (bs >= (1ll << 32)) return MXVL_ERR_UNSUPPORTED;    // seg_off: 32-bit batch strides
    const long parts = (a.batch + a.fold_bpp - 1) / a.fold_bpp;
    const long tiles32 = parts * a.G * ((a.dim / a.G + 31) / 32);
    const int fv = mxvl_scan_bwd_variant();       // tests: 1 forces the 8-wave walk at small sizes, 2 the 4-wave walk
    if ((a.dim / a.G) % 32 == 0 && (fv == 1 || (fv != 2 && tiles32 >= 256))) {
      if constexpr (sizeof(io_t) == 2) return launch_bwd1<io_t, 8, true, 16, true, true>(a, stream);     // rows by LDS-DMA
      else return launch_bwd1<io_t, 8, true, 16, true>(a, stream);
    }
    return launch_bwd1<io_t, 4, true, 16, true>(a, stream);
  }
  const int variant = mxvl_scan_bwd_variant();   // tests / A-B measurements: 0 automatic
  const bool wide = variant == 1 ? true : variant == 2 ? false : bwd_wide(a.batch, a.dim, a.G, a.L);
  if (wide) return a.vec_ok ? launch_bwd<io_t, 8, true>(a, stream) : launch_bwd<io_t, 8, false>(a, stream);
  return a.vec_ok ? launch_bwd<io_t, 4, true>(a, stream) : launch_bwd<io_t, 4, false>(a, stream);
}

}  // namespace mxvl

using namespace mxvl;

extern "C" int mxvl_scan_check(const mxvl_scan_desc* d);

// The per-tile dB/dC workspace + reduce kernel of ABI v3 measured slower than the fp32 atomics at every shape
// (profiles/r02_bwd_workspace_ab.txt: 1.47 vs 1.27 ms) and was removed in round 3: no scratch is useful, the descriptor's
// workspace fields are ignored.
extern "C" int64_t mxvl_scan_bwd_workspace_bytes(const mxvl_scan_desc* f) {
  (void)f;
  return 0;
}

extern "C" int mxvl_scan_bwd(const mxvl_scan_bwd_desc* d, void* hip_stream) {
  if (!d) return MXVL_ERR_NULL;
  const mxvl_scan_desc* f = &d->fwd;
  int rc = mxvl_scan_check(f);
  if (rc != MXVL_OK) return rc;
  if (!d->dout || !d->du || !d->ddelta || !d->dA || !d->dB || !d->dC) return MXVL_ERR_NULL;
  if (f->z && !d->dz) return MXVL_ERR_NULL;
  if (f->D && !d->dD) return MXVL_ERR_NULL;
  if (f->delta_bias && !d->ddelta_bias) return MXVL_ERR_NULL;
  const int n_ckpt = (f->seqlen + kCkptLenB - 1) / kCkptLenB;
  if (n_ckpt > 1 && !f->ckpt) return MXVL_ERR_CHECKPOINT;
  ScanBwdArgs a;
  a.batch = f->batch; a.dim = f->dim; a.L = f->seqlen; a.N = f->dstate; a.G = f->n_groups; a.n_ckpt = n_ckpt;
  a.softplus = (f->flags & MXVL_SCAN_DELTA_SOFTPLUS) ? 1 : 0;
  a.out_f32 = (f->flags & MXVL_SCAN_OUT_F32) ? 1 : 0;
  a.u_bs = f->u_bs; a.u_ds = f->u_ds; a.dl_bs = f->delta_bs; a.dl_ds = f->delta_ds; a.z_bs = f->z_bs; a.z_ds = f->z_ds;
  a.do_bs = d->dout_bs; a.do_ds = d->dout_ds; a.du_bs = d->du_bs; a.du_ds = d->du_ds;
  a.dd_bs = d->ddelta_bs; a.dd_ds = d->ddelta_ds; a.dz_bs = d->dz_bs; a.dz_ds = d->dz_ds;
  a.B_bs = f->B_bs; a.B_gs = f->B_gs; a.B_ns = f->B_ns; a.C_bs = f->C_bs; a.C_gs = f->C_gs; a.C_ns = f->C_ns;
  a.A_ds = f->A_ds; a.A_ns = f->A_ns;
  a.dB_bs = d->dB_bs; a.dB_gs = d->dB_gs; a.dB_ns = d->dB_ns; a.dC_bs = d->dC_bs; a.dC_gs = d->dC_gs; a.dC_ns = d->dC_ns;
  a.u = f->u; a.delta = f->delta; a.B = f->B; a.C = f->C; a.z = f->z; a.dout = d->dout;
  a.A = (const float*)f->A; a.D = (const float*)f->D; a.bias = (const float*)f->delta_bias; a.ckpt = (const float*)f->ckpt;
  a.du = d->du; a.ddelta = d->ddelta; a.dz = d->dz;
  a.dA = (float*)d->dA; a.dB = (float*)d->dB; a.dC = (float*)d->dC; a.dD = (float*)d->dD; a.dbias = (float*)d->ddelta_bias;
  a.dl_ratio = f->delta_group_ratio > 1 ? f->delta_group_ratio : 1;
  a.dl_magic = delta_magic(a.dl_ratio);
  if (f->dstate > 64) return MXVL_ERR_UNSUPPORTED;
  a.ablate = 0;
  a.fold_magic = 0; a.fold_bpp = 0; a.fold_cpp = 0;
  if (f->flags & MXVL_SCAN_FOLD_BATCH) {       // the forward laid the checkpoints out by virtual chunk: the same geometry here
    if (!mxvl_scan_fold_ok(f->batch, f->seqlen, f->dstate)) return MXVL_ERR_UNSUPPORTED;
    a.fold_magic = scan_fold_magic(f->seqlen);
    a.fold_bpp = scan_fold_bpp(f->batch, f->seqlen, f->dim, f->n_groups);
    a.fold_cpp = (a.fold_bpp * f->seqlen + kCkptLenB - 1) / kCkptLenB;
    if (a.fold_cpp > 1 && !f->ckpt) return MXVL_ERR_CHECKPOINT;
  }
  {
    const int64_t esz = f->io_dtype == MXVL_F32 ? 4 : 2;
    const int64_t strides[] = {f->u_bs, f->u_ds, f->delta_bs, f->delta_ds, f->z ? f->z_bs : 0, f->z ? f->z_ds : 0,
                               d->dout_bs, d->dout_ds, d->du_bs, d->du_ds, d->ddelta_bs, d->ddelta_ds,
                               f->z ? d->dz_bs : 0, f->z ? d->dz_ds : 0,
                               f->B_bs, f->B_gs, f->B_ns, f->C_bs, f->C_gs, f->C_ns};
    bool ok = true;
    for (int64_t s : strides) ok = ok && (s % 4 == 0);
    const void* ptrs[] = {f->u, f->delta, f->z, d->du, d->ddelta, d->dz, f->B, f->C};
    for (const void* q : ptrs) ok = ok && (((uintptr_t)q) % (4 * esz) == 0);
    ok = ok && (((uintptr_t)d->dout) % (4 * (a.out_f32 ? 4 : esz)) == 0);
    // 16-bit rows take kernels that move B / C (and, at L % 8 == 0, the rows) in 16-byte LDS-DMA units: those need 16-byte aligned
    // addresses.  Rows that are only 8-byte aligned (a view that starts 4 elements into its storage, L % 8 == 4) take the
    // element-wise path: a third, non-DMA vector instantiation of this 1000-line kernel per (dtype, waves, fold) for views that no
    // model of the reference produces was judged not worth its compile time -- the cliff is documented in mxvl.h instead.
    if (ok && esz == 2) {
      const int64_t s8[] = {f->u_bs, f->u_ds, f->delta_bs, f->delta_ds, f->z ? f->z_bs : 0, f->z ? f->z_ds : 0,
                            a.out_f32 ? 0 : d->dout_bs, a.out_f32 ? 0 : d->dout_ds, f->B_bs, f->B_gs, f->B_ns, f->C_bs, f->C_gs, f->C_ns};
      for (int64_t s : s8) ok = ok && (s % 8 == 0);
      const void* p16[] = {f->u, f->delta, f->z, a.out_f32 ? nullptr : d->dout, f->B, f->C};
      for (const void* q : p16) ok = ok && (((uintptr_t)q) % 16 == 0);
    }
    a.vec_ok = ok ? 1 : 0;
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  switch (f->io_dtype) {
    case MXVL_F32: return dispatch_bwd<float>(a, stream);
    case MXVL_BF16: return dispatch_bwd<bf16_t>(a, stream);
    default: return dispatch_bwd<f16_t>(a, stream);
  }
}
