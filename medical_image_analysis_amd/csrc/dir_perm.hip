// dir_perm.hip -- the scan-order re-orderings of the 4- / 6-direction Mamba mixer (bimamba v3 / v4), for gfx950.
//
// The reference materialises every direction with advanced indexing / flips / reshapes
// (CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:447-532: `xz.flip([-1])`, the middle-cls transpose :476-482 and
// its inverse :522-527, three adds and a division): ~12 tensor passes per mixer.  Both directions of the data movement are ONE
// kernel here, and each is the other's adjoint:
//   dir_gather : x (B,D,L)        -> X (B,K,D,Lp)   X[b,k,d,l] = l < L ? x[b,d,perm[k][l]] : 0      (Lp >= L: aligned rows)
//   dir_merge  : y (B,K,D,Lp)     -> out (B,D,L)    out[b,d,t] = sum_k y[b,k,d,inv[k][t]]           (fp32 sum, k ascending)
// perm / inv are (K, L) int32 permutations of [0, L).  One workgroup per (b, d) row: the row goes through LDS once, global
// reads and writes are coalesced, the permutation is applied on the LDS side.
//
// Output gate.  Every direction's scan output is gated by silu(z) of ITS ordering of z; the gate commutes with the
// re-ordering -- sum_k P_k^-1 (y_k * silu(P_k z)) = silu(z) * sum_k P_k^-1 y_k -- so the scans run ungated and the merge
// applies silu(z) * scale once, in registers (`gate`): the reference's `* silu(z)` per direction and `/ 4` (:522-529) cost
// no tensor pass of their own.  The ungated sum goes to `pre` for the backward, which is the gather with the same gate:
// d(stacked) = gather(d(out) * silu(z) * scale) and d(z) = d(out) * pre * scale * silu'(z), all from the row in LDS.
#include "mxvl_common.h"

namespace mxvl {

struct PermArgs {
  int B, D, L, Lp, K;
  long long x_bs, x_ds;          // (B,D,L) side: batch / row strides in elements (L stride 1)
  long long X_bs, X_ks, X_ds;    // (B,K,D,Lp) side
  const int* idx;                // (K, L)
  const void* src;
  void* dst;
  // optional output gate (see the header comment); all (B,D,L)-shaped with their own batch / row strides
  const void* gate;
  void* pre;                     // merge: written (may be NULL);  gather: read
  void* dgate;                   // gather only
  long long g_bs, g_ds, p_bs, p_ds, dg_bs, dg_ds;
  float scale;
};

template <typename io_t>
__global__ __launch_bounds__(256) void dir_gather_kernel(const PermArgs p) {
  extern __shared__ float srow[];
  using io = Io<io_t>;
  const int d = blockIdx.x, b = blockIdx.y;
  const io_t* x = (const io_t*)p.src + (long long)b * p.x_bs + (long long)d * p.x_ds;
  if (p.gate) {   // backward of the gated merge: x is d(out)
    const io_t* z = (const io_t*)p.gate + (long long)b * p.g_bs + (long long)d * p.g_ds;
    const io_t* pre = (const io_t*)p.pre + (long long)b * p.p_bs + (long long)d * p.p_ds;
    io_t* dz = (io_t*)p.dgate + (long long)b * p.dg_bs + (long long)d * p.dg_ds;
    for (int t = threadIdx.x; t < p.L; t += 256) {
      const float g = io::ld(x + t) * p.scale, zv = io::ld(z + t);
      const float sg = sigmoid(zv);
      srow[t] = g * (zv * sg);
      io::st(dz + t, g * io::ld(pre + t) * (sg * fmaf(zv, 1.0f - sg, 1.0f)));
    }
  } else {
    for (int t = threadIdx.x; t < p.L; t += 256) srow[t] = io::ld(x + t);
  }
  __syncthreads();
  for (int k = 0; k < p.K; ++k) {
    io_t* X = (io_t*)p.dst + (long long)b * p.X_bs + (long long)k * p.X_ks + (long long)d * p.X_ds;
    const int* ix = p.idx + (long long)k * p.L;
    for (int l = threadIdx.x; l < p.Lp; l += 256) io::st(X + l, l < p.L ? srow[ix[l]] : 0.0f);
  }
}

template <typename io_t>
__global__ __launch_bounds__(256) void dir_merge_kernel(const PermArgs p) {
  extern __shared__ float srow[];
  using io = Io<io_t>;
  constexpr int MAXT = 20;                      // L <= 256 * MAXT
  const int d = blockIdx.x, b = blockIdx.y;
  float acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) acc[i] = 0.0f;
  for (int k = 0; k < p.K; ++k) {
    const io_t* y = (const io_t*)p.src + (long long)b * p.X_bs + (long long)k * p.X_ks + (long long)d * p.X_ds;
    const int* ix = p.idx + (long long)k * p.L;
    __syncthreads();
    for (int l = threadIdx.x; l < p.L; l += 256) srow[l] = io::ld(y + l);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = threadIdx.x + i * 256;
      if (t < p.L) acc[i] += srow[ix[t]];
    }
  }
  io_t* out = (io_t*)p.dst + (long long)b * p.x_bs + (long long)d * p.x_ds;
  if (p.gate) {
    const io_t* z = (const io_t*)p.gate + (long long)b * p.g_bs + (long long)d * p.g_ds;
    io_t* pre = p.pre ? (io_t*)p.pre + (long long)b * p.p_bs + (long long)d * p.p_ds : nullptr;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      const int t = threadIdx.x + i * 256;
      if (t < p.L) {
        if (pre) io::st(pre + t, acc[i]);
        io::st(out + t, acc[i] * silu(io::ld(z + t)) * p.scale);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int t = threadIdx.x + i * 256;
    if (t < p.L) io::st(out + t, acc[i]);
  }
}

// ---- short rows (padded_len <= 256: the 197-token encoders) -----------------------------------------------------------
// One element per thread.  The generic kernels above walk a dependent chain per direction (row load -> barrier -> index load
// -> LDS -> store) with 2 bytes per lane in flight: at 200-element rows they are latency-bound (measured 0.74 TB/s merge,
// 1.26 TB/s gather on the ARM-large 224 step).  Here every global load a thread needs -- its element of all K direction rows
// and its K indices -- is issued before the first use, so one round trip covers them all.
constexpr int kMaxDirs = 6;

// A workgroup takes RPB consecutive channels of one batch element: the K indices of a step are loaded once per thread and
// reused for all of them (at one row per workgroup the index loads were twice the bytes of the row itself), and the RPB
// row loads of a thread are in flight together.
constexpr int kRowsPerBlock = 8;

template <typename io_t>
__global__ __launch_bounds__(256) void dir_gather_short_kernel(const PermArgs p) {
  constexpr int RPB = kRowsPerBlock;
  __shared__ float srow[RPB][256];
  using io = Io<io_t>;
  const int d0 = blockIdx.x * RPB, b = blockIdx.y, t = threadIdx.x;
  const bool live = t < p.L;
  // unconditional loads from clamped (always valid) addresses: no control flow between them, so all are in flight together
  const int tc = live ? t : 0;
  int ixv[kMaxDirs];
#pragma unroll
  for (int k = 0; k < kMaxDirs; ++k) ixv[k] = p.idx[(k < p.K ? k : 0) * p.L + tc];
  float v[RPB], zv[RPB], pre[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    const int d = min(d0 + r, p.D - 1);
    v[r] = io::ld((const io_t*)p.src + (long long)b * p.x_bs + (long long)d * p.x_ds + tc);
    if (p.gate) {
      zv[r] = io::ld((const io_t*)p.gate + (long long)b * p.g_bs + (long long)d * p.g_ds + tc);
      pre[r] = io::ld((const io_t*)p.pre + (long long)b * p.p_bs + (long long)d * p.p_ds + tc);
    }
  }
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    float x = v[r];
    if (p.gate) {   // backward of the gated merge: x is d(out)
      const float g = x * p.scale, sg = sigmoid(zv[r]);
      x = g * (zv[r] * sg);
      if (live && d0 + r < p.D)
        io::st((io_t*)p.dgate + (long long)b * p.dg_bs + (long long)(d0 + r) * p.dg_ds + t, g * pre[r] * (sg * fmaf(zv[r], 1.0f - sg, 1.0f)));
    }
    srow[r][t] = x;
  }
  __syncthreads();
  if (t >= p.Lp) return;
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    if (d0 + r >= p.D) break;
#pragma unroll
    for (int k = 0; k < kMaxDirs; ++k) {
      if (k < p.K) {
        io_t* X = (io_t*)p.dst + (long long)b * p.X_bs + (long long)k * p.X_ks + (long long)(d0 + r) * p.X_ds;
        io::st(X + t, live ? srow[r][ixv[k]] : 0.0f);
      }
    }
  }
}

// merge: RPB / 2 channels per workgroup (K x rows x 256 floats of LDS)
template <typename io_t>
__global__ __launch_bounds__(256) void dir_merge_short_kernel(const PermArgs p) {
  constexpr int RPB = kRowsPerBlock / 2;
  __shared__ float srow[RPB][kMaxDirs][256];
  using io = Io<io_t>;
  const int d0 = blockIdx.x * RPB, b = blockIdx.y, t = threadIdx.x;
  const bool live = t < p.L;
  // unconditional loads from clamped (always valid) addresses: no control flow between them, so all are in flight together
  const int tc = live ? t : 0;
  int ixv[kMaxDirs];
#pragma unroll
  for (int k = 0; k < kMaxDirs; ++k) ixv[k] = p.idx[(k < p.K ? k : 0) * p.L + tc];
  float yv[RPB][kMaxDirs], zv[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    const int d = min(d0 + r, p.D - 1);
#pragma unroll
    for (int k = 0; k < kMaxDirs; ++k) {
      const int kc = k < p.K ? k : 0;
      yv[r][k] = io::ld((const io_t*)p.src + (long long)b * p.X_bs + (long long)kc * p.X_ks + (long long)d * p.X_ds + tc);
    }
    zv[r] = p.gate ? io::ld((const io_t*)p.gate + (long long)b * p.g_bs + (long long)d * p.g_ds + tc) : 0.0f;
  }
#pragma unroll
  for (int r = 0; r < RPB; ++r)
#pragma unroll
    for (int k = 0; k < kMaxDirs; ++k) srow[r][k][t] = yv[r][k];
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    if (d0 + r >= p.D) break;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kMaxDirs; ++k)
      if (k < p.K) acc += srow[r][k][ixv[k]];          // k ascending, like the generic kernel
    io_t* out = (io_t*)p.dst + (long long)b * p.x_bs + (long long)(d0 + r) * p.x_ds;
    if (p.gate) {
      if (p.pre) io::st((io_t*)p.pre + (long long)b * p.p_bs + (long long)(d0 + r) * p.p_ds + t, acc);
      io::st(out + t, acc * silu(zv[r]) * p.scale);
    } else {
      io::st(out + t, acc);
    }
  }
}

// ---- short rows, 16-bit io, 16-byte rows on the stacked side ---------------------------------------------------------------
// The stacked tensor is K times the bytes of the row tensors and its rows are Lp = a multiple of 8 elements long: the kernels
// below move it with 16-byte accesses (a 16-bit element per lane is 128 bytes per wave instruction: the element-wise kernels
// above spent 16 (merge) / 32 (gather) such instructions per thread on it), keep the tile in LDS in the io dtype and take 8
// channels per workgroup.  The row side (L = 197 elements, 2-byte aligned rows with arbitrary strides) stays element-wise,
// coalesced across the lanes.  Results are bit-identical to the kernels above (same fp32 sum order, one rounding).
constexpr int kVecRows = 8;

template <typename io_t>
__global__ __launch_bounds__(256) void dir_gather_vec_kernel(const PermArgs p) {
  static_assert(sizeof(io_t) == 2, "16-bit io");
  constexpr int RPB = kVecRows;
  __shared__ __attribute__((aligned(16))) io_t srow[RPB][256];           // entries >= L are zero: the padding reads them
  __shared__ __attribute__((aligned(16))) uint8_t sidx[kMaxDirs][256];    // perm[k][l], l >= L -> 255 (a zero entry; L <= 248 here)
  using io = Io<io_t>;
  const int d0 = blockIdx.x * RPB, b = blockIdx.y, t = threadIdx.x;
  const bool live = t < p.L;
  const int tc = live ? t : 0;
  int ixv[kMaxDirs];
#pragma unroll
  for (int k = 0; k < kMaxDirs; ++k) ixv[k] = p.idx[(k < p.K ? k : 0) * p.L + tc];
  // ungated: the gate / pre loads go to the source rows instead (values unused): no branch around a load, see dir_merge_vec_kernel
  const bool gated = p.gate != nullptr;
  const io_t* gz = (const io_t*)(gated ? p.gate : p.src);
  const io_t* gp = (const io_t*)(gated ? p.pre : p.src);
  const long long gbs = gated ? p.g_bs : p.x_bs, gds = gated ? p.g_ds : p.x_ds, pbs = gated ? p.p_bs : p.x_bs, pds = gated ? p.p_ds : p.x_ds;
  io_t vr[RPB], zr[RPB], pr[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    const int d = min(d0 + r, p.D - 1);
    vr[r] = *((const io_t*)p.src + (long long)b * p.x_bs + (long long)d * p.x_ds + tc);
    zr[r] = *(gz + (long long)b * gbs + (long long)d * gds + tc);
    pr[r] = *(gp + (long long)b * pbs + (long long)d * pds + tc);
  }
  float v[RPB], zv[RPB], pre[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) { v[r] = io::ld(&vr[r]); zv[r] = io::ld(&zr[r]); pre[r] = io::ld(&pr[r]); }
#pragma unroll
  for (int r = 0; r < RPB; ++r) asm volatile("" : "+v"(v[r]), "+v"(zv[r]), "+v"(pre[r]));   // (keeps a load from being sunk into the gated branch)
#pragma unroll
  for (int k = 0; k < kMaxDirs; ++k) sidx[k][t] = live ? (uint8_t)ixv[k] : (uint8_t)255;
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    float x = v[r];
    if (gated) {   // backward of the gated merge: x is d(out)
      const float g = x * p.scale, sg = sigmoid(zv[r]);
      x = g * (zv[r] * sg);
      if (live && d0 + r < p.D)
        io::st((io_t*)p.dgate + (long long)b * p.dg_bs + (long long)(d0 + r) * p.dg_ds + t, g * pre[r] * (sg * fmaf(zv[r], 1.0f - sg, 1.0f)));
    }
    asm volatile("" : "+v"(x));     // round the product to fp32 first, like the element-wise kernel (which parks it in LDS as fp32):
                                    // hipcc otherwise folds the last multiply into the fp16 conversion (v_fma_mixlo_f16, one rounding)
    io::st(&srow[r][t], live ? x : 0.0f);
  }
  __syncthreads();
  const int cpr = p.Lp >> 3;                       // 16-byte vectors per stacked row
  const int nvec = p.K * RPB * cpr;
  for (int vv = t; vv < nvec; vv += 256) {
    const int c = vv % cpr, rk = vv / cpr, r = rk % RPB, k = rk / RPB;
    if (d0 + r >= p.D) continue;
    const uint2 ib = *(const uint2*)&sidx[k][c * 8];
    const uint16_t* row = (const uint16_t*)srow[r];
    uint4 o;
    o.x = (uint32_t)row[ib.x & 255u] | ((uint32_t)row[(ib.x >> 8) & 255u] << 16);
    o.y = (uint32_t)row[(ib.x >> 16) & 255u] | ((uint32_t)row[ib.x >> 24] << 16);
    o.z = (uint32_t)row[ib.y & 255u] | ((uint32_t)row[(ib.y >> 8) & 255u] << 16);
    o.w = (uint32_t)row[(ib.y >> 16) & 255u] | ((uint32_t)row[ib.y >> 24] << 16);
    io_t* X = (io_t*)p.dst + (long long)b * p.X_bs + (long long)k * p.X_ks + (long long)(d0 + r) * p.X_ds;
    *(uint4*)(X + c * 8) = o;
  }
}

template <typename io_t>
__global__ __launch_bounds__(256) void dir_merge_vec_kernel(const PermArgs p) {
  static_assert(sizeof(io_t) == 2, "16-bit io");
  constexpr int RPB = kVecRows;
  __shared__ __attribute__((aligned(16))) io_t sy[kMaxDirs][RPB][256];
  using io = Io<io_t>;
  const int d0 = blockIdx.x * RPB, b = blockIdx.y, t = threadIdx.x;
  const bool live = t < p.L;
  const int tc = live ? t : 0;
  const int cpr = p.Lp >> 3;
  const int nvec = p.K * RPB * cpr;
  // every global load of the thread is requested before the first use: the stacked tile (up to 5 vectors), its K indices and
  // its element of the 8 gate rows
  constexpr int MAXV = (kMaxDirs * RPB * 32 + 255) / 256;     // Lp <= 256: at most 32 vectors per row
  uint4 yv[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vv = min(t + i * 256, nvec - 1);
    const int c = vv % cpr, rk = vv / cpr, r = rk % RPB, k = rk / RPB;
    const int d = min(d0 + r, p.D - 1);
    yv[i] = *(const uint4*)((const io_t*)p.src + (long long)b * p.X_bs + (long long)k * p.X_ks + (long long)d * p.X_ds + c * 8);
  }
  int ixv[kMaxDirs];
#pragma unroll
  for (int k = 0; k < kMaxDirs; ++k) ixv[k] = p.idx[(k < p.K ? k : 0) * p.L + tc];
  io_t zr[RPB];
  {
    // the gate rows, or (ungated merge) any valid address with the same strides zeroed: no branch around a load -- hipcc waits
    // for a load at the join of the branch it sits in, which made a chain of 8 serialised round trips out of these
    const io_t* gz = (const io_t*)(p.gate ? p.gate : p.src);
    const long long gbs = p.gate ? p.g_bs : 0, gds = p.gate ? p.g_ds : 0;
#pragma unroll
    for (int r = 0; r < RPB; ++r) zr[r] = gz[(long long)b * gbs + (long long)min(d0 + r, p.D - 1) * gds + (p.gate ? tc : 0)];
  }
  // pin: all of the above are requested before the first of them is waited for (the compiler otherwise sinks each tile load
  // into the `vv < nvec` block of its LDS store: load, wait, store, six times in a row)
#pragma unroll
  for (int i = 0; i < MAXV; ++i) asm volatile("" : "+v"(yv[i].x), "+v"(yv[i].y), "+v"(yv[i].z), "+v"(yv[i].w));
  float zv[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) zv[r] = io::ld(&zr[r]);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vv = t + i * 256;
    if (vv < nvec) {
      const int c = vv % cpr, rk = vv / cpr, r = rk % RPB, k = rk / RPB;
      *(uint4*)&sy[k][r][c * 8] = yv[i];
    }
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int r = 0; r < RPB; ++r) {
    if (d0 + r >= p.D) break;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kMaxDirs; ++k)
      if (k < p.K) acc += io::ld(&sy[k][r][ixv[k]]);          // k ascending, like the generic kernel
    io_t* out = (io_t*)p.dst + (long long)b * p.x_bs + (long long)(d0 + r) * p.x_ds;
    if (p.gate) {
      if (p.pre) io::st((io_t*)p.pre + (long long)b * p.p_bs + (long long)(d0 + r) * p.p_ds + t, acc);
      io::st(out + t, acc * silu(zv[r]) * p.scale);
    } else {
      io::st(out + t, acc);
    }
  }
}

static int perm_launch(bool merge, const mxvl_dir_perm_desc* d, void* stream) {
  if (!d || !d->rows || !d->stacked || !d->index) return MXVL_ERR_NULL;
  if (d->batch <= 0 || d->dim <= 0 || d->seqlen <= 0 || d->n_dirs <= 0 || d->padded_len < d->seqlen) return MXVL_ERR_SHAPE;
  if (d->seqlen > 256 * 20 || d->batch > 65535) return MXVL_ERR_UNSUPPORTED;   // merge keeps 20 steps per thread; grid.y limit
  PermArgs a;
  a.B = d->batch; a.D = d->dim; a.L = d->seqlen; a.Lp = d->padded_len; a.K = d->n_dirs;
  a.x_bs = d->rows_bs; a.x_ds = d->rows_ds; a.X_bs = d->stacked_bs; a.X_ks = d->stacked_ks; a.X_ds = d->stacked_ds;
  a.idx = (const int*)d->index;
  a.src = merge ? d->stacked : d->rows;
  a.dst = merge ? (void*)d->rows : (void*)d->stacked;
  a.gate = d->gate; a.pre = d->pre; a.dgate = d->dgate; a.scale = d->gate_scale;
  a.g_bs = d->gate_bs; a.g_ds = d->gate_ds; a.p_bs = d->pre_bs; a.p_ds = d->pre_ds; a.dg_bs = d->dgate_bs; a.dg_ds = d->dgate_ds;
  if (d->gate && !merge && (!d->pre || !d->dgate)) return MXVL_ERR_NULL;   // the gated gather is a backward: it needs pre and dgate
  const dim3 grid(a.D, a.B);
  const size_t lds = sizeof(float) * (size_t)a.L;
  hipStream_t s = (hipStream_t)stream;
  const bool short_rows = a.Lp <= 256 && a.K <= kMaxDirs;
  // 16-byte accesses on the stacked side: 16-bit io, rows of whole 8-element vectors at 16-byte aligned addresses
  const bool vec_rows = short_rows && d->io_dtype != MXVL_F32 && a.Lp % 8 == 0 && a.L <= 248 &&
                        ((uintptr_t)d->stacked % 16 == 0) && a.X_bs % 8 == 0 && a.X_ks % 8 == 0 && a.X_ds % 8 == 0;
  const dim3 vgrid((a.D + kVecRows - 1) / kVecRows, a.B);
  const dim3 ggrid((a.D + kRowsPerBlock - 1) / kRowsPerBlock, a.B), mgrid((a.D + kRowsPerBlock / 2 - 1) / (kRowsPerBlock / 2), a.B);
#define MXVL_PERM(T)                                                                                   \
  do {                                                                                                  \
    if constexpr (sizeof(T) == 2) {                                                                     \
      if (vec_rows) {                                                                                   \
        if (merge) hipLaunchKernelGGL(dir_merge_vec_kernel<T>, vgrid, dim3(256), 0, s, a);              \
        else hipLaunchKernelGGL(dir_gather_vec_kernel<T>, vgrid, dim3(256), 0, s, a);                   \
        break;                                                                                          \
      }                                                                                                 \
    }                                                                                                   \
    if (short_rows && merge) hipLaunchKernelGGL(dir_merge_short_kernel<T>, mgrid, dim3(256), 0, s, a);   \
    else if (short_rows) hipLaunchKernelGGL(dir_gather_short_kernel<T>, ggrid, dim3(256), 0, s, a);      \
    else if (merge) hipLaunchKernelGGL(dir_merge_kernel<T>, grid, dim3(256), lds, s, a);                \
    else hipLaunchKernelGGL(dir_gather_kernel<T>, grid, dim3(256), lds, s, a);                          \
  } while (0)
  switch (d->io_dtype) {
    case MXVL_F32: MXVL_PERM(float); break;
    case MXVL_BF16: MXVL_PERM(bf16_t); break;
    case MXVL_F16: MXVL_PERM(f16_t); break;
    default: return MXVL_ERR_DTYPE;
  }
#undef MXVL_PERM
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

}  // namespace mxvl

extern "C" {
int mxvl_dir_gather(const mxvl_dir_perm_desc* d, void* hip_stream) { return mxvl::perm_launch(false, d, hip_stream); }
int mxvl_dir_merge(const mxvl_dir_perm_desc* d, void* hip_stream) { return mxvl::perm_launch(true, d, hip_stream); }
}
