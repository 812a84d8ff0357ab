// mamba_inner.hip -- mxvl_mamba_inner_fwd / mxvl_mamba_inner_bwd (ABI v11): mamba_inner_fn / mamba_inner_fn_no_out_proj behind ONE
// native call.  The reference calls the patched mamba_ssm's fused block at
// CXPMRG_Bench_MambaXray_VL/pretrain/mamba_simple.py:388-402 and arm/Finetuning/mamba_simple.py:450-511, 650-664 (third-party, not in
// the reference tree); the arithmetic is the in-repo slow path mamba_simple.py:665-709:
//     conv1d + SiLU -> x_proj -> split (dt | B | C) -> dt_proj -> selective scan (z gate, delta_bias, softplus) [-> out_proj].
// Composition, not fusion (include/mxvl.h says why): the conv and the scan are mxvl_conv1d_* / mxvl_scan_* of this library, the
// dense products are plain GEMMs.  Everything stays in the op boundary's channel-major layout (batch, channels, seqlen) -- per batch
// element a row-major (channels x seqlen) matrix IS a column-major (seqlen x channels) one, so
//     x_dbl_b (R + 2N, L) = Wx (R + 2N, D) xc_b (D, L)        and        delta_b (D, L) = Wdt (D, R) x_dbl_b[:R]
// are column-major NN products with the weight as the strided-batch-invariant operand, B_b and C_b are ROWS of x_dbl_b (the
// (batch, dstate, seqlen) arrays the scan kernels read: no transpose, no copy), and every gradient product is an NT / TN / TT form of
// the same operands.  The GEMMs are rocBLAS's (rocblas_gemm_strided_batched_ex, fp32 accumulation) -- "plain library GEMMs" -- reached
// through dlopen at the first call, so the library has no link-time dependency on it.
#include <dlfcn.h>

#include <mutex>

#include "mxvl_common.h"

namespace mxvl {

// ---- rocBLAS, by hand: the five symbols and the enum values this file needs (rocblas-types.h) ------------------------------------------
typedef void* rb_handle;
enum { RB_OP_N = 111, RB_OP_T = 112 };
enum { RB_F16 = 150, RB_F32 = 151, RB_BF16 = 168 };
typedef int (*rb_create_handle_t)(rb_handle*);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_set_pointer_mode_t)(rb_handle, int);
typedef int (*rb_gemm_sb_ex_t)(rb_handle, int, int, int32_t, int32_t, int32_t, const void*, const void*, int, int32_t, int64_t, const void*, int,
                               int32_t, int64_t, const void*, const void*, int, int32_t, int64_t, void*, int, int32_t, int64_t, int32_t, int,
                               int, int32_t, uint32_t);
struct RocBlas {
  rb_handle handle = nullptr;
  rb_set_stream_t set_stream = nullptr;
  rb_gemm_sb_ex_t gemm = nullptr;
  std::mutex mu;          // one handle, one stream at a time: the forward (main thread) and the backward (autograd thread) share it
};
static RocBlas* rocblas_get() {
  static RocBlas rb;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    void* lib = nullptr;
    for (const char* name : {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so.5", "/opt/rocm/lib/librocblas.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return;
    auto create = (rb_create_handle_t)dlsym(lib, "rocblas_create_handle");
    auto mode = (rb_set_pointer_mode_t)dlsym(lib, "rocblas_set_pointer_mode");
    rb.set_stream = (rb_set_stream_t)dlsym(lib, "rocblas_set_stream");
    rb.gemm = (rb_gemm_sb_ex_t)dlsym(lib, "rocblas_gemm_strided_batched_ex");
    if (!create || !mode || !rb.set_stream || !rb.gemm) return;
    if (create(&rb.handle) != 0 || !rb.handle) return;
    if (mode(rb.handle, 0) != 0) return;     // alpha / beta on the host
    ok = true;
  });
  return ok ? &rb : nullptr;
}

static int rb_dtype(int dt) { return dt == MXVL_F32 ? RB_F32 : dt == MXVL_BF16 ? RB_BF16 : RB_F16; }
static int64_t esz(int dt) { return dt == MXVL_F32 ? 4 : 2; }

// C (ldc, strided over the batch) = op(A) op(B) + beta C, column-major, fp32 accumulation; a / b in the io dtype, c in `c_dt`
struct Gemm {
  RocBlas* rb;
  int io;
  int run(int opa, int opb, int m, int n, int k, const void* a, int lda, int64_t sa, const void* b, int ldb, int64_t sb, float beta, void* c,
          int c_dt, int ldc, int64_t sc, int batch) const {
    const float alpha = 1.0f;
    const int rc = rb->gemm(rb->handle, opa, opb, m, n, k, &alpha, a, rb_dtype(io), lda, sa, b, rb_dtype(io), ldb, sb, &beta, c, rb_dtype(c_dt), ldc,
                            sc, c, rb_dtype(c_dt), ldc, sc, batch, RB_F32, 0, 0, 0);
    return rc == 0 ? MXVL_OK : MXVL_ERR_LAUNCH;
  }
};

// ---- the three element-wise helpers ----------------------------------------------------------------------------------------------------
// dst[b][i] = io(src[b][i]), i < n: the fp32 dB | dC accumulators of a batch element -> its d(x_dbl) rows (selective_scan.cpp:347's cast)
template <typename io_t>
__global__ __launch_bounds__(256) void mi_cast_rows_kernel(const float* src, int64_t src_bs, io_t* dst, int64_t dst_bs, int64_t n) {
  const float* s = src + (int64_t)blockIdx.y * src_bs;
  io_t* d = dst + (int64_t)blockIdx.y * dst_bs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) Io<io_t>::st(d + i, s[i]);
}
// out[r][c] = bias[c]: the rows out_proj's GEMM then accumulates onto (beta = 1: bias and product meet in fp32, one rounding)
template <typename io_t>
__global__ __launch_bounds__(256) void mi_fill_rows_kernel(io_t* out, const io_t* bias, int64_t rows, int cols) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] = bias[i % cols];
}
// acc[c] += sum_r x[r][c]: out_proj's bias gradient.  A workgroup owns 256 columns x a slab of rows; one atomic per column and slab.
template <typename io_t>
__global__ __launch_bounds__(256) void mi_colsum_kernel(const io_t* x, float* acc, int64_t rows, int cols, int rows_per_wg) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  float s = 0.0f;
  for (int64_t r = r0; r < r1; ++r) s += Io<io_t>::ld(x + r * cols + c);
  atomicAdd(acc + c, s);
}

static int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

struct FwdWs { int64_t xc, x_dbl, delta, y, ckpt, total; };
struct BwdWs { int64_t dy, du, ddelta, dBC, dx_dbl, total; };

static int mi_check(const mxvl_mamba_inner_desc* d) {
  if (!d || !d->xz || !d->conv_weight || !d->x_proj_weight || !d->dt_proj_weight || !d->A) return MXVL_ERR_NULL;
  if (d->io_dtype != MXVL_F32 && d->io_dtype != MXVL_BF16 && d->io_dtype != MXVL_F16) return MXVL_ERR_DTYPE;
  if (d->batch <= 0 || d->dim <= 0 || d->seqlen <= 0 || d->dstate <= 0 || d->dt_rank <= 0 || d->width <= 0) return MXVL_ERR_SHAPE;
  if (d->dstate > MXVL_MAX_DSTATE) return MXVL_ERR_DSTATE;
  if (d->out_proj_weight && d->d_model <= 0) return MXVL_ERR_SHAPE;
  if (d->out_proj_bias && !d->out_proj_weight) return MXVL_ERR_NULL;
  if ((int64_t)d->batch * d->dim * d->seqlen > 0x7fffffffLL * 8) return MXVL_ERR_SHAPE;
  return MXVL_OK;
}

static FwdWs fwd_ws(const mxvl_mamba_inner_desc* d) {
  const int64_t e = esz(d->io_dtype), BDL = (int64_t)d->batch * d->dim * d->seqlen, M = d->dt_rank + 2 * d->dstate;
  FwdWs w;
  int64_t o = 0;
  w.xc = o; o += al256(BDL * e);
  w.x_dbl = o; o += al256((int64_t)d->batch * M * d->seqlen * e);
  w.delta = o; o += al256(BDL * e);
  w.y = o; o += d->out_proj_weight ? al256(BDL * e) : 0;
  w.ckpt = o; o += al256((int64_t)d->batch * d->dim * mxvl_scan_n_chunks(d->seqlen, d->dstate) * d->dstate * 4);
  w.total = o;
  return w;
}
static BwdWs bwd_ws(const mxvl_mamba_inner_desc* d) {
  const int64_t e = esz(d->io_dtype), BDL = (int64_t)d->batch * d->dim * d->seqlen, M = d->dt_rank + 2 * d->dstate;
  BwdWs w;
  int64_t o = 0;
  w.dy = o; o += d->out_proj_weight ? al256(BDL * e) : 0;
  w.du = o; o += al256(BDL * e);
  w.ddelta = o; o += al256(BDL * e);
  w.dBC = o; o += al256((int64_t)d->batch * 2 * d->dstate * d->seqlen * 4);
  w.dx_dbl = o; o += al256((int64_t)d->batch * M * d->seqlen * e);
  w.total = o;
  return w;
}

// the scan descriptor both directions share: u = xc, B / C = rows of x_dbl, z = the second half of xz
static void scan_desc(const mxvl_mamba_inner_desc* d, const FwdWs& w, mxvl_scan_desc& s) {
  const int64_t e = esz(d->io_dtype), D = d->dim, L = d->seqlen, N = d->dstate, R = d->dt_rank, M = R + 2 * N;
  char* ws = (char*)d->workspace;
  s = mxvl_scan_desc{};
  s.batch = d->batch; s.dim = d->dim; s.seqlen = d->seqlen; s.dstate = d->dstate; s.n_groups = 1;
  s.io_dtype = d->io_dtype; s.flags = d->flags & MXVL_SCAN_DELTA_SOFTPLUS; s.delta_group_ratio = 0;
  s.u_bs = D * L; s.u_ds = L; s.delta_bs = D * L; s.delta_ds = L; s.z_bs = 2 * D * L; s.z_ds = L; s.out_bs = D * L; s.out_ds = L;
  s.B_bs = M * L; s.B_gs = 0; s.B_ns = L; s.C_bs = M * L; s.C_gs = 0; s.C_ns = L; s.A_ds = N; s.A_ns = 1;
  s.u = ws + w.xc; s.delta = ws + w.delta; s.A = d->A;
  s.B = ws + w.x_dbl + R * L * e; s.C = ws + w.x_dbl + (R + N) * L * e;
  s.D = d->D; s.delta_bias = d->delta_bias; s.z = (const char*)d->xz + D * L * e;
  s.ckpt = ws + w.ckpt;
}

template <typename io_t>
static void launch_fill_rows(void* out, const void* bias, int64_t rows, int cols, hipStream_t s) {
  const int64_t total = rows * cols;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL((mi_fill_rows_kernel<io_t>), dim3(grid), dim3(256), 0, s, (io_t*)out, (const io_t*)bias, rows, cols);
}
template <typename io_t>
static void launch_cast_rows(const float* src, int64_t src_bs, void* dst, int64_t dst_bs, int64_t n, int batch, hipStream_t s) {
  const unsigned gx = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL((mi_cast_rows_kernel<io_t>), dim3(gx, batch), dim3(256), 0, s, src, src_bs, (io_t*)dst, dst_bs, n);
}
template <typename io_t>
static void launch_colsum(const void* x, float* acc, int64_t rows, int cols, hipStream_t s) {
  const int rpw = 256;
  hipLaunchKernelGGL((mi_colsum_kernel<io_t>), dim3((cols + 255) / 256, (unsigned)((rows + rpw - 1) / rpw)), dim3(256), 0, s, (const io_t*)x, acc,
                     rows, cols, rpw);
}
#define MXVL_MI_BY_DTYPE(dt, CALL)               \
  do {                                           \
    switch (dt) {                                \
      case MXVL_F32: { typedef float io_t; CALL; } break;   \
      case MXVL_BF16: { typedef bf16_t io_t; CALL; } break; \
      default: { typedef f16_t io_t; CALL; } break;         \
    }                                            \
  } while (0)

}  // namespace mxvl

using namespace mxvl;

extern "C" {

int64_t mxvl_mamba_inner_workspace_bytes(const mxvl_mamba_inner_desc* d) {
  if (mi_check(d) != MXVL_OK) return -1;
  return fwd_ws(d).total;
}
int64_t mxvl_mamba_inner_bwd_workspace_bytes(const mxvl_mamba_inner_desc* d) {
  if (mi_check(d) != MXVL_OK) return -1;
  return bwd_ws(d).total;
}

int mxvl_mamba_inner_fwd(const mxvl_mamba_inner_desc* d, void* hip_stream) {
  int rc = mi_check(d);
  if (rc != MXVL_OK) return rc;
  if (!d->out || !d->workspace) return MXVL_ERR_NULL;
  const FwdWs w = fwd_ws(d);
  if (d->workspace_bytes < w.total) return MXVL_ERR_SHAPE;
  RocBlas* rb = rocblas_get();
  if (!rb) return MXVL_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)hip_stream;
  const int64_t e = esz(d->io_dtype), D = d->dim, L = d->seqlen, N = d->dstate, R = d->dt_rank, M = R + 2 * N;
  char* ws = (char*)d->workspace;

  mxvl_conv1d_desc c{};
  c.batch = d->batch; c.dim = d->dim; c.seqlen = d->seqlen; c.width = d->width; c.io_dtype = d->io_dtype; c.silu = 1;
  c.x_bs = 2 * D * L; c.x_ds = L; c.y_bs = D * L; c.y_ds = L;
  c.x = d->xz; c.weight = d->conv_weight; c.bias = d->conv_bias; c.y = ws + w.xc;
  rc = mxvl_conv1d_fwd(&c, hip_stream);
  if (rc != MXVL_OK) return rc;

  mxvl_scan_desc s;
  scan_desc(d, w, s);
  s.out = d->out_proj_weight ? (void*)(ws + w.y) : d->out;
  {
    std::lock_guard<std::mutex> lock(rb->mu);
    if (rb->set_stream(rb->handle, stream) != 0) return MXVL_ERR_LAUNCH;
    const Gemm g{rb, d->io_dtype};
    // x_dbl_b^T (L, M) = xc_b^T (L, D) Wx^T (D, M)
    rc = g.run(RB_OP_N, RB_OP_N, (int)L, (int)M, (int)D, ws + w.xc, (int)L, D * L, d->x_proj_weight, (int)D, 0, 0.0f, ws + w.x_dbl, d->io_dtype,
               (int)L, M * L, d->batch);
    if (rc != MXVL_OK) return rc;
    // delta_b^T (L, D) = dt_b^T (L, R) Wdt^T (R, D)
    rc = g.run(RB_OP_N, RB_OP_N, (int)L, (int)D, (int)R, ws + w.x_dbl, (int)L, M * L, d->dt_proj_weight, (int)R, 0, 0.0f, ws + w.delta,
               d->io_dtype, (int)L, D * L, d->batch);
    if (rc != MXVL_OK) return rc;
    rc = mxvl_scan_fwd(&s, hip_stream);
    if (rc != MXVL_OK) return rc;
    if (d->out_proj_weight) {
      if (d->out_proj_bias) MXVL_MI_BY_DTYPE(d->io_dtype, launch_fill_rows<io_t>(d->out, d->out_proj_bias, (int64_t)d->batch * L, d->d_model, stream));
      // out_b^T (d_model, L) = Wo (d_model, D) y_b (D, L)
      rc = g.run(RB_OP_T, RB_OP_T, d->d_model, (int)L, (int)D, d->out_proj_weight, (int)D, 0, ws + w.y, (int)L, D * L, d->out_proj_bias ? 1.0f : 0.0f,
                 d->out, d->io_dtype, d->d_model, (int64_t)L * d->d_model, d->batch);
      if (rc != MXVL_OK) return rc;
    }
  }
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

int mxvl_mamba_inner_bwd(const mxvl_mamba_inner_bwd_desc* b, void* hip_stream) {
  if (!b) return MXVL_ERR_NULL;
  const mxvl_mamba_inner_desc* d = &b->fwd;
  int rc = mi_check(d);
  if (rc != MXVL_OK) return rc;
  if (!d->workspace || !b->workspace || !b->dout || !b->dxz || !b->dconv_weight || !b->dx_proj_weight || !b->ddt_proj_weight || !b->dA)
    return MXVL_ERR_NULL;
  if ((d->conv_bias && !b->dconv_bias) || (d->D && !b->dD) || (d->delta_bias && !b->ddelta_bias) || (d->out_proj_weight && !b->dout_proj_weight) ||
      (d->out_proj_bias && !b->dout_proj_bias))
    return MXVL_ERR_NULL;
  const FwdWs w = fwd_ws(d);
  const BwdWs v = bwd_ws(d);
  if (d->workspace_bytes < w.total || b->workspace_bytes < v.total) return MXVL_ERR_SHAPE;
  RocBlas* rb = rocblas_get();
  if (!rb) return MXVL_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)hip_stream;
  const int64_t e = esz(d->io_dtype), D = d->dim, L = d->seqlen, N = d->dstate, R = d->dt_rank, M = R + 2 * N;
  const int io = d->io_dtype;
  char* ws = (char*)d->workspace;
  char* bs = (char*)b->workspace;
  float* dBC = (float*)(bs + v.dBC);
  if (hipMemsetAsync(dBC, 0, (size_t)d->batch * 2 * N * L * 4, stream) != hipSuccess) return MXVL_ERR_LAUNCH;

  std::lock_guard<std::mutex> lock(rb->mu);
  if (rb->set_stream(rb->handle, stream) != 0) return MXVL_ERR_LAUNCH;
  const Gemm g{rb, io};
  const void* dy = b->dout;
  if (d->out_proj_weight) {
    const int dm = d->d_model;
    // dy_b^T (L, D) = dout_b (L, d_model) Wo (d_model, D)
    rc = g.run(RB_OP_T, RB_OP_T, (int)L, (int)D, dm, b->dout, dm, (int64_t)L * dm, d->out_proj_weight, (int)D, 0, 0.0f, bs + v.dy, io, (int)L, D * L,
               d->batch);
    if (rc != MXVL_OK) return rc;
    // dWo^T (D, d_model) += y_b (D, L) dout_b (L, d_model), one batch element after the other (fp32 accumulator)
    for (int i = 0; i < d->batch; ++i) {
      rc = g.run(RB_OP_T, RB_OP_T, (int)D, dm, (int)L, ws + w.y + (int64_t)i * D * L * e, (int)L, 0, (const char*)b->dout + (int64_t)i * L * dm * e, dm, 0,
                 1.0f, b->dout_proj_weight, MXVL_F32, (int)D, 0, 1);
      if (rc != MXVL_OK) return rc;
    }
    if (d->out_proj_bias) MXVL_MI_BY_DTYPE(io, launch_colsum<io_t>(b->dout, (float*)b->dout_proj_bias, (int64_t)d->batch * L, dm, stream));
    dy = bs + v.dy;
  }

  mxvl_scan_bwd_desc sb{};
  scan_desc(d, w, sb.fwd);
  sb.dout_bs = D * L; sb.dout_ds = L; sb.du_bs = D * L; sb.du_ds = L; sb.ddelta_bs = D * L; sb.ddelta_ds = L; sb.dz_bs = 2 * D * L; sb.dz_ds = L;
  sb.dB_bs = 2 * N * L; sb.dB_gs = 0; sb.dB_ns = L; sb.dC_bs = 2 * N * L; sb.dC_gs = 0; sb.dC_ns = L;
  sb.dout = dy; sb.du = bs + v.du; sb.ddelta = bs + v.ddelta; sb.dz = (char*)b->dxz + D * L * e;
  sb.dA = b->dA; sb.dB = dBC; sb.dC = dBC + N * L; sb.dD = b->dD; sb.ddelta_bias = b->ddelta_bias;
  rc = mxvl_scan_bwd(&sb, hip_stream);
  if (rc != MXVL_OK) return rc;

  // d(x_dbl)_b rows: dt rows = Wdt^T ddelta_b, B | C rows = the fp32 accumulators in the io dtype
  rc = g.run(RB_OP_N, RB_OP_T, (int)L, (int)R, (int)D, bs + v.ddelta, (int)L, D * L, d->dt_proj_weight, (int)R, 0, 0.0f, bs + v.dx_dbl, io, (int)L, M * L,
             d->batch);
  if (rc != MXVL_OK) return rc;
  MXVL_MI_BY_DTYPE(io, launch_cast_rows<io_t>(dBC, 2 * N * L, bs + v.dx_dbl + R * L * e, M * L, 2 * N * L, d->batch, stream));
  for (int i = 0; i < d->batch; ++i) {
    // dWdt^T (R, D) += dt_b (R, L) ddelta_b^T (L, D)
    rc = g.run(RB_OP_T, RB_OP_N, (int)R, (int)D, (int)L, ws + w.x_dbl + (int64_t)i * M * L * e, (int)L, 0, bs + v.ddelta + (int64_t)i * D * L * e, (int)L, 0,
               1.0f, b->ddt_proj_weight, MXVL_F32, (int)R, 0, 1);
    if (rc != MXVL_OK) return rc;
    // dWx^T (D, M) += xc_b (D, L) d(x_dbl)_b^T (L, M)
    rc = g.run(RB_OP_T, RB_OP_N, (int)D, (int)M, (int)L, ws + w.xc + (int64_t)i * D * L * e, (int)L, 0, bs + v.dx_dbl + (int64_t)i * M * L * e, (int)L, 0,
               1.0f, b->dx_proj_weight, MXVL_F32, (int)D, 0, 1);
    if (rc != MXVL_OK) return rc;
  }
  // du_b^T (L, D) += d(x_dbl)_b^T (L, M) Wx (M, D): x_proj's data gradient lands on the scan's du (the addmm of the reference's backward)
  rc = g.run(RB_OP_N, RB_OP_T, (int)L, (int)D, (int)M, bs + v.dx_dbl, (int)L, M * L, d->x_proj_weight, (int)D, 0, 1.0f, bs + v.du, io, (int)L, D * L,
             d->batch);
  if (rc != MXVL_OK) return rc;

  mxvl_conv1d_bwd_desc cb{};
  cb.fwd.batch = d->batch; cb.fwd.dim = d->dim; cb.fwd.seqlen = d->seqlen; cb.fwd.width = d->width; cb.fwd.io_dtype = io; cb.fwd.silu = 1;
  cb.fwd.x_bs = 2 * D * L; cb.fwd.x_ds = L; cb.fwd.y_bs = D * L; cb.fwd.y_ds = L;
  cb.fwd.x = d->xz; cb.fwd.weight = d->conv_weight; cb.fwd.bias = d->conv_bias;
  cb.dy_bs = D * L; cb.dy_ds = L; cb.dx_bs = 2 * D * L; cb.dx_ds = L;
  cb.dy = bs + v.du; cb.dx = b->dxz; cb.dweight = b->dconv_weight; cb.dbias = b->dconv_bias;
  rc = mxvl_conv1d_bwd(&cb, hip_stream);
  if (rc != MXVL_OK) return rc;
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}

}  // extern "C"
