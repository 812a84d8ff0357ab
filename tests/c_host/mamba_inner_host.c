/* A plain-C host of libmxvl.so: mamba_inner_fn (conv1d + SiLU -> x_proj -> dt_proj -> selective scan -> out_proj) forward and
 * backward through the ONE entry pair mxvl_mamba_inner_fwd / mxvl_mamba_inner_bwd (include/mxvl.h, ABI v11) -- no torch, no C++:
 * hipMalloc'd buffers, plain pointers and sizes.  tests/test_mixer_gpu.py builds it (gcc + the HIP runtime), feeds it fp32 tensors
 * through files and compares what it writes back with the package's own autograd node.
 *   mamba_inner_host <in.bin> <out.bin> batch dim seqlen dstate dt_rank d_model
 * in.bin  (fp32): xz | conv_w (dim,4) | conv_b | x_proj_w | dt_proj_w | out_proj_w | out_proj_b | A | D | delta_bias | dout
 * out.bin (fp32): out | dxz | dconv_w | dconv_b | dx_proj_w | ddt_proj_w | dout_proj_w | dout_proj_b | dA | dD | ddelta_bias */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "mxvl.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static float* dev_from(FILE* f, size_t n) {
  float* h = (float*)malloc(n * sizeof(float));
  float* d = NULL;
  if (!h || fread(h, sizeof(float), n, f) != n) { fprintf(stderr, "short input\n"); exit(3); }
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess || hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) exit(4);
  free(h);
  return d;
}
static float* dev_zero(size_t n) {
  float* d = NULL;
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess || hipMemset(d, 0, n * sizeof(float)) != hipSuccess) exit(4);
  return d;
}
static void dump(FILE* f, const float* d, size_t n) {
  float* h = (float*)malloc(n * sizeof(float));
  if (!h || hipMemcpy(h, d, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess || fwrite(h, sizeof(float), n, f) != n) exit(5);
  free(h);
}

int main(int argc, char** argv) {
  if (argc != 9) { fprintf(stderr, "usage: %s in.bin out.bin batch dim seqlen dstate dt_rank d_model\n", argv[0]); return 1; }
  const int B = atoi(argv[3]), D = atoi(argv[4]), L = atoi(argv[5]), N = atoi(argv[6]), R = atoi(argv[7]), dm = atoi(argv[8]);
  const size_t M = (size_t)R + 2 * N, W = 4;
  if (mxvl_abi_version() != MXVL_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 1;
  mxvl_mamba_inner_bwd_desc b = {0};
  mxvl_mamba_inner_desc* d = &b.fwd;
  d->batch = B; d->dim = D; d->seqlen = L; d->dstate = N; d->dt_rank = R; d->width = (int)W; d->d_model = dm;
  d->io_dtype = MXVL_F32; d->flags = MXVL_SCAN_DELTA_SOFTPLUS;
  d->xz = dev_from(in, (size_t)B * 2 * D * L);
  d->conv_weight = dev_from(in, (size_t)D * W);
  d->conv_bias = dev_from(in, D);
  d->x_proj_weight = dev_from(in, M * D);
  d->dt_proj_weight = dev_from(in, (size_t)D * R);
  d->out_proj_weight = dev_from(in, (size_t)dm * D);
  d->out_proj_bias = dev_from(in, dm);
  d->A = dev_from(in, (size_t)D * N);
  d->D = dev_from(in, D);
  d->delta_bias = dev_from(in, D);
  b.dout = dev_from(in, (size_t)B * L * dm);
  fclose(in);

  float* out = dev_zero((size_t)B * L * dm);
  d->out = out;
  d->workspace_bytes = mxvl_mamba_inner_workspace_bytes(d);
  b.workspace_bytes = mxvl_mamba_inner_bwd_workspace_bytes(d);
  if (d->workspace_bytes < 0 || b.workspace_bytes < 0) { fprintf(stderr, "workspace_bytes refused the descriptor\n"); return 1; }
  CK(hipMalloc(&d->workspace, (size_t)d->workspace_bytes));
  CK(hipMalloc(&b.workspace, (size_t)b.workspace_bytes));
  hipStream_t stream;
  CK(hipStreamCreate(&stream));
  int rc = mxvl_mamba_inner_fwd(d, stream);
  if (rc != MXVL_OK) { fprintf(stderr, "mxvl_mamba_inner_fwd: %d (hip %d)\n", rc, mxvl_last_hip_error()); return 1; }

  float* dxz = dev_zero((size_t)B * 2 * D * L);
  float *dcw = dev_zero((size_t)D * W), *dcb = dev_zero(D), *dwx = dev_zero(M * D), *dwdt = dev_zero((size_t)D * R);
  float *dwo = dev_zero((size_t)dm * D), *dbo = dev_zero(dm), *dA = dev_zero((size_t)D * N), *dD = dev_zero(D), *ddb = dev_zero(D);
  b.dxz = dxz; b.dconv_weight = dcw; b.dconv_bias = dcb; b.dx_proj_weight = dwx; b.ddt_proj_weight = dwdt;
  b.dout_proj_weight = dwo; b.dout_proj_bias = dbo; b.dA = dA; b.dD = dD; b.ddelta_bias = ddb;
  rc = mxvl_mamba_inner_bwd(&b, stream);
  if (rc != MXVL_OK) { fprintf(stderr, "mxvl_mamba_inner_bwd: %d (hip %d)\n", rc, mxvl_last_hip_error()); return 1; }
  CK(hipStreamSynchronize(stream));

  FILE* of = fopen(argv[2], "wb");
  if (!of) return 1;
  dump(of, out, (size_t)B * L * dm);
  dump(of, dxz, (size_t)B * 2 * D * L);
  dump(of, dcw, (size_t)D * W); dump(of, dcb, D); dump(of, dwx, M * D); dump(of, dwdt, (size_t)D * R);
  dump(of, dwo, (size_t)dm * D); dump(of, dbo, dm); dump(of, dA, (size_t)D * N); dump(of, dD, D); dump(of, ddb, D);
  fclose(of);
  printf("ok\n");
  return 0;
}
