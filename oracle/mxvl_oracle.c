/*
 * mxvl_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, fp32, sequential restatement of the reference's CPU definitions of the
 * MambaXray-VL hot-path operators.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path
 * (medical_image_analysis_amd/) never does.
 *
 * Parity pin: checked against golden vectors produced by importing the reference's
 * own Python (tests/golden/make_golden.py, run in the build container) -- see
 * tests/test_oracle_golden.py.
 *
 * Reference definitions followed (paths relative to the reference checkout):
 *   orc_scan_fwd    selective_scan_ref
 *                   R2GenCSR/VMamba/kernels/selective_scan/test_selective_scan_easy.py:857-922
 *                   (identical to test_selective_scan.py:168-234)
 *   orc_scan_bwd    the gradient of the above (the reference obtains it by autograd through
 *                   selective_scan_ref; the CUDA statement of the same math is
 *                   csrc/selective_scan/cus/selective_scan_bwd_kernel.cuh:125-272)
 *   orc_conv1d_fwd  act(conv1d(x)[..., :L]), nn.Conv1d(D, D, W, groups=D, padding=W-1)
 *                   CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:78-86,672-673
 *   orc_conv1d_bwd  gradient of the above
 *   orc_conv1d_update / orc_state_update   mamba_simple.py:724-730 / :748-755 (decode step)
 *
 * All tensors are dense fp32, row-major, shapes as in include/mxvl.h.
 * Threading: rows (b,d) are independent; built with -fopenmp the outer loops are
 * parallel (cpu_baseline reports the thread count it used).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* F.softplus(x) with the default beta=1, threshold=20 (selective_scan_ref line 879) */
static inline float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
static inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

/*
 * selective_scan_ref, lines 857-922.
 *   delta' = softplus?(delta + bias)                          (:876-879)
 *   deltaA = exp(delta' * A); deltaB_u = delta' * B * u       (:892-901)
 *   x = deltaA[t] * x + deltaB_u[t]; y[t] = sum_n x * C[t]    (:905-913)
 *   out = y + u*D; out *= silu(z)                             (:918-920)
 * B,C: (batch, G, N, L), row d uses group d / (dim/G)         (:899,902-903)
 * last_state (batch,dim,N) = x after the last step            (:914-915), may be NULL.
 */
void orc_scan_fwd(const float *u, const float *delta, const float *A, const float *B, const float *C,
                  const float *D, const float *z, const float *delta_bias, int delta_softplus,
                  int batch, int dim, int L, int N, int G, float *out, float *last_state) {
  const int dpg = dim / G;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < batch; ++b) {
    for (int d = 0; d < dim; ++d) {
      const int g = d / dpg;
      const float *ur = u + ((size_t)b * dim + d) * L;
      const float *dr = delta + ((size_t)b * dim + d) * L;
      const float *zr = z ? z + ((size_t)b * dim + d) * L : NULL;
      const float *Br = B + ((size_t)b * G + g) * N * L;
      const float *Cr = C + ((size_t)b * G + g) * N * L;
      const float *Ar = A + (size_t)d * N;
      float *outr = out + ((size_t)b * dim + d) * L;
      float h[256];
      for (int n = 0; n < N; ++n) h[n] = 0.0f;
      const float bias = delta_bias ? delta_bias[d] : 0.0f;
      const float Dd = D ? D[d] : 0.0f;
      for (int t = 0; t < L; ++t) {
        float dt = dr[t] + bias;
        if (delta_softplus) dt = softplus_f(dt);
        const float ut = ur[t];
        float y = 0.0f;
        for (int n = 0; n < N; ++n) {
          const float a = expf(dt * Ar[n]);
          h[n] = a * h[n] + dt * Br[(size_t)n * L + t] * ut;
          y += h[n] * Cr[(size_t)n * L + t];
        }
        if (D) y += ut * Dd;
        if (zr) y *= zr[t] * sigmoid_f(zr[t]);
        outr[t] = y;
      }
      if (last_state)
        for (int n = 0; n < N; ++n) last_state[((size_t)b * dim + d) * N + n] = h[n];
    }
  }
}

/*
 * Gradient of orc_scan_fwd w.r.t. every floating input, given dout.
 *   dy_t   = dout_t * silu(z_t);  dz_t = dout_t * y_t * silu'(z_t)
 *   g_t    = C_t * dy_t + a_{t+1} * g_{t+1}                      (reverse recurrence, per n)
 *   dC_t  += dy_t * h_t ; dB_t += g_t * delta'_t * u_t           (summed over the rows of a group)
 *   du_t   = dy_t * D + delta'_t * sum_n g_t B_t
 *   ddelta'_t = u_t * sum_n g_t B_t + sum_n g_t h_{t-1} a_t A_n
 *   dA_n  += sum_t g_t h_{t-1} a_t delta'_t ; dD += sum_t dy_t u_t
 *   ddelta_t = ddelta'_t * sigmoid(delta_t + bias)  (softplus) ; dbias += sum ddelta_t
 * du, ddelta, dz are written; dA, dB, dC, dD, ddelta_bias are ZERO-FILLED here and then summed
 * (deterministic order: b, then d, then t).
 */
void orc_scan_bwd(const float *u, const float *delta, const float *A, const float *B, const float *C,
                  const float *D, const float *z, const float *delta_bias, int delta_softplus,
                  const float *dout, int batch, int dim, int L, int N, int G, float *du, float *ddelta,
                  float *dA, float *dB, float *dC, float *dD, float *dz, float *ddelta_bias) {
  const int dpg = dim / G;
  memset(dA, 0, sizeof(float) * (size_t)dim * N);
  memset(dB, 0, sizeof(float) * (size_t)batch * G * N * L);
  memset(dC, 0, sizeof(float) * (size_t)batch * G * N * L);
  if (dD) memset(dD, 0, sizeof(float) * dim);
  if (ddelta_bias) memset(ddelta_bias, 0, sizeof(float) * dim);
  /* rows of one (b, group) share dB/dC: parallelise over (b, group) so the sum order is fixed */
#pragma omp parallel
  {
    float *hs = (float *)malloc(sizeof(float) * (size_t)(L + 1) * N); /* h_{-1..L-1} */
    float *as = (float *)malloc(sizeof(float) * (size_t)L * N);
    float *dts = (float *)malloc(sizeof(float) * (size_t)L);
    float *dA_loc = (float *)calloc((size_t)dim * N, sizeof(float));
    float *dD_loc = (float *)calloc((size_t)dim, sizeof(float));
    float *db_loc = (float *)calloc((size_t)dim, sizeof(float));
#pragma omp for collapse(2) schedule(static)
    for (int b = 0; b < batch; ++b) {
      for (int g = 0; g < G; ++g) {
        const float *Br = B + ((size_t)b * G + g) * N * L;
        const float *Cr = C + ((size_t)b * G + g) * N * L;
        float *dBr = dB + ((size_t)b * G + g) * N * L;
        float *dCr = dC + ((size_t)b * G + g) * N * L;
        for (int d = g * dpg; d < (g + 1) * dpg; ++d) {
          const size_t row = ((size_t)b * dim + d) * L;
          const float *ur = u + row, *dr = delta + row, *doutr = dout + row;
          const float *zr = z ? z + row : NULL;
          const float *Ar = A + (size_t)d * N;
          const float bias = delta_bias ? delta_bias[d] : 0.0f;
          const float Dd = D ? D[d] : 0.0f;
          /* forward recompute, keeping h and a */
          for (int n = 0; n < N; ++n) hs[n] = 0.0f;
          for (int t = 0; t < L; ++t) {
            float dt = dr[t] + bias;
            if (delta_softplus) dt = softplus_f(dt);
            dts[t] = dt;
            for (int n = 0; n < N; ++n) {
              const float a = expf(dt * Ar[n]);
              as[(size_t)t * N + n] = a;
              hs[(size_t)(t + 1) * N + n] = a * hs[(size_t)t * N + n] + dt * Br[(size_t)n * L + t] * ur[t];
            }
          }
          float gacc[256];
          for (int n = 0; n < N; ++n) gacc[n] = 0.0f; /* a_{t+1} * g_{t+1} */
          for (int t = L - 1; t >= 0; --t) {
            const float dt = dts[t], ut = ur[t];
            float y = 0.0f;
            for (int n = 0; n < N; ++n) y += hs[(size_t)(t + 1) * N + n] * Cr[(size_t)n * L + t];
            if (D) y += ut * Dd;
            float dy = doutr[t];
            if (zr) {
              const float s = sigmoid_f(zr[t]);
              dz[row + t] = doutr[t] * y * s * (1.0f + zr[t] * (1.0f - s));
              dy = doutr[t] * zr[t] * s;
            }
            float gB = 0.0f, gha = 0.0f;
            for (int n = 0; n < N; ++n) {
              const float gn = Cr[(size_t)n * L + t] * dy + gacc[n];
              const float a = as[(size_t)t * N + n];
              const float ha = gn * hs[(size_t)t * N + n] * a; /* g_t h_{t-1} a_t */
              dCr[(size_t)n * L + t] += dy * hs[(size_t)(t + 1) * N + n];
              dBr[(size_t)n * L + t] += gn * dt * ut;
              gB += gn * Br[(size_t)n * L + t];
              gha += ha * Ar[n];
              dA_loc[(size_t)d * N + n] += ha * dt;
              gacc[n] = a * gn;
            }
            du[row + t] = dy * Dd + dt * gB;
            if (D) dD_loc[d] += dy * ut;
            float dd = ut * gB + gha;
            if (delta_softplus) {
              const float x = dr[t] + bias;
              dd *= (x > 20.0f) ? 1.0f : sigmoid_f(x);
            }
            ddelta[row + t] = dd;
            db_loc[d] += dd;
          }
        }
      }
    }
#pragma omp critical
    {
      for (size_t i = 0; i < (size_t)dim * N; ++i) dA[i] += dA_loc[i];
      if (dD) for (int i = 0; i < dim; ++i) dD[i] += dD_loc[i];
      if (ddelta_bias) for (int i = 0; i < dim; ++i) ddelta_bias[i] += db_loc[i];
    }
    free(hs); free(as); free(dts); free(dA_loc); free(dD_loc); free(db_loc);
  }
}

/*
 * Depthwise causal conv1d + optional SiLU (mamba_simple.py:78-86, 672-673):
 * y[b,d,t] = act(bias[d] + sum_{k<W} w[d,k] * x[b,d,t-(W-1)+k]),  x[<0] = 0.
 */
void orc_conv1d_fwd(const float *x, const float *w, const float *bias, int silu, int batch, int dim,
                    int L, int W, float *y) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < batch; ++b)
    for (int d = 0; d < dim; ++d) {
      const float *xr = x + ((size_t)b * dim + d) * L;
      float *yr = y + ((size_t)b * dim + d) * L;
      for (int t = 0; t < L; ++t) {
        float acc = bias ? bias[d] : 0.0f;
        for (int k = 0; k < W; ++k) {
          const int s = t - (W - 1) + k;
          if (s >= 0) acc += w[(size_t)d * W + k] * xr[s];
        }
        yr[t] = silu ? acc * sigmoid_f(acc) : acc;
      }
    }
}

/* gradient of orc_conv1d_fwd; dw (dim,W) and dbias (dim) are zero-filled then summed over (b,t) */
void orc_conv1d_bwd(const float *x, const float *w, const float *bias, int silu, const float *dy,
                    int batch, int dim, int L, int W, float *dx, float *dw, float *dbias) {
  memset(dw, 0, sizeof(float) * (size_t)dim * W);
  if (dbias) memset(dbias, 0, sizeof(float) * dim);
#pragma omp parallel for schedule(static)
  for (int d = 0; d < dim; ++d) {
    float *pre = (float *)malloc(sizeof(float) * (size_t)L);
    for (int b = 0; b < batch; ++b) {
      const float *xr = x + ((size_t)b * dim + d) * L;
      const float *dyr = dy + ((size_t)b * dim + d) * L;
      float *dxr = dx + ((size_t)b * dim + d) * L;
      for (int t = 0; t < L; ++t) {
        float g = dyr[t];
        if (silu) {
          float acc = bias ? bias[d] : 0.0f;
          for (int k = 0; k < W; ++k) {
            const int s = t - (W - 1) + k;
            if (s >= 0) acc += w[(size_t)d * W + k] * xr[s];
          }
          const float sg = sigmoid_f(acc);
          g *= sg * (1.0f + acc * (1.0f - sg));
        }
        pre[t] = g;
        if (dbias) dbias[d] += g;
        for (int k = 0; k < W; ++k) {
          const int s = t - (W - 1) + k;
          if (s >= 0) dw[(size_t)d * W + k] += g * xr[s];
        }
      }
      for (int s = 0; s < L; ++s) {
        float acc = 0.0f;
        for (int k = 0; k < W; ++k) {
          const int t = s + (W - 1) - k;
          if (t < L) acc += w[(size_t)d * W + k] * pre[t];
        }
        dxr[s] = acc;
      }
    }
    free(pre);
  }
}

/* decode-step conv (mamba_simple.py:724-730): roll state left, append x, dot with w, +bias, SiLU */
void orc_conv1d_update(const float *x, float *conv_state, const float *w, const float *bias, int silu,
                       int batch, int dim, int W, float *y) {
  for (int b = 0; b < batch; ++b)
    for (int d = 0; d < dim; ++d) {
      float *st = conv_state + ((size_t)b * dim + d) * W;
      for (int k = 0; k + 1 < W; ++k) st[k] = st[k + 1];
      st[W - 1] = x[(size_t)b * dim + d];
      float acc = 0.0f;
      for (int k = 0; k < W; ++k) acc += st[k] * w[(size_t)d * W + k];
      if (bias) acc += bias[d];
      y[(size_t)b * dim + d] = silu ? acc * sigmoid_f(acc) : acc;
    }
}

/*
 * decode-step SSM (mamba_simple.py:748-755):
 * dt = softplus(dt + dt_bias); state = state*exp(dt*A) + x*dt*B; y = <state,C> + D*x; y *= silu(z)
 */
void orc_state_update(float *state, const float *x, const float *dt, const float *A, const float *B,
                      const float *C, const float *D, const float *z, const float *dt_bias,
                      int dt_softplus, int batch, int dim, int N, float *out) {
  for (int b = 0; b < batch; ++b)
    for (int d = 0; d < dim; ++d) {
      float dtv = dt[(size_t)b * dim + d] + (dt_bias ? dt_bias[d] : 0.0f);
      if (dt_softplus) dtv = softplus_f(dtv);
      const float xv = x[(size_t)b * dim + d];
      float *st = state + ((size_t)b * dim + d) * N;
      float y = 0.0f;
      for (int n = 0; n < N; ++n) {
        st[n] = st[n] * expf(dtv * A[(size_t)d * N + n]) + xv * dtv * B[(size_t)b * N + n];
        y += st[n] * C[(size_t)b * N + n];
      }
      if (D) y += D[d] * xv;
      if (z) { const float zv = z[(size_t)b * dim + d]; y *= zv * sigmoid_f(zv); }
      out[(size_t)b * dim + d] = y;
    }
}

/* ---- VMamba 4-direction orderings (R2GenCSR/VMamba/classification/models/vmamba.py:25-67) --------------------------
 * cross_scan : x (B,C,H,W) -> xs (B,4,C,L): xs0 row-major, xs1 column-major, xs2 / xs3 their reversals (:27-35).
 * cross_merge: ys (B,4,C,L) -> y (B,C,L) = (ys0 + flip ys2) + transpose(ys1 + flip ys3) (:48-55), fp32 adds in that
 * association. */
void orc_cross_scan(const float *x, float *xs, int B, int C, int H, int W) {
  const long L = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float *src = x + ((long)b * C + c) * L;
      float *o0 = xs + (((long)b * 4 + 0) * C + c) * L, *o1 = xs + (((long)b * 4 + 1) * C + c) * L;
      float *o2 = xs + (((long)b * 4 + 2) * C + c) * L, *o3 = xs + (((long)b * 4 + 3) * C + c) * L;
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          const float v = src[(long)h * W + w];
          const long lr = (long)h * W + w, lc = (long)w * H + h;
          o0[lr] = v; o2[L - 1 - lr] = v; o1[lc] = v; o3[L - 1 - lc] = v;
        }
    }
}

void orc_cross_merge(const float *ys, float *y, int B, int C, int H, int W) {
  const long L = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float *i0 = ys + (((long)b * 4 + 0) * C + c) * L, *i1 = ys + (((long)b * 4 + 1) * C + c) * L;
      const float *i2 = ys + (((long)b * 4 + 2) * C + c) * L, *i3 = ys + (((long)b * 4 + 3) * C + c) * L;
      float *dst = y + ((long)b * C + c) * L;
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          const long lr = (long)h * W + w, lc = (long)w * H + h;
          const float a = i0[lr] + i2[L - 1 - lr];
          const float t = i1[lc] + i3[L - 1 - lc];
          dst[lr] = a + t;
        }
    }
}
