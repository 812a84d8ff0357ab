/*
 * mxvl.h -- C-ABI of libmxvl.so, the MI355X (gfx950) native hot path of
 * MambaXray-VL (Event-AHU/Medical_Image_Analysis).
 *
 * Every entry point takes plain device pointers + sizes + element strides and a
 * hipStream_t passed as void*.  No torch / C++ types cross this boundary.  All
 * functions return MXVL_OK (0) or a negative mxvl_status; they never throw and
 * never allocate: the caller owns every buffer.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference checkout, R2GenCSR/VMamba/kernels/selective_scan = KSS):
 *
 *   mxvl_scan_fwd        selective_scan_fwd   KSS/csrc/selective_scan/cus/selective_scan.cpp:157-239
 *                        (oflex twin          KSS/csrc/selective_scan/cusoflex/selective_scan_oflex.cpp:144-232)
 *                        = mamba_ssm selective_scan_fn forward as called at
 *                        CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:693-704
 *   mxvl_scan_bwd        selective_scan_bwd   KSS/csrc/selective_scan/cus/selective_scan.cpp:241-349
 *   mxvl_conv1d_fwd/bwd  causal_conv1d_fn     call site mamba_simple.py:676-681, definition = the in-repo
 *                        fallback act(conv1d(x)[..., :L]) at mamba_simple.py:672-673
 *   mxvl_conv1d_update   causal_conv1d_update call site mamba_simple.py:732-738, fallback :724-730
 *   mxvl_state_update    selective_state_update call site mamba_simple.py:757-759, fallback :748-755
 *
 * Tensor conventions (identical to the reference op boundary):
 *   u, delta, z, out : (batch, dim, seqlen)   seqlen stride 1
 *   A                : (dim, dstate) fp32     (= -exp(A_log))
 *   B, C             : (batch, n_groups, dstate, seqlen)  seqlen stride 1
 *   D, delta_bias    : (dim) fp32
 * "io" tensors (u, delta, z, out, B, C and their gradients) share one dtype
 * (fp32 / bf16 / fp16); weights, state and accumulators are always fp32.
 * Alignment and speed: any element-aligned view is accepted.  The vector kernels need every row of every io tensor to start on
 * a 4-element boundary (fp32: 16 bytes) and, for the 16-bit dtypes, on a 16-byte boundary (8 elements: strides % 8 == 0,
 * pointers % 16 == 0) -- forward and backward apply the same rule; rows that are only 8-byte aligned (e.g. seqlen % 8 == 4, or a
 * view starting 4 elements into its storage) run on the element-wise kernels (same results, several times slower), and
 * MXVL_SCAN_FOLD_BATCH returns MXVL_ERR_UNSUPPORTED for them.
 * dstate 1 without z (VMamba / R2GenCSR SS2D, vmamba.py:294-312; the vendored oflex kernels' own test configuration): forward and
 * backward have kernels of their own (csrc/scan_n1.h, scan_n1_bwd.h) -- rows of the whole launch walked as one flat sequence
 * (forward) / pass-major per wave (backward), wave-wide DPP scans, no LDS tiles.  Taken when seqlen % 4 == 0 and the rows of u,
 * delta, B, C start on 4-element boundaries (8 elements for 16-bit rows with seqlen % 8 == 0: 16-byte accesses); the backward also
 * needs dim / n_groups % 4 == 0.  Same descriptors, same checkpoint layout (the state entering every 128-step chunk), same
 * tolerances: nothing about the call changes, and either direction may run on the general kernels while the other does not.  Rows of
 * at most 128 steps that miss those alignments (seqlen % 4 != 0: the 7 x 7 stage) take a lane-per-row kernel pair (csrc/scan_n1_short.h)
 * when u / delta / out (and dout / du / ddelta) are dense (batch, dim, seqlen) arrays and 64 | dim / n_groups.
 *
 * Deviations from SURVEY.md section 8-b, recorded here because this header is the boundary:
 *   * `mxvl_mamba_inner_fwd / mxvl_mamba_inner_bwd` (one entry for mamba_inner_fn, CXPMRG_Bench_MambaXray_VL/pretrain/
 *     mamba_simple.py:388-402) are exported since ABI v11 as a COMPOSED entry: mxvl_conv1d_*, mxvl_scan_* and four rocBLAS GEMMs
 *     behind one call (see the entry's comment below), not as one fused kernel.  Fusing the two projections into the scan prologue
 *     would make every channel tile of the backward re-derive dB / dC contributions of the projection (3x the fp32 atomics of
 *     mxvl_scan_bwd) for a GEMM share below half a percent of the step.
 *   * Tolerance of the scan entries against the reference's selective_scan_ref (KSS/test_selective_scan_easy.py:857-922), as the
 *     parity tests assert it: fp32 io -- |got - ref| <= 1e-4 * max(1, max|ref| / 32) + 1e-5 * |ref|, i.e. north_star's flat
 *     atol 1e-4 wherever |ref| < 32 and the same bound relative to the output scale beyond (the L = 4097 golden reaches 262:
 *     the C oracle itself needs that scaling against the reference's fp32 torch loop); bf16 / fp16 io -- the reference test's own
 *     rtol / atol (test_selective_scan.py: 3e-2 / 5e-2 bf16, 3e-3 / 5e-3 fp16).  Index / ordering entries (mxvl_row_gather,
 *     mxvl_cross_scan / _merge, mxvl_dir_gather, mxvl_patch_cols, mxvl_image_preprocess, mxvl_beam_step's choices) are bit-exact.
 *   * mxvl_add_layernorm_fwd / _bwd serve EVERY row width: cols = 256 * k, k in {1, 2, 3, 4, 6, 8} (256 .. 2048: every width of the
 *     reference's ARM / VisionMamba / ViT-MAE factories) a row per wave in registers, the narrow rows 64, 128, 192, 384 four / two rows
 *     per wave (VMamba's first stage and patch embedding, the constructors' default embed_dim = 192), and since ABI v11 any other
 *     width on a wave-per-row element-wise kernel pair (same arithmetic and rounding points, several times slower): the Python mirror
 *     no longer routes any width of a HIP tensor to torch.nn.functional.layer_norm.
 *   * GEMMs: the token-major forward / data-gradient products of the blocks' linears stay on the vendor library (hipBLASLt through
 *     torch / rocBLAS: 0.95-1.48 PFLOP/s at the step's shapes, where the hand-written NT kernel mxvl_gemm_nt measured slower on 7 of
 *     8 shapes); the WEIGHT gradients dW = dy^T x -- two K-major operands, the layout the library is weak on -- are mxvl_gemm_tn
 *     (ABI v12) wherever it was measured to win (token axis >= 32 000, a tile count that fills an XCD's workgroups:
 *     selective_scan_interface.gemm_tn_wins), the library's batched split-K form elsewhere.
 */
#ifndef MXVL_H_
#define MXVL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MXVL_ABI_VERSION 12

typedef enum mxvl_status {
  MXVL_OK = 0,
  MXVL_ERR_NULL = -1,        /* required pointer is NULL                        */
  MXVL_ERR_DTYPE = -2,       /* io dtype not in {f32,bf16,f16}                  */
  MXVL_ERR_SHAPE = -3,       /* non-positive size, dim % n_groups != 0, ...     */
  MXVL_ERR_DSTATE = -4,      /* dstate > MXVL_MAX_DSTATE                        */
  MXVL_ERR_STRIDE = -5,      /* a stride that would alias / is negative         */
  MXVL_ERR_LAUNCH = -6,      /* hipLaunchKernel failed (see mxvl_last_hip_error)*/
  MXVL_ERR_UNSUPPORTED = -7, /* valid request this build has no kernel for      */
  MXVL_ERR_CHECKPOINT = -8   /* seqlen > one chunk but no checkpoint buffer     */
} mxvl_status;

typedef enum mxvl_dtype { MXVL_F32 = 0, MXVL_BF16 = 1, MXVL_F16 = 2 } mxvl_dtype;

#define MXVL_MAX_DSTATE 256

/* flag bits of mxvl_scan_desc.flags */
#define MXVL_SCAN_DELTA_SOFTPLUS 1u
/* `out` (mxvl_scan_fwd) and `dout` (mxvl_scan_bwd) are fp32 whatever io_dtype says; their strides count fp32 elements.
 * The vendored VMamba extension's "oflex" i16o32 mode (cusoflex/selective_scan_oflex.cpp:150,207): half-precision inputs,
 * the fp32 accumulator stored unrounded. */
#define MXVL_SCAN_OUT_F32 2u
/* ABI v5: fold the batch into the sequence.  A workgroup walks several batch elements of its channels as ONE sequence (the state
 * is cut at every row start), so short rows (197-token encoders: 2 chunks of 128 steps, the second 44 % full) stop paying for
 * their padding and for a prologue per (batch element, channel tile).  Same results up to fp32 summation order.  Ask with
 * mxvl_scan_fold_ok(batch, seqlen, dstate) first; needs seqlen % 8 == 0, dstate 16, rows that qualify for 16-byte access, no
 * last_state; else MXVL_ERR_UNSUPPORTED.  ckpt of a folded call is (dim, mxvl_scan_fold_slots(...), dstate) fp32, indexed by
 * (channel, part, chunk of the part's folded sequence): the backward must be called with the same flag and buffer. */
#define MXVL_SCAN_FOLD_BATCH 4u

/*
 * Forward selective scan.
 *   delta' = softplus?(delta + delta_bias)
 *   h_t    = exp(delta'_t * A) * h_{t-1} + delta'_t * B_t * u_t        (h in R^{dim x dstate}, fp32)
 *   y_t    = <C_t, h_t> + D * u_t ;  out_t = y_t * silu(z_t)  (if z)
 * Mirrors SSMParamsBase (KSS/csrc/selective_scan/selective_scan.h:26-61): sizes, element strides,
 * raw pointers.  Optional pointers may be NULL: D, delta_bias, z, last_state, ckpt.
 *
 * ckpt: (batch, dim, n_chunks, dstate) fp32, state entering each chunk of mxvl_scan_chunk_len()
 * time steps -- the analogue of the reference's `x` buffer (selective_scan.cpp:217-219); it is
 * what mxvl_scan_bwd restarts its recomputation from.  last_state: (batch, dim, dstate) fp32 = h_L.
 */
typedef struct mxvl_scan_desc {
  int32_t batch, dim, seqlen, dstate, n_groups;
  int32_t io_dtype; /* mxvl_dtype */
  uint32_t flags;
  /* 0 or 1: delta is (batch, dim, seqlen) and delta_bias (dim).  k > 1 (k divides dim): delta is (batch, dim/k, seqlen) and
   * delta_bias (dim/k); channel d reads row d / k -- the vendored oflex kernels' dim_deltagroups_ratio
   * (cusoflex/selective_scan_oflex.cpp:59, selective_scan_fwd_kernel_oflex.cuh:91-107).  mxvl_scan_bwd still produces
   * ddelta (batch, dim, seqlen) and ddelta_bias (dim) PER CHANNEL; the caller sums each group of k (the reference reduces
   * them with atomics into the small tensors, selective_scan_bwd_kernel_oflex.cuh:104-121). */
  int32_t delta_group_ratio;
  /* element strides; the seqlen stride of every io tensor is 1 */
  int64_t u_bs, u_ds;
  int64_t delta_bs, delta_ds;
  int64_t z_bs, z_ds;
  int64_t out_bs, out_ds;
  int64_t B_bs, B_gs, B_ns;
  int64_t C_bs, C_gs, C_ns;
  int64_t A_ds, A_ns;
  const void *u, *delta, *A, *B, *C;
  const void *D, *delta_bias, *z; /* optional */
  void *out;
  void *last_state; /* optional */
  void *ckpt;       /* optional (required by bwd when n_chunks > 1) */
} mxvl_scan_desc;

/*
 * Backward selective scan (same math as selective_scan_bwd, selective_scan.cpp:241-349).
 * Gradients du, ddelta, dz (io dtype) are fully written.  dA (dim,dstate), dD (dim),
 * ddelta_bias (dim), dB, dC (batch,n_groups,dstate,seqlen) are fp32 and ACCUMULATED into:
 * the caller zero-fills them first (reference contract: selective_scan.cpp:321-327).
 * `fwd` carries the forward inputs (fwd.out may be NULL; fwd.ckpt must be the buffer
 * the forward call filled, or NULL when seqlen fits one chunk).
 */
typedef struct mxvl_scan_bwd_desc {
  mxvl_scan_desc fwd;
  int64_t dout_bs, dout_ds;
  int64_t du_bs, du_ds;
  int64_t ddelta_bs, ddelta_ds;
  int64_t dz_bs, dz_ds;
  int64_t dB_bs, dB_gs, dB_ns;
  int64_t dC_bs, dC_gs, dC_ns;
  const void *dout;
  void *du, *ddelta, *dz;   /* dz required iff fwd.z */
  void *dA, *dB, *dC;       /* fp32 accumulate */
  void *dD, *ddelta_bias;   /* fp32 accumulate; optional like their forward twins */
  /* Reserved (ABI v3 layout kept): ignored.  dB / dC are sums over all channels of a group; every workgroup pre-sums its
   * 16 / 32 rows in registers + LDS and adds ONE fp32 global atomic per (n, t), as the reference's kernel does per row
   * (selective_scan_bwd_kernel.cuh:215-221).  The per-tile scratch + reduce kernel of round 2 measured slower at every
   * shape (profiles/r02_bwd_workspace_ab.txt) and was removed: mxvl_scan_bwd_workspace_bytes() returns 0. */
  void *workspace;
  int64_t workspace_bytes;
} mxvl_scan_bwd_desc;

/* depthwise causal conv1d (+ optional SiLU): y[b,d,t] = act(bias[d] + sum_k w[d,k] * x[b,d,t-W+1+k]) */
typedef struct mxvl_conv1d_desc {
  int32_t batch, dim, seqlen, width;
  int32_t io_dtype; /* dtype of x / y (and dx, dy) */
  int32_t silu;     /* 1: SiLU activation, 0: identity */
  int64_t x_bs, x_ds;
  int64_t y_bs, y_ds;
  const void *x;
  const void *weight; /* (dim, width) fp32, row stride = width */
  const void *bias;   /* (dim) fp32, optional */
  void *y;
} mxvl_conv1d_desc;

typedef struct mxvl_conv1d_bwd_desc {
  mxvl_conv1d_desc fwd; /* fwd.y unused */
  int64_t dy_bs, dy_ds;
  int64_t dx_bs, dx_ds;
  const void *dy;
  void *dx;      /* io dtype, fully written */
  void *dweight; /* (dim,width) fp32, accumulated into */
  void *dbias;   /* (dim) fp32, accumulated into, optional */
} mxvl_conv1d_bwd_desc;

/*
 * Single-token decoder step of the report generator (16-bit weights and activations, fp32 accumulation), rows =
 * batch * beams <= 80.  ABI v8: every descriptor of the step carries `dtype` (MXVL_BF16 | MXVL_F16; 0 = bf16, what a v7 caller
 * leaves there): the reference loads its LLM with torch_dtype=torch.float16 (models/MambaXrayVL_DownStream.py:72,85,92), so every
 * kernel of the step exists for both element types (csrc/decode_elt.h) -- "bf16" in the field comments below reads "dtype".  Replaces per-token HF `LlamaForCausalLM.forward` + cuBLAS GEMVs
 * (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:292-301; layer arithmetic as restated in
 * EMRRG/models/hybrid_decoder_layer.py:185-199, 266-337, 392-457).
 *
 * mxvl_decode_gemv:  y[m][n] = epi( sum_k W[n][k] * xhat[m][k] ),  xhat = x, or RMSNorm(x; norm_weight, eps) when
 * norm_weight != NULL.  epi: + bias[n]; + residual[m][n]; swiglu = 1: y = silu(W xhat) * (W2 xhat); out_f32: fp32 y.
 * W, W2: (N, K) row-major bf16 (nn.Linear layout).
 * rows <= 8: one GEMV kernel, activations staged in LDS (rows * K * 2 bytes <= 150 KiB, K <= 8192 with norm_weight).
 * rows 9..80 (ABI v6): v_mfma_f32_16x16x32_bf16 with the weight tile as the A operand, loaded from HBM straight into the MFMA
 * registers; K >= 32; norm_weight must be NULL (mxvl_decode_rmsnorm first).  Same epilogues, same rounding points.
 */
/* ABI v8: rows 1..80 with k_splits == 1 and norm_weight != NULL, K % 64 == 0, K >= 256: the RMSNorm is fused into the matrix-core
 * projection -- y = rsqrt(mean(x^2) + eps)[m] * sum_k W[n][k] * dtype(norm_weight[k] * x[m][k]): the gain rides on the activation
 * fragments, the squares are summed beside the MFMAs, the row statistic scales the fp32 sums in the epilogue.  One 16-bit rounding
 * less than the modules' dtype(dtype(x * rstd) * g) (not bit-equal; within two ulp of the output type). */
typedef struct mxvl_gemv_desc {
  int32_t rows, K, N;
  int32_t swiglu, out_f32;
  float eps;
  const void *x;            /* (rows, K) bf16 */
  const void *norm_weight;  /* (K) bf16, optional */
  const void *W, *W2;       /* (N, K) bf16; W2 only with swiglu */
  const void *bias;         /* (N) bf16, optional */
  const void *residual;     /* (rows, N) bf16, optional */
  void *y;                  /* (rows, N) bf16, or fp32 when out_f32 */
  void *split_acc;          /* optional: (k_splits, rows, N) fp32 (ABI v8; v6-v7: one (rows, N) plane added to atomically).  K is split
                               over k_splits workgroups per column block; split s WRITES its partial sums to plane s (contents on
                               entry irrelevant, every element written); no epilogue runs (swiglu / bias / residual / out_f32 /
                               norm_weight must be unset, y is ignored): mxvl_decode_rmsnorm adds the planes in a fixed order
                               (acc, acc_splits) -- a deterministic sum.  For the projections with few columns (o_proj, down_proj:
                               N = hidden) that cannot fill the chip otherwise */
  int32_t k_splits;         /* 0: kernel by row count (rows <= 8: GEMV).  != 0: the matrix-core kernels at any row count; 1..16 with
                               split_acc, 1 without */
  int32_t dtype;            /* ABI v8: mxvl_dtype of x, norm_weight, W, W2, bias, residual, y (0 = MXVL_BF16) */
  float norm_gain_scale;    /* ABI v9 (fused RMSNorm only; 0 = 1): a power of two s.  The kernel rounds dtype(norm_weight[k] * s * x[m][k]) and
                               folds 1 / s into the row's rstd -- exact, so the result is the one of s = 1 wherever that did not leave the
                               16-bit range.  With s = 2^-ceil(log2 max|norm_weight|) the scaled gain is <= 1: the product cannot overflow
                               fp16 (|x| <= 65504 already), and a layer of uniformly small gains (Llama's first input_layernorm: 1e-2 ..
                               1e-3) no longer pushes g * x into fp16's subnormals.  The modules normalise in fp32 FIRST (x * rstd is O(1)),
                               so they never had either failure; bf16 has fp32's exponent range and does not need it. */
} mxvl_gemv_desc;

/*
 * mxvl_decode_attn: RoPE on the new q/k, append k/v at *pos to the cache, one-query attention per (row, head).
 * The cache is (rows, n_kv_heads, max_len, head_dim) bf16 indexed by PHYSICAL slot; slot_table[row][t] names the slot
 * holding position t of the hypothesis that `row` continues, so beam re-ordering never moves cache lines.
 */
typedef struct mxvl_decode_attn_desc {
  int32_t rows, n_heads, n_kv_heads, head_dim, max_len;
  float scale;
  const void *qkv;          /* (rows, (n_heads + 2*n_kv_heads) * head_dim) bf16: fused q|k|v projection */
  const void *cos, *sin;    /* (rows, head_dim) fp32 at the current position */
  void *k_cache, *v_cache;
  const void *slot_table;   /* (rows, max_len) int32 */
  const void *pos;          /* device int64 scalar: position being written */
  const void *mask;         /* (rows, max_len) int64, nonzero = may attend */
  void *out;                /* (rows, n_heads * head_dim) bf16 */
  void *q_rope;             /* ABI v3, optional: (rows, n_heads * head_dim) bf16, the rotated (unscaled) query -- the input of
                               mxvl_decode_cross_attn for hybrid layers conditioned on image tokens */
  int32_t beams;            /* ABI v6: 0 / 1 = a workgroup per (head, row).  2..5 (rows % beams == 0; rows b * beams .. are the beams of
                               sample b): a workgroup per (head, sample) -- cache positions on which the beams' slot-table entries agree
                               (the prompt, the common generated prefix) are read once for all of them; both products run on the
                               matrix cores and the probabilities are rounded to bf16 before the second one (softmax(...).to(bf16) @ V
                               of the modules); needs rows * n_kv_heads * max_len * head_dim < 2^31 (32-bit cache offsets) */
  int32_t dtype;            /* ABI v8: mxvl_dtype of qkv, the caches, out, q_rope (0 = MXVL_BF16) */
} mxvl_decode_attn_desc;

/*
 * mxvl_decode_cross_attn (ABI v3): the gated image cross-attention of a hybrid decoder layer for one new token per row --
 * `all2media_cross_attn`, EMRRG/models/hybrid_decoder_layer.py:653-697 (attention :25-77, gate construction :631-640):
 *   ctx = softmax(q_rope K_img^T * scale, masked by key_mask) V_img      per (row, head), grouped-query heads
 *   out = text_state + (row_on ? ctx : 0) * gate,   gate = [tanh](gate_weight . text_state + gate_bias) * [tanh](warm_up_gate)
 * with bf16 roundings where the reference's bf16 tensor ops round.  K_img / V_img are the layer's `cross_attn_kv_proj` of the
 * (input-normed) image tokens, constant over a generation; rows of one sample (beams) share them through kv_rows_div.
 */
#define MXVL_GATE_TANH 1        /* nn.Tanh after the gate projection ("...-tanh..." gating types) */
#define MXVL_GATE_WARM_TANH 2   /* warm_up_gate enters through tanh() (the text-only variant, :745) instead of raw (:690) */
typedef struct mxvl_decode_cross_attn_desc {
  int32_t rows, n_heads, n_kv_heads, head_dim, n_keys;
  int32_t kv_rows_div;      /* rows per image sample (num_beams); K/V, key_mask, row_on are indexed by row / kv_rows_div */
  int32_t gate_flags;       /* MXVL_GATE_* */
  float scale;
  const void *q_rope;       /* (rows, n_heads * head_dim) bf16, from mxvl_decode_attn */
  const void *k, *v;        /* (rows / kv_rows_div, n_kv_heads, n_keys, head_dim) bf16 */
  const void *key_mask;     /* optional (rows / kv_rows_div, n_keys) uint8, nonzero = may attend */
  const void *row_on;       /* optional (rows / kv_rows_div) uint8, 0 = sample without an image: context zeroed (:693) */
  const void *text_state;   /* (rows, n_heads * head_dim) bf16: the self-attention output (mxvl_decode_attn's out) */
  const void *gate_weight;  /* (n_heads * head_dim) bf16 */
  const void *gate_bias;    /* (1) bf16 */
  const void *warm_up_gate; /* optional (1) bf16 */
  void *out;                /* (rows, n_heads * head_dim) bf16, must not alias text_state */
  int32_t dtype;            /* ABI v8: mxvl_dtype of every 16-bit tensor above (0 = MXVL_BF16) */
  int32_t reserved0;
} mxvl_decode_cross_attn_desc;

/*
 * mxvl_decode_prologue (ABI v5): everything a generation step does before the decoder stack, as one launch -- what HF `generate`
 * spreads over its cache re-ordering, embedding lookup and rotary module per token (MambaXrayVL_DownStream.py:292-301):
 *   pos = *cur + prompt_len - 1;  slot_table[r][:] = slot_table[beam_src[r]][:];  slot_table[r][pos] = r;  mask[r][pos] = 1;
 *   x[r] = embed[tok[r]];  cos[r] / sin[r] = cos_table / sin_table[n_real[r] + *cur - 1];  *pos_out = pos
 * The tables are filled by the caller with its own rotary module for positions 0 .. table_len - 1 (so the values are the
 * module's own bits).  tok / beam_src / cur are the search state's device tensors (mxvl_beam_step updates them in place).
 */
typedef struct mxvl_decode_prologue_desc {
  int32_t rows, hidden, max_len, head_dim, prompt_len, table_len;
  const void *tok;          /* (rows) int64 */
  const void *beam_src;     /* (rows) int64: parent row of every row */
  const void *cur;          /* device int64 scalar: tokens generated so far (the step being fed is cur - 1) */
  const void *n_real;       /* (rows) int64: RoPE position of the first generated token */
  const void *embed;        /* (vocab, hidden) bf16 */
  const void *cos_table, *sin_table;   /* (table_len, head_dim) fp32 */
  void *slot_table;         /* (rows, max_len) int32, in place */
  void *mask;               /* (rows, max_len) int64, in place */
  void *x;                  /* (rows, hidden) bf16 out */
  void *cos, *sin;          /* (rows, head_dim) fp32 out */
  void *pos;                /* device int64 scalar out */
} mxvl_decode_prologue_desc;
/*
 * mxvl_decode_rmsnorm (ABI v6): y = bf16( bf16(x * rsqrt(mean(x^2) + eps)) * weight ), fp32 statistics -- Qwen2RMSNorm / LlamaRMSNorm
 * (EMRRG/models/hybrid_decoder_layer.py:185-199) of the (rows, K) activations ahead of a projection.  mxvl_decode_gemv fuses this
 * for rows <= 8 (norm_weight); for 9..80 rows -- the reference's own decode batches: val_batch_size 6 x beam 3 = 18
 * (launch/launch_mambaclip_chexpert.sh:23), 8 x 3 = 24 (launch_mambaclip_test_cheXpert.sh:26), 16 x 5 = 80
 * (launch_mambaclip_test_iu.sh:26-27) -- the projection runs on the matrix cores and takes already-normalised rows.
 */
typedef struct mxvl_rmsnorm_desc {
  int32_t rows, K;          /* K % 8 == 0, K <= 16384 */
  float eps;
  const void *x;            /* (rows, K) bf16; ignored in fold mode */
  const void *weight;       /* (K) bf16 */
  void *y;                  /* (rows, K) bf16, may alias x */
  void *acc;                /* optional, fold mode: (acc_splits, rows, K) fp32 partial sums of a split projection
                               (mxvl_gemv_desc.split_acc).  The row normalised is  x_out = bf16(sum_s acc[s]) + residual  (the modules'
                               `residual + linear(...)`); acc is left as it is */
  const void *residual;     /* (rows, K) bf16, with acc */
  void *x_out;              /* (rows, K) bf16, with acc; may alias residual */
  int32_t dtype;            /* ABI v8: mxvl_dtype of x, weight, y, residual, x_out (0 = MXVL_BF16) */
  int32_t acc_splits;       /* ABI v8: planes of acc (0 = 1) */
} mxvl_rmsnorm_desc;
/* Diagnostic / A-B switch of the 17..80-row projections (ABI v8): 1 (default; 2 = the same) = the waves of a workgroup split N and share
 * the activation tile through LDS (csrc/decode_gemm.h decode_gemm_wide_kernel) wherever the grid has >= 64 workgroups, 0 = the K-split
 * kernels of round 4 at every row count; 3 = wide as first measured (33..80 rows, >= 160 workgroups, 3-stage ring, four waves per
 * workgroup), 4 = wide with four waves per workgroup everywhere (the default takes three where that fills the 256 CUs), 5 = wide at 33..80
 * rows and >= 160 workgroups only, 6 = wide from one row on.  Returns the previous setting.  Same results in every mode up to the order
 * of the fp32 sums. */
int mxvl_set_decode_gemm_wide(int on);
/* Diagnostic (ABI v8): which kernel mxvl_decode_gemv would launch for a descriptor -- the dispatch as a pure function, nothing is launched
 * and no GPU is needed.  out[0] = 2 the per-row GEMV kernel (<= 8 rows, k_splits == 0), 1 = decode_gemm_wide_kernel, 0 = the K-split
 * matrix-core kernels; for 1: out[1] waves per workgroup, out[2] weight tiles per wave, out[3] LDS ring stages, out[4] workgroups.
 * Same argument checks and error codes as mxvl_decode_gemv. */
int mxvl_decode_gemm_plan(const mxvl_gemv_desc *desc, int32_t out[5]);
int mxvl_decode_rmsnorm(const mxvl_rmsnorm_desc *desc, void *hip_stream);
int mxvl_decode_prologue(const mxvl_decode_prologue_desc *desc, void *hip_stream);
int mxvl_decode_gemv(const mxvl_gemv_desc *desc, void *hip_stream);
int mxvl_decode_attn(const mxvl_decode_attn_desc *desc, void *hip_stream);
int mxvl_decode_cross_attn(const mxvl_decode_cross_attn_desc *desc, void *hip_stream);

int mxvl_abi_version(void);
int mxvl_scan_fold_ok(int batch, int seqlen, int dstate);   /* 1: MXVL_SCAN_FOLD_BATCH is supported and pays at this shape */
int mxvl_scan_fold_slots(int batch, int seqlen, int dim, int n_groups);   /* folded calls: ckpt is (dim, slots, dstate) fp32 */
/* time steps covered by one checkpoint chunk for a sequence of `seqlen` steps and `dstate` states */
int mxvl_scan_chunk_len(int seqlen, int dstate);
int mxvl_scan_n_chunks(int seqlen, int dstate);

int mxvl_scan_fwd(const mxvl_scan_desc *desc, void *hip_stream);
int mxvl_scan_bwd(const mxvl_scan_bwd_desc *desc, void *hip_stream);
/* always 0 since round 3 (kept so ABI-v3 callers link): no scratch is useful, see mxvl_scan_bwd_desc.workspace */
int64_t mxvl_scan_bwd_workspace_bytes(const mxvl_scan_desc *fwd);

int mxvl_conv1d_fwd(const mxvl_conv1d_desc *desc, void *hip_stream);
int mxvl_conv1d_bwd(const mxvl_conv1d_bwd_desc *desc, void *hip_stream);
/* single-token decode step: conv_state (batch,dim,width) io dtype is rolled in place; x,y: (batch,dim) */
int mxvl_conv1d_update(const void *x, void *conv_state, const void *weight, const void *bias, void *y,
                       int batch, int dim, int width, int io_dtype, int silu, void *hip_stream);
/* single-token SSM step (selective_state_update): state (batch,dim,dstate) fp32 updated in place */
int mxvl_state_update(void *state, const void *x, const void *dt, const void *A, const void *B,
                      const void *C, const void *D, const void *z, const void *dt_bias, void *out,
                      int batch, int dim, int dstate, int io_dtype, int dt_softplus, void *hip_stream);

/* Scan-order re-orderings of the 4- / 6-direction Mamba mixer (bimamba v3 / v4; CXPMRG_Bench_MambaXray_VL/arm/Finetuning/
 * mamba_simple.py:447-532: flips, the middle-cls transpose :476-482, its inverse and the direction sum :522-527).
 *   mxvl_dir_gather: rows (batch,dim,seqlen) -> stacked (batch,n_dirs,dim,padded_len): stacked[b,k,d,l] = rows[b,d,index[k][l]], 0 for
 *                    l >= seqlen;      mxvl_dir_merge: stacked -> rows: rows[b,d,t] = sum_k stacked[b,k,d,index[k][t]] (fp32 sum).
 * index: (n_dirs, seqlen) int32 permutations; gather with the permutations and merge with their inverses are each other's adjoint.
 * Strides in elements, innermost stride 1; seqlen <= 5120.
 * Output gate (gate != NULL; the reference's `y * silu(z)` of every direction and the `/ 4`, :522-529, which commute with the
 * re-ordering): merge writes rows = (sum_k ...) * silu(gate) * gate_scale and, when pre != NULL, the ungated sum to pre;
 * gather then is that merge's backward: rows holds d(out), stacked receives gather(rows * silu(gate) * gate_scale) and
 * dgate = rows * pre * gate_scale * silu'(gate) (pre and dgate required).  gate / pre / dgate are (batch, dim, seqlen). */
typedef struct mxvl_dir_perm_desc {
  int32_t batch, dim, seqlen, padded_len, n_dirs, io_dtype;
  int64_t rows_bs, rows_ds, stacked_bs, stacked_ks, stacked_ds;
  const void *index;
  void *rows, *stacked;      /* gather reads rows and writes stacked; merge reads stacked and writes rows */
  const void *gate;          /* optional */
  void *pre, *dgate;
  int64_t gate_bs, gate_ds, pre_bs, pre_ds, dgate_bs, dgate_ds;
  float gate_scale;
  int32_t reserved0;
} mxvl_dir_perm_desc;
int mxvl_dir_gather(const mxvl_dir_perm_desc *desc, void *hip_stream);
int mxvl_dir_merge(const mxvl_dir_perm_desc *desc, void *hip_stream);

/* One beam-search update of report generation (what HF `generate(num_beams>1)` does between two decoder steps; call site
 * CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:292-301): log-softmax, repetition penalty, min-new-tokens,
 * top-`keep` over beams*vocab, live-beam / finished-pool bookkeeping, early-stop heuristic.  All state tensors are updated
 * in place; *cur is incremented; *unfinished = decoding continues.  beams <= 8, keep <= 16 (beam 5 of launch_mambaclip_test_iu.sh:27).  Dtypes: logits, scores, tables
 * fp32; sequences, cur, eos, tok, beam_src int64; fin_done, heur_open, unfinished 1-byte booleans. */
typedef struct mxvl_beam_desc {
  int32_t batch, beams, vocab, max_new, min_new, n_eos, early_stopping, keep; /* early_stopping: 1 = True, 0 = False/"never" */
  float repetition_penalty;
  int32_t reserved0;
  const void *logits;              /* (batch*beams, vocab) */
  void *run_seq, *fin_seq;         /* (batch, beams, max_new) */
  void *run_score, *fin_score;     /* (batch, beams) */
  void *fin_done, *heur_open;      /* (batch, beams), (batch) */
  void *cur;                       /* scalar */
  const void *eos;                 /* (n_eos) */
  const void *len_tab, *hyp_tab;   /* (max_new): (t+1)^length_penalty, hypothesis-length^length_penalty */
  void *tok, *beam_src;            /* (batch*beams): next token, parent row of every live beam */
  void *unfinished;                /* scalar */
  void *scratch;                   /* ABI v6, optional: (1) uint32, zero before the first call (the kernel leaves it zero): when
                                      given, one workgroup per batch element instead of one workgroup walking the batch */
  void *unfinished_log;            /* ABI v7, optional: (max_new) bytes the HOST can read (pinned, device-mapped): the step that ran at
                                      *cur == c also stores its *unfinished at [c], so a token loop that runs ahead of the host
                                      (report_decoder._search_lookahead) needs no device-to-host copy between the steps */
  void *workspace;                 /* ABI v7, optional: mxvl_beam_workspace_bytes(batch, beams, keep) bytes of device memory (contents
                                      irrelevant): when given, the two sweeps over the (beams x vocab) logits of a sample are split over
                                      slices of the vocabulary (two extra launches in front of the one-workgroup-per-sample kernel) */
  int64_t workspace_bytes;
} mxvl_beam_desc;
int mxvl_beam_step(const mxvl_beam_desc *desc, void *hip_stream);
int64_t mxvl_beam_workspace_bytes(int batch, int beams, int keep);

/* Residual add + LayerNorm of an ARM / VisionMamba block (CXPMRG_Bench_MambaXray_VL/arm/Finetuning/models_mamba.py:
 * 110-116 `x + mixer(norm1(x))`, `x + mlp(norm2(x))`; the reference's fused_add_norm path pairs them the same way):
 *   h = x + branch (branch may be NULL: h is x and is not written);  n = (h - mean) * rstd * gamma + beta.
 * res_dtype is the dtype of x, h, dx, dh; branch_dtype of branch / dbranch; out_dtype of n / dn (MXVL_F32 | MXVL_BF16,
 * combinations f32/f32/f32, f32/bf16/bf16, f32/f32/bf16, bf16/bf16/bf16, and for fp16 autocast f32/f16/f16, f32/f32/f16);
 * cols % 256 == 0, cols <= 2048.
 * Backward: dx = dh + LN'(dn) (dh may be NULL), optionally also written in the branch dtype to dbranch; dgamma/dbeta
 * leave as n_partials = mxvl_add_layernorm_partials(rows) partial rows the caller sums. */
typedef struct mxvl_add_ln_desc {
  int32_t rows, cols, res_dtype, branch_dtype, out_dtype;
  float eps;
  const void *x, *branch, *gamma, *beta; /* gamma, beta: (cols) fp32; beta optional */
  void *h, *n, *mean, *rstd;             /* mean, rstd: (rows) fp32 */
} mxvl_add_ln_desc;
typedef struct mxvl_add_ln_bwd_desc {
  int32_t rows, cols, res_dtype, branch_dtype, out_dtype, n_partials;
  const void *dn, *dh, *h, *gamma, *mean, *rstd;
  void *dx, *dbranch, *partial_dgamma, *partial_dbeta; /* partials: (n_partials, cols) fp32 */
  void *partial_dbranch;    /* ABI v6, optional (n_partials, cols) fp32: column sums of the branch gradient AS WRITTEN (rounded to the
                               branch dtype when dbranch is given, else dx) = the bias gradient of the linear layer whose output was
                               the branch (models_mamba.py:110-116: mixer out_proj / SwiGLU w3) */
} mxvl_add_ln_bwd_desc;
int mxvl_add_layernorm_fwd(const mxvl_add_ln_desc *desc, void *hip_stream);
int mxvl_add_layernorm_bwd(const mxvl_add_ln_bwd_desc *desc, void *hip_stream);
int mxvl_add_layernorm_partials(int rows);
/*
 * ViT-MAE index / loss glue (ABI v4), HD_Xray_Pretrain_MAE/pretrain/models/mae.py:
 * mxvl_row_gather:  out[n, r, :] = (idx[n, r] >= 0 ? src[n, idx[n, r], :] : fill[:]) + add[r, :]   (fill / add fp32, optional:
 *   NULL fill = zero row, NULL add = nothing added).  One kernel for `torch.gather(x, 1, ids_keep...)` of random_masking[_yiliao]
 *   (:157-253), for its backward, for forward_decoder's cat(mask tokens) -> gather(ids_restore) -> cat(cls) -> + decoder_pos_embed
 *   (:280-305: idx = 1 + ids_restore or -1, row 0 = cls) and for that one's backward.  src (N, rows_src, D) / out (N, rows_out, D)
 *   with contiguous rows and the given batch strides (elements); src_dtype == out_dtype without add is a bit copy, otherwise the
 *   value passes through fp32 (torch's type promotion / autograd's cast back).  idx int32 (N, rows_out).
 * mxvl_patch_loss:  loss[n, l] = mean_j (pred[n, l, j] - target[n, l, j])^2, target = patchify(img) (:129-141), normalised per patch
 *   with the unbiased variance and eps 1e-6 when norm_pix (:307-323); img (N, C, HW, HW) fp32 contiguous, pred (N, L, patch^2 C)
 *   contiguous io dtype.  Forward: loss != NULL (dpred NULL).  Backward: dpred != NULL, dloss (N, L) fp32 given:
 *   dpred = dloss * 2 (pred - target) / (patch^2 C).
 */
int mxvl_row_gather(const void *src, const int32_t *idx, const float *fill, const float *add, void *out, int batch, int rows_src,
                    int rows_out, int dim, int64_t src_bs, int64_t out_bs, int src_dtype, int out_dtype, void *hip_stream);
int mxvl_patch_loss(const void *img, const void *pred, const void *dloss, void *loss, void *dpred, int batch, int channels, int hw,
                    int patch, int norm_pix, int io_dtype, void *hip_stream);
/* mxvl_patch_cols (ABI v7): the im2col of a convolution whose kernel equals its stride -- the first projection of a patch embedding
 * (HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:21-43 SmallPatchEmbed conv1 16x16 / s16; finetune/DP/models/vit.py:186-208) as a GEMM:
 *   cols[n, i * gw + j, (c * patch + di) * patch + dj] = img[n, c, i * patch + di, j * patch + dj]
 * img (N, C, H, W) contiguous; cols (N, (H / patch) * (W / patch), C * patch^2) contiguous in out_dtype (== in_dtype, or a 16-bit
 * dtype for an fp32 image: the autocast cast of the GEMM input).  patch in {4, 8, 16, 32, 64}, 16-byte aligned pointers. */
int mxvl_patch_cols(const void *img, void *cols, int batch, int channels, int h, int w, int patch, int in_dtype, int out_dtype,
                    void *hip_stream);
/* ABI v11: window rows of a CHANNELS-LAST feature map with the activation in front of them fused -- the `F.relu(conv1(x))` -> `conv2`
 * hand-off of SmallPatchEmbed (HD_Xray_Pretrain_MAE/pretrain/patch_embed.py:21-41: kernel == stride convolutions as GEMMs):
 *   forward  (backward = 0): cols[n][i * gw + j][(di * k + dj) * C + c] = act(x[n][i * k + di][j * k + dj][c]),  x (N, H, W, C), act = ReLU iff relu
 *   backward (backward = 1): out = dx (N, H, W, C) = dcols gathered back, zeroed where the PRE-activation x <= 0 (x may be NULL without relu)
 * one io dtype, C % (16 bytes) == 0, 16-byte aligned buffers.  One pass each way instead of ReLU + strided copy (+ their two backward
 * passes) over a (256, 80, 80, 1024) map. */
int mxvl_window_cols(const void *x, const void *dcols, void *out, int N, int H, int W, int C, int k, int relu, int backward, int io_dtype,
                     void *hip_stream);

/*
 * mxvl_gemm_swiglu_fwd (ABI v4): the SwiGLU input projection of the block MLP as ONE MFMA GEMM with the gate in its epilogue --
 *   ab = x [w1; w2]^T + [b1 | b2],   h = silu(ab[:, :H]) * ab[:, H:]
 * replaces `self.w1(x)`, `self.w2(x)`, `self.act(x1) * x2` (CXPMRG_Bench_MambaXray_VL/arm/Finetuning/models_mamba.py:59-83;
 * pretrain/models_pretrain.py twin).  x (M, K), weight (2H, K) = nn.Linear layout of the merged [w1; w2], bias (2H) fp32 or the
 * io dtype or NULL; h (M, H) always; ab (M, 2H) only when the caller needs the pre-activations (training: the backward reads
 * them), NULL otherwise.  bf16 / fp16, fp32 accumulation; h is gated from the fp32 accumulators, ab is their io-dtype rounding.
 * K % 64 == 0; x / weight rows 16-byte aligned (strides % 8 == 0); any M, any H (ragged tiles are masked).
 */
typedef struct mxvl_gemm_swiglu_desc {
  int32_t M, K, H;
  int32_t io_dtype, bias_dtype;
  int64_t x_rs, w_rs, ab_rs, h_rs;       /* row strides in elements */
  const void *x, *weight, *bias;
  void *ab, *h;
} mxvl_gemm_swiglu_desc;
int mxvl_gemm_swiglu_fwd(const mxvl_gemm_swiglu_desc *desc, void *hip_stream);
/* ABI v8: the backward of the same MLP half fused into the dgrad GEMM of its output projection -- `self.w3(self.act(x1) * x2)` of the
 * reference's SwiGLU (models_mamba.py:59-83), differentiated: with dy (M, K = out_features) the gradient at w3's output and
 * w3t (H, K) = w3.weight^T (row-major, the hidden axis zero-padded like ab),
 *     d_h = dy w3t^T ;  dab[:, :H] = d_h * b * silu'(a) ;  dab[:, H:] = d_h * silu(a)      (a | b = ab, the saved pre-activations)
 * in ONE MFMA kernel: d_h lives only in the accumulators (rounded to the io dtype before use, as the tensor it replaces), ab is read
 * and dab written in the GEMM's epilogue, and partial (mxvl_gemm_swiglu_bwd_partials(M), 2H) fp32 receives the column sums of the
 * rounded dab over 128-token slabs (sum over its rows = the bias gradient of [w1; w2]; every element < 2H is written).
 * K % 64 == 0, H % 8 == 0, 16-byte aligned dy / w3t rows, bf16 / fp16. */
typedef struct mxvl_gemm_swiglu_bwd_desc {
  int32_t M, K, H;
  int32_t io_dtype;
  int64_t dy_rs, w_rs, ab_rs, dab_rs;  /* row strides in elements */
  const void *dy, *w3t, *ab;
  void *dab;
  void *partial;                       /* optional */
} mxvl_gemm_swiglu_bwd_desc;
int mxvl_gemm_swiglu_bwd(const mxvl_gemm_swiglu_bwd_desc *desc, void *hip_stream);
/* ABI v8: the same MFMA kernel as a plain token-major GEMM, c (M, N) = a (M, K) b^T (+ bias), b (N, K) row-major = an nn.Linear weight
 * (forward: CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:388-402 in_proj / out_proj, models_mamba.py:59-83 w3) or the
 * transposed weight (the dgrad products).  K % 64 == 0, N % 8 == 0, 16-byte aligned a / b rows, bf16 / fp16, fp32 accumulation. */
typedef struct mxvl_gemm_nt_desc {
  int32_t M, K, N;
  int32_t io_dtype, bias_dtype;
  int32_t reserved0;
  int64_t a_rs, b_rs, c_rs;            /* row strides in elements */
  const void *a, *b, *bias;            /* bias (N) fp32 or io dtype, optional */
  void *c;
} mxvl_gemm_nt_desc;
int mxvl_gemm_nt(const mxvl_gemm_nt_desc *desc, void *hip_stream);
/* ABI v12: the weight-gradient GEMM, c (M, N) fp32 (+)= a^T b for TOKEN-MAJOR operands a (K, M), b (K, N) -- autograd's
 * `grad_weight = dy^T x` of every nn.Linear of the blocks (CXPMRG_Bench_MambaXray_VL/arm/Finetuning/mamba_simple.py:76,91 in_proj /
 * out_proj, models_mamba.py:59-83 w1 | w2 | w3; K = the step's tokens).  One MFMA kernel (csrc/gemm_tn.hip): operands through
 * LDS-DMA and the gfx950 transpose read, the token axis split over the XCDs, partial tiles added into c by fp32 atomics (so the
 * low bits of c depend on the arrival order, like the dB / dC of mxvl_scan_bwd).  accumulate == 0: c is zeroed on the stream first.
 * K % 64 == 0, K >= 512, M % 8 == 0, N % 8 == 0, 16-byte aligned a / b rows, bf16 / fp16 operands, fp32 accumulation and output.
 * slices_per_xcd: 0 = chosen from the shape (1..4 token slices per XCD), q / -q = forced with / without the split of the last round
 * (measurement). */
typedef struct mxvl_gemm_tn_desc {
  int32_t M, N, K;
  int32_t io_dtype;
  int32_t accumulate;
  int32_t slices_per_xcd;
  int64_t a_rs, b_rs, c_rs;            /* row strides in elements */
  const void *a, *b;
  void *c;
} mxvl_gemm_tn_desc;
int mxvl_gemm_tn(const mxvl_gemm_tn_desc *desc, void *hip_stream);
/* ABI v12: the bias gradient beside it, autograd's `grad_bias = dy.sum(0)` over the token axis of a token-major (rows, cols) bf16 / fp16
 * tensor: partial (mxvl_colsum_partials(rows, cols), cols) fp32 receives one partial sum per row group (every element written); the
 * caller adds the rows.  cols % 8 == 0, 16-byte aligned rows (row_stride in elements). */
int mxvl_colsum_partials(int rows, int cols);
int mxvl_colsum(const void *x, void *partial, int rows, int cols, int64_t row_stride, int n_partials, int io_dtype, void *hip_stream);
int mxvl_gemm_swiglu_bwd_partials(int M);
/* SwiGLU gate of the block MLP (models_mamba.py:59-83 `act(w1 x) * w2 x`): ab (rows, 2*hidden) = [w1 x | w2 x] from ONE
 * GEMM -> y (rows, hidden) = silu(a) * b; backward writes dab (rows, 2*hidden).  Contiguous, one io dtype. */
int mxvl_swiglu_fwd(const void *ab, void *y, int rows, int hidden, int io_dtype, void *hip_stream);
int mxvl_swiglu_bwd(const void *ab, const void *dy, void *dab, int rows, int hidden, int io_dtype, void *hip_stream);
/* same backward, also leaving the column sums of dab -- the bias gradient of the w1|w2 GEMM -- as n_partials =
 * mxvl_swiglu_partials(rows, hidden) fp32 rows of `partial` (n_partials, 2*hidden) that the caller adds up; hidden must be even
 * (mxvl_swiglu_partials returns 0 otherwise). */
int mxvl_swiglu_partials(int rows, int hidden);
int mxvl_swiglu_bwd_colsum(const void *ab, const void *dy, void *dab, void *partial, int n_partials, int rows, int hidden,
                           int io_dtype, void *hip_stream);

/* ---- ABI v10: the element-wise work of a decoder layer in the TRAINING step of the report-generation stages (csrc/llm_ops.hip) ----------
 * The stage-3 / R2GenCSR step runs a frozen fp16-loaded LLM under bf16 autocast, forward and activation-gradient backward
 * (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:195-241, R2GenCSR/models/R2GenCSR.py:309-474).
 *
 * mxvl_rope: rotary position embedding of q and k in one launch -- apply_rotary_pos_emb / rotate_half
 * (EMRRG/models/hybrid_decoder_layer.py:290-323; HF's Llama classes carry the same text) followed by the cast back to the projections'
 * dtype:  y1 = x1 c1 - x2 s1,  y2 = x2 c2 + x1 s2  over the halves (x1 | x2) of every head, every product and the sum rounded to
 * promote(io_dtype, cs_dtype) (fp32 unless both are the same 16-bit type), the result to io_dtype: bit-identical to the torch
 * expression.  backward != 0: q / k carry the gradients of the outputs and q_out / k_out receive the gradients of the inputs, with
 * autograd's rounding points (csrc/llm_ops.hip header).  Any batch / token / head strides with a contiguous head_dim; head_dim % 16 == 0
 * (16-bit) / % 8 == 0 (fp32); strides multiples of 8 / 4 elements; 16-byte aligned bases.  cos / sin: (batch, seqlen, head_dim) rows,
 * cs_bs = 0 shares one table over the batch. */
typedef struct mxvl_rope_desc {
  int32_t batch, seqlen, n_q_heads, n_k_heads, head_dim;
  int32_t io_dtype, cs_dtype, backward;
  int64_t q_bs, q_ts, q_hs, k_bs, k_ts, k_hs;           /* element strides of q / k */
  int64_t qo_bs, qo_ts, qo_hs, ko_bs, ko_ts, ko_hs;     /* of q_out / k_out */
  int64_t cs_bs, cs_ts;                                  /* of cos / sin */
  const void *q, *k, *cos, *sin;
  void *q_out, *k_out;
} mxvl_rope_desc;
int mxvl_rope(const mxvl_rope_desc *desc, void *hip_stream);
/* mxvl_rmsnorm_train_fwd / _bwd: Qwen2RMSNorm / LlamaRMSNorm (EMRRG/models/hybrid_decoder_layer.py:185-199) over (rows, cols)
 * activations, forward and INPUT gradient (no weight gradient: a trainable norm weight keeps the torch expression).
 *   fwd: y = y_dtype( P( weight * x_dtype( x * rsqrt(mean(x^2) + eps) ) ) ),  P = promote(w_dtype, x_dtype), fp32 statistics; rstd (rows)
 *        fp32 is written for the backward (optional).
 *   bwd: y receives dx (x_dtype) = rstd * (gh - xhat * mean(gh * xhat)),  gh = x_dtype(P(grad * weight)),  xhat = x * rstd;  grad has
 *        y_dtype.
 * cols % 8 == 0, contiguous rows, 16-byte aligned bases. */
typedef struct mxvl_rms_train_desc {
  int32_t rows, cols;
  int32_t x_dtype, w_dtype, y_dtype;
  float eps;
  const void *x, *weight;
  const void *grad;          /* bwd only: (rows, cols) y_dtype */
  void *y;                   /* fwd: (rows, cols) y_dtype;  bwd: dx (rows, cols) x_dtype */
  void *rstd;                /* (rows) fp32: written by fwd (optional), read by bwd */
} mxvl_rms_train_desc;
int mxvl_rmsnorm_train_fwd(const mxvl_rms_train_desc *desc, void *hip_stream);
int mxvl_rmsnorm_train_bwd(const mxvl_rms_train_desc *desc, void *hip_stream);
/* mxvl_silu_mul: the gate of Qwen2MLP / LlamaMLP, `act_fn(gate_proj(x)) * up_proj(x)` (EMRRG/models/hybrid_decoder_layer.py:326-338), over
 * n contiguous elements of one dtype, with the two roundings of the two torch kernels:  y = io(io(silu(a)) * b).  With dy != NULL it is the
 * backward of that expression as autograd runs it: y receives da = io(ds * sig(a) (1 + a (1 - sig(a)))), ds = io(dy * b), and db receives
 * io(dy * io(silu(a))).  n % 8 == 0, 16-byte aligned bases. */
int mxvl_silu_mul(const void *a, const void *b, const void *dy, void *y, void *db, int64_t n, int io_dtype, void *hip_stream);

/* VMamba SS2D 4-direction orderings (R2GenCSR/VMamba/classification/models/vmamba.py:25-67, CrossScan / CrossMerge).
 * mxvl_cross_scan : x (batch,channels,height,width) -> xs (batch,4,channels,height*width): row-major, column-major
 *                   and their reversals.   mxvl_cross_merge: ys (batch,4,channels,L) -> y (batch,channels,L) =
 *                   (ys0 + flip(ys2)) + transpose(ys1 + flip(ys3)), each add rounded to the io dtype (bit-exact with
 *                   the reference's tensor adds).  Each is the other's backward.  Contiguous tensors, one io dtype. */
int mxvl_cross_scan(const void *x, void *xs, int batch, int channels, int height, int width, int io_dtype, void *hip_stream);
int mxvl_cross_merge(const void *ys, void *y, int batch, int channels, int height, int width, int io_dtype, void *hip_stream);

/* Depthwise 3x3 convolution (padding 1) + optional SiLU of VMamba's SS2D block: `self.act(self.conv2d(x))`,
 * nn.Conv2d(d_inner, d_inner, 3, padding=1, groups=d_inner) (R2GenCSR/VMamba/classification/models/vmamba.py:746-755, 1121-1123).
 * x, y, dy, dx: (batch, channels, height, width) contiguous in the io dtype; weight (channels, 9) fp32, bias (channels) fp32 or
 * NULL; dweight / dbias fp32, ACCUMULATED into (caller zero-fills).  ksize must be 3, height*width <= 4096. */
int mxvl_dwconv2d_fwd(const void *x, const void *weight, const void *bias, void *y, int batch, int channels, int height,
                      int width, int ksize, int io_dtype, int silu, void *hip_stream);
int mxvl_dwconv2d_bwd(const void *x, const void *weight, const void *bias, const void *dy, void *dx, void *dweight,
                      void *dbias, int batch, int channels, int height, int width, int ksize, int io_dtype, int silu,
                      void *hip_stream);

/* Image pre-processing of the report-generation data pipeline: `AutoImageProcessor(...)(img, return_tensors="pt",
 * size=input_size).pixel_values[0]` (CXPMRG_Bench_MambaXray_VL/dataset/data_helper.py:17-26, :70-76) = Pillow `Image.resize`
 * (libImaging/Resample.c, 8-bit two-pass fixed-point convolution) + transformers rescale/normalize.  Bit-exact integer resize;
 * the float value of each byte comes from the caller's table, so the output equals the CPU pipeline bit for bit.
 *   mxvl_resample_ksize / mxvl_resample_coeffs: HOST-ONLY (no GPU): Resample.c precompute_coeffs + normalize_coeffs_8bpc for the
 *     whole-image box.  bounds: (out_size, 2) int32 {first source index, tap count}; kk: (ksize, out_size) int32, tap-major,
 *     22-bit fixed point, zero past the tap count.  in_size == out_size gives the identity (ksize 1), i.e. a skipped pass.
 *   mxvl_image_preprocess: src (in_h, in_w, 3) uint8 -> out (3, out_h, out_w) in out_dtype; tmp is a caller-provided
 *     (in_h, out_w, 3) uint8 workspace (Pillow's intermediate image); lut (3, 256) fp32 = value of byte v in channel c after
 *     rescale + normalise.  All pointers are device pointers; in_w <= 20000. */
typedef enum mxvl_resample { MXVL_RESAMPLE_BILINEAR = 2, MXVL_RESAMPLE_BICUBIC = 3 } mxvl_resample; /* PIL.Image.Resampling values */
typedef struct mxvl_image_desc {
  int32_t in_h, in_w, out_h, out_w;
  int32_t ksize_h, ksize_v;
  int32_t out_dtype; /* mxvl_dtype */
  int32_t reserved0;
  const void *src;
  const void *bounds_h, *kk_h; /* horizontal pass: in_w -> out_w */
  const void *bounds_v, *kk_v; /* vertical pass:   in_h -> out_h */
  const void *lut;
  void *tmp;
  void *out;
} mxvl_image_desc;
int mxvl_resample_ksize(int in_size, int out_size, int filter);
int mxvl_resample_coeffs(int in_size, int out_size, int filter, int32_t *bounds, int32_t *kk);
int mxvl_image_preprocess(const mxvl_image_desc *desc, void *hip_stream);

/* Fused multi-head attention, forward and backward, on the matrix cores (csrc/attn.hip): out = softmax(q k^T * scale + mask) v
 * without materialising the (seqlen_q x seqlen_k) scores.  Replaces the score-matrix attention of
 *   CXPMRG_Bench_MambaXray_VL/pretrain/models_pretrain.py:55-83 (CrossAttention + the block-causal mask of :395-400),
 *   HD_Xray_Pretrain_MAE/finetune/DP/models/vit.py:141-163 (ViT / MAE blocks),
 *   EMRRG/models/hybrid_decoder_layer.py:25-77 (text -> image cross-attention, boolean key mask) and :392-457 (causal GQA).
 * q (batch, n_heads, seqlen_q, head_dim), k / v (batch, n_kv_heads, seqlen_k, head_dim), out like q: element strides for
 * batch / head / token, head_dim contiguous, every row 16-byte aligned (so any (B, L, H, D) / (B, H, L, D) view works without
 * a copy).  head_dim 32 / 64 (forward and backward) or 128 (forward only); io dtype fp32 (exact fp32 MFMA) / bf16 / fp16.
 * mask_mode: 0 none; 1 causal: key j visible to query i iff j <= i + seqlen_k - seqlen_q; 2 block-causal: iff
 * j / cluster <= i / cluster (mask_generate's block-lower-triangular mask with 16-token clusters).  key_mask: optional
 * (batch, seqlen_k) bytes, nonzero = may be attended.  bias: optional additive fp32 (seqlen_q, seqlen_k), broadcast over
 * batch and heads (generic slow path for masks that are none of the above).
 * lse: (batch, n_heads, seqlen_q) fp32 log2-domain log-sum-exp of the scaled logits (+inf for a row with no visible key, whose
 * output is 0); optional for inference, required for mxvl_attn_bwd, which also needs `delta` scratch of the same shape. */
typedef struct mxvl_attn_desc {
  int32_t batch, n_heads, n_kv_heads, seqlen_q, seqlen_k, head_dim;
  int32_t io_dtype;  /* mxvl_dtype */
  int32_t mask_mode, cluster;
  float scale;
  int64_t q_bs, q_hs, q_ts;
  int64_t k_bs, k_hs, k_ts;
  int64_t v_bs, v_hs, v_ts;
  int64_t o_bs, o_hs, o_ts;
  const void *q, *k, *v;
  void *out;
  void *lse;
  const void *key_mask;
  const void *bias;
  /* ABI v11: attention dropout (training): probability (query i, key j) of head (b, h) is zeroed with probability dropout_p and
   * the kept ones scaled by 1 / (1 - dropout_p), AFTER the softmax normalisation, as nn.Dropout on the attention matrix does
   * (models_pretrain.py:62,80 attn_drop; hybrid_decoder_layer.py attention_dropout; Blip2 Q-Former attention_probs_dropout_prob).
   * The keep bit is a pure function of (dropout_seed, b * n_heads + h, i, j) -- csrc/attn.hip attn_drop_hash, restated on the host by
   * flash_attention.dropout_keep_mask -- so mxvl_attn_bwd reproduces the forward's mask from the same two fields.  0 = no dropout
   * (and the only value the 64-queries-per-wave / LDS-DMA kernels serve: a call with dropout runs on the general kernels). */
  float dropout_p;
  uint32_t dropout_seed;
} mxvl_attn_desc;
typedef struct mxvl_attn_bwd_desc {
  mxvl_attn_desc fwd;   /* fwd.out and fwd.lse as the forward call left them */
  int64_t dout_bs, dout_hs, dout_ts;
  int64_t dq_bs, dq_hs, dq_ts;
  int64_t dk_bs, dk_hs, dk_ts;
  int64_t dv_bs, dv_hs, dv_ts;
  const void *dout;
  void *dq, *dk, *dv;   /* io dtype, fully written; dk / dv are (batch, n_kv_heads, seqlen_k, head_dim): summed over the group */
  void *delta;          /* (batch, n_heads, seqlen_q) fp32 scratch */
} mxvl_attn_bwd_desc;
int mxvl_attn_fwd(const mxvl_attn_desc *desc, void *hip_stream);
int mxvl_attn_bwd(const mxvl_attn_bwd_desc *desc, void *hip_stream);

/* Stage-2 contrastive step (CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py:133-148) as one kernel: L2-normalise both
 * feature sets, logits = exp(logit_scale) * img_n @ txt_n^T, loss = (CE(logits, arange) + CE(logits^T, arange)) / 2, plus the
 * gradients for d(loss) = 1.  Features fp32 (batch, dim) contiguous; logit_scale (device
 * scalar) is the parameter, i.e. the log of the multiplier; loss and d_logit_scale are device scalars.  batch <= 89. */
int mxvl_clip_loss(const float *image_features, const float *text_features, const float *logit_scale, int batch, int dim,
                   float *loss, float *d_image, float *d_text, float *d_logit_scale, void *hip_stream);

/*
 * ABI v11: mamba_inner_fn / mamba_inner_fn_no_out_proj as ONE native entry (SURVEY.md section 8-b `mxvl_mamba_inner_fwd/bwd`;
 * the reference calls the patched mamba_ssm's fused block at CXPMRG_Bench_MambaXray_VL/pretrain/mamba_simple.py:388-402 and
 * arm/Finetuning/mamba_simple.py:450-511, 650-664; its semantics are the in-repo slow path mamba_simple.py:665-709):
 *     x, z = xz[:, :dim], xz[:, dim:]                          xz (batch, 2 dim, seqlen), seqlen contiguous
 *     xc = silu(conv1d(x))                                     mxvl_conv1d_fwd
 *     x_dbl = x_proj_weight @ xc                               per batch element (dt_rank + 2 dstate, seqlen): rows dt | B | C
 *     delta = dt_proj_weight @ x_dbl[:dt_rank]                 (dim, seqlen)
 *     y = selective_scan(xc, delta, A, B, C, D, z, delta_bias, softplus)      mxvl_scan_fwd
 *     out = y                       (batch, dim, seqlen)       when out_proj_weight == NULL   (mamba_inner_fn_no_out_proj)
 *     out = y^T out_proj_weight^T (+ out_proj_bias)  (batch, seqlen, d_model)  otherwise       (mamba_inner_fn)
 * The conv and the scan are this library's kernels; the four dense products are plain GEMMs on the activations' own channel-major
 * layout (no transposes: B and C are ROWS of x_dbl, exactly the (batch, dstate, seqlen) arrays the scan reads) and go to rocBLAS
 * (rocblas_gemm_strided_batched_ex, fp32 accumulation; the library is opened with dlopen at the first call -- inside a torch
 * process that is the copy torch already mapped -- so libmxvl.so itself has no link-time dependency; MXVL_ERR_UNSUPPORTED when it
 * cannot be opened).  Weights of the projections are passed in the io dtype; conv weight / bias, A, D, delta_bias in fp32.
 * Every intermediate lives in the caller's workspace (mxvl_mamba_inner_workspace_bytes): the forward leaves xc, x_dbl, delta, y and
 * the scan checkpoints there and the backward reads them back, so the workspace of a forward call must reach the backward call
 * untouched.  The one allocation this entry ever causes is rocBLAS's own handle (once per process).
 * Backward: dxz (io dtype) fully written; every parameter gradient is fp32 and ACCUMULATED into (caller zero-fills: the contract of
 * mxvl_scan_bwd / mxvl_conv1d_bwd); needs its own scratch of mxvl_mamba_inner_bwd_workspace_bytes.
 * The Python mirror's autograd node (selective_scan_interface._MambaInnerFn) keeps composing the same kernels with torch's GEMMs
 * (hipBLASLt, tuned per shape, split-K weight gradients); `mamba_inner_fn_native` is this entry behind the same signature.
 */
typedef struct mxvl_mamba_inner_desc {
  int32_t batch, dim, seqlen, dstate, dt_rank, width;   /* dim = d_inner, width = d_conv */
  int32_t d_model;                                      /* out features of out_proj (ignored without out_proj_weight) */
  int32_t io_dtype;                                     /* xz, the projection weights, out */
  uint32_t flags;                                       /* MXVL_SCAN_DELTA_SOFTPLUS */
  int32_t reserved0;
  const void *xz;                                       /* (batch, 2 dim, seqlen) contiguous */
  const void *conv_weight, *conv_bias;                  /* (dim, width) fp32; (dim) fp32, optional */
  const void *x_proj_weight;                            /* (dt_rank + 2 dstate, dim) io dtype, row-major */
  const void *dt_proj_weight;                           /* (dim, dt_rank) io dtype */
  const void *out_proj_weight, *out_proj_bias;          /* (d_model, dim), (d_model) io dtype; optional */
  const void *A, *D, *delta_bias;                       /* (dim, dstate), (dim), (dim) fp32; D / delta_bias optional */
  void *out;
  void *workspace;
  int64_t workspace_bytes;
} mxvl_mamba_inner_desc;
typedef struct mxvl_mamba_inner_bwd_desc {
  mxvl_mamba_inner_desc fwd;      /* the forward call's descriptor (fwd.out unused, fwd.workspace as that call left it) */
  const void *dout;               /* shaped like out */
  void *dxz;                      /* (batch, 2 dim, seqlen) io dtype, fully written */
  void *dconv_weight, *dconv_bias;                      /* fp32, accumulated; dconv_bias iff conv_bias */
  void *dx_proj_weight, *ddt_proj_weight;               /* fp32, accumulated */
  void *dout_proj_weight, *dout_proj_bias;              /* fp32, accumulated; iff their forward twins */
  void *dA, *dD, *ddelta_bias;                          /* fp32, accumulated; dD / ddelta_bias iff their forward twins */
  void *workspace;
  int64_t workspace_bytes;
} mxvl_mamba_inner_bwd_desc;
int64_t mxvl_mamba_inner_workspace_bytes(const mxvl_mamba_inner_desc *desc);
int64_t mxvl_mamba_inner_bwd_workspace_bytes(const mxvl_mamba_inner_desc *fwd);
int mxvl_mamba_inner_fwd(const mxvl_mamba_inner_desc *desc, void *hip_stream);
int mxvl_mamba_inner_bwd(const mxvl_mamba_inner_bwd_desc *desc, void *hip_stream);

/* last hipError_t observed by a failing launch on this thread (0 = hipSuccess) */
int mxvl_last_hip_error(void);
/* ---- DIAGNOSTICS (not part of the drop-in surface; no caller of the reference's interface needs them) -------------------------
 * They stay in the product library on purpose: the parity tests must run against the SAME .so the product loads (the driver records
 * which libraries the test processes mapped), and forcing each instantiation of the scan kernels through the public entry points
 * is how tests/test_scan_gpu.py covers every variant against the reference goldens.  Thread-local, no effect on results.
 *
 * Scan kernel selection for tests / A-B measurements.  Bits 0..7: forward kernel shape (0 = automatic, unknown ids fall
 * back to automatic); bits 8..15: backward (0 automatic, 1 = 32-row / 8-wave workgroups, 2 = 16-row / 4-wave; other
 * ids fall back to automatic).  Every choice is a correct kernel.  Thread-local: it affects only calls made by the calling thread --
 * note that autograd runs backward on its own thread.  The product library ignores bits >= 16 (measurement builds,
 * -DMXVL_ABLATE, read ablation switches there). */
void mxvl_set_scan_variant(int variant);
/* name of the kernel the last mxvl_scan_fwd on this thread dispatched to (static string) */
const char *mxvl_last_scan_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* MXVL_H_ */
