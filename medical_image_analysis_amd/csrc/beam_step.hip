// beam_step.hip -- one beam-search update of the report decoder as ONE kernel, for gfx950.
//
// Replaces the ~100 tiny kernels HF `generate` (transformers generation/utils.py `_beam_search`, vectorised form; called from
// CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:292-301 with num_beams=3, repetition_penalty=2.0,
// length_penalty=2.0, min_new_tokens) spends per generated token on log-softmax, RepetitionPenalty / MinNewTokensLength
// processors, a sort-based top-k over beams*vocab candidates and the beam bookkeeping: ~0.5 ms of a 3.9 ms token.
// Same arithmetic, same order of the fp32 operations as report_decoder._BeamState.advance (the torch restatement that is
// checked token-exact against HF): this kernel is tested against that function.
//
// One workgroup (512 threads) per batch element when the caller provides the arrival word (`scratch`; rows > 8 in the stepper:
// 6 / 8 / 16 samples x beams), else one workgroup walks the batch elements.  The workgroups meet in ONE returning atomicAdd
// whose addend packs "arrived / any heuristic open / all candidates stopped / all pools full": the last arriver owns the
// totals, writes *unfinished and *cur and clears the word (no fence: nothing but the atomic's own value crosses workgroups).
// Per element: (1) online max / sum-exp of every beam row (32
// independent loads in flight per thread: a single workgroup is latency-, not bandwidth-limited on 384 KB of logits), (2) every
// thread keeps the `keep` best penalised candidates of its strided share, history membership through an LDS bitmap,
// (3) `keep` rounds of a block arg-max merge them, (4) a few lanes do the bookkeeping on the `keep` survivors.
#include <math.h>
#include <stdlib.h>

#include "mxvl_common.h"

namespace mxvl {

constexpr int kMaxKeep = 16, kMaxBeams = 8, kMaxEos = 4, kMaxSurv = 256;
// (num_beams 5 of launch_mambaclip_test_iu.sh:27 keeps 2 x 5 = 10 candidates)

struct BeamArgs {
  int batch, nb, V, max_new, min_new, n_eos, early, keep, ablate;   // early: 1 = early_stopping True
  int vec4;                            // V % 4 == 0 and 16-byte aligned logits: a thread's words come four at a time
  float rep_pen;
  const float* logits;                 // (batch*nb, V)
  long long *run_seq, *fin_seq;        // (batch, nb, max_new)
  float *run_score, *fin_score;        // (batch, nb)
  unsigned char *fin_done, *heur_open; // (batch, nb), (batch)
  long long* cur;                      // scalar, incremented at the end
  const long long* eos;                // (n_eos)
  const float *len_tab, *hyp_tab;      // (max_new)
  long long *tok, *beam_src;           // (batch*nb)
  unsigned char* unfinished;           // scalar
  unsigned int* ticket;                // optional: arrival word of a multi-workgroup launch (zero between launches)
  unsigned char* unf_log;              // optional, host-visible (max_new): *unfinished of the step that ran at cur, at [cur]
  // vocabulary split over workgroups (workspace given): slices of VS words; the statistics and the `keep` best candidates of every
  // (sample, slice) come from beam_stats_kernel / beam_cand_kernel, beam_step_kernel merges them and does the bookkeeping
  int S, VS;
  float* ws_stats;                     // (batch*nb, S, 2): max, sum exp(x - max) of a row's slice
  float* ws_cand_v;                    // (batch, S, keep)
  int* ws_cand_i;                      // (batch, S, keep): beam * V + token
};

__device__ inline bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// ---- the vocabulary over workgroups (round 4) ---------------------------------------------------------------------------------------
// One workgroup per sample reads its nb x V logits twice (statistics, candidates) through ONE CU: 384 KB at the 30-40 GB/s a CU
// sustains is 2 x 10 us, and the per-word candidate test on 16 waves of one CU another 10 (profiles/r04_beam_phase_ablation2.txt).
// Split: S slices of the vocabulary, (1) beam_stats_kernel -- max / sum-exp of every (row, slice); (2) beam_cand_kernel -- every
// (sample, slice) workgroup combines the S partial statistics of its nb rows (the same S values in the same order in every
// workgroup: identical log Z), sweeps its slice with the thread-local two-entry lists of the one-workgroup kernel and leaves its
// `keep` best (value, index) pairs -- the global top-keep is a subset of the union of the per-slice top-keeps; (3) beam_step_kernel
// merges S x keep candidates instead of sweeping, then does the bookkeeping.  Kernel boundaries order the three steps.
constexpr int kSliceThreads = 256, kSliceWaves = kSliceThreads / 64;
constexpr int kCandThreads = 256, kCandWaves = kCandThreads / 64;   // (one wave per (sample, slice) -- no barriers -- measured slower: 47 dependent
                                                                    //  words per thread, 52 vs 41 us per step at batch 1)
// History membership (the repetition penalty) is an LDS bitmap over a TILE of the vocabulary, rebuilt per tile -- round 5: it used
// to cover the whole vocabulary (beams x vocab / 32 words), which put Qwen1.5's 151 936 tokens at beam 3 / 5 (the reference's
// IU-Xray decoder, MambaXrayVL_DownStream.py:65-77, launch_mambaclip_test_iu.sh:26-27) past the LDS of a workgroup.  A 32 000-word
// vocabulary is still one tile in both kernels: nothing changes for Llama.
constexpr int kCandTile = 8192;

__global__ __launch_bounds__(kSliceThreads) void beam_stats_kernel(const BeamArgs p) {
  __shared__ float s_red[kSliceWaves];
  if (*p.unfinished == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, row = blockIdx.y, V = p.V;
  const int v0 = s * p.VS, v1 = v0 + p.VS < V ? v0 + p.VS : V;
  const float* x = p.logits + (size_t)row * V;
  float m = -INFINITY;
  for (int v = v0 + tid; v < v1; v += kSliceThreads) m = fmaxf(m, x[v]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  float mx = s_red[0];
#pragma unroll
  for (int w = 1; w < kSliceWaves; ++w) mx = fmaxf(mx, s_red[w]);
  __syncthreads();
  float sum = 0.0f;
  if (mx > -INFINITY)
    for (int v = v0 + tid; v < v1; v += kSliceThreads) sum += fast_exp(x[v] - mx);      // (second pass: the slice is in L1 / L2)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) s_red[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    float ss = 0.0f;
    for (int w = 0; w < kSliceWaves; ++w) ss += s_red[w];
    p.ws_stats[((size_t)row * p.S + s) * 2] = mx;
    p.ws_stats[((size_t)row * p.S + s) * 2 + 1] = ss;
  }
}

__global__ __launch_bounds__(kCandThreads) void beam_cand_kernel(const BeamArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned int cand_smem[];      // [nb][words of the slice] history membership
  __shared__ float s_red[kCandWaves];
  __shared__ int s_redi[kCandWaves];
  __shared__ float s_rowmax[kMaxBeams], s_logz[kMaxBeams], s_rscore[kMaxBeams];
  __shared__ float s_top_lp[kMaxKeep];
  __shared__ int s_top_ix[kMaxKeep];
  __shared__ float s_surv_v[kMaxSurv];
  __shared__ int s_surv_i[kMaxSurv];
  __shared__ int s_nsurv;
  if (*p.unfinished == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, b = blockIdx.y, nb = p.nb, V = p.V, keep = p.keep, S = p.S, max_new = p.max_new;
  const int v0 = s * p.VS, v1 = v0 + p.VS < V ? v0 + p.VS : V;
  const int tile = p.VS < kCandTile ? p.VS : kCandTile;        // words of the vocabulary per bitmap tile
  const int words = (tile + 31) / 32;
  const int cur = (int)*p.cur;
  unsigned int* bitmap = cand_smem;
  int eos32[kMaxEos];
#pragma unroll
  for (int e = 0; e < kMaxEos; ++e) eos32[e] = e < p.n_eos ? (int)p.eos[e] : -1;
  // log-softmax statistics of the nb rows from the S partial results: lane k holds slice k, a shuffle tree combines them (the same
  // tree in every workgroup of the sample: identical log Z everywhere)
  for (int r = 0; r < nb; ++r) {
    const float* st = p.ws_stats + (size_t)(b * nb + r) * S * 2;
    const float mk = lane < S ? st[2 * lane] : -INFINITY, sk = lane < S ? st[2 * lane + 1] : 0.0f;
    float mx = mk;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float ss = mk > -INFINITY ? sk * fast_exp(mk - mx) : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (tid == 0) { s_rowmax[r] = mx; s_logz[r] = logf(ss); s_rscore[r] = p.run_score[b * nb + r]; }
  }
  // bitmap of the history tokens that fall into [t0, t0 + tile) (workgroup-uniform call: it synchronises)
  auto build_bitmap = [&](int t0) {
    __syncthreads();
    for (int i = tid; i < nb * words; i += kCandThreads) bitmap[i] = 0u;
    __syncthreads();
    if (p.rep_pen != 1.0f && cur > 0) {
      const long long* rs = p.run_seq + (size_t)b * nb * max_new;
      for (int i = tid; i < nb * cur; i += kCandThreads) {
        const int r = i / cur, t = i - r * cur;
        const long long tk = rs[r * max_new + t];
        const int v = (int)tk - t0;
        if (tk >= t0 && v < tile && tk < v1) atomicOr(&bitmap[r * words + (v >> 5)], 1u << (v & 31));
      }
    }
    __syncthreads();
  };
  const bool mask_eos = cur < p.min_new;
  auto cand_value = [&](int r, int v, float raw, int t0) -> float {
    float x = (raw - s_rowmax[r]) - s_logz[r];                              // log_softmax
    const bool hit = (bitmap[r * words + ((v - t0) >> 5)] >> ((v - t0) & 31)) & 1u;
    if (__any(hit)) x = hit ? (x < 0.0f ? x * p.rep_pen : x / p.rep_pen) : x;
    if (mask_eos) {
#pragma unroll
      for (int e = 0; e < kMaxEos; ++e)
        if (eos32[e] == v) x = -INFINITY;
    }
    return x + s_rscore[r];
  };
  float v1st = -INFINITY, v2nd = -INFINITY;
  int i1 = 0x7fffffff, i2 = 0x7fffffff;
  for (int t0 = v0; t0 < v1; t0 += tile) {
  build_bitmap(t0);
  const int t1 = t0 + tile < v1 ? t0 + tile : v1;
  for (int r = 0; r < nb; ++r) {
    const float* row = p.logits + (size_t)(b * nb + r) * V;
    for (int vb = t0 + tid; vb < t1; vb += 4 * kCandThreads) {      // four words of the thread in flight per trip
      float raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) raw[u] = vb + u * kCandThreads < t1 ? row[vb + u * kCandThreads] : -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = vb + u * kCandThreads;
        if (v >= t1) continue;
        const float x = cand_value(r, v, raw[u], t0);
        if (x >= v2nd) {
          const int idx = r * V + v;
          const bool b1 = better(x, idx, v1st, i1), b2 = better(x, idx, v2nd, i2);
          v2nd = b1 ? v1st : (b2 ? x : v2nd);
          i2 = b1 ? i1 : (b2 ? idx : i2);
          v1st = b1 ? x : v1st;
          i1 = b1 ? idx : i1;
        }
      }
    }
  }
  }
  int head = 0;
  for (int round = 0; round < keep; ++round) {
    float v = head == 0 ? v1st : head == 1 ? v2nd : -INFINITY;
    int ix = head == 0 ? i1 : head == 1 ? i2 : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(v, off, 64);
      const int oi = __shfl_xor(ix, off, 64);
      if (better(ov, oi, v, ix)) { v = ov; ix = oi; }
    }
    if (lane == 0) { s_red[wave] = v; s_redi[wave] = ix; }
    __syncthreads();
    if (tid == 0) {
      float bv = s_red[0];
      int bi = s_redi[0];
      for (int w = 1; w < kCandWaves; ++w)
        if (better(s_red[w], s_redi[w], bv, bi)) { bv = s_red[w]; bi = s_redi[w]; }
      s_top_lp[round] = bv;
      s_top_ix[round] = bi;
    }
    __syncthreads();
    const int win = s_top_ix[round];
    if ((head == 0 && i1 == win) || (head == 1 && i2 == win)) ++head;
  }
  // exactness: a thread with both entries among the winners may hold more candidates above the keep-th value (as in beam_step_kernel)
  if (tid == 0) s_nsurv = 0;
  __syncthreads();
  const bool suspect = head == 2 && keep > 2;
  if (__syncthreads_or(suspect ? 1 : 0)) {
    const float tau = s_top_lp[keep - 1];
    const int tau_ix = s_top_ix[keep - 1];
    for (int t0 = v0; t0 < v1; t0 += tile) {
      if (v1 - v0 > tile) build_bitmap(t0);          // (a one-tile slice still holds its bitmap)
      const int t1 = t0 + tile < v1 ? t0 + tile : v1;
      if (suspect) {
        for (int r = 0; r < nb; ++r) {
          const float* row = p.logits + (size_t)(b * nb + r) * V;
          for (int v = t0 + tid; v < t1; v += kCandThreads) {
            const int idx = r * V + v;
            if (idx == i1 || idx == i2) continue;
            const float x = cand_value(r, v, row[v], t0);
            if (better(x, idx, tau, tau_ix)) {
              const int slot = atomicAdd(&s_nsurv, 1);
              if (slot < kMaxSurv) { s_surv_v[slot] = x; s_surv_i[slot] = idx; }
            }
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      const int ns = s_nsurv < kMaxSurv ? s_nsurv : kMaxSurv;
      for (int q = 0; q < ns; ++q) {
        float x = s_surv_v[q];
        int ix = s_surv_i[q];
        for (int k = 0; k < keep; ++k)
          if (better(x, ix, s_top_lp[k], s_top_ix[k])) {
            const float tv = s_top_lp[k]; s_top_lp[k] = x; x = tv;
            const int ti = s_top_ix[k]; s_top_ix[k] = ix; ix = ti;
          }
      }
    }
    __syncthreads();
  }
  if (tid < keep) {
    p.ws_cand_v[((size_t)b * S + s) * keep + tid] = s_top_lp[tid];
    p.ws_cand_i[((size_t)b * S + s) * keep + tid] = s_top_ix[tid];
  }
}

// kBeamThreads x U logits of a row are in flight per trip: 1024 x 32 covers a 32 000-word vocabulary row in ONE round trip (the
// 512 x 16 shape of round 3 walked a row in four dependent trips, twice -- statistics and candidates -- for every beam row).
template <int kBeamThreads, int U>
__global__ __launch_bounds__(kBeamThreads) void beam_step_kernel(const BeamArgs p) {
  constexpr int kBeamWaves = kBeamThreads / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned int smem_u[];
  __shared__ float s_red[16];
  __shared__ int s_redi[16];
  static_assert(kBeamThreads / 64 <= 16 && U % 4 == 0, "s_red; four words per load");
  __shared__ float s_rowmax[kMaxBeams], s_logz[kMaxBeams];
  __shared__ float s_top_lp[kMaxKeep];
  __shared__ int s_top_ix[kMaxKeep];
  __shared__ int s_sel_src[kMaxBeams], s_sel_tok[kMaxBeams];      // per new live beam: parent beam, new token
  __shared__ int s_fin_from[kMaxBeams];                            // per new finished slot: < nb old slot, else nb + candidate
  __shared__ int s_any_open, s_all_hits, s_all_done, s_nsurv;
  __shared__ float s_surv_v[kMaxSurv];
  __shared__ int s_surv_i[kMaxSurv];
  __shared__ float s_bk_f[4 * kMaxKeep + 3 * kMaxBeams];
  __shared__ int s_bk_i[6 * kMaxKeep + 3 * kMaxBeams];
  __shared__ float s_in_fscore[kMaxBeams], s_in_rscore[kMaxBeams], s_tabs[2];
  __shared__ int s_in_fdone[kMaxBeams], s_in_open;
  __shared__ float s_out_rscore[kMaxBeams], s_out_fscore[kMaxBeams];
  __shared__ int s_out_fdone[kMaxBeams], s_out_open;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = p.nb, V = p.V, keep = p.keep, max_new = p.max_new;
  constexpr int TILE = kBeamThreads * U;                           // one trip of the sweeps = one bitmap tile
  const bool merged = p.ws_cand_v != nullptr;                      // the sweeps were done by beam_stats_kernel / beam_cand_kernel
  const int words = merged ? 0 : ((V < TILE ? V : TILE) + 31) / 32;
  unsigned int* bitmap = smem_u;                                   // [nb][words] history membership of the tile being swept
  long long* old_run = (long long*)(smem_u + ((nb * words + 1) & ~1));   // [nb][max_new]
  long long* old_fin = old_run + nb * max_new;                     // [nb][max_new]
  const int cur = (int)*p.cur;
  // a search that has already stopped stays stopped: the stepper enqueues token k + 1 before the host has read token k's flag
  // (report_decoder._search_lookahead), and that speculative launch must leave the state as the last real one left it
  if (*p.unfinished == 0) return;
  if (tid == 0) { s_any_open = 0; s_all_hits = 1; s_all_done = 1; }
  int eos32[kMaxEos];                  // the EOS ids live in registers: they are tested against every candidate
#pragma unroll
  for (int e = 0; e < kMaxEos; ++e) eos32[e] = e < p.n_eos ? (int)p.eos[e] : -1;

  for (int b = blockIdx.x; b < p.batch; b += gridDim.x) {
    __syncthreads();
    const float* lg = p.logits + (size_t)b * nb * V;
    long long* rs = p.run_seq + (size_t)b * nb * max_new;
    long long* fs = p.fin_seq + (size_t)b * nb * max_new;
    // the few scalars the bookkeeping lane needs: one parallel round trip now instead of ~25 serial ones later
    if (tid < nb) {
      s_in_fscore[tid] = p.fin_score[b * nb + tid];
      s_in_rscore[tid] = p.run_score[b * nb + tid];
      s_in_fdone[tid] = p.fin_done[b * nb + tid];
    } else if (tid == 64) {
      s_in_open = p.heur_open[b];
    } else if (tid == 65) {
      s_tabs[0] = p.len_tab[cur];
    } else if (tid == 66) {
      s_tabs[1] = p.hyp_tab[cur];
    }
    for (int i = tid; i < nb * max_new; i += kBeamThreads) { old_run[i] = rs[i]; old_fin[i] = fs[i]; }
    __syncthreads();
    // bitmap of the history tokens in [t0, t0 + TILE) (workgroup-uniform call: it synchronises)
    auto build_bitmap = [&](int t0) {
      __syncthreads();
      for (int i = tid; i < nb * words; i += kBeamThreads) bitmap[i] = 0u;
      __syncthreads();
      if (p.rep_pen != 1.0f && cur > 0)
        for (int i = tid; i < nb * cur; i += kBeamThreads) {
          const int r = i / cur, t = i - r * cur;
          const long long tk = old_run[r * max_new + t];
          const int v = (int)tk - t0;
          if (tk >= t0 && v < TILE && tk < V) atomicOr(&bitmap[r * words + (v >> 5)], 1u << (v & 31));
        }
      __syncthreads();
    };
    // ---- (1) log-softmax statistics per beam row: one sweep, U independent loads in flight per thread --------------
    // word u of a thread's trip: four consecutive words per 16-byte load when the rows allow it (a quarter of the load instructions)
    auto vof = [&](int base, int u) {
      return p.vec4 ? base + ((u >> 2) * kBeamThreads + tid) * 4 + (u & 3) : base + u * kBeamThreads + tid;
    };
    auto load_trip = [&](const float* row, int base, float (&x)[U]) {
      if (p.vec4) {
#pragma unroll
        for (int q = 0; q < U / 4; ++q) {
          const int v = base + (q * kBeamThreads + tid) * 4;
          const float4 t = v < V ? *(const float4*)(row + v) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
          x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = base + u * kBeamThreads + tid;
          x[u] = v < V ? row[v] : -INFINITY;
        }
      }
    };
    for (int r = 0; r < (merged ? 0 : nb); ++r) {
      const float* row = lg + (size_t)r * V;
      float m = -INFINITY, s = 0.0f;
      for (int base = 0; base < V; base += kBeamThreads * U) {
        float x[U];
        load_trip(row, base, x);
        float cm = x[0];
#pragma unroll
        for (int u = 1; u < U; ++u) cm = fmaxf(cm, x[u]);
        const float mn = fmaxf(m, cm);
        if (mn > -INFINITY) {
          float cs = 0.0f;
#pragma unroll
          for (int u = 0; u < U; ++u) cs += fast_exp(x[u] - mn);
          s = s * fast_exp(m - mn) + cs;
          m = mn;
        }
      }
      float wm = m;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
      if (lane == 0) s_red[wave] = wm;
      __syncthreads();
      float mx = s_red[0];
#pragma unroll
      for (int w = 1; w < kBeamWaves; ++w) mx = fmaxf(mx, s_red[w]);
      __syncthreads();
      float ws = (m > -INFINITY) ? s * fast_exp(m - mx) : 0.0f;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ws += __shfl_xor(ws, off, 64);
      if (lane == 0) s_red[wave] = ws;
      __syncthreads();
      if (tid == 0) {
        float ss = 0.0f;
        for (int w = 0; w < kBeamWaves; ++w) ss += s_red[w];
        s_rowmax[r] = mx;
        s_logz[r] = logf(ss);
      }
      __syncthreads();
    }
    if (MXVL_ABL(p.ablate == 1)) continue;
    // ---- (2) every thread keeps its TWO best penalised, score-shifted candidates (branch-free insertion) ---------------
    // A sorted `keep`-deep list per thread costs a wave-wide insertion for almost every element (some lane always
    // inserts): 100 us.  Two entries per thread are exact unless one thread owns three of the final `keep` -- detected
    // below and repaired by a rescan of that thread's share (practically never taken).
    auto cand_value = [&](int r, int v, float raw, float mx, float lz, float sc, bool mask_eos, int t0) -> float {
      float x = (raw - mx) - lz;                                              // log_softmax
      // (the penalty behind a wave-uniform test: as a select, hipcc ran the fp32 division sequence for every word of the vocabulary)
      const bool hit = (bitmap[r * words + ((v - t0) >> 5)] >> ((v - t0) & 31)) & 1u;
      if (__any(hit)) x = hit ? (x < 0.0f ? x * p.rep_pen : x / p.rep_pen) : x;
      if (mask_eos) {
#pragma unroll
        for (int e = 0; e < kMaxEos; ++e)
          if (eos32[e] == v) x = -INFINITY;
      }
      return x + sc;
    };
    float v1 = -INFINITY, v2 = -INFINITY;
    int i1 = 0x7fffffff, i2 = 0x7fffffff;
    if (merged && tid < p.S * keep) {     // one entry per thread: the `keep` best of every vocabulary slice of this sample
      v1 = p.ws_cand_v[(size_t)b * p.S * keep + tid];
      i1 = p.ws_cand_i[(size_t)b * p.S * keep + tid];
    }
    for (int base = 0; base < (merged ? 0 : V); base += TILE) {
      build_bitmap(base);
      for (int r = 0; r < nb; ++r) {
        const float* row = lg + (size_t)r * V;
        const float mx = s_rowmax[r], lz = s_logz[r], sc = s_in_rscore[r];
        const bool mask_eos = cur < p.min_new;
        float xs[U];
        load_trip(row, base, xs);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = vof(base, u);
          if (v < V) {
            const float x = cand_value(r, v, xs[u], mx, lz, sc, mask_eos, base);
            if (x >= v2) {           // below the thread's second entry nothing changes: the insertion is the rare path
              const int idx = r * V + v;
              const bool b1 = better(x, idx, v1, i1), b2 = better(x, idx, v2, i2);
              v2 = b1 ? v1 : (b2 ? x : v2);
              i2 = b1 ? i1 : (b2 ? idx : i2);
              v1 = b1 ? x : v1;
              i1 = b1 ? idx : i1;
            }
          }
        }
      }
    }
    if (MXVL_ABL(p.ablate == 2)) continue;
    // ---- (3) `keep` rounds of block arg-max over the thread-local heads --------------------------------------------------
    int head = 0;
    for (int round = 0; round < keep; ++round) {
      float v = head == 0 ? v1 : head == 1 ? v2 : -INFINITY;
      int ix = head == 0 ? i1 : head == 1 ? i2 : 0x7fffffff;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(ix, off, 64);
        if (better(ov, oi, v, ix)) { v = ov; ix = oi; }
      }
      if (lane == 0) { s_red[wave] = v; s_redi[wave] = ix; }
      __syncthreads();
      if (tid == 0) {
        float bv = s_red[0];
        int bi = s_redi[0];
        for (int w = 1; w < kBeamWaves; ++w)
          if (better(s_red[w], s_redi[w], bv, bi)) { bv = s_red[w]; bi = s_redi[w]; }
        s_top_lp[round] = bv;
        s_top_ix[round] = bi;
      }
      __syncthreads();
      const int win = s_top_ix[round];
      if ((head == 0 && i1 == win) || (head == 1 && i2 == win)) ++head;
    }
    // exactness: a thread with both entries among the winners may hold more candidates above the keep-th value
    if (tid == 0) s_nsurv = 0;
    __syncthreads();
    const bool suspect = !merged && head == 2 && keep > 2;
    if (__syncthreads_or(suspect ? 1 : 0)) {
      const float tau = s_top_lp[keep - 1];
      const int tau_ix = s_top_ix[keep - 1];
      for (int base = 0; base < (merged ? 0 : V); base += TILE) {
        if (V > TILE) build_bitmap(base);            // (a one-tile vocabulary still holds its bitmap)
        if (suspect) {
          for (int r = 0; r < nb; ++r) {
            const float* row = lg + (size_t)r * V;
            const float mx = s_rowmax[r], lz = s_logz[r], sc = s_in_rscore[r];
            for (int u = 0; u < U; ++u) {            // this thread's share of the tile, word by word
              const int v = vof(base, u);
              if (v >= V) continue;
              const int idx = r * V + v;
              if (idx == i1 || idx == i2) continue;
              const float x = cand_value(r, v, row[v], mx, lz, sc, cur < p.min_new, base);
              if (better(x, idx, tau, tau_ix)) {
                const int slot = atomicAdd(&s_nsurv, 1);
                if (slot < kMaxSurv) { s_surv_v[slot] = x; s_surv_i[slot] = idx; }
              }
            }
          }
        }
      }
      __syncthreads();
      if (tid == 0) {     // insertion of the (few) survivors into the sorted winners
        const int ns = s_nsurv < kMaxSurv ? s_nsurv : kMaxSurv;
        for (int q = 0; q < ns; ++q) {
          float x = s_surv_v[q];
          int ix = s_surv_i[q];
          for (int k = 0; k < keep; ++k)
            if (better(x, ix, s_top_lp[k], s_top_ix[k])) {
              const float tv = s_top_lp[k]; s_top_lp[k] = x; x = tv;
              const int ti = s_top_ix[k]; s_top_ix[k] = ix; ix = ti;
            }
        }
      }
      __syncthreads();
    }
    if (MXVL_ABL(p.ablate == 3)) continue;
    // ---- (4) bookkeeping on the survivors (one lane; everything here is `keep` <= 8 wide) -----------------------------
    if (tid == 0) {
      float* top_lp = s_bk_f;                      // dynamically indexed: LDS, not private (scratch) arrays
      float* live_lp = s_bk_f + kMaxKeep;
      float* cand = s_bk_f + 2 * kMaxKeep;
      float* ms = s_bk_f + 3 * kMaxKeep;           // [kMaxBeams + kMaxKeep]
      float* new_score = ms + kMaxBeams + kMaxKeep;
      float* fscore = new_score + kMaxBeams;
      int* src = s_bk_i;
      int* ntok = s_bk_i + kMaxKeep;
      int* hits = s_bk_i + 2 * kMaxKeep;
      int* used = s_bk_i + 3 * kMaxKeep;
      int* md = s_bk_i + 4 * kMaxKeep;             // [kMaxBeams + kMaxKeep]
      int* mu = md + kMaxBeams + kMaxKeep;
      int* fdone = mu + kMaxBeams + kMaxKeep;
      bool all_done_b = true;
      for (int i = 0; i < nb; ++i) all_done_b = all_done_b && s_in_fdone[i];
      const bool open_b = s_in_open != 0;
      for (int k = 0; k < keep; ++k) {
        top_lp[k] = s_top_lp[k];
        src[k] = s_top_ix[k] / V;
        ntok[k] = s_top_ix[k] - src[k] * V;
        bool h = cur + 1 >= max_new;
#pragma unroll
        for (int e = 0; e < kMaxEos; ++e) h = h || (eos32[e] == ntok[k]);
        hits[k] = h ? 1 : 0;
        if (!h) s_all_hits = 0;
        live_lp[k] = top_lp[k] + (h ? 1.0f : 0.0f) * -1e9f;
        const bool just = h && k < nb;
        float c = top_lp[k] / s_tabs[0];
        if (p.early == 1) c = c + (all_done_b ? 1.0f : 0.0f) * -1e9f;
        c = (c + (open_b ? 0.0f : 1.0f) * -1e9f) + (just ? 0.0f : 1.0f) * -1e9f;
        cand[k] = c;
        used[k] = 0;
      }
      // live beams: the nb best of live_lp (descending, first index wins ties)
      for (int i = 0; i < nb; ++i) {
        int best = -1;
        for (int k = 0; k < keep; ++k)
          if (!used[k] && (best < 0 || live_lp[k] > live_lp[best])) best = k;
        used[best] = 1;
        new_score[i] = live_lp[best];
        s_sel_src[i] = src[best];
        s_sel_tok[i] = ntok[best];
      }
      // finished pool: the nb best of [old finished scores | candidate scores]
      for (int i = 0; i < nb; ++i) { ms[i] = s_in_fscore[i]; md[i] = s_in_fdone[i] != 0; mu[i] = 0; }
      for (int k = 0; k < keep; ++k) { ms[nb + k] = cand[k]; md[nb + k] = hits[k] && k < nb; mu[nb + k] = 0; }
      for (int i = 0; i < nb; ++i) {
        int best = -1;
        for (int k = 0; k < nb + keep; ++k)
          if (!mu[k] && (best < 0 || ms[k] > ms[best])) best = k;
        mu[best] = 1;
        fscore[i] = ms[best];
        fdone[i] = md[best];
        s_fin_from[i] = best;
      }
      float fmin = fscore[0];
      bool fall = true;
      for (int i = 0; i < nb; ++i) { fmin = fminf(fmin, fscore[i]); fall = fall && fdone[i]; }
      const float best_live = new_score[0] / s_tabs[1];
      bool any = false;
      for (int i = 0; i < nb; ++i) any = any || (best_live > (fdone[i] ? fmin : -1e9f));
      const bool open_new = open_b && any;
      s_out_open = open_new ? 1 : 0;
      if (open_new) s_any_open = 1;
      if (!fall) s_all_done = 0;
      for (int i = 0; i < nb; ++i) {
        s_out_rscore[i] = new_score[i];
        s_out_fscore[i] = fscore[i];
        s_out_fdone[i] = fdone[i] ? 1 : 0;
      }
    }
    __syncthreads();
    if (tid < nb) {
      p.run_score[b * nb + tid] = s_out_rscore[tid];
      p.fin_score[b * nb + tid] = s_out_fscore[tid];
      p.fin_done[b * nb + tid] = (unsigned char)s_out_fdone[tid];
      p.tok[b * nb + tid] = s_sel_tok[tid];
      p.beam_src[b * nb + tid] = (long long)s_sel_src[tid] + (long long)b * nb;
    } else if (tid == 64) {
      p.heur_open[b] = (unsigned char)s_out_open;
    }
    // sequences: new live rows = parent row + new token at `cur`; finished rows from the old pool or from a candidate
    for (int i = tid; i < nb * max_new; i += kBeamThreads) {
      const int r = i / max_new, t = i - r * max_new;
      rs[i] = (t == cur) ? (long long)s_sel_tok[r] : old_run[s_sel_src[r] * max_new + t];
      const int from = s_fin_from[r];
      long long fv;
      if (from < nb) fv = old_fin[from * max_new + t];
      else {
        const int k = from - nb;
        const int sb = s_top_ix[k] / V;
        fv = (t == cur) ? (long long)(s_top_ix[k] - sb * V) : old_run[sb * max_new + t];
      }
      fs[i] = fv;
    }
  }
  __syncthreads();
  if (MXVL_ABL(p.ablate != 0)) return;      // a truncated step must not stop the search it is timed in
  if (tid == 0) {
    bool any_open = s_any_open != 0, all_hits = s_all_hits != 0, all_done = s_all_done != 0, last = true;
    if (gridDim.x > 1) {
      const unsigned int inc = 1u | (any_open ? 1u << 8 : 0u) | (all_hits ? 1u << 16 : 0u) | (all_done ? 1u << 24 : 0u);
      const unsigned int tot = atomicAdd(p.ticket, inc) + inc;
      last = (tot & 0xffu) == gridDim.x;
      any_open = ((tot >> 8) & 0xffu) != 0;
      all_hits = ((tot >> 16) & 0xffu) == gridDim.x;
      all_done = ((tot >> 24) & 0xffu) == gridDim.x;
      if (last) atomicExch(p.ticket, 0u);
    }
    if (last) {       // every workgroup read *cur at its start, before it arrived
      bool unf = any_open && !all_hits;
      if (p.early == 1) unf = unf && !all_done;
      *p.unfinished = unf ? 1 : 0;
      if (p.unf_log && cur < max_new) __hip_atomic_store(p.unf_log + cur, (unsigned char)(unf ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      *p.cur = cur + 1;
    }
  }
}

}  // namespace mxvl

using namespace mxvl;

// slices of the vocabulary: enough workgroups to fill the chip at batch 1, S * keep <= 512 candidates to merge; a vocabulary beyond
// 64 K words (Qwen1.5: 151 936) is cut into 64 slices -- at 16 x beam 5 the candidate sweep of 16 slices took 124 us of a 2.6 ms token
// (profiles/r05_decode_timeline_qwen_b16x5_before.txt: 185 dependent words per thread), S <= 64: the statistics are combined by one wave
static int beam_slices(int batch, int vocab) { return vocab > 65536 ? 64 : (batch <= 4 ? 32 : 16); }

extern "C" int64_t mxvl_beam_workspace_bytes(int batch, int beams, int keep) {
  if (batch <= 0 || beams <= 0 || keep <= 0) return 0;
  const int64_t S = 64;                    // the largest slicing (the call does not know the vocabulary)
  return 4 * ((int64_t)batch * beams * S * 2 + 2 * (int64_t)batch * S * keep);
}

extern "C" int mxvl_beam_step(const mxvl_beam_desc* d, void* hip_stream) {
  if (!d || !d->logits || !d->run_seq || !d->fin_seq || !d->run_score || !d->fin_score || !d->fin_done || !d->heur_open ||
      !d->cur || !d->len_tab || !d->hyp_tab || !d->tok || !d->beam_src || !d->unfinished)
    return MXVL_ERR_NULL;
  if (d->n_eos > 0 && !d->eos) return MXVL_ERR_NULL;
  if (d->batch <= 0 || d->beams <= 0 || d->vocab <= 0 || d->max_new <= 0) return MXVL_ERR_SHAPE;
  if (d->n_eos > kMaxEos || d->beams > kMaxBeams || d->keep > kMaxKeep || d->keep < d->beams || (long long)d->beams * d->vocab > 0x7fffffffLL)
    return MXVL_ERR_UNSUPPORTED;
  BeamArgs a;
  a.batch = d->batch; a.nb = d->beams; a.V = d->vocab; a.max_new = d->max_new; a.min_new = d->min_new; a.n_eos = d->n_eos;
  a.early = d->early_stopping; a.keep = d->keep; a.rep_pen = d->repetition_penalty;
  a.ablate = MXVL_ABL_ENV("MXVL_BEAM_ABLATE");
  a.logits = (const float*)d->logits; a.run_seq = (long long*)d->run_seq; a.fin_seq = (long long*)d->fin_seq;
  a.run_score = (float*)d->run_score; a.fin_score = (float*)d->fin_score; a.fin_done = (unsigned char*)d->fin_done;
  a.heur_open = (unsigned char*)d->heur_open; a.cur = (long long*)d->cur; a.eos = (const long long*)d->eos;
  a.len_tab = (const float*)d->len_tab; a.hyp_tab = (const float*)d->hyp_tab; a.tok = (long long*)d->tok;
  a.beam_src = (long long*)d->beam_src; a.unfinished = (unsigned char*)d->unfinished;
  a.ticket = (unsigned int*)d->scratch;
  a.unf_log = (unsigned char*)d->unfinished_log;
  a.vec4 = (a.V % 4 == 0 && ((uintptr_t)a.logits & 15) == 0) ? 1 : 0;
  const int grid = (a.ticket && d->batch > 1) ? (d->batch < 255 ? d->batch : 255) : 1;
  a.S = 0; a.VS = 0; a.ws_stats = nullptr; a.ws_cand_v = nullptr; a.ws_cand_i = nullptr;
  if (d->workspace && d->workspace_bytes >= mxvl_beam_workspace_bytes(d->batch, d->beams, d->keep) && !MXVL_ABL_ENV("MXVL_BEAM_ONE_WG")) {
    a.S = beam_slices(d->batch, d->vocab);
    a.VS = ((a.V + a.S - 1) / a.S + 31) & ~31;
    a.ws_stats = (float*)d->workspace;
    a.ws_cand_v = a.ws_stats + (size_t)d->batch * d->beams * a.S * 2;
    a.ws_cand_i = (int*)(a.ws_cand_v + (size_t)d->batch * a.S * d->keep);
    hipLaunchKernelGGL(beam_stats_kernel, dim3(a.S, d->batch * d->beams), dim3(kSliceThreads), 0, (hipStream_t)hip_stream, a);
    const size_t lds_c = 4 * (size_t)a.nb * (((a.VS < kCandTile ? a.VS : kCandTile) + 31) / 32);     // <= 8 KB at 8 beams
    hipLaunchKernelGGL(beam_cand_kernel, dim3(a.S, d->batch), dim3(kCandThreads), lds_c, (hipStream_t)hip_stream, a);
  }
  static const int shape = MXVL_ABL_ENV("MXVL_BEAM_SHAPE");      // measurement build: 1 = the 512 x 16 shape
  // LDS: the copies of the old sequences (16 bytes x beams x max_new) + the history bitmap of ONE sweep tile (threads x U words of the
  // vocabulary; none when the sweeps were done by the slice kernels) -- independent of the vocabulary size
  auto launch = [&](auto kern, int threads, int tile) -> int {
    const size_t words = a.ws_cand_v ? 0 : (size_t)((a.V < tile ? a.V : tile) + 31) / 32;
    const size_t lds = 4 * ((a.nb * words + 1) & ~(size_t)1) + 8 * (size_t)2 * a.nb * a.max_new;
    if (lds > 160 * 1024) return MXVL_ERR_UNSUPPORTED;          // max_new in the thousands
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return MXVL_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, (hipStream_t)hip_stream, a);
    return MXVL_OK;
  };
  int rc;
  if (a.ws_cand_v && a.S * a.keep <= 256)       // merge + bookkeeping only: four waves (barriers and the serial lane are all that is left)
    rc = launch(beam_step_kernel<256, 8>, 256, 256 * 8);
  else if (MXVL_ABL(shape == 1) || (a.ws_cand_v && a.S * a.keep <= 512))
    rc = launch(beam_step_kernel<512, 16>, 512, 512 * 16);
  else rc = launch(beam_step_kernel<1024, 32>, 1024, 1024 * 32);
  if (rc != MXVL_OK) return rc;
  return hipGetLastError() == hipSuccess ? MXVL_OK : MXVL_ERR_LAUNCH;
}
