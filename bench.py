#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native MambaXray-VL hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM.
Rank 0 prints ONE JSON line (contract in the task statement): whole-job images/sec, plus
  "roofline":     algorithmic HBM bytes of the dominant kernel / its mean launch duration (HIP events
                  on the launch stream, inside the timed region) against the 8 TB/s HBM3E peak;
  "cpu_baseline": the CPU oracle (oracle/mxvl_oracle.c, a restatement of the reference's
                  selective_scan_ref) timed on this box's host cores on a bounded sample (rank 0, N=1).

Workloads (BASELINE.json configs):
  arm_pretrain_large_1024  configs[2]: MambaXray-VL-Large stage-1 pre-training step (forward + backward + clip +
                   AdamW) on 1024x1024 synthetic X-rays, bf16 autocast, DDP over RCCL; images/sec
  arm_pretrain_base_192    the reference factory arm_base_pz16 (192x192, L=128) -- quick variant
  scan_fwd_cfg2    configs[1]: MambaXray-VL-Base selective-scan forward, B=32 L=196 D=768 N=16 fp32
  scan_fwd_target  north_star roofline shape: B=8 L=4096 D=1536 N=16 fp32 (one 1024x1024 X-ray = 4096 tokens)
  scan_fwd_target_bf16  same with bf16 io
Multi-GPU: the scan shards over independent batch elements -- every rank runs the same per-GPU
batch (weak scaling), no data-path collective; value = N * images / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md)


MFMA_KINDS = ("gemm_tn",)      # timed kernels whose work figure is FLOPs (selective_scan_interface.gemm_tn); the others report bytes


def _kernel_stats(timers):
    stats = {}
    for kind, e0, e1, work in timers:
        d = stats.setdefault(kind, [0.0, 0, 0])
        d[0] += e0.elapsed_time(e1)
        d[1] += 1
        d[2] += work
    return stats


def _mfma_kernel_object(stats, timer_steps, step_ms):
    """The weight-gradient MFMA kernel's own roofline object (an extra to the contract's `roofline`, which is the dominant kernel's)."""
    if "gemm_tn" not in stats:
        return None
    ms, calls, flops = stats["gemm_tn"]
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "gemm_tn_kernel (weight gradients dW = dy^T x of the blocks' token-major linears, csrc/gemm_tn.hip)", "bound": "mfma",
            "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "traffic": None,
            "kernel_ms": ms / calls, "launches_timed": calls, "algorithmic_flops_per_launch": flops // calls,
            "step_share": round(ms / timer_steps / step_ms, 4)}


def pmc_traffic(workload):
    """HBM bytes per launch of the workload's dominant kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json,
    written by tools/gpu_round.sh: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same bench command, KB units,
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  None when no pass was recorded for the workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f).get(workload)
        except (OSError, ValueError):
            continue
        if rec:
            return {"bytes": rec["bytes"], "source": os.path.basename(path), "fetch_kb": rec["fetch_kb"], "write_kb": rec["write_kb"],
                    "kernel": rec.get("kernel")}
    return None


def _dp_label(world):
    if os.environ.get("MXVL_BENCH_ONE_GPU") == "1" and world > 1:
        return f"dp{world} (DEV CHECK: all ranks on cuda:0, gloo all-reduce -- not a scaling number)"
    return f"dp{world} (DDP, RCCL all-reduce of fp32 grads, 256 MiB buckets)"


def attach_traffic(roofline, workload, command=None):
    """roofline.traffic = HBM bytes per launch of the dominant kernel.  It is NOT re-measured by this process (PMC needs
    rocprofv3 around the command): the number is read from the committed PMC record of the same command and labelled so."""
    pmc = pmc_traffic(workload)
    if pmc is None:
        return
    roofline["traffic"] = pmc["bytes"]
    roofline["traffic_from_profile"] = (f"profiles/{pmc['source']} ({pmc.get('kernel')}): FETCH_SIZE {pmc['fetch_kb']} KB x2 (gfx950 "
                                        f"correction) + WRITE_SIZE {pmc['write_kb']} KB per launch, separate rocprofv3 --pmc passes of "
                                        f"`{command or 'bench.py --workload ' + workload}`; copied from the profile, not re-measured in this run")

WORKLOADS = {
    # name: (B, D, L, N, torch dtype name, description)
    "scan_fwd_cfg2": (32, 768, 196, 16, "float32",
                      "configs[1]: MambaXray-VL-Base selective-scan forward B=32 L=196 D=768 N=16 (z, D, delta_bias, softplus)"),
    "scan_fwd_target": (8, 1536, 4096, 16, "float32",
                        "north_star roofline shape: selective-scan forward B=8 L=4096 D=1536 N=16 (z, D, delta_bias, softplus)"),
    "scan_fwd_target_bf16": (8, 1536, 4096, 16, "bfloat16",
                             "north_star roofline shape with bf16 io: B=8 L=4096 D=1536 N=16"),
    # the dominant hand-written kernel of the default workload, alone: same shape, dtype and entry point (mxvl_scan_bwd) as the
    # 24 launches per step of arm_pretrain_large_1024 at per-GPU batch 16 -- what the PMC traffic passes are taken on
    "scan_bwd_pretrain": (16, 1024, 4080, 16, "bfloat16",
                          "selective-scan BACKWARD of the ARM-large 1024x1024 pre-training step: B=16 L=4080 D=1024 N=16 bf16 io"),
}
PRETRAIN_WORKLOADS = {
    # name: (img, patch, embed_dim, depth, dec_dim, per-GPU batch, description)
    "arm_pretrain_large_1024": (1024, 16, 1024, 24, 512, 16,
                                "configs[2]: MambaXray-VL-Large (VisionMamba 1024x24, dec 512x4) stage-1 ARM pre-training "
                                "step, 1024x1024 synthetic X-rays (4096 patches, 4080-token scan), bf16 autocast"),
    "arm_pretrain_base_192": (192, 16, 768, 12, 512, 64,
                              "reference factory arm_base_pz16 (192x192, 128-token scan) stage-1 pre-training step, bf16 autocast"),
}
MAE_WORKLOADS = {
    # name: (per-GPU batch, description)
    "mae_vit_large_1280": (256, "HD_Xray_Pretrain_MAE: mae_vit_large_patch16 (1280x1280 1-channel X-rays, 64x64 patches -> 400 tokens, "
                                "encoder 1024x24x16h, decoder 512x8x16h), chest-region masking (mask_type 1, ratios 0.85 / 0.95), the reference's "
                                "arithmetic: fp16 autocast + GradScaler (pretrain/main.py:211-213,317); per-GPU batch sized for the 288 GB of an "
                                "MI355X (the reference's default of 2 per 24 GB card leaves the 47-token encoder GEMMs launch-bound)"),
}
VMAMBA_WORKLOADS = {
    # name: (per-GPU batch, description)
    "arm_encoder_large_224": (64, "stage-2/3 visual encoder MambaXray-VL-Large (arm_large_pz16: 1024 x 24, bimamba v3 = 4 scan directions, "
                                  "middle cls token) at 224x224 (197 tokens), encoder training step on a synthetic feature loss, bf16 autocast"),
    "vmamba_base_224": (32, "configs[4]: R2GenCSR visual encoder VMamba-base (vssm1_base_0229: dims 128..1024, depths [2,2,15,2], d_state 1, "
                            "SS2D v3noz) at 224x224, encoder training step (forward + backward + AdamW on a synthetic pooled-feature loss), "
                            "bf16 autocast"),
}
FINETUNE_WORKLOADS = {
    # name: (kind, per-GPU batch, description) -- the training steps of the reference's report-generation stages, whole pipeline
    "finetune_stage3_llama7b": ("mambaxray", 6,
                                "CXPMRG_Bench stage 3 as launch/launch_mambaclip_chexpert.sh runs it: MambaXray-VL-Large encoder (arm_large_pz16, 4 scan "
                                "directions, 224x224, TRAINABLE: --freeze_vm False) -> llama_proj -> LayerNorm -> [bos, prompt, 197 image tokens, prompt, "
                                "report (max_length 100)] -> FROZEN Llama-2-7B-shaped LLM in fp16 (torch_dtype of MambaXrayVL_DownStream.py:85-92) under "
                                "bf16 autocast, causal-LM loss, backward through the LLM into the encoder, AdamW(lr 1e-4); batch_size 6"),
    "r2gencsr_step": ("r2gencsr", 36,
                      "BASELINE configs[4] as R2GenCSR/scripts/mimic.sh trains it: VMamba-base encoder (TRAINABLE) + linear projector + 3 + 3 context "
                      "studies encoded under no_grad, pooled-feature residuals wrapped in their prompts (R2GenCSR.py:376-474), 49 image tokens, report "
                      "(max_length 100) -> FROZEN Llama-2-7B-shaped LLM in fp16 under bf16 autocast, loss + backward + AdamW(lr 1e-4); batch_size 36"),
}
DECODE_WORKLOADS = {
    # name: (vocab, hidden, inter, layers, heads, kv_heads, prompt_len, new_tokens, beams, batch, description)
    "decode_llama7b_128": (32000, 4096, 11008, 32, 32, 32, 230, 128, 3, 1,
                           "configs[3]: report generation with a Llama-2-7B-shaped decoder (random-init bf16 weights), 230-embedding "
                           "prompt [bos, prompt, 197 image tokens, prompt], beam 3, 128 new tokens (min = max = 128), repetition/length penalty 2.0"),
    # the batches the reference's launch scripts decode at (rows = batch x beams): validation 6 x 3 (launch_mambaclip_chexpert.sh:23,
    # launch_mambaclip_mimic.sh:25), test 8 x 3 (launch_mambaclip_test_cheXpert.sh:26), IU test 16 x beam 5 (launch_mambaclip_test_iu.sh:26-27)
    "decode_llama7b_b6x3": (32000, 4096, 11008, 32, 32, 32, 230, 128, 3, 6,
                            "configs[3] at the reference's validation batch: val_batch_size 6 x beam 3 = 18 rows per decoder step, Llama-2-7B-shaped "
                            "decoder (random-init bf16), 230-embedding prompts, 128 new tokens (min = max = 128), repetition/length penalty 2.0"),
    "decode_llama7b_b8x3": (32000, 4096, 11008, 32, 32, 32, 230, 128, 3, 8,
                            "configs[3] at the reference's test batch: test_batch_size 8 x beam 3 = 24 rows per decoder step, otherwise as decode_llama7b_b6x3"),
    "decode_llama7b_b16x5": (32000, 4096, 11008, 32, 32, 32, 230, 128, 5, 16,
                             "configs[3] at the reference's IU-Xray test batch: 16 x beam 5 = 80 rows per decoder step, otherwise as decode_llama7b_b6x3"),
    "decode_llama7b_b16x3": (32000, 4096, 11008, 32, 32, 32, 230, 128, 3, 16,
                             "configs[3] at the reference's config default: batch 16 x beam 3 = 48 rows per decoder step (configs/config.py:11-12,50), "
                             "otherwise as decode_llama7b_b6x3"),
    # the reference's own LLM dtype: torch_dtype=torch.float16 (MambaXrayVL_DownStream.py:72,85,92) -- the fp16 instantiations of every decode kernel
    "decode_llama7b_128_fp16": (32000, 4096, 11008, 32, 32, 32, 230, 128, 3, 1,
                                "configs[3] in the dtype the reference loads its LLM in: Llama-2-7B-shaped decoder, random-init FP16 weights, "
                                "230-embedding prompt, beam 3, 128 new tokens (min = max = 128), repetition/length penalty 2.0"),
    "decode_llama7b_b6x3_fp16": (32000, 4096, 11008, 32, 32, 32, 230, 128, 3, 6,
                                 "decode_llama7b_b6x3 with fp16 weights / activations (the reference's torch_dtype)"),
    # the reference's IU-Xray decoder: Qwen1.5-1.8B-Chat in fp16 (MambaXrayVL_DownStream.py:65-77), test_batch_size 16 x beam 5, min 40 / max 100 new
    # tokens, repetition / length penalty 2.0 (launch/launch_mambaclip_test_iu.sh:26-35)
    "decode_qwen1p8b_b16x5": (151936, 2048, 5504, 24, 16, 16, 230, 100, 5, 16,
                              "the reference's IU-Xray report decoder: Qwen1.5-1.8B-shaped (hidden 2048, 16 heads of 128, q/k/v biases, rope_theta 1e6, "
                              "vocabulary 151 936; random-init FP16 weights), test_batch_size 16 x beam 5 = 80 rows per decoder step, 230-embedding "
                              "prompts, 100 new tokens (min = max = 100), repetition/length penalty 2.0"),
    "decode_qwen1p8b_b1x5": (151936, 2048, 5504, 24, 16, 16, 230, 100, 5, 1,
                             "decode_qwen1p8b_b16x5 at batch 1 (5 rows per decoder step)"),
}
# per-workload extras: activation / weight dtype (default bf16) and constructor arguments beyond the widths
DECODE_EXTRA = {
    "decode_llama7b_128_fp16": dict(dtype="fp16"), "decode_llama7b_b6x3_fp16": dict(dtype="fp16"),
    "decode_qwen1p8b_b16x5": dict(dtype="fp16", ctor=dict(rope_theta=1000000.0, rms_norm_eps=1e-6, max_position_embeddings=32768)),
    "decode_qwen1p8b_b1x5": dict(dtype="fp16", ctor=dict(rope_theta=1000000.0, rms_norm_eps=1e-6, max_position_embeddings=32768)),
}


def _decode_dtype(workload):
    name = DECODE_EXTRA.get(workload, {}).get("dtype", "bf16")
    return name, (torch.float16 if name == "fp16" else torch.bfloat16)
DEFAULT_WORKLOAD = "arm_pretrain_large_1024"


def scan_bytes(B, D, L, N, G, elt, has_z=True):
    """elt*(4*B*D*L [u,delta,z,out] + 2*B*G*N*L [B,C]) + 4*(D*N + 2*D) [A,D,delta_bias]."""
    return elt * ((4 if has_z else 3) * B * D * L + 2 * B * G * N * L) + 4 * (D * N + 2 * D)


def make_scan_inputs(B, D, L, N, dtype, device, seed):
    """Reference test distribution (KSS/test_selective_scan.py:409-444)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = (-0.5 * torch.rand(D, N, generator=g)).to(device)
    mk = lambda *s: torch.randn(*s, generator=g).to(device=device, dtype=dtype)
    u, z, Bm, Cm = mk(B, D, L), mk(B, D, L), mk(B, 1, N, L), mk(B, 1, N, L)
    delta = (0.5 * torch.rand(B, D, L, generator=g)).to(device=device, dtype=dtype)
    Dv = torch.randn(D, generator=g).to(device)
    bias = (0.5 * torch.rand(D, generator=g)).to(device)
    return dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=Dv, z=z, delta_bias=bias)


def cpu_baseline_scan(B, D, L, N, budget_s=12.0):
    """Time the C oracle (selective_scan_ref restatement, fp32, OpenMP over (b,d) rows) on the host."""
    from oracle import oracle as orc
    # physical cores of one socket, the same count the training-step baseline uses (all 256 hardware threads of the pool's
    # 2-socket hosts oversubscribe the OpenMP team and made the number box-dependent)
    cores = min(host_physical_cores(), 64)
    orc.set_threads(cores)
    # bounded sample: shrink the batch until one call is ~<= 2 s, then repeat within the budget
    Bs = B
    x = make_scan_inputs(Bs, D, L, N, torch.float32, "cpu", seed=0)
    t0 = time.perf_counter()
    orc.selective_scan_ref(x["u"][:1], x["delta"][:1], x["A"], x["B"][:1], x["C"][:1], x["D"], x["z"][:1],
                           x["delta_bias"], True)
    one = time.perf_counter() - t0
    Bs = max(1, min(B, int(2.0 / max(one, 1e-6))))
    xs = {k: (v[:Bs].contiguous() if k in ("u", "delta", "B", "C", "z") else v) for k, v in x.items()}
    reps, elapsed = 0, 0.0
    orc.selective_scan_ref(xs["u"], xs["delta"], xs["A"], xs["B"], xs["C"], xs["D"], xs["z"], xs["delta_bias"], True)
    while elapsed < budget_s and reps < 200:
        t0 = time.perf_counter()
        orc.selective_scan_ref(xs["u"], xs["delta"], xs["A"], xs["B"], xs["C"], xs["D"], xs["z"],
                               xs["delta_bias"], True)
        elapsed += time.perf_counter() - t0
        reps += 1
    per_image = elapsed / (reps * Bs)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": 1.0 / per_image, "unit": "images/sec", "cores": cores, "kind": "port", "cpu": cpu_model,
        "sample": f"oracle/mxvl_oracle.c orc_scan_fwd (restatement of selective_scan_ref), {reps} x batch {Bs} "
                  f"of the same (D={D}, L={L}, N={N}) fp32 scan, OpenMP over rows, {elapsed:.1f} s of CPU work",
        "cpu": cpu_model,
    }


def _decode_kernel_text(layers, heads, B, beams, fused_norm):
    """What one decode step launches, from the stepper's actual mode (the old string claimed the fused norm AND the norm launches)."""
    rows = B * beams
    proj = ("decode_gemm_dma_kernel (16x16x32 MFMA, LDS-DMA weight stream, the waves split K)" if rows <= 16 else
            "decode_gemm_wide_kernel (16x16x32 MFMA, LDS-DMA weight ring, the waves split N and share the activation tile through LDS)")
    if fused_norm:
        norm = ("; RMSNorm fused into the consuming projection (the gain on the activation fragments, rstd in the epilogue), o_proj / down_proj add "
                "their residual in their own epilogue: no decode_rmsnorm_kernel launches, no K-split planes")
    else:
        norm = f"; o_proj / down_proj K-split into fp32 planes, folded by {2 * layers + 1} decode_rmsnorm_kernel launches"
    attn = ("decode_attn_kernel (a workgroup per (head, row))" if heads * B < 128 else
            "decode_attn_beams_mfma_kernel (a workgroup per (head, sample), both products on MFMA)")
    return (f"decode step = {4 * layers + 1} {proj}{norm} + {layers} {attn} launches + beam_step_kernel, one hipGraph replay per token; "
            "the time per token includes the prompt prefill's share")


def measure_decode(workload, steps, warmup, rank, world, dev, dist):
    """Autoregressive report decoding: one 'step' = one full generate() of `new_tokens` tokens per sample.
    Replicas only across GPUs (the reference decodes on a single device, MambaXrayVL_DownStream.py:407).
    Returns the result dict on rank 0 (None elsewhere)."""
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    vocab, hidden, inter, layers, heads, kvh, plen, new, beams, B, desc = DECODE_WORKLOADS[workload]
    dt_name, dt = _decode_dtype(workload)
    ctor = DECODE_EXTRA.get(workload, {}).get("ctor", {})
    torch.manual_seed(0)
    with torch.device(dev):
        m = ReportDecoder(vocab, hidden, inter, layers, heads, kvh, **ctor).to(dt).eval()
    if ctor:      # Qwen2: the q / k / v biases are real parameters (nn.Linear leaves them at its own init; make them count)
        with torch.no_grad():
            for layer in m.model.layers:
                for proj in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj):
                    proj.bias.normal_(0.0, 0.1)
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    emb = (0.02 * torch.randn(B, plen, hidden, generator=g)).to(dev, dt)
    kw = dict(num_beams=beams, min_new_tokens=new, max_new_tokens=new, repetition_penalty=2.0, length_penalty=2.0,
              eos_token_id=2, pad_token_id=0)
    for _ in range(warmup):
        m.generate(emb, **kw)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = m.generate(emb, **kw)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    stepper = type(next(iter(m._steppers.values()))).__name__ if getattr(m, "_steppers", None) else "eager"
    fused_norm = bool(getattr(next(iter(m._steppers.values())), "fused_norm", False)) if getattr(m, "_steppers", None) else False
    n_params = sum(p.numel() for p in m.parameters())
    n_out = int(out.shape[1])
    del m
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    tokens = B * world * steps * n_out
    n_params -= vocab * hidden                              # the embedding table is looked up (rows x hidden), not streamed
    wbytes = 2 * n_params                                   # every decode step streams the 16-bit weights once
    step_s = wall / (steps * n_out)                         # per generated token (beam batch of `beams` rows)
    achieved = wbytes / step_s / 1e9
    # the KV cache a step has to read on top of the weights: the prompt's keys / values once per SAMPLE (the beams share them), the
    # generated positions once per ROW, averaged over the n_out steps -- 1.6 % of the weight bytes at batch 1 x 3, a third at 16 x 5
    kv_pos = B * plen + B * beams * (n_out / 2.0)
    kv_bytes = kv_pos * layers * 2 * kvh * (hidden // heads) * 2
    achieved_kv = (wbytes + kv_bytes) / step_s / 1e9
    return {
        "metric": "report-generation decode tokens/sec (returned tokens; each step advances all beams)",
        "value": tokens / wall, "unit": "tokens/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dt_name, "data": "synthetic prompt embeddings (seed 1000+rank), random-init weights (seed 0)",
        "config": {"workload": f"{workload}: {desc}", "streamed_params": n_params, "batch": B, "num_beams": beams,
                   "new_tokens": n_out, "parallelism": f"replicas x{world} (no collective)", "stepper": stepper},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": _decode_kernel_text(layers, heads, B, beams, fused_norm),
                     "algorithmic_bytes_per_launch": wbytes, "kernel_ms": step_s * 1e3,
                     "kv_cache_bytes_per_token": int(kv_bytes), "frac_with_kv_cache": achieved_kv / HBM_PEAK_GBS}}


def cpu_baseline_decode(workload):
    """Reported baseline of the decode lines (the HF-generate CPU leg): the SAME decoder arithmetic on the host cores
    -- this package's torch module path (ReportDecoder.forward + the torch restatement of HF beam search), bf16 weights, which is
    what `generate` runs on CPU tensors; HF itself is not on the GPU box, so kind "port".  A Llama-2-7B on the CPU takes minutes to
    build and seconds per token: the sample is a 2-layer and a 6-layer decoder of the same widths (embedding + lm_head full size),
    a 16-embedding prompt and 2 vs 12 new tokens (the less disturbed of two repeats); per-token time = fixed part + 32 x
    per-layer part, both from those timings."""
    from medical_image_analysis_amd.report_decoder import ReportDecoder
    vocab, hidden, inter, layers, heads, kvh, plen, new, beams, B, desc = DECODE_WORKLOADS[workload]
    cores = min(host_physical_cores(), 64)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)
    dt_name, dt = _decode_dtype(workload)
    dt = torch.bfloat16             # (the host's fp16 GEMMs are emulated: the CPU leg runs the same decoder in bf16 whatever the GPU line's dtype)
    ctor = DECODE_EXTRA.get(workload, {}).get("ctor", {})
    emb = (0.02 * torch.randn(B, 16, hidden, generator=g)).to(dt)
    per_tok = {}
    t_begin = time.perf_counter()
    lo, hi = 2, 6
    for nl in (lo, hi):
        torch.manual_seed(0)
        m = ReportDecoder(vocab, hidden, inter, nl, heads, kvh, **ctor).to(dt).eval()

        def timed(n_new):
            kw = dict(num_beams=beams, min_new_tokens=n_new, max_new_tokens=n_new, repetition_penalty=2.0, length_penalty=2.0,
                      eos_token_id=2, pad_token_id=0)
            t0 = time.perf_counter()
            m.generate(emb, **kw)
            return time.perf_counter() - t0

        timed(2)                              # warms the allocator / thread pools up
        best, t_model = None, time.perf_counter()
        for rep in range(2):                  # the difference of two host timings, extrapolated 16x: keep the less disturbed repeat
            d = (timed(12) - timed(2)) / 10.0
            best = d if best is None else min(best, d)
            if time.perf_counter() - t_model > 8.0:
                break
        per_tok[nl] = best
        del m
    per_layer = max((per_tok[hi] - per_tok[lo]) / (hi - lo), 0.0)
    fixed = max(per_tok[lo] - lo * per_layer, 0.0)
    step_s = fixed + layers * per_layer
    return {"value": B / step_s, "unit": "tokens/sec", "cores": cores, "kind": "port",
            "sample": f"torch CPU path of the same decoder (bf16): {B} x beam {beams}, {lo}- and {hi}-layer models of the workload's widths, "
                      f"per-token time {per_tok[lo] * 1e3:.0f} / {per_tok[hi] * 1e3:.0f} ms -> {fixed * 1e3:.0f} ms + {layers} x {per_layer * 1e3:.1f} ms "
                      f"= {step_s * 1e3:.0f} ms per token for the {layers}-layer model; {time.perf_counter() - t_begin:.0f} s of CPU work"}


def run_decode(args, rank, world, dev, dist):
    steps = args.steps if args.steps > 0 else 3
    warmup = args.warmup if args.warmup >= 0 else 1
    res = measure_decode(args.workload, steps, warmup, rank, world, dev, dist)
    if res is not None:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_decode(args.workload)
        print(json.dumps(res))


def host_physical_cores():
    """Distinct (socket, core) pairs of /proc/cpuinfo; falls back to half the hardware threads."""
    pairs, phys = set(), None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    pairs.add((phys, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return len(pairs) or max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline_pretrain(sd, img, patch, depth, trainable, budget_s=20.0):
    """The SAME unit of work as the GPU value -- forward + backward + grad-clip + AdamW of the same model on one image -- on
    the host: oracle/models_ref.py (functional restatement of the reference's VisionMamba.forward over the C scan / conv
    oracles and their C gradients, wrapped as autograd functions in oracle/oracle.py), fp32, one socket's cores.  The
    reference's own training step cannot run without its CUDA wheels, so this is kind "port"."""
    from oracle import models_ref
    from oracle import oracle as orc
    # one socket's worth of physical cores: with every hardware thread of a 2-socket host (256 on the pool's EPYC 9575F
    # boxes) torch's intra-op pools oversubscribe and the forward ran 5x slower than on 8 cores
    cores = min(host_physical_cores(), 64)
    orc.set_threads(cores)
    torch.set_num_threads(cores)
    trainable = set(trainable)
    params = {k: (v.clone().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    opt = torch.optim.AdamW([v for k, v in params.items() if k in trainable], lr=1e-4, weight_decay=0.05)
    x = torch.randn(1, 3, img, img, generator=torch.Generator().manual_seed(0))
    n, elapsed, fwd_s = 0, 0.0, 0.0
    while elapsed < budget_s and n < 8:
        t0 = time.perf_counter()
        loss, _, _ = models_ref.visionmamba_forward_ref(params, x, patch=patch, depth=depth)
        t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss.mean().backward()
        torch.nn.utils.clip_grad_norm_([v for k, v in params.items() if k in trainable], 3.0)
        opt.step()
        elapsed += time.perf_counter() - t0
        fwd_s += t1 - t0
        n += 1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n / elapsed, "unit": "images/sec", "cores": cores, "kind": "port", "cpu": cpu_model,
            "sample": f"{n} x (forward + backward + grad-clip + AdamW) of the same model on one {img}x{img} image, fp32 "
                      f"(oracle/models_ref.py over oracle/mxvl_oracle.c and its C gradients, torch intra-op + OpenMP threads = "
                      f"{cores}); {elapsed:.1f} s of CPU work, of which forward {fwd_s:.1f} s"}


def _cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_mae(budget_s=15.0):
    """BASELINE configs[0]'s route -- "HD_Xray_Pretrain_MAE ... on CPU reference path": the MAE mirror's own torch-CPU arithmetic (the
    restatement of pretrain/models/mae.py over finetune/DP/models/vit.py's Block that tests/test_mae_vitb_cfg1.py pins to the
    reference's golden), the same mae_vit_large_patch16 on one 1280 x 1280 image: forward + backward + AdamW in fp32, host cores."""
    from medical_image_analysis_amd.mae import mae_vit_large_patch16
    cores = min(host_physical_cores(), 64)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = mae_vit_large_patch16()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=0.05)
    x = torch.randn(1, 1, 1280, 1280, generator=torch.Generator().manual_seed(0))
    n, elapsed = 0, 0.0
    while elapsed < budget_s and n < 16:
        t0 = time.perf_counter()
        loss, mask = net(x, 1, 0.85, 0.95)
        opt.zero_grad(set_to_none=True)
        ((loss * mask).sum() / mask.sum()).backward()
        opt.step()
        elapsed += time.perf_counter() - t0
        n += 1
    return {"value": n / elapsed, "unit": "images/sec", "cores": cores, "kind": "port", "cpu": _cpu_model_name(),
            "sample": f"{n} x (forward + backward + AdamW) of the same mae_vit_large_patch16 on one 1280x1280 image, fp32, the mirror's "
                      f"torch-CPU arithmetic (no GradScaler: fp32), torch intra-op threads = {cores}; {elapsed:.1f} s of CPU work"}


def cpu_baseline_vssm(sd, depths, budget_s=15.0):
    """VMamba-base forward on the host: oracle/models_ref.vssm_forward_ref over the C scan oracle, one 224 x 224 image, fp32.
    FORWARD ONLY (the model-level oracle has no SS2D backward): the unit is images/sec of the encoder forward, said so in `sample`."""
    from oracle import models_ref
    from oracle import oracle as orc
    cores = min(host_physical_cores(), 64)
    orc.set_threads(cores)
    torch.set_num_threads(cores)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    n, elapsed = 0, 0.0
    with torch.no_grad():
        while elapsed < budget_s and n < 16:
            t0 = time.perf_counter()
            models_ref.vssm_forward_ref(sd, x, depths, "v3noz", global_features=True)
            elapsed += time.perf_counter() - t0
            n += 1
    return {"value": n / elapsed, "unit": "images/sec (forward only)", "cores": cores, "kind": "port", "cpu": _cpu_model_name(),
            "sample": f"{n} x FORWARD of the same vssm1_base_0229 on one 224x224 image, fp32 (oracle/models_ref.vssm_forward_ref over "
                      f"oracle/mxvl_oracle.c; the GPU value is a training step: forward + backward + AdamW), threads = {cores}; {elapsed:.1f} s of CPU work"}


def _graph_kw(args, world):
    if not getattr(args, "graph", False):
        return {}
    if world > 1:
        raise SystemExit("--graph: single-process steps only")
    return {"use_graph": True}


def _timed_steps(eng, batches, steps, warmup, ssi, dist):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize pairs.  Eager engine: the scan launches of the timed steps
    are bracketed by HIP events (ssi.KERNEL_TIMERS) for the roofline object.  Graph engine: the timed steps are graph launches (nothing in
    Python to bracket), the kernel events come from two more EAGER steps behind the timed region."""
    graph = getattr(eng, "use_graph", False)
    if graph:
        warmup = max(warmup, eng.graph_warmup + 2)          # the capture and the first replay belong to the warm-up
    for i in range(warmup):
        eng.step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    timers = []
    if not graph:
        ssi.KERNEL_TIMERS = timers
    t0 = time.perf_counter()
    for i in range(steps):
        loss = eng.step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ssi.KERNEL_TIMERS = None
    loss = loss.clone()
    if graph:
        ssi.KERNEL_TIMERS = timers
        for i in range(2):
            eng._eager_step(batches[i % 2])
        torch.cuda.synchronize()
        ssi.KERNEL_TIMERS = None
    eng.timer_steps = 2 if graph else steps               # how many steps the kernel events cover
    return loss, wall, timers, warmup


def run_pretrain(args, rank, world, dev, dist):
    """Stage-1 pre-training step: VisionMamba forward+backward+clip+AdamW, bf16 autocast, DDP gradient all-reduce."""
    import medical_image_analysis_amd.selective_scan_interface as ssi
    from medical_image_analysis_amd.models_pretrain import VisionMamba
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    img, patch, embed, depth, dec, B, desc = PRETRAIN_WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    north = None
    if args.workload == DEFAULT_WORKLOAD and not args.no_secondary:
        # north_star's one numeric kernel target (BASELINE.json: ">= 40 % of MI355X HBM roofline on the selective-scan kernel at L=4096,
        # D=1536"): the forward scan alone at B8 x L4096 x D1536 x N16 fp32, ~0.1 s, so that the driver's own line carries it.  Measured
        # FIRST, before the training leg has driven the chip to its power limit (a memory-bound kernel right behind it runs ~3 % below
        # its standalone rate until the clocks recover: profiles/r05_secondary_warmup.txt) -- the standalone `--workload scan_fwd_target`
        # line is the same measurement
        north = measure_scan("scan_fwd_target", 200, 20, rank, world, dev, dist, no_cpu_baseline=True)
    torch.manual_seed(0)  # identical random-init weights on every rank (DDP broadcasts rank 0's anyway)
    model = VisionMamba(img_size=img, patch_size=patch, stride=patch, embed_dim=embed, depth=depth, dec_embed_dim=dec,
                        rms_norm=True, residual_in_fp32=True, fused_add_norm=True, if_abs_pos_embed=True,
                        bimamba_type="None").to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    eng = PretrainEngine(model, device=dev, **_graph_kw(args, world))
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    batches = [torch.randn(B, 3, img, img, generator=g).to(dev) for _ in range(2)]
    steps = args.steps
    loss, wall, timers, warmup = _timed_steps(eng, batches, steps, args.warmup, ssi, dist)
    timer_steps = eng.timer_steps
    if dist is not None:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    final_loss = float(loss)
    cpu_sd = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        cpu_trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    secondary = None
    if args.workload == DEFAULT_WORKLOAD and not args.no_secondary:
        # second half of BASELINE.json's metric ("MAE pretrain images/sec + report-gen decode tokens/sec"): the report
        # decoder of configs[3], measured by the same process right after the training steps (replicas on every rank)
        del eng, model, batches
        torch.cuda.empty_cache()
        secondary = measure_decode("decode_llama7b_128", 5, args.secondary_warmup, rank, world, dev, dist)
        if secondary is not None and world == 1 and not args.no_cpu_baseline:
            secondary["cpu_baseline"] = cpu_baseline_decode("decode_llama7b_128")
    if rank != 0:
        return
    stats = _kernel_stats(timers)
    mfma_obj = _mfma_kernel_object(stats, timer_steps, wall / steps * 1e3)
    # dominant hand-written kernel of the step = the timed kernel with the largest total time.  At this shape the weight-gradient MFMA
    # kernel (priced against the matrix cores) and the scan backward (priced against HBM, as the contract asks) are within a few
    # per cent of each other: whichever leads on this box is `roofline`, the other one is reported beside it (`mfma_kernel` / `hbm_kernel`)
    dominant = max(stats, key=lambda k: stats[k][0])
    kind = max((k for k in stats if k not in MFMA_KINDS), key=lambda k: stats[k][0])
    tot_ms, calls, tot_bytes = stats[kind]
    achieved = tot_bytes / (tot_ms * 1e-3) / 1e9
    L = (img // patch) ** 2 - 16
    out = {
        "metric": "pre-training images/sec (forward + backward + grad-clip + AdamW)",
        "value": B * world * steps / wall, "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic N(0,1) images (seed 1000+rank), random-init weights (seed 0)",
        "config": {"workload": f"{args.workload}: {desc}", "per_gpu_batch": B, "global_batch": B * world,
                   "seq_len": L, "params": n_params, "parallelism": _dp_label(world),
                   "final_loss": final_loss,
                   "timed_step": "forward + backward (DDP bucketed grad all-reduce overlapped) + loss all-reduce + clip + fused AdamW",
                   "launch": "one hipGraph launch per step (PretrainEngine use_graph)" if getattr(args, "graph", False) else "eager launches"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kind,
                     "kernel_ms": tot_ms / calls, "launches_timed": calls,
                     "algorithmic_bytes_per_launch": tot_bytes // calls,
                     "step_share": {k: round(v[0] / timer_steps / (wall / steps * 1e3), 4) for k, v in stats.items()},
                     "limited_by": "VALU issue rate of the fp32 recurrence (5 VALU + 1 v_exp per step and state), not HBM traffic (DESIGN.md 4.1 / 4.3); the HBM fraction is what the contract asks for"},
    }
    attach_traffic(out["roofline"], "scan_bwd_pretrain" if (kind == "scan_bwd" and args.workload == DEFAULT_WORKLOAD and B == 16)
                   else args.workload)
    if secondary is not None:
        out["secondary"] = {k: secondary[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                      "dtype", "config", "roofline", "cpu_baseline") if k in secondary}
    if mfma_obj is not None and args.workload == DEFAULT_WORKLOAD and B == 16:
        # the weight-gradient kernel's HBM bytes per launch, averaged over its launches of THIS step (PMC passes of the step itself)
        attach_traffic(mfma_obj, "gemm_tn_pretrain", "bench.py --workload arm_pretrain_large_1024 --steps 2 --warmup 1")
        mfma_obj["algorithmic_bytes_per_launch"] = ("2 K (M + N) + 4 M N: 874 / 504 MB for the two SwiGLU gradients (5504 x 1024, 1024 x 2752 at K = 65 280), "
                                                    "610 MB averaged over the step's 60 launches; the counters also see the split-K atomics "
                                                    "(8-16 partial tiles of 4 M N bytes added into the output, fetched and written by the memory-side atomic unit)")
    if mfma_obj is not None and dominant in MFMA_KINDS:
        hbm_obj = out["roofline"]
        out["roofline"] = dict(mfma_obj, step_share=hbm_obj["step_share"],
                               limited_by="matrix-core issue behind LDS transpose reads and barriers: MFMA busy 0.41 of the peak-clock cycles "
                                          "(profiles/r06_wgrad_tn_bench.txt); the same structure as the guide's best plain-HIP GEMMs (0.53-0.62)")
        out["hbm_kernel"] = hbm_obj
    elif mfma_obj is not None:
        out["mfma_kernel"] = mfma_obj
    if north is not None:
        out["north_star_kernel"] = {k: north[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline")}
    if cpu_sd is not None:
        out["cpu_baseline"] = cpu_baseline_pretrain(cpu_sd, img, patch, depth, cpu_trainable)
    print(json.dumps(out))


class _SyntheticTokenizer:
    """Whitespace tokenizer with the HF call surface the models use (ids: 0 pad, 1 bos, 2 eos = '</s>', words hashed into the
    vocabulary): the reference's tokenizer files are not in this image and no text statistic enters a timed number."""
    pad_token_id, bos_token_id, eos_token_id, cls_token_id, padding_side = 0, 1, 2, 1, "right"

    class _Toks(dict):
        __getattr__ = dict.__getitem__

        def to(self, device):
            return type(self)({k: v.to(device) for k, v in self.items()})

    def __init__(self, vocab=32000):
        self.vocab = vocab

    def _ids(self, text):
        return [2 if w == "</s>" else 3 + (sum(map(ord, w)) * 31 + len(w)) % (self.vocab - 3) for w in text.replace("</s>", " </s>").split()]

    def __call__(self, text, return_tensors="pt", padding=False, truncation=False, max_length=None, add_special_tokens=False):
        rows = [self._ids(t) for t in ([text] if isinstance(text, str) else text)]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (width - len(r)) for r in rows])
        mask = torch.tensor([[1] * len(r) + [0] * (width - len(r)) for r in rows])
        return self._Toks(input_ids=ids, attention_mask=mask)


def run_finetune(args, rank, world, dev, dist):
    """The training step of the report-generation stages, whole pipeline (encoder forward + backward THROUGH a frozen 7B LLM):
    CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:195-241 (training_step -> forward -> loss) with
    configure_optimizers :425-428, and R2GenCSR/models/R2GenCSR.py:309-474 + its training_step.  Data-parallel replicas + DDP on the
    trainable parameters (the reference: Lightning `--strategy deepspeed` stage 2 / ddp on one device)."""
    from medical_image_analysis_amd import mambaxray_vl as mx
    kind, B, desc = FINETUNE_WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    torch.manual_seed(0)
    tok = _SyntheticTokenizer(32000)
    with torch.device(dev):
        llm = mx.build_report_decoder("llama2-7b")                 # fp16, the reference's torch_dtype
    words = ["heart", "size", "is", "normal", "lungs", "are", "clear", "no", "acute", "cardiopulmonary", "process", "pleural", "effusion",
             "pneumothorax", "seen", "mild", "opacity", "left", "right", "lower", "lobe", "stable", "unchanged", "since", "prior", "."]
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    texts = [" ".join(words[int(i)] for i in torch.randint(0, len(words), (int(n),), generator=g))
             for n in torch.randint(40, 99, (B,), generator=g)]
    if kind == "mambaxray":
        a = mx.default_args(vision_model="Large-None", type="large", freeze_vm=False, max_length=100)
        model = mx.MambaXrayVLDownStream(a, tokenizer=tok, llm=llm).to(dev)
        n_img_tokens = 197
    else:
        from medical_image_analysis_amd.r2gencsr import R2GenCSR
        a = mx.default_args(vision_model="None", freeze_vm=False, max_length=100, context_pair=3, chosen="vmamba", proj="linear", llm="llama2",
                            positive="Note: <Img><ImageHere></Img> with desease. ", negative="Note: <Img><ImageHere></Img> normal. ",
                            use_feature_mean=True)
        model = R2GenCSR(a, tokenizer=tok, llm=llm).to(dev)
        model.set_context_samples(torch.randn(3, 3, 224, 224, generator=g).to(dev), torch.randn(3, 3, 224, 224, generator=g).to(dev))
        n_img_tokens = 49
    model.llama_model.to(torch.float16)                          # (.to(dev) above keeps dtypes; explicit for the record)
    trainable = [p for p in model.parameters() if p.requires_grad]
    n_train = sum(p.numel() for p in trainable)
    n_llm = sum(p.numel() for p in model.llama_model.parameters())
    net = model
    if world > 1:
        from medical_image_analysis_amd.pretrain_engine import wrap_ddp
        net = wrap_ddp(model, dev)        # trainable parameters only; find_unused_parameters from the model's own flag
    opt = torch.optim.AdamW(trainable, lr=1e-4, fused=True)
    batches = [{"id": [f"s{i}" for i in range(B)], "image": [torch.randn(B, 3, 224, 224, generator=g).to(dev)], "input_text": texts} for _ in range(2)]

    def step(batch):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = net(batch)["loss"]
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    steps, warmup = (args.steps if args.steps > 0 else 6), (args.warmup if args.warmup >= 0 else 2)
    for i in range(warmup):
        step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    if rank != 0:
        return
    with torch.no_grad():
        seq = model._prefix(batches[0])[0].shape[1] + 100
    # model flops of a step: the frozen LLM's forward + activation-gradient backward (2 + 2 flops per weight and token, no weight
    # gradients), the trainable side's forward + both backward products (6 per weight and token; VMamba: 49 tokens is the LAST stage --
    # its flops are counted from the encoder's own per-image figure instead)
    T_llm = B * seq
    enc_params = n_train
    enc_tokens = B * n_img_tokens
    flops = 4.0 * (n_llm - 32000 * 4096) * T_llm + (6.0 * enc_params * enc_tokens if kind == "mambaxray" else 15.4e9 * (3 * B + 6))   # VMamba-base: 15.4 GFLOP per 224 x 224 forward (its paper's figure), x 3 training; 6 context images forward only
    achieved = flops / (wall / steps) / 1e12
    print(json.dumps({
        "metric": "report-generation fine-tuning studies/sec (encoder + projector training step through a frozen LLM: forward + backward + AdamW)",
        "value": B * world * steps / wall, "unit": "studies/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 autocast over fp16 LLM weights",
        "data": "synthetic N(0,1) images and synthetic report text (seed 1000+rank), random-init weights (seed 0)",
        "config": {"workload": f"{args.workload}: {desc}", "per_gpu_batch": B, "global_batch": B * world, "llm_sequence": seq,
                   "trainable_params": n_train, "frozen_llm_params": n_llm, "parallelism": _dp_label(world), "final_loss": float(loss)},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                     "traffic": None, "kernel": "whole step (model flops: 4 x frozen-LLM weights x LLM tokens + 6 x trainable weights x image tokens); "
                     "the LLM's projections are library GEMMs under autocast, its attention csrc/attn.hip, the encoder the scan / conv / add+LN kernels",
                     "model_flops_per_step": flops}}))


def run_mae(args, rank, world, dev, dist):
    """ViT-MAE pre-training step (HD_Xray_Pretrain_MAE/pretrain/main.py:319-323: loss = sum(loss*mask)/sum(mask)).  The
    transformer blocks are library GEMMs + the MFMA flash-attention kernels (csrc/attn.hip); masking gather, mask-token
    un-shuffle and the per-patch loss are HIP kernels (csrc/mae_ops.hip) -- the roofline object reports the model-level MFMA
    rate (analytic flops of the visible tokens / step time against the 2.5 PFLOP/s dense bf16 peak)."""
    import torch.nn as nn
    from medical_image_analysis_amd.mae import mae_vit_large_patch16
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    B, desc = MAE_WORKLOADS[args.workload]
    if args.batch:
        B = args.batch

    class MaeLoss(nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = net
            self.kept = 0

        def forward(self, imgs):
            loss, mask = self.net(imgs, 1, 0.85, 0.95)
            self.kept = int(mask.shape[1] - mask[0].sum())
            return ((loss * mask).sum() / mask.sum()).reshape(1)

    torch.manual_seed(0)
    model = MaeLoss(mae_vit_large_patch16()).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    eng = PretrainEngine(model, device=dev, amp_dtype=torch.float16)     # autocast() + GradScaler, as main.py:211-213,317
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    batches = [torch.randn(B, 1, 1280, 1280, generator=g).to(dev) for _ in range(2)]
    steps, warmup = args.steps, args.warmup
    for i in range(warmup):
        eng.step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = eng.step(batches[i % 2])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    if rank != 0:
        return
    net = model.net
    enc = sum(p.numel() for n, p in net.named_parameters() if n.startswith("blocks."))
    dec = sum(p.numel() for n, p in net.named_parameters() if n.startswith("decoder_blocks.") or n.startswith("decoder_pred") or n.startswith("decoder_embed"))
    kept, L = model.kept + 1, 401
    flops = 6.0 * B * (enc * kept + dec * L) + 12.0 * B * (24 * kept * kept * 1024 + 8 * L * L * 512)   # GEMMs + attention, fwd+bwd
    step_s = wall / steps
    cpu_b = None
    if world == 1 and not args.no_cpu_baseline:
        del eng, batches
        torch.cuda.empty_cache()
        cpu_b = cpu_baseline_mae()
    print(json.dumps({
        **({"cpu_baseline": cpu_b} if cpu_b is not None else {}),
        "metric": "pre-training images/sec (forward + backward + grad-clip + AdamW)", "value": B * world * steps / wall,
        "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": step_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16",
        "data": "synthetic N(0,1) 1-channel images (seed 1000+rank), random-init weights (seed 0)",
        "config": {"workload": f"{args.workload}: {desc}", "per_gpu_batch": B, "global_batch": B * world, "visible_tokens": kept,
                   "params": n_params, "parallelism": _dp_label(world),
                   "final_loss": float(loss)},
        "roofline": {"bound": "mfma", "achieved": flops / step_s / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": flops / step_s / 1e12 / 2500.0, "traffic": None,
                     "kernel": "whole step (library GEMMs + mxvl flash attention + mxvl add+LayerNorm / MAE index / loss kernels; analytic flops of the visible tokens; fp16 autocast + GradScaler)"}}))


def run_vmamba(args, rank, world, dev, dist):
    """R2GenCSR's VMamba encoder (R2GenCSR/models/R2GenCSR.py:75-100 builds it; :233 calls it with global_features)."""
    import torch.nn as nn
    import medical_image_analysis_amd.selective_scan_interface as ssi
    from medical_image_analysis_amd.vmamba import vssm1_base_0229
    from medical_image_analysis_amd.pretrain_engine import PretrainEngine
    B, desc = VMAMBA_WORKLOADS[args.workload]
    if args.batch:
        B = args.batch

    class PooledLoss(nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, imgs):
            return self.net(imgs, global_features=True).float().square().mean(-1)

    torch.manual_seed(0)
    if args.workload == "arm_encoder_large_224":
        from medical_image_analysis_amd.models_mamba import arm_large_pz16

        class FeatLoss(nn.Module):
            def __init__(self, net):
                super().__init__()
                self.net = net

            def forward(self, imgs):
                return self.net(imgs).float().square().mean((1, 2))

        model = FeatLoss(arm_large_pz16("large", drop_path_rate=0.0)).to(dev)
    else:
        model = PooledLoss(vssm1_base_0229(drop_path_rate=0.0)).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    eng = PretrainEngine(model, device=dev, **_graph_kw(args, world))
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    batches = [torch.randn(B, 3, 224, 224, generator=g).to(dev) for _ in range(2)]
    steps = args.steps
    loss, wall, timers, warmup = _timed_steps(eng, batches, steps, args.warmup, ssi, dist)
    timer_steps = eng.timer_steps
    if dist is not None:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    if rank != 0:
        return
    stats = _kernel_stats(timers)
    mfma_obj = _mfma_kernel_object(stats, timer_steps, wall / steps * 1e3)
    kind = max((k for k in stats if k not in MFMA_KINDS), key=lambda k: stats[k][0])
    tot_ms, calls, tot_bytes = stats[kind]
    achieved = tot_bytes / (tot_ms * 1e-3) / 1e9
    cpu_b = None
    if world == 1 and not args.no_cpu_baseline and args.workload == "vmamba_base_224":
        sd = {k: v.detach().float().cpu() for k, v in model.net.state_dict().items()}
        cpu_b = cpu_baseline_vssm(sd, [2, 2, 15, 2])
    print(json.dumps({
        **({"cpu_baseline": cpu_b} if cpu_b is not None else {}),
        **({"mfma_kernel": mfma_obj} if mfma_obj is not None else {}),
        "metric": "encoder training images/sec (forward + backward + grad-clip + AdamW)", "value": B * world * steps / wall,
        "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": wall / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic N(0,1) images (seed 1000+rank), random-init weights (seed 0)",
        "config": {"workload": f"{args.workload}: {desc}", "per_gpu_batch": B, "global_batch": B * world, "params": n_params,
                   "parallelism": _dp_label(world), "final_loss": float(loss.mean()),
                   "launch": "one hipGraph launch per step (PretrainEngine use_graph)" if getattr(args, "graph", False) else "eager launches"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": kind + (" (4 directions stacked: one launch, n_groups 4, L = 197)" if args.workload.startswith("arm_")
                                                         else " (all SS2D stages: L = 3136 / 784 / 196 / 49, 4 direction groups, d_state 1)"),
                     "kernel_ms": tot_ms / calls, "launches_timed": calls, "algorithmic_bytes_per_launch": tot_bytes // calls,
                     "step_share": {k: round(v[0] / timer_steps / (wall / steps * 1e3), 4) for k, v in stats.items()}}}))


def measure_scan(workload, steps, warmup, rank, world, dev, dist, no_cpu_baseline=False, one_gpu=False):
    """One selective-scan micro-benchmark line (the kernel alone, inputs resident in HBM); returns the dict on rank 0."""
    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd.selective_scan_interface import scan_fwd_raw
    B, D, L, N, dtname, desc = WORKLOADS[workload]
    dtype = getattr(torch, dtname)
    x = make_scan_inputs(B, D, L, N, dtype, dev, seed=rank)
    step = lambda: scan_fwd_raw(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], x["z"], x["delta_bias"], True)
    backward = workload.startswith("scan_bwd")
    if backward:
        from medical_image_analysis_amd.selective_scan_interface import scan_algorithmic_bytes, scan_bwd_raw
        _, _, ckpt = scan_fwd_raw(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], x["z"], x["delta_bias"], True, want_ckpt=True)
        dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(7 + rank)).to(dev, dtype)
        step = lambda: scan_bwd_raw(x["u"], x["delta"], x["A"], x["B"], x["C"], x["D"], x["z"], x["delta_bias"], True, ckpt, dout)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / steps  # mean launch-to-launch duration of the scan kernel
    if dist is not None:
        t = torch.tensor([wall, kern_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(t[0]), float(t[1])

    if rank == 0:
        images = B * world * steps
        elt = torch.empty((), dtype=dtype).element_size()
        nbytes = scan_bytes(B, D, L, N, 1, elt)
        if backward:
            nbytes = scan_algorithmic_bytes(B, D, L, N, 1, elt, True, True, ckpt.shape[2])
        achieved = nbytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "images/sec (one image = one (D x L) patch-token sequence through the selective scan)",
            "value": images / wall, "unit": "images/sec", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": wall / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"float32": "f32", "bfloat16": "bf16"}[dtname],
            "data": "synthetic (reference test distribution, seed = rank), inputs resident in HBM",
            "config": {"workload": f"{workload}: {desc}", "per_gpu_batch": B, "seq_len": L, "d_inner": D,
                       "d_state": N, "parallelism": f"dp{world} (independent batch shards, no collective)" + (
                           " -- DEV CHECK: all ranks on cuda:0, gloo barriers, not a scaling number" if one_gpu and world > 1 else ""),
                       "kernel": _abi.load().mxvl_last_scan_kernel().decode()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": ("scan_bwd_kernel + zero-fill of the fp32 accumulators (one mxvl_scan_bwd call)"
                                    if backward else "scan_fwd_stream_kernel"),
                         "algorithmic_bytes_per_launch": nbytes, "kernel_ms": kern_ms,
                         "limited_by": "VALU issue rate of the fp32 recurrence (5 VALU + 1 v_exp per step and state), not HBM traffic (DESIGN.md 4.1 / 4.3); the HBM fraction is what the contract asks for"},
        }
        attach_traffic(out["roofline"], workload)
        if world == 1 and not no_cpu_baseline and not backward:
            out["cpu_baseline"] = cpu_baseline_scan(B, D, L, N)
        return out
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = workload default (20 training steps / 200 kernel launches)")
    ap.add_argument("--warmup", type=int, default=-1, help="-1 = workload default (3 / 20)")
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD,
                    choices=sorted(WORKLOADS) + sorted(PRETRAIN_WORKLOADS) + sorted(DECODE_WORKLOADS) + sorted(MAE_WORKLOADS) + sorted(VMAMBA_WORKLOADS)
                    + sorted(FINETUNE_WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override for the pre-training workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the decode tokens/sec leg of the default workload")
    ap.add_argument("--graph", action="store_true", help="training-step workloads, one process: the whole step captured once into a hipGraph and "
                    "replayed (PretrainEngine(use_graph=True)); the launch-bound small-image steps gain, the 1024 x 1024 headline step is "
                    "GPU-bound either way -- off by default so that the N = 1 line is the same program as the N > 1 lines")
    ap.add_argument("--secondary-warmup", type=int, default=8,
                    help="untimed generate() calls of the decode leg: it starts on a chip the training leg has just driven at its power limit, and "
                         "the first ~3 s of (memory-bound) decoding run 3 %% below the standalone decode line until the clocks recover -- "
                         "profiles/r05_secondary_warmup.txt: 350.8 tok/s with 1 warm-up call, 354.2 with 3, 361.2 after 3 s idle, 359.3 standalone")
    ap.add_argument("--mlp-bwd", choices=["fused", "unfused"], default=None,
                    help="A/B switch of the training steps: SwiGLU backward inside w3's dgrad GEMM (fused: the default for fp16 autocast) or the round-4 "
                         "two-kernel backward (unfused: the default for bf16 autocast, the reference's training dtype)")
    ap.add_argument("--llm-shadows", choices=["on", "off"], default=None,
                    help="A/B switch of the fine-tuning steps: cached autocast-dtype copies of the frozen LLM's weights (default) or per-call casts")
    ap.add_argument("--llm-ops", choices=["on", "off"], default=None,
                    help="A/B switch of the fine-tuning steps: rotary embedding + RMSNorm of the frozen LLM's layers as csrc/llm_ops.hip kernels (default) "
                         "or as the torch expressions of hybrid_decoder_layer.py")
    ap.add_argument("--decode-gemm", choices=["wide", "ksplit", "wide_pf3", "wide_nw4", "wide_mt3", "wide1"], default=None,
                    help="A/B switch of the 17..80-row decode projections: waves split N + LDS-shared activations (wide, default), the round-4 K-split "
                         "kernels at every row count (ksplit), the wide kernel as first measured (wide_pf3: 33..80 rows, >= 160 workgroups, 3-stage ring, "
                         "four waves), four waves per workgroup everywhere (wide_nw4), 33..80 rows and >= 160 workgroups only (wide_mt3), from one row on (wide1)")
    ap.add_argument("--scan-variant", type=int, default=0,
                    help="A/B switch: mxvl_set_scan_variant (forward kernel choice in bits 0..7, backward in bits 8..15; 0 = automatic)")
    ap.add_argument("--decode-norm", choices=["fused", "split"], default=None,
                    help="A/B switch of the decode step: RMSNorm fused into the consuming projection (default) or the round-4 path "
                         "(K-split o_proj / down_proj folded by explicit norm launches)")
    args = ap.parse_args()
    if args.graph and (args.gpus > 1 or not (args.workload in PRETRAIN_WORKLOADS or args.workload in VMAMBA_WORKLOADS)):
        raise SystemExit("--graph: one process, a VisionMamba / ARM / VMamba training-step workload")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves, exactly the way the driver does
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    # MXVL_BENCH_ONE_GPU=1 (dev check of the multi-rank code path on a 1-GPU box): every rank on cuda:0, gloo collectives
    one_gpu = os.environ.get("MXVL_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)

    from medical_image_analysis_amd import _abi
    from medical_image_analysis_amd.selective_scan_interface import scan_fwd_raw

    if args.scan_variant:
        _abi.load().mxvl_set_scan_variant(args.scan_variant)
    if args.mlp_bwd:
        from medical_image_analysis_amd import fused_ops
        fused_ops._MlpSwiGLU.FUSED_BWD = args.mlp_bwd == "fused"
    if args.llm_shadows:
        from medical_image_analysis_amd.report_decoder import ReportDecoder
        ReportDecoder.autocast_shadows = args.llm_shadows == "on"
    if args.llm_ops:
        from medical_image_analysis_amd import fused_ops
        fused_ops.LLM_OPS = args.llm_ops == "on"
    if args.decode_gemm:
        _abi.load().mxvl_set_decode_gemm_wide({"wide": 1, "ksplit": 0, "wide_pf3": 3, "wide_nw4": 4, "wide_mt3": 5, "wide1": 6}[args.decode_gemm])
    if args.decode_norm:
        from medical_image_analysis_amd.report_decoder import _KernelStepper
        _KernelStepper.norm_mode = args.decode_norm
    if args.workload in FINETUNE_WORKLOADS:
        run_finetune(args, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.workload in DECODE_WORKLOADS:
        run_decode(args, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    pre = args.workload in PRETRAIN_WORKLOADS or args.workload in MAE_WORKLOADS or args.workload in VMAMBA_WORKLOADS
    if args.steps <= 0:
        args.steps = 20 if pre else 200
    if args.warmup < 0:
        args.warmup = 3 if pre else 20
    if pre:
        runner = run_mae if args.workload in MAE_WORKLOADS else run_vmamba if args.workload in VMAMBA_WORKLOADS else run_pretrain
        runner(args, rank, world, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return

    out = measure_scan(args.workload, args.steps, args.warmup, rank, world, dev, dist, args.no_cpu_baseline, one_gpu)
    if out is not None:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
