"""Dev tool: time the MFMA flash-attention kernels (fwd / dq / dkv) at the pre-training decoder's shape and the library path
beside them.  usage: python tools/attn_bench.py [B H L D] [--sdpa]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd import flash_attention as fa


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B, H, L, D = (int(x) for x in args[:4]) if len(args) >= 4 else (16, 8, 4080, 64)
    mask = args[4] if len(args) > 4 else "block_causal"
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, L, H, D, generator=g).to(dev, torch.bfloat16).transpose(1, 2)
    kv = torch.randn(B, L, 2, H, D, generator=g).to(dev, torch.bfloat16)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
    do = torch.randn(B, L, H, D, generator=g).to(dev, torch.bfloat16).transpose(1, 2)
    mm = fa.MASKS[mask]
    scale = D ** -0.5
    out, lse, saved = fa.attn_fwd_raw(q, k, v, scale, mm, 16)
    frac = {"none": 1.0, "causal": 0.5, "block_causal": 0.5}[mask]
    flops_f = 4.0 * B * H * L * L * D * frac
    t_f = timeit(lambda: fa.attn_fwd_raw(q, k, v, scale, mm, 16))
    t_b = timeit(lambda: fa.attn_bwd_raw(saved, out, lse, do, scale, mm, 16))
    print(f"mxvl attention B={B} H={H} L={L} D={D} {mask}: fwd {t_f:8.1f} us ({flops_f / t_f * 1e-6:6.1f} TFLOP/s)   "
          f"bwd {t_b:8.1f} us ({2.5 * flops_f / t_b * 1e-6:6.1f} TFLOP/s, 5 GEMMs of useful work)")


if __name__ == "__main__":
    main()
