"""mxvl_gemm_tn (csrc/gemm_tn.hip: the K-major x K-major weight-gradient MFMA kernel, split over the XCDs, atomic epilogue) against the
round-2..5 form of the same product (one batched library GEMM over token slices into fp32 planes + a sum over the planes,
selective_scan_interface.splitk_wgrad) at the wgrad shapes of the headline training step (ARM-large, 65 280 tokens per GPU).
    python tools/wgrad_tn_bench.py [--tuned] [--q]        (--q: also force 1..4 slices per XCD)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_analysis_amd import selective_scan_interface as ssi      # noqa: E402


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    if "--tuned" in sys.argv:
        from medical_image_analysis_amd.pretrain_engine import enable_tuned_gemms
        enable_tuned_gemms()
    dev = "cuda:0"
    K = 65280
    shapes = [("w1|w2", 5504, 1024), ("w3", 1024, 2752), ("in_proj", 4096, 1024), ("in_proj half", 2048, 1024), ("out_proj", 1024, 2048),
              ("square 1024", 1024, 1024), ("decoder fc1", 2048, 512), ("w1|w2 base", 4096, 768), ("K 32640 w1|w2", 5504, 1024)]
    dt = torch.float16 if "--fp16" in sys.argv else torch.bfloat16
    if "--mae" in sys.argv:        # mae_vit_large_1280 at per-GPU batch 256: 101 visible tokens per image, 401 in the decoder
        shapes = [("enc qkv", 3072, 1024, 25856), ("enc proj", 1024, 1024, 25856), ("enc fc1", 4096, 1024, 25856), ("enc fc2", 1024, 4096, 25856),
                  ("dec qkv", 1536, 512, 102656), ("dec proj", 512, 512, 102656), ("dec fc1", 2048, 512, 102656), ("dec fc2", 512, 2048, 102656),
                  ("K 16384 fc1", 4096, 1024, 16384), ("K 8192 fc1", 4096, 1024, 8192), ("K 131072 fc1", 4096, 1024, 131072)]
    g = torch.Generator().manual_seed(0)
    for shp in shapes:
        name, M, N = shp[:3]
        k = shp[3] if len(shp) > 3 else (32640 if name.startswith("K 32640") else K)
        a = torch.randn(k, M, generator=g).to(dev, dt)
        b = torch.randn(k, N, generator=g).to(dev, dt)
        ref = torch.matmul(a[:4096].double().t(), b[:4096].double())
        got = ssi.gemm_tn(a[:4096], b[:4096])
        err = float((got.double() - ref).abs().max()) / float(ref.abs().max())
        t_lib = timed(lambda: ssi.splitk_wgrad_library(a, b, torch.float32))
        t_me = timed(lambda: ssi.gemm_tn(a, b))
        fl = 2.0 * k * M * N
        line = (f"{name:14s} M={M:5d} N={N:5d} K={k}: library split-K + sum {t_lib:7.1f} us ({fl / t_lib / 1e6:5.0f} TF) | "
                f"mxvl_gemm_tn {t_me:7.1f} us ({fl / t_me / 1e6:5.0f} TF) x{t_lib / t_me:.2f}  rel.err(K=4096) {err:.1e}")
        if "--q" in sys.argv:
            line += "  q:" + " ".join(f"{q}={timed(lambda: ssi.gemm_tn(a, b, slices_per_xcd=q), n=10):.0f}" for q in (1, 2, -1, -2))
        print(line, flush=True)


if __name__ == "__main__":
    main()
