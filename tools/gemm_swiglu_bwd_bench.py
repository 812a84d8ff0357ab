"""A/B of the SwiGLU-backward-in-dgrad kernel (csrc/gemm_swiglu.hip MODE 1) against the two kernels it replaces (library dgrad GEMM +
mxvl_swiglu_bwd_colsum) at the ARM-large layer shape of the headline step: 65 280 tokens, K = 1024, hidden 2752.
    python tools/gemm_swiglu_bwd_bench.py [M K H]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_analysis_amd import _abi                      # noqa: E402
from medical_image_analysis_amd.fused_ops import gemm_swiglu_bwd_raw   # noqa: E402


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
    M, K, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (65280, 1024, 2752)
    dev = "cuda:0"
    lib = _abi.load()
    for dtype in (torch.bfloat16, torch.float16):
        g = torch.Generator().manual_seed(0)
        dy = (0.5 * torch.randn(M, K, generator=g)).to(dev, dtype)
        w3 = (K ** -0.5 * torch.randn(K, H, generator=g)).to(dev, dtype)          # nn.Linear(H, K).weight
        ab = torch.randn(M, 2 * H, generator=g).to(dev, dtype)
        w3t = w3.t().contiguous()
        dab = torch.empty_like(ab)
        n_part = lib.mxvl_swiglu_partials(M, H)
        partial = torch.empty((n_part, 2 * H), dtype=torch.float32, device=dev)

        def unfused():
            dh = torch.matmul(dy, w3)
            _abi.check(lib.mxvl_swiglu_bwd_colsum(ab.data_ptr(), dh.data_ptr(), dab.data_ptr(), partial.data_ptr(), n_part, M, H,
                                                  _abi.dtype_code(dtype), _abi.stream_ptr(torch.device(dev))), "swiglu_bwd_colsum")
            return partial.sum(0)

        def fused():
            return gemm_swiglu_bwd_raw(dy, w3.t().contiguous(), ab)

        t_g = timed(lambda: torch.matmul(dy, w3))
        t_u, t_f = timed(unfused), timed(fused)
        flops = 2.0 * M * K * H
        bytes_epi = 2 * M * 2 * H * ab.element_size()
        print(f"{str(dtype):16s} M={M} K={K} H={H}: library dgrad alone {t_g:7.1f} us ({flops / t_g / 1e6:6.0f} TFLOP/s) | dgrad + swiglu_bwd_colsum "
              f"{t_u:7.1f} us | fused {t_f:7.1f} us ({flops / t_f / 1e6:6.0f} TFLOP/s, epilogue stream {bytes_epi / t_f / 1e6:5.2f} TB/s)  -> x{t_u / t_f:.2f}")


if __name__ == "__main__":
    main()
