"""mxvl_gemm_nt (csrc/gemm_swiglu.hip MODE 2: the persistent 256 x 256 MFMA kernel with a plain store epilogue) against the library GEMM
torch picks (hipBLASLt, with the repository's offline-tuned solutions when --tuned) at the token-major GEMM shapes of the headline
training step (ARM-large, 65 280 tokens): forward products x W^T and dgrad products dy W (as dy (W^T)^T with a transposed weight copy).
    python tools/gemm_nt_bench.py [--tuned]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_analysis_amd import _abi      # noqa: E402


def gemm_nt(a, b, bias=None):
    lib = _abi.load()
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=a.dtype, device=a.device)
    d = _abi.GemmNtDesc()
    d.M, d.K, d.N, d.io_dtype = M, K, N, _abi.dtype_code(a.dtype)
    d.bias_dtype = _abi.dtype_code(bias.dtype) if bias is not None else 0
    d.a_rs, d.b_rs, d.c_rs = a.stride(0), b.stride(0), c.stride(0)
    d.a, d.b, d.bias, d.c = a.data_ptr(), b.data_ptr(), _abi.ptr(bias), c.data_ptr()
    _abi.check(lib.mxvl_gemm_nt(ctypes.byref(d), _abi.stream_ptr(a.device)), "mxvl_gemm_nt")
    return c


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
    M = 65280
    shapes = [("in_proj fwd", 1024, 4096), ("out_proj fwd", 2048, 1024), ("w3 fwd", 2752, 1024), ("in_proj dgrad", 4096, 1024),
              ("out_proj dgrad", 1024, 2048), ("w3 dgrad", 1024, 2752), ("w12 dgrad", 5504, 1024), ("square 4096", 4096, 4096)]
    g = torch.Generator().manual_seed(0)
    for name, K, N in shapes:
        a = (0.5 * torch.randn(M, K, generator=g)).to(dev, torch.bfloat16)
        b = (K ** -0.5 * torch.randn(N, K, generator=g)).to(dev, torch.bfloat16)
        bt = b.t().contiguous()                                  # (K, N): what a dgrad's torch.matmul(dy, W) sees
        ref = torch.matmul(a, b.t())
        got = gemm_nt(a, b)
        err = float((got.float() - ref.float()).abs().max()) / float(ref.float().abs().max())
        t_nt = timed(lambda: torch.nn.functional.linear(a, b))
        t_nn = timed(lambda: torch.matmul(a, bt))
        t_me = timed(lambda: gemm_nt(a, b))
        fl = 2.0 * M * K * N
        print(f"{name:16s} M={M} K={K:5d} N={N:5d}: library NT {t_nt:7.1f} us ({fl / t_nt / 1e6:5.0f} TF) NN {t_nn:7.1f} us ({fl / t_nn / 1e6:5.0f} TF) | "
              f"mxvl_gemm_nt {t_me:7.1f} us ({fl / t_me / 1e6:5.0f} TF)  rel.err {err:.1e}")


if __name__ == "__main__":
    main()
