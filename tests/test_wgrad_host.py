"""Host logic around the weight-gradient kernel (mxvl_gemm_tn) that needs no GPU: the routing predicates refuse CPU tensors and shapes
the kernel does not take, the library form is what runs then, `linear_module` / `linear_tokens` are the plain module call off the GPU,
and the C-ABI exports the entry with the descriptor layout the binding mirrors (tests/test_abi.py checks sizeof / offsetof)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from medical_image_analysis_amd import _abi
from medical_image_analysis_amd import selective_scan_interface as ssi


def test_gemm_tn_is_exported_and_routing_refuses_cpu_tensors():
    lib = _abi.load()
    assert hasattr(lib, "mxvl_gemm_tn")
    a = torch.randn(4096, 256).to(torch.bfloat16)
    b = torch.randn(4096, 512).to(torch.bfloat16)
    assert not ssi.gemm_tn_takes(a, b)                         # CPU tensors: never the kernel
    dw = ssi.splitk_wgrad(a, b, torch.float32)                 # ... the library form runs
    ref = a.float().t() @ b.float()
    assert dw.dtype == torch.float32 and dw.shape == (256, 512)
    assert float((dw - ref).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())


def test_gemm_tn_routing_rule():
    """the measured win region (profiles/r06_wgrad_tn_bench.txt): long token axes and tile counts that fill an XCD's 32 workgroups"""
    w = ssi.gemm_tn_wins
    assert w(65280, 5504, 1024) and w(65280, 1024, 2752) and w(65280, 2048, 512) and w(32640, 5504, 1024) and w(102656, 512, 2048)
    assert not w(25856, 4096, 1024) and not w(8192, 4096, 1024)            # short token axis: the library's batched split-K wins
    assert not w(102656, 1536, 512) and not w(65280, 512, 512)             # 12 / 4 tiles: a quarter / half of the workgroups idle
    assert not w(65280, 128, 1024)


def test_wgrad_splits_rule():
    assert ssi.wgrad_splits(1024, 4096, 4096) == 1                                  # short token axis: one GEMM
    assert ssi.wgrad_splits(65280, 5504, 1024) == 16                                # the measured exception
    s = ssi.wgrad_splits(65280, 1024, 1024)
    assert s >= 2 and 65280 % s == 0 and s & (s - 1) == 0                           # a power of two dividing K


def test_linear_module_and_linear_tokens_are_the_module_call_off_the_gpu():
    torch.manual_seed(0)
    lin = nn.Linear(32, 48)
    x = torch.randn(5000, 32, requires_grad=True)
    y = ssi.linear_module(lin, x)
    assert torch.equal(y, lin(x))
    w = torch.randn(48, 32, requires_grad=True)
    assert torch.equal(ssi.linear_tokens(x, w, None), F.linear(x, w))
    y.sum().backward()
    assert lin.weight.grad is not None and x.grad is not None
