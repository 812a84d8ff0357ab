"""Dev probe: the decoder MLP's gate / up projections at prompt-prefill shapes (M = batch x 230 tokens, K 4096, H 11008):
two library GEMMs + silu * mul (the module path) against one mxvl_gemm_swiglu_fwd over [gate; up] (no pre-activations out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd import fused_ops

dev = "cuda:0"


def timed(f, iters=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


K, H = 4096, 11008
g = torch.Generator().manual_seed(0)
w = (torch.randn(2 * H, K, generator=g) * K ** -0.5).to(dev, torch.bfloat16)
wg, wu = w[:H], w[H:]
for B in (1, 6, 8, 16):
    M = B * 230
    x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
    mod = lambda: torch.nn.functional.silu(torch.nn.functional.linear(x, wg)) * torch.nn.functional.linear(x, wu)
    fused = lambda: fused_ops.gemm_swiglu_fwd_raw(x, w, None, False)[0]
    a, b = mod(), fused()
    ref = torch.nn.functional.silu(x.float() @ wg.float().t()) * (x.float() @ wu.float().t())
    ea, eb = float((a.float() - ref).abs().max()), float((b.float() - ref).abs().max())
    tm, tf = timed(mod), timed(fused)
    fl = 2 * 2 * M * K * H
    print(f"M={M:5d}: module path {tm:8.1f} us ({fl / tm / 1e6:6.0f} TFLOP/s, max err {ea:.3e})   fused {tf:8.1f} us ({fl / tf / 1e6:6.0f} TFLOP/s, max err {eb:.3e})")
