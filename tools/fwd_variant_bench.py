"""Dev tool: forward scan kernel variants (mxvl_set_scan_variant low byte) at short-sequence shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd import _abi
from medical_image_analysis_amd.selective_scan_interface import scan_fwd_raw
lib = _abi.load()
dev = torch.device("cuda:0")
for (B, D, L, N, dt) in [(32, 768, 196, 16, torch.float32), (32, 768, 196, 16, torch.bfloat16), (64, 4096, 200, 16, torch.bfloat16), (32, 4096, 196, 1, torch.bfloat16)]:
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, dt)
    u, z, Bm, Cm = mk(B, D, L), mk(B, D, L), mk(B, 1, N, L), mk(B, 1, N, L)
    delta = (0.5 * torch.rand(B, D, L, generator=g)).to(dev, dt)
    A = (-0.5 * torch.rand(D, N, generator=g)).to(dev); Dv = torch.randn(D, generator=g).to(dev); bias = (0.5 * torch.rand(D, generator=g)).to(dev)
    res = []
    for v in (0, 10, 11, 14, 1, 3, 6, 9):
        lib.mxvl_set_scan_variant(v)
        f = lambda: scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=True)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        res.append((v, e0.elapsed_time(e1) / 50 * 1e3, lib.mxvl_last_scan_kernel().decode()))
    lib.mxvl_set_scan_variant(0)
    print(f"B={B} D={D} L={L} N={N} {str(dt)[6:]}: " + "  ".join(f"v{v}:{us:6.1f}us" for v, us, _ in res) + f"   auto = {res[0][2]}")
