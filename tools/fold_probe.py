"""Dev probe: what would folding the batch into the sequence buy the 4-direction encoder's scans?  The same number of (row, step) pairs as
B64 x L200 (the 197-token rows, padded to 200) as ONE long sequence per channel: no half-empty second 128-step chunk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd.selective_scan_interface import scan_bwd_raw, scan_fwd_raw

dev = torch.device("cuda:0")


def timed(f, iters=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


from medical_image_analysis_amd import selective_scan_interface as ssi
import sys
SHAPES = ((64, 200), (256, 144)) if len(sys.argv) < 2 else tuple((int(a.split('x')[0]), int(a.split('x')[1])) for a in sys.argv[1:])
for (B, L, fold) in [(b, l, f) for (b, l) in SHAPES for f in (True, False)]:
    ssi.FOLD_SHORT_ROWS = fold
    G, D, N = 4, 8192, 16
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16)
    u, z, dout = mk(B, D, L), mk(B, D, L), mk(B, D, L)
    Bm, Cm = mk(B, G, N, L), mk(B, G, N, L)
    delta = (0.5 * torch.rand(B, D, L, generator=g)).to(dev, torch.bfloat16)
    A = (-0.5 * torch.rand(D, N, generator=g)).to(dev)
    Dv = torch.randn(D, generator=g).to(dev)
    bias = (0.5 * torch.rand(D, generator=g)).to(dev)
    _, _, ckpt = scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=True)
    dB = torch.zeros(Bm.shape, dtype=torch.float32, device=dev); dC = torch.zeros_like(dB)
    tf = timed(lambda: scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=True))
    tb = timed(lambda: scan_bwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, ckpt, dout, dB=dB, dC=dC))
    print(f"B={B:3d} L={L:6d} D={D} G={G} folded={ckpt is not None and ckpt.dim() == 3}: fwd {tf:8.1f} us   bwd {tb:8.1f} us")
