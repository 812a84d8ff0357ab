"""Dev tool: time mxvl_scan_bwd alone (set MXVL_BWD_ABLATE=bits to skip parts: 1 LDS atomics, 2 global atomics, 4 state loop)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# ablation switches live only in the measurement build (csrc/mxvl_common.h): point the binding at it when one is requested
if any(os.environ.get(k) for k in ['MXVL_BWD_ABLATE', 'MXVL_BWD_WAVES']):
    from medical_image_analysis_amd import _abi as _abi_sel, build as _build_sel
    _abi_sel.LIB_PATH = _build_sel.build(ablate=True)
import torch
from medical_image_analysis_amd.selective_scan_interface import scan_fwd_raw, scan_bwd_raw, scan_algorithmic_bytes

import medical_image_analysis_amd.selective_scan_interface as ssi


def run(B, D, L, N, dtype, iters=10, workspace=False, variant=0):
    ssi.USE_BWD_WORKSPACE = workspace
    from medical_image_analysis_amd import _abi
    _abi.load().mxvl_set_scan_variant(variant << 8)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    A = (-0.5 * torch.rand(D, N, generator=g)).to(dev)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, dtype)
    u, z, Bm, Cm, dout = mk(B, D, L), mk(B, D, L), mk(B, 1, N, L), mk(B, 1, N, L), mk(B, D, L)
    delta = (0.5 * torch.rand(B, D, L, generator=g)).to(dev, dtype)
    Dv = torch.randn(D, generator=g).to(dev); bias = (0.5 * torch.rand(D, generator=g)).to(dev)
    out, _, ckpt = scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=True)
    for _ in range(2):
        scan_bwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, ckpt, dout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        scan_bwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, ckpt, dout)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    nb = scan_algorithmic_bytes(B, D, L, N, 1, u.element_size(), True, True, ckpt.shape[2])
    print(f"bwd B={B} D={D} L={L} N={N} {str(dtype)[6:]} ablate={os.environ.get('MXVL_BWD_ABLATE','0')} dBdC={'workspace+reduce' if workspace else 'atomics'} variant={variant}: {us:9.1f} us  {nb/us*1e-6:6.3f} TB/s (incl. torch.zeros of the accumulators)")

if __name__ == "__main__":
    if len(sys.argv) > 4:
        run(*map(int, sys.argv[1:5]), getattr(torch, sys.argv[5]) if len(sys.argv) > 5 else torch.float32)
        sys.exit(0)
    # variant: 1 = 8-wave (32-row) workgroups, 2 = 4-wave
    for rep in range(2):
        for v in (1, 2):
            run(16, 1024, 4080, 16, torch.bfloat16, variant=v)
    run(16, 1024, 4080, 16, torch.float32, variant=1)
    for v in (1, 2):
        run(8, 1024, 4080, 16, torch.bfloat16, variant=v)
        run(64, 4096, 200, 16, torch.bfloat16, variant=v)   # arm_encoder_large_224: 4 directions stacked, L = 197 -> 200
        run(32, 768, 196, 16, torch.bfloat16, variant=v)
