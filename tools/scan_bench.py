"""Quick A/B of the scan kernel variants on the GPU (dev tool; bench.py is the contract).
usage: python tools/scan_bench.py [B D L N dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd import _abi
from medical_image_analysis_amd.selective_scan_interface import scan_fwd_raw

def run(B, D, L, N, dtype, variants=(10, 15, 16), iters=10, rounds=5):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    A = (-0.5 * torch.rand(D, N, generator=g)).to(dev)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, dtype)
    u, z, Bm, Cm = mk(B, D, L), mk(B, D, L), mk(B, 1, N, L), mk(B, 1, N, L)
    delta = (0.5 * torch.rand(B, D, L, generator=g)).to(dev, dtype)
    Dv = torch.randn(D, generator=g).to(dev); bias = (0.5 * torch.rand(D, generator=g)).to(dev)
    elt = u.element_size()
    bytes_ = elt * (4 * B * D * L + 2 * B * N * L) + 4 * (D * N + 2 * D)
    lib = _abi.load()
    import statistics
    times = {v: [] for v in variants}
    names = {}
    for rnd in range(rounds):          # interleaved rounds: DVFS / box noise hits every variant alike
        for v in variants:
            lib.mxvl_set_scan_variant(v)
            for _ in range(2):
                scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / iters * 1e3)
            names[v] = lib.mxvl_last_scan_kernel().decode()
    for v in variants:
        us, med = min(times[v]), statistics.median(times[v])
        print(f"B={B} D={D} L={L} N={N} {str(dtype)[6:]:8s} variant {v:5d} {names[v]:36s} "
              f"min {us:8.1f} us  med {med:8.1f} us  {bytes_ / us * 1e-6:6.3f} TB/s  ({bytes_ / us * 1e-6 / 8 * 100:5.1f}% of 8 TB/s)")
    lib.mxvl_set_scan_variant(0)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        if os.environ.get("MXVL_LIB"):
            _abi.LIB_PATH = os.environ["MXVL_LIB"]
        vs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [14]
        dt = getattr(torch, sys.argv[3]) if len(sys.argv) > 3 else torch.float32
        for ab in (0, 1, 8, 32, 64, 128, 256, 32 | 8, 1 | 32 | 8):
            print("ablate bits", ab, "(1=no n-loop 2=half n-loop 4=no softplus/silu 8=no out store 32=no global loads after chunk 0; one class only: 64=B/C tile 128=z 256=u/delta)")
            run(8, 1536, 4096, 16, dt, variants=[v | (ab << 16) for v in vs], rounds=3)
    elif len(sys.argv) > 1:
        B, D, L, N = map(int, sys.argv[1:5]); dt = getattr(torch, sys.argv[5]) if len(sys.argv) > 5 else torch.float32
        run(B, D, L, N, dt)
    else:
        run(8, 1536, 4096, 16, torch.float32)
        run(8, 1536, 4096, 16, torch.bfloat16)
        run(32, 768, 196, 16, torch.float32)
        run(32, 768, 197, 16, torch.float32)
