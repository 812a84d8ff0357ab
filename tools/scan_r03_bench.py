"""Dev tool (round 3): forward variants and backward workgroup shapes of the scan kernels at the shapes that matter,
interleaved rounds inside ONE process (cdna_hip_programming.md 5.4 rule 24): median and min per variant.

    python tools/scan_r03_bench.py [fwd|bwd|all] [rounds]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from medical_image_analysis_amd import _abi
from medical_image_analysis_amd.selective_scan_interface import scan_algorithmic_bytes, scan_bwd_raw, scan_fwd_raw

lib = _abi.load()
dev = torch.device("cuda:0")
PRODUCT = _abi.LIB_PATH


def use_lib(path):
    """Switch the binding to another build of the library (A/B arms of build.py --exp N) inside this process."""
    global lib
    _abi._lib = None
    _abi.LIB_PATH = path
    lib = _abi.load()
    return lib


def inputs(B, D, L, N, dt, seed=0, G=1):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, dt)
    u, z, Bm, Cm, dout = mk(B, D, L), mk(B, D, L), mk(B, G, N, L), mk(B, G, N, L), mk(B, D, L)
    delta = (0.5 * torch.rand(B, D, L, generator=g)).to(dev, dt)
    A = (-0.5 * torch.rand(D, N, generator=g)).to(dev)
    Dv = torch.randn(D, generator=g).to(dev)
    bias = (0.5 * torch.rand(D, generator=g)).to(dev)
    return u, delta, A, Bm, Cm, Dv, z, bias, dout


def timed(f, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def fwd(rounds):
    shapes = [(8, 1536, 4096, 16, torch.float32, False), (8, 1536, 4096, 16, torch.bfloat16, False),
              (16, 1024, 4080, 16, torch.bfloat16, True), (32, 768, 196, 16, torch.float32, False),
              (64, 4096, 200, 16, torch.bfloat16, True)]
    for (B, D, L, N, dt, ck) in shapes:
        u, delta, A, Bm, Cm, Dv, z, bias, _ = inputs(B, D, L, N, dt)
        variants = (0, 10, 14, 12, 20, 15, 16, 17, 18)
        res = {v: [] for v in variants}
        names = {}
        for r in range(rounds + 1):
            for v in variants:
                lib.mxvl_set_scan_variant(v)
                f = lambda: scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=ck)
                f()
                torch.cuda.synchronize()
                t = timed(f, 20)
                names[v] = lib.mxvl_last_scan_kernel().decode()
                if r > 0:
                    res[v].append(t)
        lib.mxvl_set_scan_variant(0)
        nb = scan_algorithmic_bytes(B, D, L, N, 1, u.element_size(), True, False, (L + 127) // 128 if ck else 0)
        print(f"fwd B={B} D={D} L={L} N={N} {str(dt)[6:]} ckpt={ck}: algorithmic {nb / 1e6:.1f} MB; auto = {names[0]}")
        for v in variants:
            med, mn = statistics.median(res[v]), min(res[v])
            print(f"   v{v:<3d} med {med:8.1f} us  min {mn:8.1f} us   {nb / med * 1e-6:6.3f} TB/s = {nb / med * 1e-6 / 8 * 100:5.1f} %   {names[v]}")


def ab(rounds, libs, what):
    """interleaved A/B of whole library builds: automatic kernel choice, headline shapes"""
    shapes = [(16, 1024, 4080, 16, torch.bfloat16), (64, 4096, 200, 16, torch.bfloat16), (8, 1536, 4096, 16, torch.float32)]
    for (B, D, L, N, dt) in shapes:
        u, delta, A, Bm, Cm, Dv, z, bias, dout = inputs(B, D, L, N, dt)
        use_lib(PRODUCT)
        _, _, ckpt = scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=True)
        dB = torch.zeros(Bm.shape, dtype=torch.float32, device=dev)
        dC = torch.zeros_like(dB)
        res = {l: [] for l in libs}
        ref_out = None
        for r in range(rounds + 1):
            for l in libs:
                use_lib(l)
                if what == "fwd" and r == 0:       # every arm must produce the product library's bits
                    o = scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=(dt != torch.float32))[0]
                    ref_out = o.clone() if ref_out is None else ref_out
                    print(f"      {os.path.basename(l)}: max |out - product out| = {float((o.float() - ref_out.float()).abs().max()):.3e}")
                if what == "bwd" and r == 0:       # every arm's gradients against the product library's (relative L2)
                    o = [t.float().clone() for t in scan_bwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, ckpt, dout) if t is not None]
                    ref_out = o if ref_out is None else ref_out
                    rel = [float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(o, ref_out)]
                    print(f"      {os.path.basename(l)}: rel L2 vs product (du ddelta dA dB dC dD dz dbias) = " + " ".join(f"{x:.1e}" for x in rel))
                if what == "bwd":
                    f = lambda: scan_bwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, ckpt, dout, dB=dB, dC=dC)
                else:
                    f = lambda: scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=(dt != torch.float32))
                f()
                torch.cuda.synchronize()
                t = timed(f, 10)
                if r > 0:
                    res[l].append(t)
        use_lib(PRODUCT)
        print(f"A/B {what} B={B} D={D} L={L} N={N} {str(dt)[6:]}")
        for l in libs:
            print(f"   {os.path.basename(l):24s} med {statistics.median(res[l]):8.1f} us  min {min(res[l]):8.1f} us")


def bwd(rounds):
    shapes = [(16, 1024, 4080, 16, torch.bfloat16), (64, 4096, 200, 16, torch.bfloat16), (32, 768, 196, 16, torch.float32),
              (8, 1536, 4096, 16, torch.float32), (32, 4096, 196, 1, torch.bfloat16)]
    for (B, D, L, N, dt) in shapes:
        u, delta, A, Bm, Cm, Dv, z, bias, dout = inputs(B, D, L, N, dt)
        _, _, ckpt = scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, want_ckpt=True)
        dB = torch.zeros(Bm.shape, dtype=torch.float32, device=dev)
        dC = torch.zeros_like(dB)
        variants = (0, 1, 2)
        res = {v: [] for v in variants}
        for r in range(rounds + 1):
            for v in variants:
                lib.mxvl_set_scan_variant(v << 8)
                f = lambda: scan_bwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True, ckpt, dout, dB=dB, dC=dC)
                f()
                torch.cuda.synchronize()
                t = timed(f, 10)
                if r > 0:
                    res[v].append(t)
        lib.mxvl_set_scan_variant(0)
        nb = scan_algorithmic_bytes(B, D, L, N, 1, u.element_size(), True, True, ckpt.shape[2])
        print(f"bwd B={B} D={D} L={L} N={N} {str(dt)[6:]}: algorithmic {nb / 1e6:.1f} MB (times include torch.zeros of dA / dD / dbias)")
        for v in variants:
            med, mn = statistics.median(res[v]), min(res[v])
            print(f"   variant {v} ({'auto' if v == 0 else '8 waves x 32 rows' if v == 1 else '4 waves x 16 rows'}): med {med:8.1f} us  min {mn:8.1f} us   "
                  f"{nb / med * 1e-6:6.3f} TB/s = {nb / med * 1e-6 / 8 * 100:5.1f} %")


def n1(rounds):
    """dstate 1, 4 direction groups, no z (VMamba's SS2D stages at batch 32): the flat-row kernel (variant 0) vs the general kernels (30)"""
    for (B, D, L, dt) in [(32, 4096, 196, torch.bfloat16), (32, 2048, 784, torch.bfloat16), (32, 1024, 3136, torch.bfloat16),
                          (32, 4096, 196, torch.float32)]:
        u, delta, A, Bm, Cm, Dv, _, bias, _ = inputs(B, D, L, 1, dt, G=4)
        res, names = {0: [], 30: []}, {}
        for r in range(rounds + 1):
            for v in (0, 30):
                lib.mxvl_set_scan_variant(v)
                f = lambda: scan_fwd_raw(u, delta, A, Bm, Cm, Dv, None, bias, True, want_ckpt=True)
                f()
                torch.cuda.synchronize()
                t = timed(f, 20)
                names[v] = lib.mxvl_last_scan_kernel().decode()
                if r > 0:
                    res[v].append(t)
        lib.mxvl_set_scan_variant(0)
        nb = scan_algorithmic_bytes(B, D, L, 1, 4, u.element_size(), False, False, (L + 127) // 128)
        print(f"fwd N=1 B={B} D={D} L={L} G=4 {str(dt)[6:]} ckpt=True: algorithmic {nb / 1e6:.1f} MB")
        for v in (0, 30):
            med, mn = statistics.median(res[v]), min(res[v])
            print(f"   v{v:<3d} med {med:8.1f} us  min {mn:8.1f} us   {nb / med * 1e-6:6.3f} TB/s = {nb / med * 1e-6 / 8 * 100:5.1f} %   {names[v]}")
        # backward: the pass-major kernel (variant 0) vs the general kernel (3); the times include the zero fill of dA | dD | dbias
        _, _, ckpt = scan_fwd_raw(u, delta, A, Bm, Cm, Dv, None, bias, True, want_ckpt=True)
        dout = torch.randn(B, D, L, generator=torch.Generator().manual_seed(5)).to(dev, dt)
        dB = torch.zeros(Bm.shape, dtype=torch.float32, device=dev)
        dC = torch.zeros_like(dB)
        resb = {0: [], 3: []}
        for r in range(rounds + 1):
            for v in (0, 3):
                lib.mxvl_set_scan_variant(v << 8)
                f = lambda: scan_bwd_raw(u, delta, A, Bm, Cm, Dv, None, bias, True, ckpt, dout, dB=dB, dC=dC)
                f()
                torch.cuda.synchronize()
                t = timed(f, 10)
                if r > 0:
                    resb[v].append(t)
        lib.mxvl_set_scan_variant(0)
        nbb = scan_algorithmic_bytes(B, D, L, 1, 4, u.element_size(), False, True, ckpt.shape[2])
        print(f"bwd N=1 B={B} D={D} L={L} G=4 {str(dt)[6:]}: algorithmic {nbb / 1e6:.1f} MB")
        for v in (0, 3):
            med, mn = statistics.median(resb[v]), min(resb[v])
            print(f"   variant {v} ({'scan_n1_bwd' if v == 0 else 'general kernel'}): med {med:8.1f} us  min {mn:8.1f} us   {nbb / med * 1e-6:6.3f} TB/s = {nbb / med * 1e-6 / 8 * 100:5.1f} %")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if what.startswith("ab"):      # ab_bwd / ab_fwd  rounds  exp numbers...
        here = os.path.dirname(PRODUCT)
        libs = [PRODUCT] + [os.path.join(here, "build", f"libmxvl_exp{e}.so") for e in sys.argv[3:]]
        ab(rounds, libs, what[3:])
        sys.exit(0)
    if what == "n1":
        n1(rounds)
        sys.exit(0)
    if what in ("fwd", "all"):
        fwd(rounds)
    if what in ("bwd", "all"):
        bwd(rounds)
