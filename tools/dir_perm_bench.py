"""Dev tool (round 3): the four scan-order re-ordering launches of a bimamba-v3 mixer (csrc/dir_perm.hip) at the 197-token
encoder shape, product library against A/B builds (build.py --exp N) interleaved inside ONE process; every arm must produce
the product library's bits.

    python tools/dir_perm_bench.py [rounds] [exp ...]            # e.g. `3 8`: product vs build/libmxvl_exp8.so
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from medical_image_analysis_amd import _abi
from medical_image_analysis_amd import mamba_simple as ms

dev = torch.device("cuda:0")
PRODUCT = _abi.LIB_PATH


def use_lib(path):
    _abi._lib = None
    _abi.LIB_PATH = path
    return _abi.load()


def timed(f, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medical_image_analysis_amd")
    libs = [PRODUCT] + [os.path.join(here, "build", f"libmxvl_exp{e}.so") for e in sys.argv[2:]]
    for (B, D, S, K, dt) in [(64, 1024, 14, 4, torch.bfloat16), (256, 1024, 12, 4, torch.bfloat16), (64, 1024, 14, 6, torch.float16)]:
        L = S * S + 1
        Lp = (L + 7) // 8 * 8
        g = torch.Generator().manual_seed(0)
        perm = torch.stack([torch.randperm(L, generator=g) for _ in range(K)]).to(dev, torch.int32)
        inv = torch.argsort(perm.long(), dim=1).to(torch.int32)
        x = torch.randn(B, 2 * D, L, generator=g).to(dev, dt)          # xz of the mixer: x and z are its channel halves
        z = x[:, D:]
        dout = torch.randn(B, D, L, generator=g).to(dev, dt)
        X = torch.empty(K, D, B, Lp, dtype=dt, device=dev).permute(2, 0, 1, 3)     # direction-channel-major, like the mixer node
        y = torch.randn(K, D, B, Lp, generator=g).to(dev, dt).permute(2, 0, 1, 3)
        out, pre, dz = (torch.empty(B, D, L, dtype=dt, device=dev) for _ in range(3))
        dy = torch.empty_like(X)
        dx = torch.empty(B, D, L, dtype=dt, device=dev)
        calls = {
            "gather          x -> X        ": (lambda: ms._dir_perm(False, x[:, :D], X, perm, L, Lp), lambda: [X], (1 + K) * B * D * L),
            "merge * silu(z) y -> out, pre ": (lambda: ms._dir_perm(True, out, y, inv, L, Lp, gate=z, pre=pre, scale=0.25), lambda: [out, pre], (K + 3) * B * D * L),
            "gated gather    dout -> dy, dz": (lambda: ms._dir_perm(False, dout, dy, perm, L, Lp, gate=z, pre=pre, dgate=dz, scale=0.25), lambda: [dy, dz], (K + 4) * B * D * L),
            "merge           dX -> dx      ": (lambda: ms._dir_perm(True, dx, y, inv, L, Lp), lambda: [dx], (K + 1) * B * D * L),
        }
        print(f"B={B} D={D} L={L} Lp={Lp} K={K} {str(dt)[6:]}")
        for name, (f, outs, elems) in calls.items():
            res = {l: [] for l in libs}
            ref = None
            for r in range(rounds + 1):
                for l in libs:
                    use_lib(l)
                    t = timed(f)
                    if r == 0:
                        got = [o.clone() for o in outs()]
                        if ref is None:
                            ref = got
                        else:
                            same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ref, got))
                            print(f"      {os.path.basename(l)}: bits equal to the product library: {same}")
                    else:
                        res[l].append(t)
            use_lib(PRODUCT)
            nb = elems * x.element_size()
            print("   " + name + "  ".join(f"{os.path.basename(l)[7:-3] or 'product':8s} {statistics.median(res[l]):7.1f} us {nb / statistics.median(res[l]) * 1e-6:5.2f} TB/s"
                                           for l in libs))


if __name__ == "__main__":
    main()
