"""Dev tool (round 3): flash-attention forward / backward of the product library against A/B builds (build.py --exp N), interleaved
inside ONE process.      python tools/attn_ab.py [rounds] [exp ...]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd import _abi
from medical_image_analysis_amd import flash_attention as fa

PRODUCT = _abi.LIB_PATH


def use_lib(path):
    _abi._lib = None
    _abi.LIB_PATH = path
    return _abi.load()


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medical_image_analysis_amd")
    libs = [PRODUCT] + [os.path.join(here, "build", f"libmxvl_exp{e}.so") for e in sys.argv[2:]]
    dev = "cuda:0"
    for (B, H, L, D, mask) in [(16, 8, 4080, 64, "block_causal"), (32, 16, 400, 64, "none"), (4, 32, 2048, 128, "causal"), (64, 16, 400, 32, "none")]:
        g = torch.Generator().manual_seed(0)
        q = torch.randn(B, L, H, D, generator=g).to(dev, torch.bfloat16).transpose(1, 2)
        kv = torch.randn(B, L, 2, H, D, generator=g).to(dev, torch.bfloat16)
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
        do = torch.randn(B, L, H, D, generator=g).to(dev, torch.bfloat16).transpose(1, 2)
        mm, scale = fa.MASKS[mask], D ** -0.5
        frac = {"none": 1.0, "causal": 0.5, "block_causal": 0.5}[mask]
        flops = 4.0 * B * H * L * L * D * frac
        res = {l: ([], []) for l in libs}
        ref = None
        for r in range(rounds + 1):
            for l in libs:
                use_lib(l)
                out, lse, saved = fa.attn_fwd_raw(q, k, v, scale, mm, 16)
                tf = timeit(lambda: fa.attn_fwd_raw(q, k, v, scale, mm, 16))
                tb = timeit(lambda: fa.attn_bwd_raw(saved, out, lse, do, scale, mm, 16))
                if r == 0:
                    grads = fa.attn_bwd_raw(saved, out, lse, do, scale, mm, 16)
                    got = [out.clone()] + [t.clone() for t in grads if torch.is_tensor(t)]
                    if ref is None:
                        ref = got
                    else:
                        print(f"      {os.path.basename(l)}: max |diff| vs product: " + " ".join(f"{float((a.float() - b.float()).abs().max()):.2e}" for a, b in zip(ref, got)))
                else:
                    res[l][0].append(tf); res[l][1].append(tb)
        use_lib(PRODUCT)
        print(f"B={B} H={H} L={L} D={D} {mask}")
        for l in libs:
            f, b = statistics.median(res[l][0]), statistics.median(res[l][1])
            print(f"   {os.path.basename(l):22s} fwd {f:8.1f} us ({flops / f * 1e-6:6.1f} TFLOP/s)   bwd {b:8.1f} us ({2.5 * flops / b * 1e-6:6.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
