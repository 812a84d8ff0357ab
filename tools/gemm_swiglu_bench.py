"""Dev tool (round 3): the fused SwiGLU projection (mxvl_gemm_swiglu_fwd: GEMM + bias + gate, h and ab out) against the
round-2 path (library GEMM over [w1; w2] + mxvl_swiglu_fwd) at the ARM-large / base / huge layer shapes: correctness vs fp32
and time, interleaved rounds in one process.

    python tools/gemm_swiglu_bench.py [rounds]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from medical_image_analysis_amd import _abi, fused_ops
from medical_image_analysis_amd.pretrain_engine import enable_tuned_gemms

lib = _abi.load()
dev = torch.device("cuda:0")
enable_tuned_gemms()


def timed(f, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    shapes = [(16 * 4080, 1024, 2730, "ARM-large 1024^2, batch 16"), (8 * 4080, 1024, 2730, "ARM-large, batch 8"),
              (64 * 197, 1024, 2730, "ARM-large encoder 224^2, batch 64"), (64 * 128, 768, 2048, "ARM-base 192^2, batch 64"),
              (8 * 4080, 1536, 4096, "ARM-huge, batch 8")]
    for M, K, H, what in shapes:
        g = torch.Generator().manual_seed(M + H)
        x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16)
        w = (torch.randn(2 * H, K, generator=g) * K ** -0.5).to(dev, torch.bfloat16)
        b = (0.5 * torch.randn(2 * H, generator=g)).to(dev)
        bb = b.to(torch.bfloat16)

        def lib_path():
            ab = torch.nn.functional.linear(x, w, bb)
            y = torch.empty((M, H), dtype=ab.dtype, device=dev)
            _abi.check(lib.mxvl_swiglu_fwd(ab.data_ptr(), y.data_ptr(), M, H, _abi.dtype_code(ab.dtype), _abi.stream_ptr(dev)), "swiglu")
            return y, ab

        fused = lambda: fused_ops.gemm_swiglu_fwd_raw(x, w, b, True)
        fused_noab = lambda: fused_ops.gemm_swiglu_fwd_raw(x, w, b, False)
        gemm_only = lambda: torch.nn.functional.linear(x, w, bb)
        # correctness on a slab (fp32 reference on the bf16 inputs)
        h, ab = fused()
        rows = torch.randint(0, M, (256,), generator=g).to(dev)
        ref_ab = x[rows].float() @ w.float().t() + b
        ref_h = torch.nn.functional.silu(ref_ab[:, :H]) * ref_ab[:, H:]
        e_ab = float((ab[rows].float() - ref_ab).abs().max()) / float(ref_ab.abs().max())
        e_h = float((h[rows].float() - ref_h).abs().max()) / float(ref_h.abs().max())
        h2, _ = fused_noab()
        same = bool(torch.equal(h, h2))
        res = {k: [] for k in ("library GEMM + gate", "library GEMM alone", "fused (h + ab)", "fused (h only)")}
        fs = {"library GEMM + gate": lib_path, "library GEMM alone": gemm_only, "fused (h + ab)": fused, "fused (h only)": fused_noab}
        for r in range(rounds + 1):
            for k, f in fs.items():
                f()
                torch.cuda.synchronize()
                t = timed(f)
                if r:
                    res[k].append(t)
        fl = 2.0 * M * 2 * H * K
        print(f"{what}: M={M} K={K} H={H}  rel err ab {e_ab:.2e} h {e_h:.2e}  h identical with/without ab: {same}")
        for k, v in res.items():
            med = statistics.median(v)
            print(f"   {k:22s} med {med:8.1f} us  min {min(v):8.1f} us   {fl / med * 1e-6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
