"""Dev helper (GPU box): SwiGLU backward with a separate bias-gradient reduction vs mxvl_swiglu_bwd_colsum, ARM-large shape.
    PYTHONPATH=. python tools/swiglu_bench.py"""
import torch

from medical_image_analysis_amd import _abi


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _abi.load()
    dev = torch.device("cuda:0")
    for rows, H in [(65280, 2730), (32640, 2730), (12608, 2730)]:
        ab = torch.randn(rows, 2 * H, device=dev).bfloat16()
        dy = torch.randn(rows, H, device=dev).bfloat16()
        dab = torch.empty_like(ab)
        st = _abi.stream_ptr(dev)
        code = _abi.dtype_code(ab.dtype)
        n_part = lib.mxvl_swiglu_partials(rows, H)
        partial = torch.empty((n_part, 2 * H), dtype=torch.float32, device=dev)

        def separate():
            _abi.check(lib.mxvl_swiglu_bwd(ab.data_ptr(), dy.data_ptr(), dab.data_ptr(), rows, H, code, st), "bwd")
            return dab.sum(0, dtype=torch.float32)

        def fused():
            _abi.check(lib.mxvl_swiglu_bwd_colsum(ab.data_ptr(), dy.data_ptr(), dab.data_ptr(), partial.data_ptr(), n_part, rows, H, code, st), "colsum")
            return partial.sum(0)

        a, b = separate(), fused()
        err = float((a - b).abs().max()) / float(a.abs().max())
        print(f"rows {rows} H {H}: separate {timed(separate):.1f} us, fused {timed(fused):.1f} us (n_partials {n_part}, rel diff {err:.2e})")


if __name__ == "__main__":
    main()
