"""Dev tool: forward-scan A/B of experiment builds (build.py --exp N) x kernel variants at one shape, interleaved rounds in ONE
process; every arm's output is compared with the product library's bits.
usage: python tools/scan_exp_ab.py "14,10" "1,2,3" [B D L N dtype rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import scan_r03_bench as sb
from medical_image_analysis_amd.selective_scan_interface import scan_algorithmic_bytes, scan_fwd_raw

variants = [int(v) for v in sys.argv[1].split(",")]
exps = [e for e in sys.argv[2].split(",") if e]
B, D, L, N = (int(x) for x in sys.argv[3:7]) if len(sys.argv) > 6 else (8, 1536, 4096, 16)
dt = getattr(torch, sys.argv[7]) if len(sys.argv) > 7 else torch.float32
rounds = int(sys.argv[8]) if len(sys.argv) > 8 else 5
here = os.path.dirname(sb.PRODUCT)
libs = [("product", sb.PRODUCT)] + [(f"exp{e}", os.path.join(here, "build", f"libmxvl_exp{e}.so")) for e in exps]
u, delta, A, Bm, Cm, Dv, z, bias, _ = sb.inputs(B, D, L, N, dt)
res, names, ref = {}, {}, None
for r in range(rounds + 1):
    for tag, path in libs:
        lib = sb.use_lib(path)
        for v in variants:
            lib.mxvl_set_scan_variant(v)
            f = lambda: scan_fwd_raw(u, delta, A, Bm, Cm, Dv, z, bias, True)
            o = f()[0]
            torch.cuda.synchronize()
            if r == 0:
                ref = o.clone() if ref is None else ref
                names[(tag, v)] = (lib.mxvl_last_scan_kernel().decode(), float((o.float() - ref.float()).abs().max()))
            t = sb.timed(f, 20)
            if r > 0:
                res.setdefault((tag, v), []).append(t)
        lib.mxvl_set_scan_variant(0)
sb.use_lib(sb.PRODUCT)
nb = scan_algorithmic_bytes(B, D, L, N, 1, u.element_size(), True, False, 0)
print(f"fwd B={B} D={D} L={L} N={N} {str(dt)[6:]}: algorithmic {nb / 1e6:.1f} MB")
for key, ts in res.items():
    med, mn = statistics.median(ts), min(ts)
    print(f"   {key[0]:8s} v{key[1]:<3d} med {med:8.1f} us  min {mn:8.1f} us  {nb / med * 1e-6 / 8 * 100:5.1f} % of 8 TB/s   {names[key][0]}  max|out - first arm| {names[key][1]:.2e}")
