"""Dev tool: per-token decode time (Llama-2-7B shape, beam 3) of library builds and of the per-launch step, interleaved in ONE
process on one model.   python tools/decode_ab.py [rounds] [exp numbers of build.py --exp ...]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medical_image_analysis_amd import _abi, report_decoder
from medical_image_analysis_amd.report_decoder import ReportDecoder

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
here = os.path.dirname(_abi.LIB_PATH)
PRODUCT = _abi.LIB_PATH
arms = [("stack kernel", PRODUCT, True), ("per-launch step", PRODUCT, False)]
arms += [(f"stack kernel exp{e}", os.path.join(here, "build", f"libmxvl_exp{e}.so"), True) for e in sys.argv[2:]]
torch.manual_seed(0)
with torch.device(dev):
    m = ReportDecoder(32000, 4096, 11008, 32, 32, 32).to(torch.bfloat16).eval()
emb = (0.02 * torch.randn(1, 230, 4096)).to(dev, torch.bfloat16)
kw = dict(num_beams=3, min_new_tokens=128, max_new_tokens=128, repetition_penalty=2.0, length_penalty=2.0, eos_token_id=2, pad_token_id=0)
res = {a[0]: [] for a in arms}
for r in range(rounds + 1):
    for name, lib, stacked in arms:
        _abi._lib, _abi.LIB_PATH = None, lib
        report_decoder._STACKED = stacked
        m.__dict__["_steppers"] = {}
        try:
            m.generate(emb, **kw)
        except RuntimeError as e:          # timing-only experiment arms may trip the barrier check
            print(name, "->", str(e)[:80])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            out = m.generate(emb, **kw)
        except RuntimeError:
            pass
        torch.cuda.synchronize()
        if r:
            res[name].append((time.perf_counter() - t0) / 128 * 1e3)
_abi._lib, _abi.LIB_PATH = None, PRODUCT
for name, v in res.items():
    med = statistics.median(v)
    print(f"{name:28s} {med:7.3f} ms/token  ({1e3 / med:6.1f} tok/s, {13.4776 / med * 1e3 / 8000 * 100:5.1f} % of 8 TB/s)")
