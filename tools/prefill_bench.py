"""Dev tool: prompt prefill (torch-module forward) vs the whole generate call.   python tools/prefill_bench.py [batch]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medical_image_analysis_amd.report_decoder import ReportDecoder, KVCache
dev = "cuda:0"
torch.manual_seed(0)
with torch.device(dev):
    m = ReportDecoder(32000, 4096, 11008, 32, 32, 32).to(torch.bfloat16).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
emb = (0.02 * torch.randn(B, 230, 4096)).to(dev, torch.bfloat16)
kw = dict(num_beams=3, min_new_tokens=128, max_new_tokens=128, repetition_penalty=2.0, length_penalty=2.0, eos_token_id=2, pad_token_id=0)
with torch.no_grad():
    for _ in range(2): m.generate(emb, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.generate(emb, **kw); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    for _ in range(2): m(emb, past_key_values=KVCache())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): m(emb, past_key_values=KVCache())
    torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 5
print(f"batch {B}: generate {tg*1e3:.1f} ms; prefill forward {tp*1e3:.1f} ms; per-token {(tg - tp) / 128 * 1e3:.3f} ms")
