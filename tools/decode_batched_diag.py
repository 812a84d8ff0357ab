"""Dev diagnostic: kernel stepper vs torch-module stepper, teacher-forced on the HF greedy tokens of decode_llama_hd128_batched."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from conftest import load_golden
from keyed_fill import keyed_fill_llama_
from medical_image_analysis_amd.report_decoder import ReportDecoder, KVCache, _KernelStepper, _GraphStepper
dev = "cuda:0"
g = load_golden("decode_llama_hd128_batched")
shape = {k[4:]: int(v) for k, v in g.items() if k.startswith("cfg_")}
m = ReportDecoder(rms_norm_eps=1e-6, max_position_embeddings=128, **shape)
keyed_fill_llama_(m, int(g["weight_seed"]))
m = m.to(dev).to(torch.bfloat16).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
emb = g["inputs_embeds_bf16"].view(torch.bfloat16)[:B].to(dev)
att = g["attention_mask"][:B].to(dev)
toks = g["greedy_b16"][:B].to(dev)
steps = toks.shape[1]
with torch.no_grad():
    c1, c2 = KVCache(), KVCache()
    m(emb, attention_mask=att, past_key_values=c1)
    m(emb, attention_mask=att, past_key_values=c2)
    ks = _KernelStepper(m, B, att, c1, steps, torch.bfloat16)
    ts = _GraphStepper(m, B, att, c2, steps, torch.bfloat16)
    ident = torch.arange(B, device=dev)
    for k in range(steps - 1):
        a = ks.step(toks[:, k], ident, k).float()
        b = ts.step(toks[:, k], ident, k).float()
        d = (a - b).abs()
        top2 = b.topk(2, dim=-1).values
        marg = top2[:, 0] - top2[:, 1]
        flips = (a.argmax(-1) != b.argmax(-1)).nonzero().flatten().tolist()
        print(f"step {k}: scale {float(b.abs().max()):.2f} max|diff| {float(d.max()):.4f} rows max {[round(float(x), 3) for x in d.max(-1).values]} min margin {float(marg.min()):.3f} flips {flips} margins@flips {[round(float(marg[i]),3) for i in flips]}")
