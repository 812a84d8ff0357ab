"""Deterministic stand-ins for the third-party HF tokenizer / Bio_ClinicalBERT text encoder the CLIP stage calls
(CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_CLIP.py:117-131).  Shared by make_golden.py (driving the reference's own
forward) and the tests (driving this package's mirror), so both see the same text branch."""
import types

import torch


class Toks(dict):
    def to(self, device):
        return Toks({k: v.to(device) for k, v in self.items()})


class ToyTokenizer:
    bos_token_id = 1
    cls_token_id = 1

    def __call__(self, texts, padding=None, truncation=None, return_tensors=None, max_length=128):
        ids = torch.zeros(len(texts), max_length, dtype=torch.long)
        att = torch.zeros(len(texts), max_length, dtype=torch.long)
        for i, t in enumerate(texts):
            w = [1] + [3 + (sum(map(ord, x)) % 60) for x in str(t).split()][:max_length - 2] + [2]
            ids[i, :len(w)] = torch.tensor(w)
            att[i, :len(w)] = 1
        return Toks(input_ids=ids, attention_mask=att)


class ToyText(torch.nn.Module):
    """embedding + one linear mix; returns the mapping the reference indexes (`[...]["last_hidden_state"]`)."""

    def __init__(self, hidden=32):
        super().__init__()
        self.config = types.SimpleNamespace(hidden_size=hidden)
        self.emb = torch.nn.Embedding(64, hidden)
        self.mix = torch.nn.Linear(hidden, hidden)

    def forward(self, input_ids, attention_mask=None):
        h = self.emb(input_ids)
        m = attention_mask[..., None].float()
        ctx = (h * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)
        return {"last_hidden_state": torch.tanh(self.mix(h + ctx))}
