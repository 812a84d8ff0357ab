"""Report-quality metrics of the validation / test loop: BLEU-1..4, ROUGE-L and CIDEr.

Host-side mirror of the scorers `MambaXrayVLDownStream.score` runs on the decoded reports
(CXPMRG_Bench_MambaXray_VL/models/MambaXrayVL_DownStream.py:134-157; definitions under evalcap/: bleu/bleu.py:24-47 +
bleu_scorer.py:181-246 corpus BLEU with the "closest" reference length, rouge/rouge.py:50-105 ROUGE-L with beta 1.2,
cider/cider.py:27-54 + cider_scorer.py:90-192 CIDEr with n = 4, sigma = 6).  Same class names, `compute_score(gts, res)`
signature, return shapes and numbers (tests/golden/report_metrics.npz holds the reference's outputs); these are small string /
counting jobs that stay on the CPU, as SURVEY.md §8-f.4 says.  METEOR is not provided: the reference shells out to
`meteor-1.5.jar` (evalcap/meteor/meteor.py:17-47), a Java binary it does not ship.
"""
from __future__ import annotations

import math
from collections import Counter

import numpy as np


def _check(gts, res):
    if gts.keys() != res.keys():
        raise AssertionError("references and hypotheses must cover the same ids")
    for k in gts:
        if not isinstance(res[k], list) or len(res[k]) != 1:
            raise AssertionError("every id needs exactly one hypothesis (a 1-element list)")
        if not isinstance(gts[k], list) or len(gts[k]) < 1:
            raise AssertionError("every id needs at least one reference")


def _ngram_counts(sentence: str, n: int) -> tuple[int, Counter]:
    """Whitespace tokens -> (length, counts of all 1..n-grams)."""
    words = sentence.split()
    counts = Counter()
    for k in range(1, n + 1):
        counts.update(tuple(words[i:i + k]) for i in range(len(words) - k + 1))
    return len(words), counts


class Bleu:
    """Corpus BLEU-1..n with clipped n-gram counts, per-sentence closest reference length and the pycocoevalcap
    smoothing constants (tiny 1e-15 on matches, small 1e-9 on guesses)."""

    TINY, SMALL = 1e-15, 1e-9

    def __init__(self, n: int = 4):
        self._n = n

    def compute_score(self, gts, res, verbose=0):
        _check(gts, res)
        n = self._n
        tot_guess, tot_correct = [0] * n, [0] * n
        tot_test = tot_ref = 0
        per_sentence = [[] for _ in range(n)]
        for key in gts:
            hyp_len, hyp = _ngram_counts(res[key][0], n)
            ref_lens, ref_max = [], {}
            for r in gts[key]:
                rl, rc = _ngram_counts(r, n)
                ref_lens.append(rl)
                for g, c in rc.items():
                    if c > ref_max.get(g, 0):
                        ref_max[g] = c
            ref_len = min((abs(l - hyp_len), l) for l in ref_lens)[1]
            guess = [max(0, hyp_len - k) for k in range(n)]
            correct = [0] * n
            for g, c in hyp.items():
                correct[len(g) - 1] += min(ref_max.get(g, 0), c)
            tot_test += hyp_len
            tot_ref += ref_len
            for k in range(n):
                tot_guess[k] += guess[k]
                tot_correct[k] += correct[k]
            for k, b in enumerate(self._geo_means(correct, guess, hyp_len, ref_len)):
                per_sentence[k].append(b)
        return self._geo_means(tot_correct, tot_guess, tot_test, tot_ref), per_sentence

    def _geo_means(self, correct, guess, test_len, ref_len):
        out, prod = [], 1.0
        for k in range(self._n):
            prod *= (float(correct[k]) + self.TINY) / (float(guess[k]) + self.SMALL)
            out.append(prod ** (1.0 / (k + 1)))
        ratio = (test_len + self.TINY) / (ref_len + self.SMALL)
        if ratio < 1:
            bp = math.exp(1 - 1 / ratio)
            out = [b * bp for b in out]
        return out

    def method(self):
        return "Bleu"


def _lcs_len(a, b) -> int:
    if len(a) < len(b):
        a, b = b, a
    row = [0] * (len(b) + 1)
    for x in a:
        diag = 0
        for j, y in enumerate(b, 1):
            keep = row[j]
            row[j] = diag + 1 if x == y else max(row[j], row[j - 1])
            diag = keep
    return row[len(b)]


class Rouge:
    """ROUGE-L: F_beta (beta = 1.2) of the best LCS precision and the best LCS recall over the references, averaged over ids.
    Tokens are split on single spaces (`split(" ")`), as the reference does."""

    def __init__(self):
        self.beta = 1.2

    def calc_score(self, candidate, refs):
        tok_c = candidate[0].split(" ")
        prec, rec = [], []
        for r in refs:
            tok_r = r.split(" ")
            lcs = _lcs_len(tok_r, tok_c)
            prec.append(lcs / float(len(tok_c)))
            rec.append(lcs / float(len(tok_r)))
        p, r = max(prec), max(rec)
        if p != 0 and r != 0:
            return ((1 + self.beta ** 2) * p * r) / float(r + self.beta ** 2 * p)
        return 0.0

    def compute_score(self, gts, res):
        _check(gts, res)
        scores = np.array([self.calc_score(res[k], gts[k]) for k in gts])
        return np.mean(scores), scores

    def method(self):
        return "Rouge"


class Cider:
    """CIDEr: tf-idf weighted n-gram cosine similarity (n = 1..4, document frequency over the reference sets of the corpus),
    hypothesis weights clipped to the reference's, Gaussian length penalty (sigma 6), x10, mean over references and ids."""

    def __init__(self, test=None, refs=None, n: int = 4, sigma: float = 6.0):
        self._n, self._sigma = n, sigma

    def compute_score(self, gts, res):
        _check(gts, res)
        n = self._n
        hyps = [_ngram_counts(res[k][0], n)[1] for k in gts]
        refs = [[_ngram_counts(r, n)[1] for r in gts[k]] for k in gts]
        doc_freq = Counter()
        for group in refs:
            doc_freq.update({g for r in group for g in r})
        if len(hyps) < max(doc_freq.values(), default=0):
            raise AssertionError("document frequency exceeds the corpus size")
        log_corpus = 1 if len(refs) == 1 else np.log(float(len(refs)))

        def weigh(counts):
            vec = [dict() for _ in range(n)]
            norm = [0.0] * n
            length = 0
            for g, tf in counts.items():
                k = len(g) - 1
                w = float(tf) * (log_corpus - np.log(max(1.0, float(doc_freq.get(g, 0)))))
                vec[k][g] = w
                norm[k] += pow(w, 2)
                if k == 1:
                    length += tf
            return vec, [np.sqrt(v) for v in norm], length

        def similarity(hv, hn, hl, rv, rn, rl):
            delta = float(hl - rl)
            val = np.zeros(n)
            for k in range(n):
                for g, w in hv[k].items():
                    rw = rv[k].get(g, 0.0)
                    val[k] += min(w, rw) * rw
                if hn[k] != 0 and rn[k] != 0:
                    val[k] /= (hn[k] * rn[k])
                val[k] *= np.e ** (-(delta ** 2) / (2 * self._sigma ** 2))
            return val

        scores = []
        for hyp, group in zip(hyps, refs):
            hv, hn, hl = weigh(hyp)
            acc = np.zeros(n)
            for r in group:
                acc += similarity(hv, hn, hl, *weigh(r))
            scores.append(np.mean(acc) / len(group) * 10.0)
        scores = np.array(scores)
        return np.mean(scores), scores

    def method(self):
        return "CIDEr"


def score(ref, hypo, dataset: str | None = None):
    """`MambaXrayVLDownStream.score(ref, hypo)` (:134-157) without METEOR: {"Bleu_1".."Bleu_4", "ROUGE_L", "CIDEr"}.
    ref / hypo: {id: [sentence, ...]} / {id: [sentence]}; the 'chinese' dataset joins its characters with spaces first."""
    if dataset == "chinese":
        hypo = {k: [" ".join(vi) for vi in v] for k, v in hypo.items()}
        ref = {k: [" ".join(vi) for vi in v] for k, v in ref.items()}
    out = {}
    bleu, _ = Bleu(4).compute_score(ref, hypo)
    for i, b in enumerate(bleu, 1):
        out[f"Bleu_{i}"] = b
    out["ROUGE_L"] = Rouge().compute_score(ref, hypo)[0]
    out["CIDEr"] = Cider().compute_score(ref, hypo)[0]
    return out
