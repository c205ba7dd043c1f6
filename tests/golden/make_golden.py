#!/usr/bin/env python3
"""Generates tests/golden/reference_golden.npz + reference_golden.json by executing the REAL reference
modules (/root/reference/timing.py, retokenize.py, metrics.py) in the build container.

The reference imports third-party packages that are absent offline (openai-whisper, num2words); they are
replaced by in-process stub modules:
  * whisper.audio constants, whisper.model.disable_sdpa = nullcontext
  * whisper.timing.median_filter / dtw  -> the oracle's restatement (so what this pins is the reference's
    OWN arithmetic around them: head scores, tuple-ordered top-k, both aggregations, the [sot:-1] slice and
    the jump -> word-time arithmetic of force_align, plus retokenize.py and metrics.py verbatim)
  * a minimal byte-level tokenizer object (oracle.tokenizer_ref.CharTokenizer) stands in for whisper's.
  * get_attentions (timing.py:45-67) and default_find_alignment (timing.py:116-186) run against a stub torch
    nn.Module shaped like whisper.model.Whisper (hookable decoder.blocks[i].cross_attn whose forward returns
    (out, qk), a sparse `alignment_heads` buffer, `dims.n_text_layer`, `device`): the QK logits are seeded
    arrays, so what is pinned is the reference's hook wiring (outs[-1] / outs[-1][0]), cat / stack order,
    [:max_frames] slice, median -> *qk_scale -> softmax order, std/mean normalisation, [sot:-1] slice, word
    split and jump arithmetic, and the 5-tuple each function returns.
Only arrays / strings (inputs and the reference's outputs) are stored; no reference source is copied.
Run from the repo root:  python tests/golden/make_golden.py
"""
import contextlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import timing_ref, tokenizer_ref  # noqa: E402


def install_stubs():
    n2w = types.ModuleType("num2words")
    n2w.num2words = lambda n: (_ for _ in ()).throw(RuntimeError("num2words is not available offline"))
    sys.modules["num2words"] = n2w
    w = types.ModuleType("whisper")
    wm = types.ModuleType("whisper.model")
    wm.disable_sdpa = contextlib.nullcontext
    wt = types.ModuleType("whisper.timing")
    wt.median_filter = timing_ref.median_filter
    wt.dtw = lambda x: tuple(timing_ref.dtw(x))
    wa = types.ModuleType("whisper.audio")
    wa.HOP_LENGTH, wa.SAMPLE_RATE, wa.TOKENS_PER_SECOND = 160, 16000, 50
    w.model, w.timing, w.audio = wm, wt, wa
    sys.modules.update({"whisper": w, "whisper.model": wm, "whisper.timing": wt, "whisper.audio": wa})


class _StubCrossAttn(torch.nn.Module):
    """whisper.model.MultiHeadAttention stand-in: forward returns (out, qk) with a preset qk [1, H, n, S]."""

    def __init__(self, qk):
        super().__init__()
        self.qk = qk

    def forward(self, x, xa=None):
        return x, self.qk


class _StubBlock(torch.nn.Module):
    def __init__(self, qk):
        super().__init__()
        self.cross_attn = _StubCrossAttn(qk)


class _StubDecoder(torch.nn.Module):
    def __init__(self, qks):
        super().__init__()
        self.blocks = torch.nn.ModuleList([_StubBlock(qk) for qk in qks])


class StubWhisper(torch.nn.Module):
    """The attributes timing.get_attentions / default_find_alignment touch on a whisper.model.Whisper."""

    def __init__(self, qks, logits, alignment_heads=None):
        super().__init__()
        self.dims = types.SimpleNamespace(n_text_layer=len(qks))
        self.decoder = _StubDecoder(qks)
        self.logits = logits  # [n, V]
        L, H = len(qks), qks[0].shape[1]
        mask = torch.zeros(L, H, dtype=torch.bool)
        for l, h in (alignment_heads or []):
            mask[l, h] = True
        self.register_buffer("alignment_heads", mask.to_sparse(), persistent=False)

    @property
    def device(self):
        return torch.device("cpu")

    def forward(self, mel, tokens):
        assert mel.dim() == 3 and tokens.dim() == 2 and tokens.shape[-1] == self.logits.shape[0]
        x = torch.zeros(1, tokens.shape[-1], 4)
        for blk in self.decoder.blocks:
            x = blk.cross_attn(x, None)[0]  # module __call__: the reference's forward hooks fire here
        return self.logits[None]


def _grid_randn(g, *shape, scale=3.0, step=1.0 / 32):
    """Seeded logits on a coarse grid (compresses well; plenty of exact ties for the median)."""
    return (torch.randn(*shape, generator=g) * scale / step).round() * step


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import metrics as ref_metrics        # noqa: E402  (the reference's files)
    import retokenize as ref_retokenize  # noqa: E402
    import timing as ref_timing          # noqa: E402
    assert os.path.dirname(os.path.abspath(ref_timing.__file__)) == REF

    arrays, meta = {}, {}
    tok = tokenizer_ref.CharTokenizer()

    # ---- filter_attention: scores + tuple-sorted selection (timing.py:13-43)
    weight_sets = [(1.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 1.0, 1.0), (0.5, 2.0, 0.25)]
    shapes = [(4, 6, 12, 50), (6, 8, 40, 120), (2, 3, 9, 200)]
    fa = []
    for si, shape in enumerate(shapes):
        g = torch.Generator().manual_seed(100 + si)
        A = torch.softmax(torch.randn(*shape, generator=g) * 3, -1)
        arrays[f"fa_A{si}"] = A.numpy()
        for wi, (wc, wr, wv) in enumerate(weight_sets):
            for topk in (3, 7, shape[0] * shape[1]):
                _sel, scored = ref_timing.filter_attention(A, topk=topk, w_colnorm=wc, w_rownorm=wr, w_coverage=wv)
                fa.append(dict(A=f"fa_A{si}", w=[wc, wr, wv], topk=topk, scores=[s for s, _, _ in scored],
                               heads=[list(lh) for _, lh, _ in scored], names=[n for _, _, n in scored]))
    # exact ties: identical heads -> order decided by the (l, h) tuple
    Atie = torch.softmax(torch.randn(1, 1, 10, 30, generator=torch.Generator().manual_seed(7)), -1).repeat(3, 4, 1, 1)
    arrays["fa_Atie"] = Atie.numpy()
    _sel, scored = ref_timing.filter_attention(Atie, topk=5)
    fa.append(dict(A="fa_Atie", w=[1, 1, 0], topk=5, scores=[s for s, _, _ in scored], heads=[list(lh) for _, lh, _ in scored],
                   names=[n for _, _, n in scored]))
    meta["filter_attention"] = fa

    # ---- force_align: aggregation, slice, DTW (oracle), word split (real retokenize.py), jump arithmetic
    texts = ["artificial intelligence is for real", "a bc", "x", "it's a dog's life ok"]
    cases = []
    for ti, text in enumerate(texts):
        tt = ref_retokenize.encode(text, tok, "char")
        n = len(tt) + 5
        for ai, (aggr, topk) in enumerate([("mean", -1), ("topk", 3), ("topk", 10)]):
            g = torch.Generator().manual_seed(1000 + 10 * ti + ai)
            L, H, F = 4, 6, 40 + 25 * ti
            ws = torch.softmax(torch.randn(L, H, n, F, generator=g) * 4, -1)
            key = f"fo_ws_{ti}_{ai}"
            arrays[key] = ws.numpy()
            out = ref_timing.force_align(ws, list(tt), tok, aligned_unit_type="char", aggregation=aggr, topk=topk)
            words, st, en, matrix, scores = out
            if isinstance(matrix, torch.Tensor):
                arrays[key + "_matrix"] = matrix.numpy()
                arrays[key + "_start"] = np.asarray(st, dtype=np.float64)
                arrays[key + "_end"] = np.asarray(en, dtype=np.float64)
            cases.append(dict(ws=key, text=text, tokens=list(tt), aggregation=aggr, topk=topk, words=list(words),
                              degenerate=not isinstance(matrix, torch.Tensor),
                              heads=[list(lh) for _, lh, _ in scores] if scores else None))
    meta["force_align"] = cases

    # ---- get_attentions (timing.py:45-67) on the stub model
    ga = []
    for ci, (L, H, n, S, F, w, qs) in enumerate([(2, 3, 12, 96, 60, 7, 1.0), (3, 2, 9, 80, 80, 3, 1.0), (2, 2, 7, 64, 5, 1, 0.5),
                                                  (1, 4, 6, 40, 3, 7, 1.0), (2, 2, 10, 72, 41, 5, 2.0)]):
        g = torch.Generator().manual_seed(3000 + ci)
        qks = [_grid_randn(g, 1, H, n, S) for _ in range(L)]
        logits = torch.randn(n, 64, generator=g)
        model = StubWhisper(qks, logits)
        weights, out_logits = ref_timing.get_attentions(torch.zeros(80, 3000), torch.arange(n), model, tok, F, medfilt_width=w, qk_scale=qs)
        assert tuple(weights.shape) == (L, H, n, F) and torch.equal(out_logits, logits)
        arrays[f"ga_qk_{ci}"] = torch.cat(qks).numpy()
        arrays[f"ga_w_{ci}"] = weights.numpy()
        ga.append(dict(qk=f"ga_qk_{ci}", weights=f"ga_w_{ci}", max_frames=F, medfilt_width=w, qk_scale=qs,
                       logits_passthrough=True))
    meta["get_attentions"] = ga

    # ---- default_find_alignment (timing.py:116-186) on the stub model
    dfa = []
    for ci, (text, L, H, S, F, w, heads) in enumerate([
            ("artificial intelligence is for real", 3, 4, 120, 100, 7, [(1, 0), (2, 3), (2, 1)]),
            ("it's a dog's life", 2, 2, 90, 64, 3, [(1, 1)]),
            ("x", 2, 2, 40, 30, 7, [(0, 1), (1, 0)]),
            ("", 2, 2, 40, 30, 7, [(0, 1), (1, 0)]),
            ("ab cd ef", 4, 2, 64, 6, 7, [(3, 1), (0, 0), (2, 0), (1, 1)])]):
        tt = ref_retokenize.encode(text, tok, "char")
        n = len(tt) + 5
        g = torch.Generator().manual_seed(4000 + ci)
        qks = [_grid_randn(g, 1, H, n, S) for _ in range(L)]
        logits = torch.randn(n, 51865, generator=g)
        model = StubWhisper(qks, logits, heads)
        out = ref_timing.default_find_alignment(model, tok, list(tt), torch.zeros(80, 3000), F, medfilt_width=w, qk_scale=1.0)
        words, st, en, weights, last = out
        assert last is None
        rec = dict(qk=f"dfa_qk_{ci}", text=text, tokens=list(tt), max_frames=F, medfilt_width=w, heads=[list(h) for h in heads],
                   heads_order=[list(map(int, lh)) for lh in model.alignment_heads.indices().T.tolist()],
                   words=list(words), degenerate=not isinstance(weights, torch.Tensor))
        arrays[f"dfa_qk_{ci}"] = torch.cat(qks).numpy()
        if isinstance(weights, torch.Tensor):
            assert tuple(weights.shape) == (len(heads), n, F)
            arrays[f"dfa_w_{ci}"] = weights.numpy()
            arrays[f"dfa_start_{ci}"] = np.asarray(st, dtype=np.float64)
            arrays[f"dfa_end_{ci}"] = np.asarray(en, dtype=np.float64)
        dfa.append(rec)
    meta["default_find_alignment"] = dfa

    # ---- retokenize.py (char mode) on the stand-in tokenizer
    rt = []
    for text in ["hello world", " leading and  double  spaces ", "don't stop", "a", "", "mixed CASE words"]:
        tt = ref_retokenize.encode(text, tok, "char")
        words, wts = ref_retokenize.split_tokens_on_spaces(tt + [tok.eot], tok, "char")
        rt.append(dict(text=text, tokens=tt, words=words, word_tokens=wts))
    meta["retokenize"] = rt
    meta["remove_punctuation"] = [[t, ref_retokenize.remove_punctuation(t)] for t in
                                  ["Hello, world!", "it's a dog's life...", "semi-colon; dash - here", "(parens) [brackets] {braces}",
                                   "quotes \"double\" 'single'", "under_score and #hash"]]

    # ---- metrics.py
    mt = {}
    g = torch.Generator().manual_seed(5)
    attn = torch.softmax(torch.randn(12, 50, generator=g) * 2, -1)
    arrays["cov_attn"] = attn.numpy()
    mt["coverage_penalty"] = [float(ref_metrics.coverage_penalty(attn)), float(ref_metrics.coverage_penalty(attn, 0.1))]
    y = [0.1, 0.5, 0.9, 1.4, 2.0]
    yh = [0.11, 0.48, 1.0, 1.39, 1.7, 2.05]
    mt["eval_n1"] = [dict(y=y, yhat=yh, tol=t, out=list(ref_metrics.eval_n1(y, yh, t))) for t in (0.02, 0.05, 0.2)]
    mt["eval_n1"].append(dict(y=y, yhat=[], tol=0.02, out=list(ref_metrics.eval_n1(y, [], 0.02))))
    words = ["The", "cat,", "sat", "on", "mat"]
    words_h = ["the", "cat", "on", "sat", "mat", "extra"]
    mt["eval_n1_strict"] = [dict(y=y, yhat=yh, words=words, words_hat=words_h, tol=t,
                                 out=list(ref_metrics.eval_n1_strict(y, yh, words, words_h, t))) for t in (0.02, 0.05, 0.5)]
    mt["get_seg_metrics"] = [dict(args=a, out=[float(v) for v in ref_metrics.get_seg_metrics(*a)])
                             for a in [(10, 10, 12, 15), (0, 0, 5, 5), (7, 7, 7, 7), (3, 3, 0, 9)]]
    meta["metrics"] = mt

    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **arrays)
    with open(os.path.join(HERE, "reference_golden.json"), "w") as f:
        json.dump(meta, f)
    print("wrote %d arrays, %d force_align cases, %d filter_attention cases" % (len(arrays), len(cases), len(fa)))


if __name__ == "__main__":
    main()
