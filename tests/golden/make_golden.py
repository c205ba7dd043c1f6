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
