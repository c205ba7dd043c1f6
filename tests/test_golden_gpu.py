"""`-m gpu`: the HIP path (through the C ABI / the drop-in Python API) against the fixtures that
tests/golden/make_golden.py produced by EXECUTING the real reference (/root/reference/timing.py,
retokenize.py) -- filter_attention incl. the exact-tie case, force_align, get_attentions' post-capture half
(timing.py:63-66) and default_find_alignment (timing.py:116-186) -- plus the fixed-seed randomised sweeps of
the integer / order-statistic kernels (DTW, median filter, tuple-ordered top-k) that used to live in
tools/fuzz_parity.py."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SCORE_NOISE = 4e-5  # relative: fp32 reductions of the head scores run in a different order on the GPU


@pytest.fixture(scope="module")
def gold():
    arrays = np.load(os.path.join(GOLD, "reference_golden.npz"))
    with open(os.path.join(GOLD, "reference_golden.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="module")
def mods(wca):
    m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
    dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
    eng = wca.WhisperAMD(dims, device="cuda:0", max_batch=1, precision="f16")  # weight-less engine: the ops below take no model
    return m("timing"), m("tokenizer").get_tokenizer(True, language="English"), eng


def check_selection(got, ref_sorted, all_ref_scores, H):
    """got: [(score, (l, h), name)] ascending from the GPU; ref_sorted: the reference's list. Position by position the
    GPU's head must be the reference's head, or a head whose reference score is within the reduction-order noise of
    it (never an arbitrary one); heads must be distinct. Exact ties (identical maps -> bit-identical scores on both
    sides) are therefore checked strictly: the (l, h) tuple order decides (timing.py:36)."""
    assert len(got) == len(ref_sorted)
    assert len({lh for _, lh, _ in got}) == len(got)
    for (gs, glh, gname), (rs, rlh, rname) in zip(got, ref_sorted):
        eps = SCORE_NOISE * max(1.0, abs(rs))
        assert abs(gs - all_ref_scores[tuple(glh)]) <= eps, (gs, glh)
        if tuple(glh) != tuple(rlh):
            assert abs(all_ref_scores[tuple(glh)] - rs) <= eps and all_ref_scores[tuple(glh)] != rs, (glh, rlh, rs)
        assert gname == "sample_layer%d_head%d" % tuple(glh)


def test_filter_attention_reference_golden(gold, mods):
    """All 46 reference cases (5 weightings x 3 shapes x 3 k, + exact ties) through wca_filter_attention."""
    arrays, meta = gold
    tm, tok, eng = mods
    by_input = {}
    for case in meta["filter_attention"]:
        A = torch.from_numpy(arrays[case["A"]])
        L, H = A.shape[:2]
        key = (case["A"], tuple(case["w"]))
        if key not in by_input:  # the reference's scores of ALL heads for this input / weighting (the k = L*H case)
            full = [c for c in meta["filter_attention"] if c["A"] == case["A"] and c["w"] == case["w"] and c["topk"] >= L * H]
            by_input[key] = {tuple(lh): s for lh, s in zip(full[0]["heads"], full[0]["scores"])} if full else None
        sel, scored = tm.filter_attention(A.cuda(), case["topk"], *case["w"])
        ref_sorted = list(zip(case["scores"], [tuple(h) for h in case["heads"]], case["names"]))
        all_scores = by_input[key] or {tuple(lh): s for lh, s in zip(case["heads"], case["scores"])}
        if by_input[key] is None:
            assert [tuple(lh) for _, lh, _ in scored] == [lh for _, lh, _ in ref_sorted]
        else:
            check_selection(scored, ref_sorted, all_scores, H)
        for t, (_, (l, h), _) in zip(sel, scored):
            assert torch.equal(t.cpu(), A[l, h].unsqueeze(0))


def test_filter_attention_exact_ties_tuple_order(gold, mods):
    """fa_Atie: 12 identical heads -> identical scores; python's tuple sort keeps the LAST k (l, h) pairs in row-major
    order (timing.py:36). The GPU's scores are bit-identical across the copies, so its order must match exactly."""
    arrays, meta = gold
    tm, tok, eng = mods
    case = [c for c in meta["filter_attention"] if c["A"] == "fa_Atie"][0]
    sel, scored = tm.filter_attention(torch.from_numpy(arrays["fa_Atie"]).cuda(), case["topk"])
    assert len({s for s, _, _ in scored}) == 1
    assert [list(lh) for _, lh, _ in scored] == case["heads"]
    assert [n for _, _, n in scored] == case["names"]
    # every k from 1 to L*H + 3 (k larger than the head count returns all heads)
    A = torch.from_numpy(arrays["fa_Atie"])
    L, H = A.shape[:2]
    order = [(l, h) for l in range(L) for h in range(H)]
    for k in range(1, L * H + 4):
        _, scored = tm.filter_attention(A.cuda(), k)
        assert [lh for _, lh, _ in scored] == order[-k:]


def test_force_align_reference_golden(gold, mods):
    """The reference's force_align outputs (both aggregations, 4 texts incl. the degenerate one) through
    wca_force_align + the host tail: words and start / end times identical, matrix within fp32 reduction noise."""
    arrays, meta = gold
    tm, tok, eng = mods
    for case in meta["force_align"]:
        ws = torch.from_numpy(arrays[case["ws"]])
        out = tm.force_align(ws.cuda(), list(case["tokens"]), tok, "char", case["aggregation"], case["topk"])
        if case["degenerate"]:
            assert out == [[], [], [], [], None]
            continue
        words, st, en, matrix, scores = out
        assert words == case["words"]
        np.testing.assert_allclose(matrix.numpy(), arrays[case["ws"] + "_matrix"], rtol=3e-5, atol=1e-7)
        np.testing.assert_array_equal(st, arrays[case["ws"] + "_start"])
        np.testing.assert_array_equal(en, arrays[case["ws"] + "_end"])
        if case["heads"] is not None:
            assert [list(lh) for _, lh, _ in scores] == case["heads"]


def test_attention_weights_reference_golden(gold, mods):
    """timing.py:63-66 as run by the REAL get_attentions on the stub model: the same captured logits through
    wca_attention_weights. The median is an order statistic (exact); exp / sum / divide are fp32 on both sides."""
    arrays, meta = gold
    tm, tok, eng = mods
    for case in meta["get_attentions"]:
        qk = torch.from_numpy(arrays[case["qk"]])
        w = tm.attention_weights(qk.cuda(), case["max_frames"], case["medfilt_width"], case["qk_scale"])
        want = arrays[case["weights"]]
        assert tuple(w.shape) == want.shape
        np.testing.assert_allclose(w.cpu().numpy(), want, rtol=2e-6, atol=1e-7)


def test_default_find_alignment_reference_golden(gold, mods):
    """timing.py:116-186 as run by the REAL reference on the stub model: same logits -> wca_attention_weights ->
    wca_default_find_alignment + host tail. Words / start / end identical; the normalised weights (4th return)
    within fp32 noise of the reference's."""
    arrays, meta = gold
    tm, tok, eng = mods
    for ci, case in enumerate(meta["default_find_alignment"]):
        qk = torch.from_numpy(arrays[case["qk"]])
        heads = sorted(tuple(h) for h in case["heads"])
        assert [list(h) for h in heads] == case["heads_order"]
        w = tm.attention_weights(qk.cuda(), case["max_frames"], case["medfilt_width"], 1.0)
        out = tm._default_alignment_from_weights(eng, w, heads, list(case["tokens"]), tok)
        if case["degenerate"]:
            assert out == [[], [], [], [], None]
            continue
        words, st, en, weights, last = out
        assert last is None and words == case["words"]
        assert tuple(weights.shape) == arrays[f"dfa_w_{ci}"].shape
        np.testing.assert_allclose(weights.cpu().numpy(), arrays[f"dfa_w_{ci}"], rtol=2e-4, atol=2e-5)
        np.testing.assert_array_equal(st, arrays[f"dfa_start_{ci}"])
        np.testing.assert_array_equal(en, arrays[f"dfa_end_{ci}"])


# ----------------------------------------------------------------------------- fixed-seed sweeps (bit-exact kernels)
def test_fuzz_dtw_bit_exact(mods):
    """150 DTW problems up to 448 x 1500: gaussian, small-integer (every step ties), constant and rank-1 matrices.
    Backtrace indices must be bit-exact (north_star) w.r.t. the dtw_cpu restatement."""
    from oracle import timing_ref
    tm, tok, eng = mods
    rng = np.random.default_rng(20250)
    for i in range(150):
        N = int(rng.integers(1, 449)) if i % 5 else int(rng.integers(1, 12))
        M = int(rng.integers(1, 1501)) if i % 7 else int(rng.integers(1, 12))
        kind = i % 4
        if kind == 0:
            x = rng.standard_normal((N, M))
        elif kind == 1:
            x = rng.integers(0, 3, (N, M)).astype(np.float64)
        elif kind == 2:
            x = np.full((N, M), float(rng.integers(-2, 3)))
        else:
            x = np.round(rng.standard_normal((N, 1)) @ rng.standard_normal((1, M)), 1)
        xt = torch.from_numpy(x.astype(np.float32))
        ti, tj = timing_ref.dtw(-xt)
        gi, gj = tm.dtw(-xt.cuda())
        assert np.array_equal(ti, gi) and np.array_equal(tj, gj), (i, N, M, kind)


def test_fuzz_median_filter_bit_exact(mods):
    from oracle import timing_ref
    tm, tok, eng = mods
    rng = np.random.default_rng(20251)
    for i in range(150):
        F = int(rng.integers(1, 1501)) if i % 6 else int(rng.integers(1, 8))
        w = int(rng.choice([1, 3, 5, 7, 9, 15, 33]))
        rows = int(rng.integers(1, 40))
        a = torch.from_numpy(rng.standard_normal((1, 1, rows, F)).astype(np.float32))
        if i % 3 == 0:
            a = torch.round(a * 2) / 2  # many equal values
        assert torch.equal(timing_ref.median_filter(a, w), tm.median_filter(a.cuda(), w).cpu()), (i, rows, F, w)


def test_fuzz_selection_and_force_align(mods):
    """60 random head-selection / force_align problems; every 4th has exactly tied heads (duplicated maps), where the
    tuple order decides. Selection is checked position by position against the reference ordering of the oracle's
    scores (no skip for close scores); word times must be within one frame."""
    from oracle import timing_ref
    tm, tok, eng = mods
    rng = np.random.default_rng(20252)
    for i in range(60):
        L, H = int(rng.integers(1, 5)), int(rng.integers(1, 7))
        n = int(rng.integers(6, 60))
        F = int(rng.integers(8, 400))
        w = torch.softmax(torch.from_numpy(rng.standard_normal((L, H, n, F)).astype(np.float32)) * float(rng.uniform(0.5, 6)), -1)
        if i % 4 == 0 and H > 1:
            w[:, 1] = w[:, 0]
        k = int(rng.integers(1, L * H + 1))
        wc, wr, wv = [(1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1)][i % 4]
        sel, scores = tm.filter_attention(w.cuda(), k, wc, wr, wv)
        _rsel, rscores = timing_ref.filter_attention(w, k, wc, wr, wv)
        allref = {lh: s for s, lh, _ in timing_ref.filter_attention(w, L * H, wc, wr, wv)[1]}
        check_selection(scores, rscores, allref, H)
        tt = [64] * (n - len(tok.sot_sequence) - 2)
        aggr = "topk" if i % 2 else "mean"
        out = tm.force_align(w.cuda(), tt, tok, "char", aggr, topk=k, w_colnorm=wc, w_rownorm=wr, w_coverage=wv)
        ref = timing_ref.force_align(w, tt, tok, "char", aggr, k, wc, wr, wv)
        assert out[0] == ref[0]
        if len(out[1]):
            assert np.max(np.abs(np.asarray(out[1]) - np.asarray(ref[1]))) <= 0.02 + 1e-9
            assert np.max(np.abs(np.asarray(out[2]) - np.asarray(ref[2]))) <= 0.02 + 1e-9
            # the SAME (GPU) matrix through the oracle's DTW + jump arithmetic: identical times (bit-exact path)
            ti, tj = timing_ref.dtw(-out[3])
            from oracle.tokenizer_ref import split_tokens_on_spaces
            _w, word_tokens = split_tokens_on_spaces(tt + [tok.eot], tok, "char")
            rst, ren = timing_ref.jumps_to_times(ti, tj, word_tokens)
            assert np.array_equal(np.asarray(out[1]), rst) and np.array_equal(np.asarray(out[2]), ren)
