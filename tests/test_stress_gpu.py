"""`-m gpu`: BASELINE configs[3] and configs[4] at their FULL sizes.

configs[3]  AMI long-form segments, whisper-large-v2, subword align, 1500-frame DTW stress
            (/root/reference/infer_ali.py:25-26,78-81: max_frames <= 1500, len(tokens) <= 448): large-v2 dimensions
            (d = 1280, 20 heads, 32 + 32 layers, seeded random weights), 30 s audio -> F = 1500, n = 448 decoder
            tokens, char AND subword units (synthetic tiktoken vocabulary), top-k and mean aggregation.
configs[4]  probe_oracle.py full L x H head sweep, whisper-large-v3 (/root/reference/probe_oracle.py:83-90): all
            640 heads' maps resident (1.72 GB at n = 448, F = 1500), one DTW per head in ONE launch.

Integer results are held to the bit-exact bar: every DTW path must equal the oracle's dtw_cpu restatement run on the
SAME (device-computed) matrix. Timings go to gpurun_out/r03_stress.txt (copied to profiles/)."""
import importlib
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _m(n):
    return importlib.import_module("whisper-char-alignment_amd." + n)


def _log(line):
    print(line, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "r03_stress.txt"), "a") as f:
            f.write(line + "\n")


def _oracle_times(matrix, toks, tok, unit, split):
    from oracle import timing_ref
    ti, tj = timing_ref.dtw(-matrix)
    words, word_tokens = split(toks + [tok.eot], tok, unit)
    return (ti, tj) + tuple(timing_ref.jumps_to_times(ti, tj, word_tokens))


@pytest.fixture(scope="module")
def large_v2(wca):
    syn = _m("synthetic")
    dims = wca.dims_for("large-v2")
    t0 = time.time()
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.05)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=2, precision="f16").load_state_dict(sd)
    del sd
    _log("large-v2 dims (d=1280, 20 heads, 32+32 layers): engine + random weights ready in %.1f s" % (time.time() - t0))
    yield dims, model
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("aggr,topk", [("topk", 10), ("mean", -1)])
def test_config3_large_v2_char_max_sizes(wca, large_v2, aggr, topk):
    """n = 448 tokens (443 characters), F = 1500 frames: fused batch path == step-by-step API == oracle DTW of the
    device matrix (444 x 1500, bit-exact), both aggregations."""
    syn, tk, rt, tm, audio = _m("synthetic"), _m("tokenizer"), _m("retokenize"), _m("timing"), _m("audio")
    dims, model = large_v2
    tok = tk.get_tokenizer(True, language="English")
    B = 2
    pcm = np.stack([syn.synth_audio(900 + u, 480000) for u in range(B)])
    texts = [syn.synth_text(900 + u, 443) for u in range(B)]
    tts = [rt.encode(t, tok, "char") for t in texts]
    toks = np.array([[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts], dtype=np.int64)
    assert toks.shape == (B, 448)
    opts = model.make_opts(aggregation=aggr, topk=topk, sot_len=3, medfilt_width=7)
    pcm_d, tok_d = torch.from_numpy(pcm).cuda(), torch.from_numpy(toks).cuda()
    model.align_batch(pcm_d, [480000] * B, tok_d, [448] * B, [1500] * B, opts)  # warm-up (buffer growth)
    torch.cuda.synchronize()
    t1 = time.time()
    jump, sel = model.align_batch(pcm_d, [480000] * B, tok_d, [448] * B, [1500] * B, opts)
    dt = time.time() - t1
    _log("configs[3] char   aggr=%-4s B=%d n=448 F=1500 fused align_batch: %.1f ms per batch (%.2f utt/s)" % (aggr, B, dt * 1e3, B / dt))
    # the step-by-step API on the SAME micro-batch (same GEMM kernels and summation orders as the fused path: a batch of one
    # takes other kernels -- split-K fc2, few-row decoder GEMMs -- whose last-bit differences can move a near-tied DTW step)
    mels = torch.stack([audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm[b])), 80, model=model) for b in range(B)]).cuda()
    w_all, _ = model.get_attentions(mels, tok_d, [1500] * B, 7, 1.0, want_logits=False)
    for b in range(B):
        w = w_all[b]
        assert tuple(w.shape) == (32, 20, 448, 1500)
        t2 = time.time()
        words, st, en, matrix, scores = tm.force_align(w, tts[b], tok, "char", aggr, topk=topk)
        t_fa = time.time() - t2
        del w
        assert tuple(matrix.shape) == (444, 1500)
        ti, tj, rst, ren = _oracle_times(matrix, tts[b], tok, "char", rt.split_tokens_on_spaces)
        assert np.array_equal(st, rst) and np.array_equal(en, ren)          # DTW rows bit-exact on the same matrix
        jm = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
        assert np.array_equal(jump[b, :444], tj[jm])                          # fused path: the same frames
        w2, st2, en2 = tm.words_from_jump_frames(jump[b], tts[b], tok, "char")
        assert w2 == words and np.array_equal(st2, st) and np.array_equal(en2, en)
        assert len(st) == len(texts[b].split()) and st[0] == 0.0 and np.all(en >= st) and en[-1] <= 29.98 + 1e-9
        if aggr == "topk":
            assert list(sel[b]) == [l * 20 + h for _, (l, h), _ in scores]
    _log("configs[3] char   aggr=%-4s wca_force_align (640 heads x 448 x 1500 scores/select/aggregate/DTW/D2H): %.1f ms" % (aggr, t_fa * 1e3))


def test_config3_large_v2_reference_precision_mode(wca, large_v2):
    """configs[3] at full size in the reference-precision (split) mode: d = 1280 (20 heads, K doubled to 2560 / 10240), n = 448,
    F = 1500. Fused batch path == step-by-step API == oracle DTW of the device matrix (bit-exact), and the softmaxed maps agree
    with the default mode's to its f16 operand noise (the two modes run different kernels end to end)."""
    syn, tk, rt, tm, audio = _m("synthetic"), _m("tokenizer"), _m("retokenize"), _m("timing"), _m("audio")
    dims, model = large_v2
    tok = tk.get_tokenizer(True, language="English")
    B = 2
    pcm = np.stack([syn.synth_audio(900 + u, 480000) for u in range(B)])
    tts = [rt.encode(syn.synth_text(900 + u, 443), tok, "char") for u in range(B)]
    toks = np.array([[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts], dtype=np.int64)
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=7)
    pcm_d, tok_d = torch.from_numpy(pcm).cuda(), torch.from_numpy(toks).cuda()
    mels = torch.stack([audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm[b])), 80, model=model) for b in range(B)]).cuda()
    w16, _ = model.get_attentions(mels[:1], tok_d[:1], [1500], 7, 1.0, want_logits=False)
    w16 = w16[0, -1].clone()          # last decoder layer's 20 heads in the default mode
    model.set_precision("split")
    try:
        model.align_batch(pcm_d, [480000] * B, tok_d, [448] * B, [1500] * B, opts)  # warm-up (buffer growth)
        torch.cuda.synchronize()
        t1 = time.time()
        jump, sel = model.align_batch(pcm_d, [480000] * B, tok_d, [448] * B, [1500] * B, opts)
        dt = time.time() - t1
        _log("configs[3] char   aggr=topk B=%d n=448 F=1500 fused align_batch, SPLIT mode: %.1f ms per batch (%.2f utt/s)" % (B, dt * 1e3, B / dt))
        mels_s = torch.stack([audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm[b])), 80, model=model) for b in range(B)]).cuda()
        w_all, _ = model.get_attentions(mels_s, tok_d, [1500] * B, 7, 1.0, want_logits=False)
        assert (w_all[0, -1] - w16).abs().max().item() < 2e-2
        for b in range(B):
            w = w_all[b]
            words, st, en, matrix, scores = tm.force_align(w, tts[b], tok, "char", "topk", topk=10)
            del w
            ti, tj, rst, ren = _oracle_times(matrix, tts[b], tok, "char", rt.split_tokens_on_spaces)
            assert np.array_equal(st, rst) and np.array_equal(en, ren)
            jm = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
            assert np.array_equal(jump[b, :444], tj[jm])
            assert list(sel[b]) == [l * 20 + h for _, (l, h), _ in scores]
    finally:
        model.set_precision("f16")


def test_config3_large_v2_subword_1500_frames(wca, large_v2, fake_vocab):
    """--aligned_unit_type subword (infer_ali.py:164) at the 1500-frame limit: BPE tokens from a (synthetic) tiktoken
    vocabulary, n as close to 448 as the text allows, word merge by tokenizer.split_to_word_tokens."""
    syn, tk, rt, tm, audio = _m("synthetic"), _m("tokenizer"), _m("retokenize"), _m("timing"), _m("audio")
    dims, model = large_v2
    tok = tk.get_tokenizer(True, language="English", vocab_path=fake_vocab)
    text = syn.synth_text(77, 1200)
    tt = rt.encode(text, tok, "subword")
    while len(tt) > 443:
        text = text[:text.rindex(" ")]
        tt = rt.encode(text, tok, "subword")
    tokens = [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]
    assert 400 <= len(tokens) <= 448
    pcm = syn.synth_audio(77, 480000)
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    w, _ = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, 1500, medfilt_width=7)
    t0 = time.time()
    words, st, en, matrix, scores = tm.force_align(w, tt, tok, "subword", "topk", topk=15)
    _log("configs[3] subword n=%d F=1500 topk=15 wca_force_align: %.1f ms" % (len(tokens), (time.time() - t0) * 1e3))
    assert tuple(matrix.shape) == (len(tt) + 1, 1500)
    ti, tj, rst, ren = _oracle_times(matrix, tt, tok, "subword", lambda t, k, u: k.split_to_word_tokens(t))
    assert np.array_equal(st, rst) and np.array_equal(en, ren)
    assert "".join(words[:-1]) == text and words[-1] == "<|endoftext|>"
    assert len(st) == len(words) - 1 and np.all(np.diff(st) >= 0)
    # the fused path in subword mode (host tail uses the same split)
    tarr = torch.tensor([tokens], dtype=torch.int64).cuda()
    opts = model.make_opts(aggregation="topk", topk=15, sot_len=3, medfilt_width=7)
    jump, sel = model.align_batch(torch.from_numpy(pcm[None]).cuda(), [480000], tarr, [len(tokens)], [1500], opts)
    w2, st2, en2 = tm.words_from_jump_frames(jump[0], tt, tok, "subword")
    assert w2 == words and np.array_equal(st2, st) and np.array_equal(en2, en)


def test_config4_probe_all_640_heads(wca):
    """probe_oracle.py:83-90 at large-v3 size: 32 x 20 = 640 heads, n = 448, F = 1500 (1.72 GB of maps resident):
    wca_probe_heads == one force_align per head. Scores for all heads vs the oracle; paths bit-exact vs the oracle DTW
    of the device matrix for a spread of 40 heads; structural invariants for all 640."""
    from oracle import timing_ref
    tm, tk, probe = _m("timing"), _m("tokenizer"), _m("probe_oracle")
    tok = tk.get_tokenizer(True, language="English")
    L, H, n, F = 32, 20, 448, 1500
    g = torch.Generator(device="cuda").manual_seed(13)
    w = torch.empty(L, H, n, F, device="cuda")
    for l in range(L):  # layer by layer: bounds the temporary memory of randn + softmax
        w[l] = torch.softmax(torch.randn(H, n, F, device="cuda", generator=g) * 4, -1)
    eng = _m("engine").default_engine(0)
    probe.probe_heads(eng, w, 3)  # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    scores, jumps = probe.probe_heads(eng, w, 3)
    dt = time.time() - t0
    _log("configs[4] probe  L*H=640 n=448 F=1500: wca_probe_heads %.1f ms (%.1f us per head: scores + column norm + 444x1500 DTW + "
         "backtrace, D2H included)" % (dt * 1e3, dt * 1e6 / 640))
    assert scores.shape == (640,) and jumps.shape == (640, 444)
    assert np.all(jumps[:, 0] == 0) and np.all(np.diff(jumps, axis=1) >= 0) and jumps.max() <= 1499
    # scores of every head against torch (fp32 reductions, different order)
    for l in range(0, L, 4):
        ref = timing_ref.filter_attention(w[l:l + 1].cpu(), H)[1]
        for s, (_, h), _ in ref:
            assert abs(scores[l * H + h] - s) <= 4e-5 * abs(s)
    tt = [64] * (n - 5)
    for hd in list(range(0, 640, 17)) + [639, 1]:
        l, h = divmod(hd, H)
        words, st, en, matrix, _ = tm.force_align(w[l, h][None, None], tt, tok, "char", "mean", topk=1)
        ti, tj = timing_ref.dtw(-matrix)
        jm = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
        assert np.array_equal(jumps[hd], tj[jm]), hd
    del w
    torch.cuda.empty_cache()
