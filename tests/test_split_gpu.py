"""Reference-precision ("split") mode on the MI355X (`-m gpu`): wca_set_precision(e, WCA_PRECISION_SPLIT) carries every GEMM /
attention operand as an f16 (hi, lo) pair against the exact f16 weights, so the forward of timing.py:58 (`model(mel, tokens)`, an
fp32 forward) is reproduced to fp32 summation noise on the f16 matrix pipe. Kernel level: against float64 references, with the
tolerance of an fp32 computation (stated per test). End to end: against the fp32 CPU oracle at the bench configuration, on the
utterances the f16-operand mode gets wrong (bench ids 10007, 10030, 10035, 10076, 10113, 10120 -- VERDICT round 2) plus ids
100-131: scores, aggregated matrices and every word boundary."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _mods():
    m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
    return m("synthetic"), m("tokenizer"), m("retokenize"), m("timing"), m("audio")


def _split(x):
    """fp32 tensor -> [..., 2K] f16 rows [hi | lo] (what the engine's split-mode producers write)."""
    hi = x.half()
    lo = (x - hi.float()).half()
    return torch.cat([hi, lo], dim=-1).contiguous()


def _join(x2):
    k = x2.shape[-1] // 2
    return x2[..., :k].double() + x2[..., k:].double()


@pytest.fixture(scope="module")
def eng(wca):
    syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
    dims = wca.ModelDimensions(80, 1500, 384, 6, 2, 51865, 448, 384, 6, 2)
    m = wca.WhisperAMD(dims, device="cuda:0", max_batch=2, precision="f16")
    m.load_state_dict(syn.random_state_dict(dims, seed=1))
    m._bind_stream()
    return m


# ------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1500, 1152, 384), (77, 130, 64), (3000, 1024, 1024), (40, 512, 512)])
@pytest.mark.parametrize("tile", [0, 128, 256, 257])
def test_split_gemm_is_fp32_accurate(eng, lib, wca, M, N, K, tile):
    """[A_hi | A_lo] . [W | W]^T through the UNCHANGED GEMM kernels (K doubled) against float64: the error must be that of an
    fp32 GEMM (a few 1e-7 of the row's |a|.|w|), three orders below the f16-operand call on the same data."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g) * 0.7
    w = (torch.randn(N, K, generator=g) * 0.1).half()
    bias = torch.randn(N, generator=g)
    ref = a.double() @ w.double().T + bias.double()
    scale = (a.abs().double() @ w.abs().double().T).max().item()
    a2, w2 = _split(a).cuda(), torch.cat([w, w], dim=1).contiguous().cuda()
    bd = bias.cuda()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    wca._lib.check(lib.wca_test_gemm(eng._h, _vp(a2), _vp(w2), _vp(bd), _vp(out), M, N, 2 * K, 0, 1 | (tile << 8)))
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 4e-7 * scale, (err, scale)
    # pair output with the erf GELU (out_mode 4): hi + lo against float64 gelu, fp32 accuracy of the stored value
    # (the lo half sits N elements after the hi half: the 16-byte stores need N % 8 == 0, as every width of the model is)
    if N % 8 == 0:
        out2 = torch.full((M, 2 * N), float("nan"), dtype=torch.float16, device="cuda")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(a2), _vp(w2), _vp(bd), _vp(out2), M, N, 2 * K, 1, 4 | (tile << 8)))
        torch.cuda.synchronize()
        gref = torch.nn.functional.gelu(ref)
        gerr = (_join(out2.cpu()) - gref).abs().max().item()
        assert gerr < 4e-7 * scale + 3e-7 * gref.abs().max().item(), gerr
    # contrast: the default mode on the f16-rounded activations
    outh = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    ah = a.half().cuda()
    wd = w.cuda()
    wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ah), _vp(wd), _vp(bd), _vp(outh), M, N, K, 0, 1 | (tile << 8)))
    torch.cuda.synchronize()
    herr = (outh.cpu().double() - ref).abs().max().item()
    assert herr > 20 * err, (herr, err)


@pytest.mark.parametrize("M,N,K,tile", [(6144, 2048, 128, 0), (6100, 2100, 256, 0), (24000, 1024, 1024, 0), (24000, 1024, 1024, 258), (12288, 1024, 4096, 0),
                                        (24064, 3072, 1024, 0)])
def test_pair_gemm_w_tile_staged_once(eng, lib, wca, switch, M, N, K, tile):
    """The persistent 256 x 256 kernel in its SPLITW form -- A rows [hi | lo], the PLAIN W, every W K-tile staged once and its
    fragments re-used by the lo step -- against float64 at the tolerance of an fp32 GEMM, for every output mode the engine uses
    (f32 store, f32 read-modify-write, f16 store, pair store with the erf GELU), one tile per workgroup and the persistent walk
    (more tiles than CUs), ragged M / N edges; and bit-for-bit determinism. Round 5: the default LDS ring (three A slots + one W slot, the A tile
    requested three steps ahead) must give the bits of round 4's two-slot rings (switch gemm_ring = 1): same MFMAs in the same order."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a = (torch.randn(M, K, generator=g) * 0.7).cuda()
    w = (torch.randn(N, K, generator=g) * 0.1).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = a.double() @ w.double().T + bias.double()
    scale = (a.abs().double() @ w.abs().double().T).max().item()
    a2 = _split(a)
    ft = tile << 8
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(out), M, N, K, 0, 1 | ft))
    torch.cuda.synchronize()
    err = (out.double() - ref).abs().max().item()
    assert err < 4e-7 * scale, (err, scale)
    out_b = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(out_b), M, N, K, 0, 1 | ft))
    torch.cuda.synchronize()
    assert torch.equal(out, out_b)
    for ring in (1,):   # round 4's two-slot rings
        switch("gemm_ring", ring)
        out_r = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(out_r), M, N, K, 0, 1 | ft))
        torch.cuda.synchronize()
        switch("gemm_ring", 0)
        assert torch.equal(out, out_r), ring
    # read-modify-write of an f32 residual
    x0 = torch.randn(M, N, generator=torch.Generator().manual_seed(5)).cuda()
    x = x0.clone()
    wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(x), M, N, K, 0, 2 | ft))
    torch.cuda.synchronize()
    assert (x.double() - (x0.double() + ref)).abs().max().item() < 4e-7 * scale + 1e-6
    if N % 8 == 0:
        out2 = torch.full((M, 2 * N), float("nan"), dtype=torch.float16, device="cuda")
        wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(out2), M, N, K, 1, 4 | ft))
        torch.cuda.synchronize()
        gref = torch.nn.functional.gelu(ref)
        gerr = (_join(out2) - gref).abs().max().item()
        assert gerr < 4e-7 * scale + 3e-7 * gref.abs().max().item(), gerr
        outh = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(outh), M, N, K, 0, 0 | ft))
        torch.cuda.synchronize()
        assert (outh.double() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()   # one f16 rounding of the stored value
    # shapes the kernel does not take are refused, not mis-computed
    with pytest.raises(RuntimeError):
        wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(out), 300, N, K, 0, 1))
    with pytest.raises(RuntimeError):
        wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(out), M, N, 64, 0, 1))


def _attn_ref64(q, k, v, H, causal):
    B, nq, d = q.shape
    nk = k.shape[1]
    qh = q.double().view(B, nq, H, 64).permute(0, 2, 1, 3)
    kh = k.double().view(B, nk, H, 64).permute(0, 2, 1, 3)
    vh = v.double().view(B, nk, H, 64).permute(0, 2, 1, 3)
    qk = (qh @ kh.transpose(-1, -2)) * 0.125
    s = qk.clone()
    if causal:
        s = s + torch.full((nq, nk), float("-inf"), dtype=torch.float64).triu_(1)
    o = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B, nq, d)
    return o, qk


@pytest.mark.parametrize("B,H,nq,nk,causal,cap_cols", [
    (2, 3, 150, 200, 0, 0), (1, 2, 70, 70, 1, 0), (2, 6, 69, 1500, 0, 500), (1, 4, 448, 1500, 0, 1500), (1, 2, 300, 300, 1, 0),
    (1, 1, 5, 1500, 0, 145), (2, 6, 1500, 1500, 0, 0), (1, 2, 97, 33, 0, 0), (3, 1, 129, 64, 1, 0), (1, 5, 2, 65, 0, 64),
    (1, 2, 128, 64, 0, 0), (2, 2, 257, 129, 0, 0), (1, 3, 300, 1, 0, 0), (1, 2, 600, 130, 0, 0), (1, 1, 512, 64, 0, 0), (2, 1, 700, 128, 0, 0),
    (1, 2, 1030, 260, 0, 0)])
@pytest.mark.parametrize("variant", [0, 1])
def test_split_attention_is_fp32_accurate(eng, lib, wca, switch, B, H, nq, nk, causal, cap_cols, variant):
    """attn_split_kernel / attn_split32_kernel (three MFMA passes per product on hi / lo pairs, fp32 online softmax on the exact logits)
    against a float64 attention: output and captured logits at fp32 accuracy. A few large logits exercise the running-maximum update.
    variant 0: the launcher's choice (the 32x32x16 kernel for unmasked, capture-free calls of >= 64 query rows); 1: the 16x16x32 kernel
    everywhere (switch attn_split_variant = 1, wca_test_set_switch)."""
    if variant:
        if causal or cap_cols or nq < 64:
            pytest.skip("the 16x16x32 kernel is the launcher's choice here already")
        switch("attn_split_variant", variant)
    g = torch.Generator().manual_seed(nq * 13 + nk)
    d = H * 64
    q = torch.randn(B, nq, d, generator=g)
    k = torch.randn(B, nk, d, generator=g)
    v = torch.randn(B, nk, d, generator=g)
    q[:, nq // 2, :64] *= 4.0
    if nk > 700:
        k[:, 650, :64] = q[:, nq // 2, :64] * 0.5   # a late key far above everything before it: forced rescale at key tile 10
    # the kernel sees hi + lo of each operand; the reference takes exactly those values
    q2, k2, v2 = _split(q), _split(k), _split(v)
    o_ref, qk_ref = _attn_ref64(_join(q2), _join(k2), _join(v2), H, causal)
    out2 = torch.full((B, nq, 2 * d), float("nan"), dtype=torch.float16, device="cuda")
    cap_ld = (cap_cols + 3) & ~3
    cap = torch.full((B, H, nq, max(cap_ld, 4)), float("nan"), device="cuda") if cap_cols else None
    qd, kd, vd = q2.cuda(), k2.cuda(), v2.cuda()
    wca._lib.check(lib.wca_test_attention_split(eng._h, _vp(qd), _vp(kd), _vp(vd), _vp(out2), _vp(cap), cap_ld, cap_cols, B, H, nq, nk, causal))
    torch.cuda.synchronize()
    got = _join(out2.cpu())
    assert torch.isfinite(got).all()
    err = (got - o_ref).abs().max().item()
    assert err < 3e-6, err            # |o| <= max |v| ~ 4; fp32 softmax + 2^-22 operand pairs
    if cap_cols:
        cerr = (cap.cpu()[..., :cap_cols].double() - qk_ref[..., :cap_cols]).abs().max().item()
        assert cerr < 4e-7 * qk_ref.abs().max().item() + 1e-6, cerr


def test_split_layernorm_pairs(eng, lib, wca):
    g = torch.Generator().manual_seed(3)
    for d in (384, 512, 768, 1024, 1280):   # 384: the four-wide pair kernel; the others: eight elements per lane (odd chunk counts at 768 / 1280)
        x = torch.randn(37, d, generator=g) * 3 + 1
        gm, bt = torch.randn(d, generator=g), torch.randn(d, generator=g)
        out2 = torch.empty(37, 2 * d, dtype=torch.float16, device="cuda")
        xd, gd, bd = x.cuda(), gm.cuda(), bt.cuda()
        wca._lib.check(lib.wca_test_layernorm_split(eng._h, _vp(xd), _vp(gd), _vp(bd), _vp(out2), 37, d))
        torch.cuda.synchronize()
        ref = torch.nn.functional.layer_norm(x.double(), (d,), gm.double(), bt.double(), 1e-5)
        assert (_join(out2.cpu()) - ref).abs().max().item() < 3e-6


# ------------------------------------------------------------------------------- forward against the fp32 oracle
def _utt(syn, rt, tok, uid, n_samples, n_chars):
    pcm = syn.synth_audio(uid, n_samples)
    text = syn.synth_text(uid, n_chars)
    tt = rt.encode(text, tok, "char")
    return pcm, text, tt, [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]


def test_split_forward_vs_oracle_small_dims(wca):
    """Small model (256 wide, 3 + 3 layers): log-mel, encoder output, softmaxed maps, logits and word times of the split mode
    against the fp32 CPU oracle, with the tolerances of two fp32 implementations of the same forward -- 100-1000x tighter than
    the default mode's (test_e2e_gpu.py) -- and the mode switch itself (f16 -> split -> f16 gives the first result again)."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 256, 4, 3, 51865, 448, 256, 4, 3)
    sd = syn.random_state_dict(dims, seed=5, cross_qk_std=0.08)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=3, precision="f16").load_state_dict(sd)
    tok, rtok = tk.get_tokenizer(True, language="English"), tokenizer_ref.CharTokenizer()
    ref = whisper_ref.WhisperRef(sd, dims)
    pcm, text, tt, tokens = _utt(syn, rt, tok, 7, 80000, 40)
    F = len(pcm) // 320
    mel16 = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    w16, _ = tm.get_attentions(mel16, torch.tensor(tokens).cuda(), model, tok, F, medfilt_width=3)
    assert model.precision == "f16"
    model.set_precision("split")
    assert model.precision == "split"
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    ref_mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm)), audio.mel_filters(80))
    mel_err = (mel.cpu() - ref_mel).abs().max().item()
    enc = model.encode(ref_mel[None].cuda()).cpu()[0]
    renc = ref.encoder(ref_mel[None])[0]
    enc_err = (enc - renc).abs().max().item()
    w, logits = tm.get_attentions(ref_mel.cuda(), torch.tensor(tokens).cuda(), model, tok, F, medfilt_width=3)
    rw, rlogits = timing_ref.get_attentions(ref_mel, torch.tensor(tokens), ref, F, 3, 1.0)
    w_err = (w.cpu() - rw).abs().max().item()
    l_err = ((logits.cpu() - rlogits).abs().max() / rlogits.abs().max()).item()
    w16_err = (w16.cpu() - rw).abs().max().item()
    print("split vs oracle: log-mel %.2e, encoder %.2e, maps %.2e (f16 mode %.2e), logits %.2e rel" % (mel_err, enc_err, w_err, w16_err, l_err))
    # measured on MI355X: log-mel 2.0e-6, encoder 1.2e-6, maps 2.3e-7 (default mode: 1.6e-4), logits 8e-7
    assert mel_err < 1e-5
    assert enc_err < 1e-5          # LayerNorm-ed outputs, O(1)
    assert w_err < 2e-6            # softmaxed maps in [0, 1]
    assert l_err < 1e-5
    words, st, en, matrix, scores = tm.force_align(w, tt, tok, "char", "topk", topk=4)
    rwords, rst, ren, rmatrix, rscores = timing_ref.force_align(rw, tt, rtok, "char", "topk", 4)
    assert words == rwords and np.array_equal(st, rst) and np.array_equal(en, ren)
    assert [lh for _, lh, _ in scores] == [lh for _, lh, _ in rscores]
    assert max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(scores, rscores)) < 1e-5
    # fused batch path in split mode == the step-by-step API in split mode (ragged batch)
    specs = [(31, 48000, 25), (32, 80000, 40), (33, 32000, 12)]
    utts = [_utt(syn, rt, tok, u, n, c) for u, n, c in specs]
    n_max, smax = max(len(u[3]) for u in utts), max(len(u[0]) for u in utts)
    pb = np.zeros((3, smax), dtype=np.float32)
    tarr = np.full((3, n_max), tok.eot, dtype=np.int64)
    for i, (p, _, _, toks) in enumerate(utts):
        pb[i, :len(p)] = p
        tarr[i, :len(toks)] = toks
    n_samples, n_tok, frames = [len(u[0]) for u in utts], [len(u[3]) for u in utts], [len(u[0]) // 320 for u in utts]
    opts = model.make_opts(aggregation="topk", topk=4, sot_len=3, medfilt_width=3)
    jump, sel = model.align_batch(torch.from_numpy(pb).cuda(), n_samples, torch.from_numpy(tarr).cuda(), n_tok, frames, opts)
    for i, (p, text_i, tt_i, toks) in enumerate(utts):
        rmel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(p)), audio.mel_filters(80))
        rw_i, _ = timing_ref.get_attentions(rmel, torch.tensor(toks), ref, frames[i], 3, 1.0)
        _rw, rst_i, ren_i, _m, rsc_i = timing_ref.force_align(rw_i, tt_i, rtok, "char", "topk", 4)
        _w2, st2, en2 = tm.words_from_jump_frames(jump[i], tt_i, tok, "char")
        assert np.array_equal(st2, rst_i) and np.array_equal(en2, ren_i), (i, st2, rst_i)
        assert list(sel[i]) == [l * dims.n_text_head + h for _, (l, h), _ in rsc_i]
    # back to the default mode: the arena is re-created, the result is the first one again
    model.set_precision("f16")
    w16b, _ = tm.get_attentions(mel16, torch.tensor(tokens).cuda(), model, tok, F, medfilt_width=3)
    assert torch.equal(w16b, w16)
    del model


OFFENDER_IDS = [10007, 10030, 10035, 10076, 10113, 10120]   # outside one frame in round 2's f16-operand mode (profiles/r02_parity_leg_128utt.json)


def test_split_mode_closes_the_headline_parity_gap(wca):
    """The bench configuration (whisper-medium dims, peaky seeded weights, 10 s audio, 64 chars, top-10, medfilt 3) through the
    FUSED wca_align_batch at B = 64 with EVERY site split (wca_set_precision SPLIT), on the six utterances round 2's f16-operand
    mode put outside the tolerance plus ids 100-131, against the fp32 CPU oracle (word times: the committed oracle fixture
    tests/golden/oracle_word_times_medium_peaky.npz; scores / matrices of the six: the live oracle):
      * every word boundary within one 20 ms frame, no exception -- including 10076, whose 10th / 11th oracle scores are 4e-6 apart;
      * selection scores and aggregated matrix of the step-by-step API within 1e-5 / 2e-5 relative of the oracle's.
    The same batch in the f16 mode is run for contrast and its offenders are printed (which utterances those are changes with
    every last-bit change of the f16 forward)."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    syn, tk, rt, tm, audio = _mods()
    dims = wca.dims_for("medium")
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
    B = 64
    ids = OFFENDER_IDS + list(range(100, 132))
    fill = (ids * ((B + len(ids) - 1) // len(ids)))[:B]
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=B, precision="f16").load_state_dict(sd)
    ref = whisper_ref.WhisperRef(sd, dims)
    tok, rtok = tk.get_tokenizer(True, language="English"), tokenizer_ref.CharTokenizer()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_word_times_medium_peaky.npz"))
    utts = [_utt(syn, rt, tok, u, 160000, 64) for u in fill]
    pcm = np.stack([u[0] for u in utts])
    tarr = np.asarray([u[3] for u in utts], dtype=np.int64)
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    args = (torch.from_numpy(pcm).cuda(), [160000] * B, torch.from_numpy(tarr).cuda(), [69] * B, [500] * B, opts)
    jump16, sel16 = model.align_batch(*args)
    model.set_precision("split")
    jump, sel = model.align_batch(*args)
    LH = dims.n_text_layer * dims.n_text_head
    total = ident = 0
    off_split, off_f16 = [], []
    max_dscore = max_dmatrix = 0.0
    for i, uid in enumerate(ids):
        p, text, tt, tokens = utts[i]
        rst, ren = gold["st_%d" % uid], gold["en_%d" % uid]
        sc = np.sort(gold["sc_%d" % uid].astype(np.float64))
        gap = (sc[-10] - sc[-11]) / abs(sc[-10])     # relative distance of the oracle's 10th and 11th head
        for jj, store in ((jump, off_split), (jump16, off_f16)):
            words, st, en = tm.words_from_jump_frames(jj[i], tt, tok, "char")
            n_off = int(np.sum(np.abs(np.asarray(st) - rst) > 0.02 + 1e-9) + np.sum(np.abs(np.asarray(en) - ren) > 0.02 + 1e-9))
            if jj is jump:
                total += 2 * len(st)
                ident += int((np.asarray(st) == rst).sum() + (np.asarray(en) == ren).sum())
            if n_off:
                store.append((uid, n_off, gap))
        if uid in OFFENDER_IDS:
            # step-by-step API in split mode on the utterance alone: scores and matrix against the LIVE oracle's
            mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(p)), audio.mel_filters(80))
            rw, _ = timing_ref.get_attentions(mel, torch.tensor(tokens), ref, 500, 3, 1.0)
            rwords, rst_l, ren_l, rmatrix, rscores = timing_ref.force_align(rw, tt, rtok, "char", "topk", 10)
            w, _ = tm.get_attentions(mel.cuda(), torch.tensor(tokens).cuda(), model, tok, 500, medfilt_width=3)
            _wd, _s, _e, matrix, scores = tm.force_align(w, tt, tok, "char", "topk", topk=10)
            gs = {lh: s_ for s_, lh, _n in tm.filter_attention(w, LH)[1]}
            rs = {lh: s_ for s_, lh, _n in timing_ref.filter_attention(rw, LH)[1]}
            dscore = max(abs(gs[lh] - rs[lh]) / abs(rs[lh]) for lh in rs)
            max_dscore = max(max_dscore, dscore)
            if [lh for _, lh, _ in scores] == [lh for _, lh, _ in rscores]:
                dm = ((matrix.cpu() - rmatrix).norm() / rmatrix.norm()).item()
                max_dmatrix = max(max_dmatrix, dm)
            print("utt %d: oracle 10th/11th score gap %.2e, max rel score deviation %.2e" % (uid, gap, dscore))
    print("split mode, medium B=64 fused: %d boundaries over %d utterances, identical %d, utterances with a boundary outside one frame: split %s | f16 %s; "
          "max rel |dscore| %.2e, rel |dmatrix| %.2e" % (total, len(ids), ident, off_split, off_f16, max_dscore, max_dmatrix))
    assert max_dscore < 1e-5, max_dscore
    assert max_dmatrix < 2e-5, max_dmatrix
    assert not off_split, off_split
    del model


SITE_ROWS = [
    # (sites, first encoder block of the ENC_* bits)
    ("capture", 0), ("dec", 0), ("cross_kv", 0), ("cross_kv,capture", 0), ("dec,cross_kv,capture", 0), ("logmel,conv", 0), ("enc_attn", 0),
    ("enc_gemm", 0), ("enc_gemm,enc_attn", 2), ("enc_attn,dec,cross_kv,capture", 1), ("enc_gemm,capture", 1), ("logmel,conv,enc_gemm,enc_attn", 0),
]


def test_precision_sites_seams_small_dims(wca):
    """wca_set_precision_sites: every stage can be switched to (hi, lo) operand pairs on its own; the seams between split and
    single-precision stages need no conversion pass (a producer writes pairs exactly when its consumer is split; a single-precision
    consumer of a pair buffer reads the hi halves). Small model (256 wide, 3 + 3 layers), step-by-step API and the fused batch path:
      * mask ALL == wca_set_precision(SPLIT), mask 0 == F16, bit for bit;
      * every mixed row gives finite maps whose distance to the fp32 oracle is at most the f16 mode's (plus noise) and at least ~the
        full split mode's, and the fused batch path agrees with the step-by-step API of the same row;
      * the decoder-side rows (CAPTURE + CROSS_KV + DEC) bring the captured logits' contribution down: maps error well below f16's."""
    from oracle import timing_ref, whisper_ref
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 256, 4, 3, 51865, 448, 256, 4, 3)
    sd = syn.random_state_dict(dims, seed=5, cross_qk_std=0.08)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=3, precision="f16").load_state_dict(sd)
    tok = tk.get_tokenizer(True, language="English")
    ref = whisper_ref.WhisperRef(sd, dims)
    pcm, text, tt, tokens = _utt(syn, rt, tok, 7, 80000, 40)
    F = len(pcm) // 320
    ref_mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm)), audio.mel_filters(80))
    rw, _ = timing_ref.get_attentions(ref_mel, torch.tensor(tokens), ref, F, 3, 1.0)
    tdev = torch.tensor(tokens).cuda()

    def maps():
        w, _ = tm.get_attentions(ref_mel.cuda(), tdev, model, tok, F, medfilt_width=3)
        return w.cpu()

    w_f16 = maps()
    model.set_precision("split")
    assert model.precision_sites == (["logmel", "conv", "enc_gemm", "enc_attn", "cross_kv", "dec", "capture"], 0)
    w_split = maps()
    model.set_precision_sites("all", 0)
    assert model.precision == "split" and torch.equal(maps(), w_split)
    model.set_precision_sites(0)
    assert model.precision == "f16" and torch.equal(maps(), w_f16)
    e16, esp = (w_f16 - rw).abs().max().item(), (w_split - rw).abs().max().item()
    # fused batch path, ragged batch
    specs = [(31, 48000, 25), (32, 80000, 40), (33, 32000, 12)]
    utts = [_utt(syn, rt, tok, u, n, c) for u, n, c in specs]
    n_max, smax = max(len(u[3]) for u in utts), max(len(u[0]) for u in utts)
    pb = np.zeros((3, smax), dtype=np.float32)
    tarr = np.full((3, n_max), tok.eot, dtype=np.int64)
    for i, (p, _, _, toks) in enumerate(utts):
        pb[i, :len(p)] = p
        tarr[i, :len(toks)] = toks
    n_samples, n_tok, frames = [len(u[0]) for u in utts], [len(u[3]) for u in utts], [len(u[0]) // 320 for u in utts]
    opts = model.make_opts(aggregation="topk", topk=4, sot_len=3, medfilt_width=3)
    report = []
    for sites, first in SITE_ROWS:
        model.set_precision_sites(sites, first)
        assert model.precision == "mixed" and sorted(model.precision_sites[0]) == sorted(sites.split(",")) and model.precision_sites[1] == first
        w = maps()
        assert torch.isfinite(w).all(), sites
        err = (w - rw).abs().max().item()
        report.append("%s@%d %.1e" % (sites, first, err))
        assert err <= 1.5 * e16 + 1e-6, (sites, err, e16)
        assert err >= 0.2 * esp, (sites, err, esp)
        if "capture" in sites and "cross_kv" in sites and "dec" in sites and "enc" not in sites:
            assert err < 0.7 * e16, (sites, err, e16)   # the decoder side's own rounding is gone; the encoder's remains
        jump, sel = model.align_batch(torch.from_numpy(pb).cuda(), n_samples, torch.from_numpy(tarr).cuda(), n_tok, frames, opts)
        # the step-by-step API on the SAME micro-batch (same kernels on the same shapes: bit-identical captured logits; a single-utterance
        # forward takes other GEMM kernels, and in a row whose decoder runs on f16 operands a last-bit difference may flip a DTW near-tie)
        mel_b = model.log_mel(torch.from_numpy(pb).cuda(), n_samples)
        wb, _ = model.get_attentions(mel_b, torch.from_numpy(tarr).cuda(), frames, medfilt_width=3, n_tok=n_tok, want_logits=False)
        for i, (p, _t, tt_i, toks) in enumerate(utts):
            w_i = wb[i, :, :, :n_tok[i], :frames[i]].contiguous()
            _wd, st_i, en_i, _m, sc_i = tm.force_align(w_i, tt_i, tok, "char", "topk", topk=4)
            _w2, st2, en2 = tm.words_from_jump_frames(jump[i], tt_i, tok, "char")
            assert np.array_equal(st2, st_i) and np.array_equal(en2, en_i), (sites, i)
            assert list(sel[i]) == [l * dims.n_text_head + h for _, (l, h), _ in sc_i], (sites, i)
    print("maps vs fp32 oracle: f16 %.1e, split %.1e; mixed: %s" % (e16, esp, "; ".join(report)))
    with pytest.raises(ValueError):
        model.set_precision_sites("nope")
    with pytest.raises(RuntimeError):
        model.set_precision_sites("enc_gemm", 99)
    model.set_precision("f16")
    assert torch.equal(maps(), w_f16)
    del model


def test_failed_precision_switch_leaves_a_working_engine(wca, switch):
    """wca_set_precision allocates the NEW arena (and the K-doubled weight copies) before it releases the old ones and
    commits the mode only when both allocations succeeded. With an allocation failure injected (wca_test_set_switch("fail_precision_alloc", 1)) the switch
    must return WCA_ERR_HIP, the engine must stay in its previous mode with every arena pointer intact -- same maps bit for bit -- and a
    later switch must work; the same from the split mode back to f16."""
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 256, 4, 2, 51865, 448, 256, 4, 2)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=2, precision="f16").load_state_dict(syn.random_state_dict(dims, seed=9, cross_qk_std=0.08))
    tok = tk.get_tokenizer(True, language="English")
    pcm, text, tt, tokens = _utt(syn, rt, tok, 3, 64000, 30)
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    tdev = torch.tensor(tokens).cuda()

    def maps():
        return tm.get_attentions(mel, tdev, model, tok, 200, medfilt_width=3)[0].cpu()

    w16 = maps()
    switch("fail_precision_alloc", 1)
    with pytest.raises(wca._lib.WcaError, match="keeps its previous mode"):
        model.set_precision("split")
    assert model.precision == "f16" and torch.equal(maps(), w16)
    switch("fail_precision_alloc", 0)
    model.set_precision("split")
    wsp = maps()
    assert model.precision == "split" and (wsp - w16).abs().max().item() < 1e-2 and not torch.equal(wsp, w16)
    switch("fail_precision_alloc", 1)
    with pytest.raises(wca._lib.WcaError, match="keeps its previous mode"):
        model.set_precision("f16")
    assert model.precision == "split" and torch.equal(maps(), wsp)
    switch("fail_precision_alloc", 0)
    model.set_precision("f16")
    assert torch.equal(maps(), w16)
    del model


def test_new_engine_is_in_the_contract_mode(wca, lib):
    """the safe mode is the constructor's. wca_engine_create and WhisperAMD() start with every site on pairs; the fast f16 mode
    is the explicit opt-in (wca_engine_create_ex(..., WCA_PRECISION_F16) / WhisperAMD(precision='f16'))."""
    import ctypes as C
    dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
    m = wca.WhisperAMD(dims, device="cuda:0", max_batch=1)
    assert m.precision == "split" and m.precision_sites[0] == ["logmel", "conv", "enc_gemm", "enc_attn", "cross_kv", "dec", "capture"]
    m16 = wca.WhisperAMD(dims, device="cuda:0", max_batch=1, precision="f16")
    assert m16.precision == "f16" and m16.precision_sites[0] == []
    cd = wca._lib.ModelDims(**__import__("dataclasses").asdict(dims))
    h = C.c_void_p(0)
    wca._lib.check(lib.wca_engine_create(C.byref(cd), 0, 1, C.byref(h)))
    assert lib.wca_get_precision(h) == 1   # WCA_PRECISION_REFERENCE
    lib.wca_engine_destroy(h)
    h = C.c_void_p(0)
    wca._lib.check(lib.wca_engine_create_ex(C.byref(cd), 0, 1, 0, C.byref(h)))
    assert lib.wca_get_precision(h) == 0
    lib.wca_engine_destroy(h)
    with pytest.raises(wca._lib.WcaError):
        wca._lib.check(lib.wca_engine_create_ex(C.byref(cd), 0, 1, 7, C.byref(h)))
    del m, m16


def test_fp32_checkpoint_that_is_not_f16_exact_runs_at_fp32_accuracy(wca):
    """VERDICT r4 item 4 / "missing" 3 (/root/reference/infer_ali.py:36-37: whisper.load_model upcasts the checkpoint to fp32 parameters, so a
    fine-tuned fp32 .pt runs in true fp32 there). The engine stores weight matrices f16; an fp32 state dict whose values are NOT f16-representable
    keeps its remainders in the W_lo slab and the contract mode multiplies the extra term A_hi W_lo^T: attention maps at fp32 accuracy against the
    fp32 oracle ON THE TRUE fp32 WEIGHTS and identical word times, where the f16-rounded weights are visibly further away. `weights_inexact`
    counts what was not exact; allow_rounded_weights=True gives exactly the rounded checkpoint's bits; an fp32 state dict whose values ARE
    f16-representable (an openai checkpoint upcast by whisper.load_model) gives the f16 checkpoint's bits and allocates nothing; the f16 mode runs."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 256, 4, 2, 51865, 448, 256, 4, 2)
    sd32 = syn.random_state_dict(dims, seed=11, cross_qk_std=0.08, dtype=torch.float32)   # N(0, 0.02) fp32 draws: almost none is an f16 value
    is_mat = lambda k, v: "weight" in k and v.ndim >= 2 and "positional" not in k   # noqa: E731
    sd16 = {k: (v.half() if v.dtype == torch.float32 and is_mat(k, v) else v) for k, v in sd32.items()}
    sd32_exact = {k: v.float() for k, v in sd16.items()}
    tok = tk.get_tokenizer(True, language="English")
    pcm, text, tt, tokens = _utt(syn, rt, tok, 5, 64000, 30)
    tdev = torch.tensor(tokens).cuda()

    def maps(model):
        mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
        return tm.get_attentions(mel, tdev, model, tok, 200, medfilt_width=3)[0].cpu()

    m = wca.WhisperAMD(dims, device="cuda:0", max_batch=3).load_state_dict(sd32)
    n_t, n_v, first = m.weights_inexact
    n_mats = sum(1 for k, v in sd32.items() if is_mat(k, v))
    assert m.precision == "split" and n_t == n_mats and n_v > 0.9 * sum(v.numel() for k, v in sd32.items() if is_mat(k, v)) and first
    w_exact32 = maps(m)
    m_allow = wca.WhisperAMD(dims, device="cuda:0", max_batch=1).load_state_dict(sd32, allow_rounded_weights=True)
    m_16 = wca.WhisperAMD(dims, device="cuda:0", max_batch=1).load_state_dict(sd16)
    assert m_allow.weights_inexact[0] == n_mats and m_16.weights_inexact == (0, 0, "")
    w_allow, w_16 = maps(m_allow), maps(m_16)
    assert torch.equal(w_allow, w_16)                      # the opt-in: exactly the rounded checkpoint
    m_ex = wca.WhisperAMD(dims, device="cuda:0", max_batch=1).load_state_dict(sd32_exact)
    assert m_ex.weights_inexact == (0, 0, "") and torch.equal(maps(m_ex), w_16)
    m_allow.load_state_dict(sd16)                           # exact tensors over inexact ones clear the record
    assert m_allow.weights_inexact == (0, 0, "") and torch.equal(maps(m_allow), w_16)
    # against the fp32 oracle
    mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm)), audio.mel_filters(80))
    rw32, _ = timing_ref.get_attentions(mel, torch.tensor(tokens), whisper_ref.WhisperRef(sd32, dims), 200, 3, 1.0)
    rw16, _ = timing_ref.get_attentions(mel, torch.tensor(tokens), whisper_ref.WhisperRef(sd16, dims), 200, 3, 1.0)
    e32, e16, e_rounded = (w_exact32 - rw32).abs().max().item(), (w_16 - rw16).abs().max().item(), (w_16 - rw32).abs().max().item()
    print("maps vs the fp32 oracle: fp32 checkpoint through the W_lo slab %.2e, f16 checkpoint %.2e, fp32 checkpoint through ROUNDED weights %.2e" % (e32, e16, e_rounded))
    assert e32 < 5e-6 and e16 < 5e-6 and e_rounded > 10 * e32
    # fused path on three utterances of the fp32 checkpoint: word times identical to the oracle's on the true fp32 weights
    rtok = tokenizer_ref.CharTokenizer()
    ref32 = whisper_ref.WhisperRef(sd32, dims)
    utts = [_utt(syn, rt, tok, 70 + u, 64000 + 16000 * u, 24 + 6 * u) for u in range(3)]
    smax, n_max = max(len(u[0]) for u in utts), max(len(u[3]) for u in utts)
    pcm3 = np.zeros((3, smax), dtype=np.float32)
    tarr = np.full((3, n_max), tok.eot, dtype=np.int64)
    for i, u in enumerate(utts):
        pcm3[i, :len(u[0])] = u[0]
        tarr[i, :len(u[3])] = u[3]
    opts = m.make_opts(aggregation="topk", topk=4, sot_len=3, medfilt_width=3)
    jump, _sel = m.align_batch(torch.from_numpy(pcm3).cuda(), [len(u[0]) for u in utts], torch.from_numpy(tarr).cuda(), [len(u[3]) for u in utts],
                               [len(u[0]) // 320 for u in utts], opts)
    for i, (p, _text, tti, toks) in enumerate(utts):
        melr = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(p)), audio.mel_filters(80))
        rw, _ = timing_ref.get_attentions(melr, torch.tensor(toks), ref32, len(p) // 320, 3, 1.0)
        _rwd, rst, ren, _m, _s = timing_ref.force_align(rw, tti, rtok, "char", "topk", 4)
        _w, st, en = tm.words_from_jump_frames(jump[i], tti, tok, "char")
        assert np.array_equal(np.asarray(st), rst) and np.array_equal(np.asarray(en), ren), i
    m.set_precision("f16")
    assert torch.isfinite(maps(m)).all()                    # the f16 mode ignores the remainders
    del m, m_allow, m_16, m_ex


def test_w_lo_term_on_the_bench_sized_kernels(wca):
    """The W_lo term through the kernels the headline configuration uses: whisper-medium WIDTH (1024, 16 heads), one encoder and one decoder layer, 8
    utterances (M = 12 000 rows: the accumulating out-projection / fc2 keep the persistent pair kernel and get a second accumulating launch; QKV / fc1 /
    cross-K/V take the K-doubled call with the pre-activation addend), fp32 weights that are not f16-exact: the encoder output against the fp32 oracle
    on the true weights at fp32 accuracy, the rounded weights visibly off."""
    from oracle import whisper_ref
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 1024, 16, 1, 51865, 448, 1024, 16, 1)
    sd32 = syn.random_state_dict(dims, seed=12, cross_qk_std=0.08, dtype=torch.float32)
    B = 8
    m = wca.WhisperAMD(dims, device="cuda:0", max_batch=B).load_state_dict(sd32)
    m_r = wca.WhisperAMD(dims, device="cuda:0", max_batch=B).load_state_dict(sd32, allow_rounded_weights=True)
    pcm = np.stack([syn.synth_audio(400 + u, 96000) for u in range(B)])
    mel = m.log_mel(torch.from_numpy(pcm).cuda())
    enc, enc_r = m.encode(mel).cpu(), m_r.encode(mel).cpu()
    ref = whisper_ref.WhisperRef(sd32, dims)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    want = ref.encoder(mel[:2].cpu())
    err = (enc[:2] - want).abs().max().item() / want.abs().max().item()
    err_r = (enc_r[:2] - want).abs().max().item() / want.abs().max().item()
    print("encoder output vs the fp32 oracle on fp32 weights (medium width, B = 8): W_lo slab %.2e, rounded weights %.2e" % (err, err_r))
    assert err < 3e-6 and err_r > 10 * err
    del m, m_r




def test_pair_gemm_ring_race_screen(eng, lib, wca, switch):
    """A new synchronisation structure is screened for races (cdna_hip_programming.md, "a sync-structure edit makes a NEW template"): the three-A-slot ring of the pair GEMM
    (A tile requested two steps ahead and left in flight across the step's barrier by a counted vmcnt) on shapes with many tiles per workgroup and ragged edges, 40 launches
    each while a second stream keeps the memory system busy -- every launch must give the bits of round 4's two-slot rings."""
    side = torch.cuda.Stream()
    noise_a = torch.randn(64 * 1024 * 1024 // 4, device="cuda")
    for M, N, K in ((24064, 3072, 1024), (30000, 1024, 512), (17000, 2300, 128)):
        g = torch.Generator().manual_seed(M + N)
        a2 = _split((torch.randn(M, K, generator=g) * 0.7).cuda())
        w = (torch.randn(N, K, generator=g) * 0.1).half().cuda()
        bias = torch.randn(N, generator=g).cuda()
        switch("gemm_ring", 1)
        want = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(want), M, N, K, 0, 1))
        torch.cuda.synchronize()
        switch("gemm_ring", 0)
        out = torch.empty_like(want)
        for it in range(40):
            out.fill_(float("nan"))
            with torch.cuda.stream(side):
                noise_b = noise_a * 1.0001 + float(it)   # unrelated traffic beside the launch
            wca._lib.check(lib.wca_test_gemm_pairs(eng._h, _vp(a2), _vp(w), _vp(bias), _vp(out), M, N, K, 0, 1))
            torch.cuda.synchronize()
            assert torch.equal(out, want), (M, N, K, it)
        del noise_b
