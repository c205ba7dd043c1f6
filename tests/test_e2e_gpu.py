"""End-to-end parity on the MI355X (`-m gpu`): the engine's forward (encoder, decoder with captured
cross-attention logits, post-processing, DTW) against the CPU oracle on seeded tiny-dims models, and the
fused wca_align_batch path against the step-by-step drop-in API."""
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
    return m("synthetic"), m("tokenizer"), m("retokenize"), m("timing"), m("audio")


@pytest.fixture(scope="module")
def setup(wca):
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 256, 4, 3, 51865, 448, 256, 4, 3)
    sd = syn.random_state_dict(dims, seed=5, cross_qk_std=0.08)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=3, precision="f16").load_state_dict(sd)
    tok = tk.get_tokenizer(True, language="English")
    return dims, sd, model, tok


def _utt(syn, rt, tok, uid, n_samples, n_chars):
    pcm = syn.synth_audio(uid, n_samples)
    text = syn.synth_text(uid, n_chars)
    tt = rt.encode(text, tok, "char")
    tokens = [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]
    return pcm, text, tt, tokens


def test_encoder_vs_oracle(wca, setup):
    from oracle import whisper_ref
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    pcm = np.stack([syn.synth_audio(u, 48000) for u in range(2)])
    mel = model.log_mel(torch.from_numpy(pcm).cuda())
    got = model.encode(mel).cpu()
    ref = whisper_ref.WhisperRef(sd, dims).encoder(mel.cpu())
    # f16 GEMM operands / f16 attention probabilities vs fp32: LayerNorm-ed outputs are O(1)
    err = (got - ref).abs()
    assert err.max().item() < 3e-2 and err.mean().item() < 2e-3, (err.max().item(), err.mean().item())


@pytest.mark.parametrize("medfilt,secs,chars", [(3, 3.0, 20), (7, 5.0, 40), (1, 2.0, 9)])
def test_get_attentions_vs_oracle(wca, setup, medfilt, secs, chars):
    from oracle import timing_ref, whisper_ref
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    pcm, text, tt, tokens = _utt(syn, rt, tok, 7, int(secs * 16000), chars)
    max_frames = len(pcm) // 320
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    w, logits = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, max_frames, medfilt_width=medfilt)
    ref = whisper_ref.WhisperRef(sd, dims)
    rw, rlogits = timing_ref.get_attentions(mel.cpu(), torch.tensor(tokens), ref, max_frames, medfilt, 1.0)
    assert tuple(w.shape) == tuple(rw.shape) == (dims.n_text_layer, dims.n_text_head, len(tokens), max_frames)
    # softmaxed maps (values in [0,1]); f16 operand rounding in the forward
    assert (w.cpu() - rw).abs().max().item() < 5e-3
    # logits: O(1) values through 3 decoder layers
    assert (logits.cpu() - rlogits).abs().max().item() < 5e-2
    # rows sum to one
    assert torch.allclose(w.sum(-1), torch.ones_like(w.sum(-1)), atol=1e-5)


def test_force_align_word_times_within_one_frame(wca, setup):
    """north_star tolerance: EVERY word start/end within one 20 ms frame of the CPU reference path (fp32 forward,
    timing.py:58) -- 12 utterances, char and top-k / mean aggregation."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    ref = whisper_ref.WhisperRef(sd, dims)
    rtok = tokenizer_ref.CharTokenizer()
    n_words = n_ident = 0
    offenders = []
    for uid in range(12):
        aggr, k = (("topk", 4), ("topk", 10), ("mean", -1))[uid % 3]
        pcm, text, tt, tokens = _utt(syn, rt, tok, 20 + uid, 48000 + 16000 * (uid % 4), 20 + 5 * (uid % 5))
        max_frames = len(pcm) // 320
        mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
        w, _ = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, max_frames, medfilt_width=3)
        words, st, en, matrix, scores = tm.force_align(w, tt, tok, "char", aggr, topk=k)
        rw, _ = timing_ref.get_attentions(mel.cpu(), torch.tensor(tokens), ref, max_frames, 3, 1.0)
        rwords, rst, ren, rmatrix, rscores = timing_ref.force_align(rw, tt, rtok, "char", aggr, k)
        assert words == rwords
        # given the SAME matrix the GPU DTW is bit-exact
        ti, tj = timing_ref.dtw(-matrix)
        jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
        wb = np.pad(np.cumsum([len(t) for t in rt.split_tokens_on_spaces(tt + [tok.eot], tok, "char")[1][:-1]]), (1, 0))
        assert np.array_equal(st, (tj[jumps] / 50)[wb[:-1]]) and np.array_equal(en, (tj[jumps] / 50)[wb[1:]])
        for a, b, kind in ((st, rst, "start"), (en, ren, "end")):
            n_words += len(a)
            n_ident += int((a == b).sum())
            d = np.abs(np.asarray(a) - np.asarray(b))
            offenders += [(uid, kind, i, float(a[i]), float(b[i])) for i in np.nonzero(d > 0.02 + 1e-9)[0]]
    print("boundaries %d identical %d outside-one-frame %d" % (n_words, n_ident, len(offenders)))
    # end to end (f16-operand forward vs fp32 CPU forward): every boundary within one frame
    assert not offenders, offenders


def test_align_batch_matches_stepwise_api(wca, setup):
    """The fused micro-batch path (ragged lengths; softmaxed maps never materialised, selected heads re-derived from the
    captured logits) must give EXACTLY the jump frames of the step-by-step API (get_attentions -> force_align) run on the
    same micro-batch: same kernels on the same shapes, so the captured logits are bit-identical and so is everything after
    them. (A single-utterance forward takes other GEMM kernels -- M <= 64 rows go to the weight-streaming one, whose K
    split sums in another order -- so its logits differ in the last bits, which a DTW near-tie may amplify.)"""
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    specs = [(31, 48000, 25), (32, 80000, 40), (33, 32000, 12)]
    utts = [_utt(syn, rt, tok, u, n, c) for u, n, c in specs]
    n_max = max(len(u[3]) for u in utts)
    smax = max(len(u[0]) for u in utts)
    pcm = np.zeros((3, smax), dtype=np.float32)
    tarr = np.full((3, n_max), tok.eot, dtype=np.int64)
    for i, (p, _, _, toks) in enumerate(utts):
        pcm[i, :len(p)] = p
        tarr[i, :len(toks)] = toks
    n_samples, n_tok, frames = [len(u[0]) for u in utts], [len(u[3]) for u in utts], [len(u[0]) // 320 for u in utts]
    opts = model.make_opts(aggregation="topk", topk=4, sot_len=3, medfilt_width=3)
    jump, sel = model.align_batch(torch.from_numpy(pcm).cuda(), n_samples, torch.from_numpy(tarr).cuda(), n_tok, frames, opts)
    mel = model.log_mel(torch.from_numpy(pcm).cuda(), n_samples)
    wb, _ = model.get_attentions(mel, torch.from_numpy(tarr).cuda(), frames, medfilt_width=3, n_tok=n_tok, want_logits=False)
    for i, (p, text, tt, toks) in enumerate(utts):
        w = wb[i, :, :, :n_tok[i], :frames[i]].contiguous()
        words, st, en, matrix, scores = tm.force_align(w, tt, tok, "char", "topk", topk=4)
        w2, st2, en2 = tm.words_from_jump_frames(jump[i], tt, tok, "char")
        assert w2 == words
        assert np.array_equal(st2, st) and np.array_equal(en2, en)
        assert list(sel[i]) == [l * dims.n_text_head + h for _, (l, h), _ in scores]
        # the drop-in single-utterance call takes other GEMM kernels (another fp32 summation order); in this f16-operand mode a
        # last-bit difference of one GEMM moves f16 roundings downstream, so the maps agree to a fraction of the mode's own
        # 3e-3 operand noise (measured 1e-4 ... 2e-4), not to fp32 precision (the split mode does: tests/test_batch_invariance_gpu.py)
        mel1 = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(p)), 80, model=model)
        w1, _ = tm.get_attentions(mel1, torch.tensor(toks).cuda(), model, tok, frames[i], medfilt_width=3)
        assert (w1 - w).abs().max().item() < 1e-3


def test_cu_partition_changes_nothing_but_the_streams(wca, setup):
    """wca_set_cu_partition (the co-scheduling experiment, profiles/r04_cu_partition.txt): phase 2 on CU-masked streams owning 64 CUs,
    phase 1 on the rest with its persistent GEMM grids sized to match -- the results must be those of the unpartitioned engine bit for
    bit (same kernels, same arithmetic, another workgroup -> tile walk), two batches in flight included; 0 lifts it; bad sizes are refused."""
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    specs = [(41, 48000, 25), (42, 80000, 40), (43, 32000, 12)]
    utts = [_utt(syn, rt, tok, u, n, c) for u, n, c in specs]
    n_max, smax = max(len(u[3]) for u in utts), max(len(u[0]) for u in utts)
    pcm = np.zeros((3, smax), dtype=np.float32)
    tarr = np.full((3, n_max), tok.eot, dtype=np.int64)
    for i, (p, _, _, toks) in enumerate(utts):
        pcm[i, :len(p)] = p
        tarr[i, :len(toks)] = toks
    args = (torch.from_numpy(pcm).cuda(), [len(u[0]) for u in utts], torch.from_numpy(tarr).cuda(), [len(u[3]) for u in utts],
            [len(u[0]) // 320 for u in utts], model.make_opts(aggregation="topk", topk=4, sot_len=3, medfilt_width=3))
    jump0, sel0 = model.align_batch(*args)
    try:
        model.set_cu_partition(64)
        jump1, sel1 = model.align_batch(*args)
        model.align_batch(*args, enqueue_only=True)      # two in flight on the masked streams
        model.align_batch(*args, enqueue_only=True)
        j2, s2 = model.fetch(3, n_max, args[5])
        j3, s3 = model.fetch(3, n_max, args[5])
        with pytest.raises(RuntimeError):
            model.set_cu_partition(65)                    # a multiple of 8 below the CU count
    finally:
        model.set_cu_partition(0)
    jump4, sel4 = model.align_batch(*args)
    for j, s_ in ((jump1, sel1), (j2, s2), (j3, s3), (jump4, sel4)):
        assert np.array_equal(j, jump0) and np.array_equal(s_, sel0)
    # after the lift the engine runs on TORCH'S CURRENT stream again, not on its private non-blocking one -- an entry
    # point with device-side outputs and no trailing sync (log_mel) must be ordered with torch work on a side stream: the producer of its
    # input is delayed by a long torch kernel on that stream, the consumer follows immediately
    side = torch.cuda.Stream()
    ref = model.log_mel(torch.from_numpy(pcm[:1]).cuda(), [len(utts[0][0])]).clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        big = torch.randn(4096, 4096, device="cuda")
        for _ in range(20):
            big = big @ big * 1e-3                 # ~100 ms of queued work ahead of the copy below
        x = torch.zeros(1, smax, device="cuda")
        x.copy_(torch.from_numpy(pcm[:1]).cuda(), non_blocking=True)
        got = model.log_mel(x, [len(utts[0][0])])   # reads x on `side` (the bound stream): must see the copy
        out = got.clone()
    side.synchronize()
    assert torch.equal(out, ref)
    # while a partition is active a foreign stream is only recorded (WCA_STATUS_PARTITIONED), and fuse_ln is refused together with it
    model.set_cu_partition(64)
    try:
        with pytest.raises(wca._lib.WcaError, match="partitioned"):
            model.set_fuse_ln(True)
        with torch.cuda.stream(side):
            got2 = model.log_mel(x, [len(utts[0][0])]).clone()
        side.synchronize()
        assert torch.equal(got2, ref)
    finally:
        model.set_cu_partition(0)
    model.set_fuse_ln(True)
    with pytest.raises(wca._lib.WcaError, match="every CU"):
        model.set_cu_partition(64)
    model.set_fuse_ln(False)


def test_too_long_is_rejected(wca, setup):
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    mel = torch.zeros(80, 3000, device="cuda")
    with pytest.raises(wca._lib.TooLongError):
        tm.get_attentions(mel, torch.zeros(449, dtype=torch.int64).cuda(), model, tok, 100)
    with pytest.raises(wca._lib.TooLongError):
        tm.get_attentions(mel, torch.zeros(10, dtype=torch.int64).cuda(), model, tok, 1501)


def test_token_id_outside_vocabulary_is_an_error_not_a_fault(wca, setup):
    """A token id >= n_vocab (tokenizer / checkpoint mismatch) must surface as WCA_ERR_INVALID -- through the synchronous
    get_attentions and through the fused path's fetch -- instead of an out-of-bounds read of the embedding table."""
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    pcm, text, tt, tokens = _utt(syn, rt, tok, 3, 32000, 12)
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    bad = list(tokens)
    bad[5] = dims.n_vocab + 7
    with pytest.raises(wca._lib.WcaError, match="vocabulary"):
        tm.get_attentions(mel, torch.tensor(bad).cuda(), model, tok, 100, medfilt_width=3)
    opts = model.make_opts(aggregation="topk", topk=4, sot_len=3, medfilt_width=3)
    with pytest.raises(wca._lib.WcaError, match="vocabulary"):
        model.align_batch(torch.from_numpy(pcm[None]).cuda(), [len(pcm)], torch.tensor([bad]).cuda(), [len(bad)], [100], opts)
    # the engine is still usable and the good tokens give the usual result
    w, _ = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, 100, medfilt_width=3)
    assert torch.isfinite(w).all()


def test_get_attentions_does_not_clobber_a_queued_encode(wca, setup):
    """wca_get_attentions takes a FREE cross-K/V slot: a batch queued by encode_batch (slot 0) and aligned afterwards
    must give the same frames as when nothing ran in between."""
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    pa, _, tta, toksa = _utt(syn, rt, tok, 61, 48000, 20)
    pb, _, ttb, toksb = _utt(syn, rt, tok, 62, 64000, 26)
    opts = model.make_opts(aggregation="topk", topk=4, sot_len=3, medfilt_width=3)
    ta = torch.tensor([toksa]).cuda()
    want, _ = model.align_batch(torch.from_numpy(pa[None]).cuda(), [len(pa)], ta, [len(toksa)], [150], opts)
    model.encode_batch(pcm=torch.from_numpy(pa[None]).cuda(), n_samples=[len(pa)])
    melb = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pb)), 80, model=model)
    tm.get_attentions(melb, torch.tensor(toksb).cuda(), model, tok, 200, medfilt_width=3)   # must not overwrite A's K/V
    got, _ = model.align_batch(None, None, ta, [len(toksa)], [150], opts)
    assert np.array_equal(got, want)


def test_layernorm_epilogue_fusion_equals_separate_launches(wca):
    """The encoder with the LayerNorms inside the residual GEMMs' epilogues (wca_set_fuse_ln) against the same engine with
    separate LayerNorm launches (the default), at a batch large enough for the fused form (medium width, 3 layers, B = 12): encoder outputs agree
    to f16 rounding and the fused alignment path gives the same jump frames either way."""
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 1024, 16, 3, 51865, 448, 1024, 16, 3)
    sd = syn.random_state_dict(dims, seed=4, cross_qk_std=0.08)
    B = 12
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=B, precision="f16").load_state_dict(sd)
    tok = tk.get_tokenizer(True, language="English")
    utts = [_utt(syn, rt, tok, 300 + u, 96000, 40) for u in range(B)]
    pcm = torch.from_numpy(np.stack([u[0] for u in utts])).cuda()
    mel = model.log_mel(pcm)
    outs = {}
    for fuse in (True, False):
        model.set_fuse_ln(fuse)
        enc = model.encode(mel).cpu()
        opts = model.make_opts(aggregation="topk", topk=6, sot_len=3, medfilt_width=3)
        tarr = torch.tensor([u[3] for u in utts], dtype=torch.int64).cuda()
        jump, sel = model.align_batch(pcm, [96000] * B, tarr, [len(u[3]) for u in utts], [300] * B, opts)
        outs[fuse] = (enc, jump.copy(), sel.copy())
    model.set_fuse_ln(False)
    assert torch.isfinite(outs[True][0]).all()
    assert (outs[True][0] - outs[False][0]).abs().max().item() < 2e-2     # ln_post outputs, O(1), through 3 layers of f16 operands
    agree = np.mean(outs[True][1][:, :41] == outs[False][1][:, :41])
    assert agree > 0.95, agree   # identical up to a few near-tie frames (the f16 xn differ in the last bit here and there)
    del model


GATE_IDS = list(range(100, 132)) + list(range(10000, 10032))


def _gate_batch(wca):
    syn, tk, rt, tm, audio = _mods()
    dims = wca.dims_for("medium")
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
    B = 64
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=B, precision="f16").load_state_dict(sd)
    tok = tk.get_tokenizer(True, language="English")
    utts = [_utt(syn, rt, tok, u, 160000, 64) for u in GATE_IDS]
    assert all(len(u[3]) == 69 for u in utts)
    pcm = torch.from_numpy(np.stack([u[0] for u in utts])).cuda()
    tarr = torch.from_numpy(np.asarray([u[3] for u in utts], dtype=np.int64)).cuda()
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_word_times_medium_peaky.npz"))
    return dims, sd, model, tok, utts, (pcm, [160000] * B, tarr, [69] * B, [500] * B, opts), gold


def _boundary_stats(tm, tok, utts, jump, gold):
    total = within = ident = 0
    offenders = []
    for i, uid in enumerate(GATE_IDS):
        _p, _text, tt, _tokens = utts[i]
        _w, st, en = tm.words_from_jump_frames(jump[i], tt, tok, "char")
        off = 0
        for a, b in ((np.asarray(st), gold["st_%d" % uid]), (np.asarray(en), gold["en_%d" % uid])):
            assert len(a) == len(b), uid
            total += len(a)
            within += int(np.sum(np.abs(a - b) <= 0.02 + 1e-9))
            ident += int(np.sum(a == b))
            off += int(np.sum(np.abs(a - b) > 0.02 + 1e-9))
        if off:
            offenders.append((uid, off))
    return total, within, ident, offenders


def test_contract_mode_parity_no_exclusions(wca):
    """THE GATE of the contract line (north_star: word start / end times within one 20 ms encoder frame of the reference CPU path on
    the same audio + text). The headline configuration -- whisper-medium dimensions, PEAKY seeded weights (cross_qk_std = 0.08), 10 s
    audio, 64-char text, topk 10, medfilt 3 -- through the FUSED wca_align_batch at B = 64 in the engine's REFERENCE precision mode
    (wca_set_precision(WCA_PRECISION_REFERENCE) = every site split; what bench.py's `value` runs), on ids 100-131 + the bench's own ids 10000-10031,
    which include utterances whose oracle head scores are tied to 4e-6 / 3e-4 and every known miss of the f16 mode. EVERY boundary
    of EVERY utterance must be within one frame of the fp32 CPU oracle's: no acceptance set, no excused utterance.
    The oracle's word times come from tests/golden/oracle_word_times_medium_peaky.npz (tests/golden/make_oracle_word_times.py);
    the LIVE oracle is run on three of the utterances and must reproduce the fixture, and the step-by-step API's scores / matrix on
    one of them are compared with the live oracle's."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok, utts, args, gold = _gate_batch(wca)
    model.set_precision("reference")
    assert model.precision == "split" and model.precision_sites[0] == ["logmel", "conv", "enc_gemm", "enc_attn", "cross_kv", "dec", "capture"]
    jump, sel = model.align_batch(*args)
    total, within, ident, offenders = _boundary_stats(tm, tok, utts, jump, gold)
    print("contract mode, medium B=64 fused: %d boundaries over %d utterances, within one frame %d, identical %d, offenders %s"
          % (total, len(GATE_IDS), within, ident, offenders))
    assert total > 1200 and within == total and not offenders, offenders
    # head selection: the oracle's top-10, or heads whose oracle score is within fp32 noise (1e-5 relative) of its 10th best
    H = dims.n_text_head
    for i, uid in enumerate(GATE_IDS):
        sc = gold["sc_%d" % uid].astype(np.float64)
        kth = np.sort(sc)[-10]
        assert all(sc[int(h)] >= kth - 1e-5 * abs(kth) for h in sel[i][:10]), uid
    # the live oracle reproduces the fixture (and the engine's scores / matrix agree with it to fp32 noise)
    ref = whisper_ref.WhisperRef(sd, dims)
    rtok = tokenizer_ref.CharTokenizer()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    for i in (0, 33, 40):   # ids 100, 10001, 10008
        uid = GATE_IDS[i]
        p, text, tt, tokens = utts[i]
        mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(p)), audio.mel_filters(80))
        rw, _ = timing_ref.get_attentions(mel, torch.tensor(tokens), ref, 500, 3, 1.0)
        rwords, rst, ren, rmatrix, rscores = timing_ref.force_align(rw, tt, rtok, "char", "topk", 10)
        assert np.array_equal(np.asarray(rst), gold["st_%d" % uid]) and np.array_equal(np.asarray(ren), gold["en_%d" % uid]), uid
        if i == 0:
            w, _ = tm.get_attentions(mel.cuda(), torch.tensor(tokens).cuda(), model, tok, 500, medfilt_width=3)
            _wd, st, en, matrix, scores = tm.force_align(w, tt, tok, "char", "topk", topk=10)
            assert [lh for _, lh, _ in scores] == [lh for _, lh, _ in rscores]
            assert max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(scores, rscores)) < 1e-5
            assert ((matrix.cpu() - rmatrix).norm() / rmatrix.norm()).item() < 2e-5
            assert (w.cpu() - rw).abs().max().item() < 2e-5   # peaky maps (values up to 0.5): measured 5.9e-6
    del model


def test_f16_operating_point_medium_dims(wca):
    """The f16-operand fast mode (the engine's construction default, `f16_operating_point` in the bench line) on the same batch:
    NOT the contract line -- operand rounding moves a few near-tied head selections / ill-conditioned paths (measured on the
    301-utterance leg: 98.5 % of the boundaries within one frame, 292 of 301 utterances clean, profiles/r04_precision_ablation.txt).
    Held to a statistical bar against the same oracle fixture, plus the step-by-step API's maps / logits against the live oracle."""
    from oracle import timing_ref, whisper_ref
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok, utts, args, gold = _gate_batch(wca)
    assert model.precision == "f16"
    jump, sel = model.align_batch(*args)
    total, within, ident, offenders = _boundary_stats(tm, tok, utts, jump, gold)
    print("f16 mode, medium B=64 fused: %d boundaries, within one frame %d, identical %d, utterances with a boundary off: %s" % (total, within, ident, offenders))
    assert within >= 0.96 * total, (within, total)
    assert len(offenders) <= 6, offenders
    p, text, tt, tokens = utts[0]
    ref = whisper_ref.WhisperRef(sd, dims)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(p)), 80, model=model)
    w, logits = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, 500, medfilt_width=3)
    rw, rlogits = timing_ref.get_attentions(mel.cpu(), torch.tensor(tokens), ref, 500, 3, 1.0)
    assert tuple(w.shape) == (24, 16, 69, 500)
    assert (w.cpu() - rw).abs().max().item() < 1e-2          # measured 3e-3 (peaky maps, values up to 0.5)
    assert ((logits.cpu() - rlogits).abs().max() / rlogits.abs().max()).item() < 5e-3   # measured 8e-4
    del model


def test_alignment_like_model_parity_medium_dims(wca):
    """The same configuration on a checkpoint whose cross-attention LOOKS like a trained Whisper's alignment heads
    (synthetic.aligned_state_dict: a sharp monotonic ridge in 12 planted heads, well separated head scores, words spread over
    the audio instead of piling up at its end as with random weights). This is the regime the method is used in; here the
    DTW is well conditioned and the selection unambiguous, so f16 operands must not move ANYTHING: the selected heads must be
    the oracle's in the oracle's order and every word time identical (not just within a frame), for all 16 checked utterances."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    syn, tk, rt, tm, audio = _mods()
    dims = wca.dims_for("medium")
    sd = syn.aligned_state_dict(dims, seed=0)
    B, n_ref = 64, 16
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=B, precision="f16").load_state_dict(sd)
    ref = whisper_ref.WhisperRef(sd, dims)
    tok, rtok = tk.get_tokenizer(True, language="English"), tokenizer_ref.CharTokenizer()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    utts = [_utt(syn, rt, tok, 300 + u, 160000, 64) for u in range(B)]
    pcm = np.stack([u[0] for u in utts])
    tarr = np.asarray([u[3] for u in utts], dtype=np.int64)
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    jump, sel = model.align_batch(torch.from_numpy(pcm).cuda(), [160000] * B, torch.from_numpy(tarr).cuda(), [69] * B, [500] * B, opts)
    H = dims.n_text_head
    total = ident = 0
    spread = []
    for i in range(n_ref):
        p, text, tt, tokens = utts[i]
        mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(p)), audio.mel_filters(80))
        rw, _ = timing_ref.get_attentions(mel, torch.tensor(tokens), ref, 500, 3, 1.0)
        rwords, rst, ren, rmatrix, rscores = timing_ref.force_align(rw, tt, rtok, "char", "topk", 10)
        words, st, en = tm.words_from_jump_frames(jump[i], tt, tok, "char")
        assert words == rwords
        assert [int(h) for h in sel[i]] == [l * H + h for _, (l, h), _ in rscores], (i, list(sel[i]), rscores)
        assert all(h == 0 and l >= dims.n_text_layer // 2 for _, (l, h), _ in rscores)     # the planted heads win the selection
        total += 2 * len(st)
        ident += int((np.asarray(st) == rst).sum() + (np.asarray(en) == ren).sum())
        spread.append(float(ren[-1] - rst[1]))
    print("alignment-like medium B=64 fused: %d boundaries over %d utterances, identical %d; mean span of the aligned words %.2f s"
          % (total, n_ref, ident, float(np.mean(spread))))
    assert ident == total
    assert float(np.mean(spread)) > 5.0          # the words really are spread over the audio (ridge at 7 frames per token)
    del model


def test_large_v3_shape_family(wca):
    """n_mels=128, d=1280, 20 heads, vocab 51866 (large-v3 shapes, 1 layer each to keep the oracle fast)."""
    from oracle import timing_ref, whisper_ref
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)
    sd = syn.random_state_dict(dims, seed=2, cross_qk_std=0.06)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=2, precision="f16").load_state_dict(sd)
    tok = tk.get_tokenizer(True, language="English")
    pcm, text, tt, tokens = _utt(syn, rt, tok, 3, 48000, 24)
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 128, model=model)
    ref_mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm)), audio.mel_filters(128))
    assert (mel.cpu() - ref_mel).abs().max().item() < 2e-4
    w, logits = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, 150, medfilt_width=7)
    rw, rlogits = timing_ref.get_attentions(mel.cpu(), torch.tensor(tokens), whisper_ref.WhisperRef(sd, dims), 150, 7, 1.0)
    assert tuple(w.shape) == (1, 20, len(tokens), 150)
    assert (w.cpu() - rw).abs().max().item() < 5e-3
    assert ((logits.cpu() - rlogits).abs().max() / rlogits.abs().max()).item() < 5e-3
    del model


def test_edge_cases_short_text_and_single_word(wca, setup):
    """T = 1 (one character): one word + eot -> one (start, end); T = 0 -> the degenerate empty return."""
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    pcm = syn.synth_audio(5, 16000)
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    for text in ["a", "ab cd"]:
        tt = rt.encode(text, tok, "char")
        tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]).cuda()
        w, _ = tm.get_attentions(mel, tokens, model, tok, 50, medfilt_width=3)
        words, st, en, matrix, scores = tm.force_align(w, tt, tok, "char", "mean")
        assert words[-1] == "<|endoftext|>" and len(st) == len(words) - 1 == len(text.split())
        assert tuple(matrix.shape) == (len(tt) + 1, 50) and st[0] == 0.0 and np.all(en >= st)
    # empty text: 5 framing tokens only -> N = 1 DTW row, and force_align returns the degenerate value
    tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, tok.eot]).cuda()
    w, _ = tm.get_attentions(mel, tokens, model, tok, 50, medfilt_width=3)
    assert tm.force_align(w, [], tok, "char", "topk", topk=2) == [[], [], [], [], None]
    # max_frames = 1 and medfilt wider than the row (returned unfiltered, like whisper.timing.median_filter)
    tt = rt.encode("hi", tok, "char")
    tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]).cuda()
    w, _ = tm.get_attentions(mel, tokens, model, tok, 2, medfilt_width=7)
    assert tuple(w.shape)[-1] == 2 and torch.allclose(w.sum(-1), torch.ones_like(w.sum(-1)), atol=1e-5)


def test_default_find_alignment_vs_oracle(wca, setup):
    """--default_whisper_timing path (timing.py:116-186): std/mean-normalised alignment heads + DTW."""
    from oracle import timing_ref, whisper_ref
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = setup
    pcm, text, tt, tokens = _utt(syn, rt, tok, 41, 64000, 28)
    max_frames = len(pcm) // 320
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    model.set_alignment_heads([(1, 0), (2, 3), (2, 1)])
    assert model.alignment_heads == [(1, 0), (2, 1), (2, 3)]  # row-major, like alignment_heads.indices().T
    words, st, en, weights, _ = tm.default_find_alignment(model, tok, tt, mel, max_frames, medfilt_width=7)
    # oracle on the engine's own maps: isolates the normalisation + DTW from the forward's f16 noise
    w, _ = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, max_frames, medfilt_width=7)
    hw = torch.stack([w.cpu()[l][h] for l, h in model.alignment_heads])
    std, mean = torch.std_mean(hw, dim=-2, keepdim=True, unbiased=False)
    ref_w = (hw - mean) / std
    assert tuple(weights.shape) == (3, len(tokens), max_frames)     # the reference's 4th return (timing.py:186)
    np.testing.assert_allclose(weights.cpu().numpy(), ref_w.numpy(), rtol=2e-4, atol=2e-5)
    matrix = weights.cpu().mean(0)[3:-1]
    np.testing.assert_allclose(matrix.numpy(), timing_ref.default_alignment_matrix(w.cpu(), model.alignment_heads, 3).numpy(),
                               rtol=2e-4, atol=2e-5)
    ti, tj = timing_ref.dtw(-matrix)
    jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
    _, word_tokens = tok.split_to_word_tokens(tt + [tok.eot])
    wb = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    assert np.array_equal(st, (tj[jumps] / 50)[wb[:-1]]) and np.array_equal(en, (tj[jumps] / 50)[wb[1:]])
    assert [w_.strip() for w_ in words[:-1]] == text.split()
    model.set_alignment_heads([(l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer) for h in range(dims.n_text_head)])


def test_force_align_subword_mode(wca, fake_vocab):
    """--aligned_unit_type subword (needs a vocabulary file): same GPU pipeline, word merge by tokenizer.split_to_word_tokens
    (retokenize.py:22); against the CPU oracle with the same word split."""
    syn, tk, retok, tm, audio = _mods()
    from oracle import timing_ref
    tok = tk.get_tokenizer(True, language="English", vocab_path=fake_vocab)
    text = "the quick brown fox jumps"
    tt = retok.encode(text, tok, "subword")
    n = len(tok.sot_sequence) + 1 + len(tt) + 1
    g = torch.Generator().manual_seed(11)
    L, H, F = 2, 4, 180
    w = torch.softmax(torch.randn(L, H, n, F, generator=g) * 4, -1)
    words, st, en, matrix, scores = tm.force_align(w.cuda(), tt, tok, "subword", "topk", topk=3)
    rwords, rst, ren, rmatrix, rscores = timing_ref.force_align(
        w, tt, tok, "subword", "topk", 3, split_fn=lambda tokens, tokenizer, unit: tokenizer.split_to_word_tokens(tokens))
    assert words == rwords and "".join(words[:-1]) == text
    assert len(st) == len(words) - 1
    assert np.max(np.abs(np.asarray(st) - np.asarray(rst))) <= 0.02 + 1e-9
    assert np.max(np.abs(np.asarray(en) - np.asarray(ren))) <= 0.02 + 1e-9


def test_head_stats_lean_kernel_changes_no_bit_of_the_alignment(wca, switch):
    """The fused path (capture -> head statistics -> top-k -> aggregation -> DTW) with the lean head-statistics kernel against the general one
    (switch head_stats_general, wca_test_set_switch): selected heads in score order and jump frames identical, in both precision modes and for ragged lengths (the
    aggregation re-materialises the selected heads with the general kernel's arithmetic: row maxima and sums must agree exactly), and the
    step-by-step API's weights bit for bit."""
    syn, tk, rt, tm, audio = _mods()
    dims = wca.ModelDimensions(80, 1500, 256, 4, 3, 51865, 448, 256, 4, 3)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=8, precision="f16").load_state_dict(syn.random_state_dict(dims, seed=4, cross_qk_std=0.08))
    tok = tk.get_tokenizer(True, language="English")
    utts = [_utt(syn, rt, tok, 700 + i, 40000 + 23000 * i, 20 + 7 * i) for i in range(8)]
    n_max, smax = max(len(u[3]) for u in utts), max(len(u[0]) for u in utts)
    pcm = np.zeros((8, smax), dtype=np.float32)
    tarr = np.full((8, n_max), tok.eot, dtype=np.int64)
    for i, (p, _, _, toks) in enumerate(utts):
        pcm[i, :len(p)] = p
        tarr[i, :len(toks)] = toks
    n_samples, n_tok, frames = [len(u[0]) for u in utts], [len(u[3]) for u in utts], [len(u[0]) // 320 for u in utts]
    pcm_d, tarr_d = torch.from_numpy(pcm).cuda(), torch.from_numpy(tarr).cuda()
    for mode in ("f16", "reference"):
        model.set_precision(mode)
        for w in (3, 7):
            opts = model.make_opts(aggregation="topk", topk=5, sot_len=3, medfilt_width=w)
            switch("head_stats_general", 1)
            jump0, sel0 = model.align_batch(pcm_d, n_samples, tarr_d, n_tok, frames, opts)
            mel = model.log_mel(pcm_d, n_samples)
            wb0, _ = model.get_attentions(mel, tarr_d, frames, medfilt_width=w, n_tok=n_tok, want_logits=False)
            wb0 = wb0.clone()
            switch("head_stats_general", 0)
            jump1, sel1 = model.align_batch(pcm_d, n_samples, tarr_d, n_tok, frames, opts)
            wb1, _ = model.get_attentions(mel, tarr_d, frames, medfilt_width=w, n_tok=n_tok, want_logits=False)
            assert np.array_equal(sel0, sel1) and np.array_equal(jump0, jump1), (mode, w)
            for i in range(8):
                a, b = wb0[i, :, :, :n_tok[i], :frames[i]], wb1[i, :, :, :n_tok[i], :frames[i]]
                assert torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32)), (mode, w, i)
    del model


@pytest.fixture(scope="module")
def medium_contract(wca):
    """One whisper-medium-dims engine in the contract mode (peaky seeded weights, the headline checkpoint) shared by the fixture-sized parity legs."""
    syn, tk, rt, tm, audio = _mods()
    dims = wca.dims_for("medium")
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=64).load_state_dict(sd)
    model.set_precision("reference")
    tok = tk.get_tokenizer(True, language="English")
    yield dims, sd, model, tok
    del model


def test_contract_mode_parity_1033_fixture_utterances(wca, medium_contract):
    """Every utterance of BOTH committed fixed-shape oracle fixtures through the contract mode (the builder-run legs under
    the driver's eyes): ids 100-131 + 10000-10300 (tests/golden/oracle_word_times_medium_peaky.npz, 333 utterances) and ids 10301-11000
    (oracle_word_times_medium_peaky_700.npz, the second leg: it contains 10830 / 10918, whose 10th / 11th oracle head scores are 6.5e-6 /
    1.1e-5 apart and which the cheaper site set misses) -- the headline configuration (whisper-medium dims, peaky seeded weights, 10 s audio,
    64 characters, topk 10, medfilt 3), fused wca_align_batch at B = 64, 17 micro-batches. North_star's bar: EVERY word start / end within one
    20 ms frame of the fp32 CPU oracle's; measured and asserted here: every boundary IDENTICAL. Head selection: the oracle's top-10 or heads
    within fp32 noise of its 10th score."""
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = medium_contract
    assert model.precision == "split"
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    golds = [(np.load(os.path.join(here, "oracle_word_times_medium_peaky.npz")), list(range(100, 132)) + list(range(10000, 10301))),
             (np.load(os.path.join(here, "oracle_word_times_medium_peaky_700.npz")), list(range(10301, 11001)))]
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    B = 64
    total = within = ident = n_utt = 0
    offenders = []
    for gold, ids in golds:
        for lo in range(0, len(ids), B):
            chunk = ids[lo:lo + B]
            fill = (chunk * ((B + len(chunk) - 1) // len(chunk)))[:B]   # the last batch is filled by repetition
            utts = [_utt(syn, rt, tok, u, 160000, 64) for u in fill]
            pcm = torch.from_numpy(np.stack([u[0] for u in utts])).cuda()
            tarr = torch.from_numpy(np.asarray([u[3] for u in utts], dtype=np.int64)).cuda()
            jump, sel = model.align_batch(pcm, [160000] * B, tarr, [69] * B, [500] * B, opts)
            for j, uid in enumerate(chunk):
                _w, st, en = tm.words_from_jump_frames(jump[j], utts[j][2], tok, "char")
                d = np.concatenate([np.abs(np.asarray(st) - gold["st_%d" % uid]), np.abs(np.asarray(en) - gold["en_%d" % uid])])
                total += d.size
                within += int((d <= 0.02 + 1e-9).sum())
                ident += int((d == 0).sum())
                n_utt += 1
                if (d > 0.02 + 1e-9).any():
                    offenders.append(uid)
                sc = gold["sc_%d" % uid].astype(np.float64)
                kth = np.sort(sc)[-10]
                assert all(sc[int(h)] >= kth - 1e-5 * abs(kth) for h in sel[j][:10]), uid
    print("contract mode, 333 + 700 fixture utterances at B=64: %d boundaries over %d utterances, within one frame %d, identical %d, offenders %s"
          % (total, n_utt, within, ident, offenders))
    assert n_utt == 1033 and total == 21050, (n_utt, total)   # 632 + 6 184 + 14 234
    assert within == total and not offenders, offenders
    assert ident == total, (ident, total)   # measured since round 4: 6 184 + 14 234 = 20 418 of 20 418 identical on the two legs


@pytest.mark.parametrize("leg", ["A", "B"])
def test_contract_mode_parity_ragged_lengths(wca, medium_contract, leg):
    """The contract mode on RAGGED micro-batches at whisper-medium dims: all 128 utterances of tools/parity_ragged.py (ids 20000-20127, 2-29 s
    audio, 9-220 characters: a different frame count, decoder length and reflect-padding position per utterance) through the fused
    wca_align_batch at B = 32. Leg A: north-star settings (char units, topk 10, medfilt 3); leg B: the reference CLI's DEFAULTS
    (/root/reference/infer_ali.py:160-162: medfilt 7, aggr mean) in char units. Every word boundary within one frame of the fp32 CPU oracle's
    (measured and asserted: all 3 376 identical; the f16 operating point misses 3 / 1 of the 128 utterances). Oracle word times from
    tests/golden/oracle_word_times_ragged_{A,B}.npz (tests/golden/make_oracle_word_times_ragged.py); on leg A the live oracle re-derives the
    shortest utterance and must reproduce the fixture."""
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_ragged
    syn, tk, rt, tm, audio = _mods()
    dims, sd, model, tok = medium_contract
    cfg = parity_ragged.LEGS[leg]
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_word_times_ragged_%s.npz" % leg))
    opts = model.make_opts(aggregation=cfg["aggr"], topk=cfg["topk"], sot_len=3, medfilt_width=cfg["medfilt"])
    ids = list(range(20000, 20128))
    total, within, ident, offenders = _ragged_leg(model, tok, parity_ragged, ids, 32, opts, gold)
    print("contract mode, ragged leg %s (medfilt %d, aggr %s), medium dims: %d boundaries over %d utterances, within one frame %d, identical %d, offenders %s"
          % (leg, cfg["medfilt"], cfg["aggr"], total, len(ids), within, ident, offenders))
    assert total == 3376 and within == total and not offenders, offenders
    assert ident == total, (ident, total)
    if leg == "A":   # the live oracle reproduces the fixture on the shortest utterance
        u = min(ids, key=lambda v: parity_ragged.spec(v)[0])
        ns, ch = parity_ragged.spec(u)
        rtok = tokenizer_ref.CharTokenizer()
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(syn.synth_audio(u, ns))), audio.mel_filters(80))
        tt = tokenizer_ref.encode_char(syn.synth_text(u, ch), rtok)
        rw, _ = timing_ref.get_attentions(mel, torch.tensor([*rtok.sot_sequence, rtok.no_timestamps, *tt, rtok.eot]), whisper_ref.WhisperRef(sd, dims), ns // 320, 3, 1.0)
        _rwords, rst, ren, _m, _s = timing_ref.force_align(rw, tt, rtok, "char", "topk", 10)
        assert np.array_equal(np.asarray(rst), gold["st_%d" % u]) and np.array_equal(np.asarray(ren), gold["en_%d" % u]), u


def _ragged_leg(model, tok, parity_ragged, ids, B, opts, gold):
    syn, tk, rt, tm, audio = _mods()
    total = within = ident = 0
    offenders = []
    for lo in range(0, len(ids), B):
        chunk = ids[lo:lo + B]
        sp = [parity_ragged.spec(u) for u in chunk]
        pcm = np.zeros((len(chunk), max(s for s, _ in sp)), dtype=np.float32)
        tts = []
        for j, (u, (ns, ch)) in enumerate(zip(chunk, sp)):
            pcm[j, :ns] = syn.synth_audio(u, ns)
            tts.append(rt.encode(syn.synth_text(u, ch), tok, "char"))
        rows = [[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts]
        tarr = np.full((len(chunk), max(len(r) for r in rows)), tok.eot, dtype=np.int64)
        for j, r in enumerate(rows):
            tarr[j, :len(r)] = r
        jump, _ = model.align_batch(torch.from_numpy(pcm).cuda(), [s for s, _ in sp], torch.from_numpy(tarr).cuda(), [len(r) for r in rows],
                                    [s // 320 for s, _ in sp], opts)
        for j, u in enumerate(chunk):
            _w, st, en = tm.words_from_jump_frames(jump[j], tts[j], tok, "char")
            d = np.concatenate([np.abs(np.asarray(st) - gold["st_%d" % u]), np.abs(np.asarray(en) - gold["en_%d" % u])])
            total += d.size
            within += int((d <= 0.02 + 1e-9).sum())
            ident += int((d == 0).sum())
            if (d > 0.02 + 1e-9).any():
                offenders.append(u)
    return total, within, ident, offenders


def test_contract_mode_parity_large_v3_dimensions(wca):
    """configs[3] / [4]'s model family against the oracle's FORWARD (at large dimensions the suite had self-consistency checks
    only): 24 ragged utterances (ids 21000-21023, 2.1-26.1 s, 7-219 characters) at whisper-large-v3 DIMENSIONS -- 128 mel bins, 1280 wide, 20
    heads, 32 + 32 layers = 640 captured heads, vocabulary 51 866, peaky seeded weights -- through the fused wca_align_batch at B = 24 in the
    contract mode, north-star settings (char, topk 10, medfilt 3): every boundary within one frame of the fp32 CPU oracle's (measured and
    asserted: 498 / 498 identical). Oracle word times: tests/golden/oracle_word_times_ragged_A_large_v3.npz
    (`python tests/golden/make_oracle_word_times_ragged.py A large-v3`, oracle/ alone, ~70 CPU-minutes)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_ragged
    syn, tk, rt, tm, audio = _mods()
    dims = wca.dims_for("large-v3")
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=24).load_state_dict(sd)
    del sd
    model.set_precision("reference")
    tok = tk.get_tokenizer(True, language="English")
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_word_times_ragged_A_large_v3.npz"))
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    ids = list(range(21000, 21024))
    total, within, ident, offenders = _ragged_leg(model, tok, parity_ragged, ids, 24, opts, gold)
    print("contract mode, large-v3 dimensions, 24 ragged utterances at B=24: %d boundaries, within one frame %d, identical %d, offenders %s" % (total, within, ident, offenders))
    assert total == 498 and within == total and not offenders, offenders
    assert ident == total, (ident, total)
    del model
