"""BASELINE.json configs[2] / [3] / [4] on the MI355X (`-m gpu`): the LibriSpeech FLAC corpus and the AMI `ami_kaldi.pkl` segments
(dataset.py:67-122, README.md:64-71) through the infer_ali.py driver on the GPU, word times against the fp32 CPU oracle; one
whisper-large-v3-shaped forward (128 mel bins, 32 + 32 layers, vocabulary 51 866) at B = 1 with the reference's token framing for
that model (infer_ali.py:41 builds the tokenizer without num_languages; infer_ali.py:159: --n_mels must be 128)."""
import importlib
import os
import pickle
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _m(n):
    return importlib.import_module("whisper-char-alignment_amd." + n)


def _speech(u, n):
    pcm = np.load(os.path.join(GOLD, "sample_pcm_int16.npy")).astype(np.int64)
    x = np.roll(pcm, 3000 * u)
    reps = (n + len(x) - 1) // len(x)
    return np.tile(x, reps)[:n]


def _small_model(wca, max_batch=4):
    syn = _m("synthetic")
    dims = wca.ModelDimensions(80, 1500, 256, 4, 3, 51865, 448, 256, 4, 3)
    sd = syn.random_state_dict(dims, seed=5, cross_qk_std=0.08)
    return dims, sd, wca.WhisperAMD(dims, device="cuda:0", max_batch=max_batch, precision="f16").load_state_dict(sd)


def _oracle_times(sd, dims, pcm_f32, token_ids, sot_len, word_tokens, max_frames, medfilt, aggregation, topk):
    """fp32 CPU restatement: log-mel -> forward with capture -> median / softmax -> aggregation -> DTW -> jump times."""
    from oracle import timing_ref, whisper_ref
    audio = _m("audio")
    ref = whisper_ref.WhisperRef(sd, dims)
    mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm_f32)), audio.mel_filters(dims.n_mels))
    w, _ = timing_ref.get_attentions(mel, torch.tensor(token_ids), ref, max_frames, medfilt, 1.0)
    matrix, _scores = timing_ref.aggregate(w, aggregation, topk)
    ti, tj = timing_ref.dtw(-matrix[sot_len:-1].cpu())
    return timing_ref.jumps_to_times(ti, tj, word_tokens)


def test_infer_ali_librispeech_flac_corpus_on_gpu(wca, tmp_path):
    """configs[2] shape: `<root>/test-clean/<spk>/<chap>/<fid>.flac` (decoded by csrc/flac.cpp inside libwca.so), `ls_alignment_*.txt`
    ground truth, char alignment, top-k aggregation, reference-precision forward -> `infer_ali.py --dataset LibriSpeech` on the GPU.
    Every predicted word start / end must equal the fp32 oracle's on the decoded audio; the result schema is the reference's."""
    import glob
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import flac_fixture as ff
    import joblib
    infer, tk, rt, audio = _m("infer_ali"), _m("tokenizer"), _m("retokenize"), _m("audio")
    dims, sd, model = _small_model(wca)
    root = tmp_path / "LibriSpeech"
    split = "test-clean"
    utts = {"1089-134686-0000": (52000, "HE HOPED THERE WOULD BE STEW"), "1089-134686-0001": (33000, "STUFF IT INTO YOU"),
            "121-127105-0003": (47000, "HELLO BIG WORLD'S END")}
    scp_lines, ali_lines, pcm_of = [], [], {}
    for i, (fid, (n, text)) in enumerate(utts.items()):
        spk, chap, _ = fid.split("-")
        d = root / split / spk / chap
        d.mkdir(parents=True, exist_ok=True)
        pcm = _speech(i, n)
        pcm_of[fid] = pcm
        (d / (fid + ".flac")).write_bytes(ff.encode(pcm, blocksize=4096, kinds=("lpc", ("fixed", 2)), rice2=bool(i & 1)))
        with open(d / ("%s-%s.trans.txt" % (spk, chap)), "a") as f:
            f.write("%s %s\n" % (fid, text))
        words = text.split()
        step = n / 16000.0 / (len(words) + 1)
        ali = [("", 0.0, round(step * 0.5, 3))] + [(w, round(step * (j + 0.5), 3), round(step * (j + 1.5), 3)) for j, w in enumerate(words)]
        ali_lines.append("%s %r\n" % (fid, ali))
        scp_lines.append("%s %s\n" % (fid, d / (fid + ".flac")))
    scp = tmp_path / "test-clean.wav.scp"
    scp.write_text("".join(scp_lines))
    ali_file = tmp_path / ("ls_alignment_%s.txt" % split)
    ali_file.write_text("".join(ali_lines))
    out = tmp_path / "out"
    args = infer.parse_args(["--model", "tiny", "--random_init", "--dataset", "LibriSpeech", "--scp", str(scp), "--alignment_file", str(ali_file),
                             "--output_dir", str(out), "--aggr", "topk", "--topk", "4", "--aligned_unit_type", "char", "--medfilt_width", "3",
                             "--batch_size", "2", "--save_prediction", "--tolerance", "0.05", "--teacher", "text", "--forward_precision", "split"])
    infer.infer_dataset(args, model=model)
    res = json.load(open(glob.glob(str(out / "*.json"))[0]))
    assert {"precision", "recall", "f1", "r_value"} <= set(res)
    preds = joblib.load(glob.glob(str(out / "*-predictions.pkl"))[0])
    assert sorted(preds) == [0, 1, 2]
    tok = tk.get_tokenizer(True, language="English")
    for idx, (fid, (n, text)) in enumerate(utts.items()):
        p = preds[idx]
        assert p["fids"] == fid and p["texts"] == rt.remove_punctuation(text).split()
        assert len(p["ends"]) == len(text.split())   # the '' (silence) entries of the alignment file are dropped
        tt = rt.encode(rt.remove_punctuation(text), tok, "char")
        tokens = [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]
        _words, word_tokens = rt.split_tokens_on_spaces(tt + [tok.eot], tok, "char")
        y, sr = audio.load_audio(str(root / split / fid.split("-")[0] / fid.split("-")[1] / (fid + ".flac")))
        assert sr == 16000 and np.array_equal(np.round(y * 32768).astype(np.int64), pcm_of[fid])   # bit-exact FLAC decode
        rst, ren = _oracle_times(sd, dims, np.asarray(y, dtype=np.float32), tokens, len(tok.sot_sequence), word_tokens, n // 320, 3, "topk", 4)
        assert np.array_equal(np.asarray(p["starts_hat"]), rst) and np.array_equal(np.asarray(p["ends_hat"]), ren), (fid, p["ends_hat"], ren)
        assert [w.strip() for w in p["predwords"][:-1]] == rt.remove_punctuation(text).split() and p["predwords"][-1] == "<|endoftext|>"
    del model


def test_infer_ali_ami_pkl_subword_on_gpu(wca, tmp_path, fake_vocab):
    """configs[3] shape: AMI segments as RIFF WAV + `ami_kaldi.pkl` word references (README.md:64-71), SUBWORD units (BPE over a
    synthetic tiktoken vocabulary of the right size), `--aggr mean` -> `infer_ali.py --dataset AMI` on the GPU in the reference-precision
    forward. Word times against the fp32 oracle on the same token ids (the BPE merge itself is host logic pinned by
    tests/test_host.py and the reference-executed fixtures)."""
    import glob
    import joblib
    infer, tk, rt = _m("infer_ali"), _m("tokenizer"), _m("retokenize")
    dims, sd, model = _small_model(wca)
    segs = {"AMI_TS3003d_H03_MTD012ME_0255148_0255515": (58000, [("okay", 0.1, 0.5), ("", 0.5, 0.6), ("so", 0.6, 0.9), ("abcd", 0.9, 1.4), ("efgh", 1.4, 2.2)]),
            "AMI_ES2004a_H00_MEE006_0010000_0010300": (41000, [("yeah", 0.05, 0.6), ("abab", 0.7, 1.3)])}
    scp_lines, pcm_of = [], {}
    for i, (sid, (n, _ref)) in enumerate(segs.items()):
        pcm = _speech(i + 3, n).astype("<i2")
        pcm_of[sid] = pcm
        data = pcm.tobytes()
        wav = tmp_path / (sid + ".wav")
        wav.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) +
                        b"data" + struct.pack("<I", len(data)) + data)
        scp_lines.append("%s %s\n" % (sid, wav))
    scp = tmp_path / "ami.scp"
    scp.write_text("".join(scp_lines))
    pkl = tmp_path / "ami_kaldi.pkl"
    pkl.write_bytes(pickle.dumps({sid: ref for sid, (_n, ref) in segs.items()}))
    out = tmp_path / "out"
    args = infer.parse_args(["--model", "tiny", "--random_init", "--dataset", "AMI", "--scp", str(scp), "--alignment_file", str(pkl), "--output_dir", str(out),
                             "--aggr", "mean", "--aligned_unit_type", "subword", "--vocab", fake_vocab, "--medfilt_width", "5", "--batch_size", "2",
                             "--save_prediction", "--strict", "--tolerance", "0.1", "--teacher", "text", "--forward_precision", "split"])
    infer.infer_dataset(args, model=model)
    preds = joblib.load(glob.glob(str(out / "*-predictions.pkl"))[0])
    assert sorted(preds) == [0, 1]
    tok = tk.get_tokenizer(True, language="English", vocab_path=fake_vocab)
    for idx, (sid, (n, ref)) in enumerate(segs.items()):
        p = preds[idx]
        text = " ".join(w for w, _s, _e in ref if w)
        assert p["fids"] == sid and p["texts"] == text.split() and p["starts"] == [s for w, s, _e in ref if w]
        tt = rt.encode(rt.remove_punctuation(text), tok, "subword")
        words_sub, _wt = rt.split_tokens_on_spaces(tt + [tok.eot], tok, "subword")
        assert [w.strip() for w in words_sub[:-1]] == text.split()   # split_to_word_tokens of the subword path (retokenize.py:19-39)
        tokens = [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]
        _words, word_tokens = rt.split_tokens_on_spaces(tt + [tok.eot], tok, "subword")
        rst, ren = _oracle_times(sd, dims, pcm_of[sid].astype(np.float32) / 32768.0, tokens, len(tok.sot_sequence), word_tokens, n // 320, 5, "mean", -1)
        assert np.array_equal(np.asarray(p["starts_hat"]), rst) and np.array_equal(np.asarray(p["ends_hat"]), ren), (sid, p["ends_hat"], ren)
    del model


def test_large_v3_shaped_forward_b1(wca):
    """configs[4]'s model: whisper-large-v3 DIMENSIONS (128 mel bins, 1280 wide, 20 heads, 32 + 32 layers, vocabulary 51 866) with
    seeded random weights, one utterance: the forward with capture runs at full depth, the maps are proper distributions over the
    frames, 640 heads are captured, the token framing is the reference's (the 99-language special ids: see below), and the fused path
    agrees with the step-by-step API."""
    syn, tk, rt, tm, audio = _m("synthetic"), _m("tokenizer"), _m("retokenize"), _m("timing"), _m("audio")
    dims = wca.dims_for("large-v3")
    assert (dims.n_mels, dims.n_vocab, dims.n_audio_layer, dims.n_text_layer, dims.n_text_head) == (128, 51866, 32, 32, 20)
    sd = syn.random_state_dict(dims, seed=2, cross_qk_std=0.08)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=1, precision="f16").load_state_dict(sd)
    # the quirk (infer_ali.py:41): the reference builds the tokenizer WITHOUT num_languages, so a large-v3 run frames its text with
    # the 99-language special ids (transcribe 50359, no_timestamps 50363) although the v3 vocabulary has 100 languages (50360 / 50364);
    # the drop-in keeps that behaviour, and --n_mels must be 128 (infer_ali.py:159)
    assert model.is_multilingual and model.num_languages == 100
    tok = tk.get_tokenizer(model.is_multilingual, language="English")
    tok_v3 = tk.get_tokenizer(True, num_languages=model.num_languages, language="English")
    assert tok.eot == tok_v3.eot == 50257 and tok.no_timestamps == 50363 and tok_v3.no_timestamps == 50364
    assert tuple(tok.sot_sequence) == (50258, 50259, 50359) and tuple(tok_v3.sot_sequence) == (50258, 50259, 50360)
    n = 64000
    pcm = syn.synth_audio(77, n)
    tt = rt.encode("large model check", tok, "char")
    tokens = [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 128, model=model)
    assert tuple(mel.shape) == (128, 3000)
    w, logits = tm.get_attentions(mel, torch.tensor(tokens).cuda(), model, tok, n // 320, medfilt_width=3)
    assert tuple(w.shape) == (32, 20, len(tokens), n // 320) and tuple(logits.shape) == (len(tokens), 51866)
    assert torch.isfinite(w).all() and torch.isfinite(logits).all()
    assert (w.sum(-1) - 1).abs().max().item() < 1e-4 and w.min().item() >= 0
    words, st, en, matrix, scores = tm.force_align(w, tt, tok, "char", "topk", topk=10)
    assert [x.strip() for x in words[:-1]] == ["large", "model", "check"] and len(scores) == 10
    assert np.all(np.diff(en) >= 0) and np.all(st[1:] == en[:-1]) and en[-1] <= n / 16000 + 1e-9
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=len(tok.sot_sequence), medfilt_width=3)
    jump, sel = model.align_batch(torch.from_numpy(pcm)[None].cuda(), [n], torch.tensor([tokens]).cuda(), [len(tokens)], [n // 320], opts)
    _w2, st2, en2 = tm.words_from_jump_frames(jump[0], tt, tok, "char")
    assert np.array_equal(st2, st) and np.array_equal(en2, en)
    assert sorted(int(h) for h in sel[0]) == sorted(l * 20 + h for _s, (l, h), _n in scores)
    del model
