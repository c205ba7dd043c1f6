"""CPU tests of the product's host-side logic (no GPU, no oracle in the product path): tokenizer,
retokenize, metrics, audio readers -- compared with golden vectors produced by the REAL reference files."""
import importlib
import json
import os
import struct

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _m(name):
    return importlib.import_module("whisper-char-alignment_amd." + name)


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GOLD, "reference_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tok():
    return _m("tokenizer").get_tokenizer(True, language="English")


def test_tokenizer_special_ids(tok):
    tk = _m("tokenizer")
    assert tok.sot_sequence == (50258, 50259, 50359) and tok.eot == 50257 and tok.no_timestamps == 50363
    assert tok.timestamp_begin == 50364 and tok.n_vocab == 51865
    en = tk.get_tokenizer(False)
    assert en.sot_sequence == (50257,) and en.eot == 50256 and en.no_timestamps == 50362 and en.n_vocab == 51864
    v3 = tk.get_tokenizer(True, num_languages=100, language="en")
    assert v3.n_vocab == 51866 and v3.no_timestamps == 50364
    assert tok.encode(" ") == [220] and tok.encode("a") == [64] and tok.encode("A") == [32] and tok.encode("'") == [6]
    assert tok.decode([64, 220, 65]) == "a b"
    with pytest.raises(tk.NeedVocabError):
        tok.encode("hello")
    with pytest.raises(ValueError):
        tk.get_tokenizer(True, language="klingon")


def test_retokenize_matches_reference(meta, tok):
    rt = _m("retokenize")
    for case in meta["retokenize"]:
        tt = rt.encode(case["text"], tok, "char")
        assert tt == case["tokens"]
        words, wts = rt.split_tokens_on_spaces(tt + [tok.eot], tok, "char")
        assert words == case["words"] and wts == case["word_tokens"]
        starts = rt.char_word_starts(tt + [tok.eot], tok)
        assert list(starts) == list(np.cumsum([0] + [len(w) for w in wts[:-1]]))
    for text, want in meta["remove_punctuation"]:
        assert rt.remove_punctuation(text) == want


def test_number_to_words():
    rt = _m("retokenize")
    assert rt.number_to_words(0) == "zero" and rt.number_to_words(42) == "forty-two"
    assert rt.number_to_words(101) == "one hundred and one" and rt.number_to_words(1001) == "one thousand and one"
    assert rt.number_to_words(1234) == "one thousand, two hundred and thirty-four"
    assert rt.remove_punctuation("room 42") == "room fortytwo"  # the final translate() drops the hyphen, as in the reference


def test_metrics_match_reference(meta):
    mt = _m("metrics")
    g = meta["metrics"]
    arrays = np.load(os.path.join(GOLD, "reference_golden.npz"))
    attn = torch.from_numpy(arrays["cov_attn"])
    assert float(mt.coverage_penalty(attn)) == pytest.approx(g["coverage_penalty"][0], rel=1e-6)
    assert float(mt.coverage_penalty(attn, 0.1)) == pytest.approx(g["coverage_penalty"][1], rel=1e-6)
    for c in g["eval_n1"]:
        assert list(mt.eval_n1(c["y"], c["yhat"], c["tol"])) == c["out"]
    for c in g["eval_n1_strict"]:
        assert list(mt.eval_n1_strict(c["y"], c["yhat"], c["words"], c["words_hat"], c["tol"])) == c["out"]
    for c in g["get_seg_metrics"]:
        assert [float(v) for v in mt.get_seg_metrics(*c["args"])] == c["out"]


def test_words_from_jump_frames_equals_reference_arithmetic(meta, tok):
    """The fused path's host tail must reproduce timing.py:108-113 given the DTW path."""
    tm = _m("timing")
    arrays = np.load(os.path.join(GOLD, "reference_golden.npz"))
    from oracle import timing_ref
    for case in meta["force_align"]:
        if case["degenerate"]:
            w, st, en = tm.words_from_jump_frames(np.zeros(len(case["tokens"]) + 1, dtype=np.int32), case["tokens"], tok, "char")
            assert w == [] and len(st) == 0
            continue
        matrix = torch.from_numpy(arrays[case["ws"] + "_matrix"])
        ti, tj = timing_ref.dtw(-matrix)
        jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
        words, st, en = tm.words_from_jump_frames(tj[jumps], case["tokens"], tok, "char")
        assert words == case["words"]
        np.testing.assert_array_equal(st, arrays[case["ws"] + "_start"])
        np.testing.assert_array_equal(en, arrays[case["ws"] + "_end"])


def test_audio_readers(tmp_path):
    au = _m("audio")
    pcm = (np.load(os.path.join(GOLD, "sample_pcm_int16.npy")))
    # RIFF/WAVE 16-bit
    p = tmp_path / "a.wav"
    data = pcm.astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
    p.write_bytes(hdr + b"data" + struct.pack("<I", len(data)) + data)
    x, sr = au.load_audio(str(p))
    assert sr == 16000 and np.array_equal(np.round(x * 32768).astype(np.int16), pcm)
    # NIST SPHERE (what TIMIT's ".wav" files are, like sample/test.wav)
    s = tmp_path / "b.wav"
    head = ("NIST_1A\n   1024\nsample_count -i %d\nsample_rate -i 16000\nchannel_count -i 1\nsample_n_bytes -i 2\n"
            "sample_byte_format -s2 01\nsample_coding -s3 pcm\nend_head\n" % len(pcm)).encode()
    s.write_bytes(head + b" " * (1024 - len(head)) + data)
    y, sr = au.load_audio(str(s))
    assert sr == 16000 and np.array_equal(np.round(y * 32768).astype(np.int16), pcm)
    assert len(pcm) == 46592 and len(pcm) // 320 == 145  # sample/test.wav -> 145 frames
    with pytest.raises(ValueError):
        bad = tmp_path / "c.wav"
        bad.write_bytes(b"garbage")
        au.load_audio(str(bad))


def test_pad_or_trim_and_filters():
    au = _m("audio")
    assert au.pad_or_trim(np.ones(10), 16).shape == (16,) and au.pad_or_trim(np.ones(20), 16).shape == (16,)
    t = au.pad_or_trim(torch.ones(2, 10), 16)
    assert t.shape == (2, 16) and float(t[:, 10:].abs().sum()) == 0.0
    f80, f128 = au.mel_filters(80), au.mel_filters(128)
    assert f80.shape == (80, 201) and f128.shape == (128, 201) and (f80 >= 0).all()
    from transformers.audio_utils import mel_filter_bank
    hf = mel_filter_bank(201, 80, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    assert np.abs(hf - f80).max() < 1e-7


def test_synthetic_inputs_are_deterministic():
    syn = _m("synthetic")
    a, b = syn.synth_audio(3, 16000), syn.synth_audio(3, 16000)
    assert np.array_equal(a, b) and a.dtype == np.float32 and np.abs(a).max() <= 1.0
    assert (a[2000:4000] == 0).all() and np.abs(a[:2000]).max() > 0  # 4 Hz gate: 125 ms on / 125 ms off
    for u in range(20):
        t = syn.synth_text(u, 64)
        assert len(t) == 64 and t == t.strip() and "  " not in t and all(2 <= len(w) for w in t.split())


def test_product_refuses_to_run_without_gpu():
    """No CPU fallback: constructing an engine on a GPU-less box must fail loudly."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    wca = importlib.import_module("whisper-char-alignment_amd")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        wca.WhisperAMD(wca.dims_for("tiny"), precision="f16")
    tm = _m("timing")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tm.filter_attention(torch.rand(2, 2, 4, 8).softmax(-1), topk=1)


# ------------------------------------------------------------------ greedy-decode host logic (decoding.py)
def test_decoding_suppress_lists_and_masks():
    import importlib
    import numpy as np
    decoding = importlib.import_module("whisper-char-alignment_amd.decoding")
    tokmod = importlib.import_module("whisper-char-alignment_amd.tokenizer")
    tok = tokmod.get_tokenizer(True, language="en", task="transcribe")
    # whisper's multilingual special-token numbering
    assert (tok.eot, tok.sot, tok.transcribe, tok.translate) == (50257, 50258, 50359, 50358)
    assert (tok.sot_lm, tok.sot_prev, tok.no_speech, tok.no_timestamps, tok.timestamp_begin) == (50360, 50361, 50362, 50363, 50364)
    assert tok.sot_sequence == (50258, 50259, 50359)
    opts = decoding.DecodingOptions(language="en")
    ids = decoding.suppress_token_ids(tok, opts)
    for t in (tok.transcribe, tok.translate, tok.sot, tok.sot_prev, tok.sot_lm, tok.no_speech):
        assert t in ids
    assert tok.eot not in ids and tok.no_timestamps not in ids
    assert tok.encode("(")[0] in ids and tok.encode("a")[0] not in ids  # a non-speech symbol vs a letter
    assert list(ids) == sorted(set(ids))
    sup, blank = decoding.filter_masks(tok, opts, 51865)
    assert sup.dtype == np.uint8 and sup.sum() == len(ids) + 1 and sup[tok.no_timestamps] == 1  # + <|notimestamps|>
    assert sorted(np.nonzero(blank)[0].tolist()) == sorted(tok.encode(" ") + [tok.eot])
    sup2, blank2 = decoding.filter_masks(tok, decoding.DecodingOptions(language="en", without_timestamps=True, suppress_blank=False), 51865)
    assert sup2[tok.no_timestamps] == 0 and blank2 is None
    # explicit list instead of "-1": no non-speech expansion
    ids3 = decoding.suppress_token_ids(tok, decoding.DecodingOptions(language="en", suppress_tokens=(11, 12)))
    assert 11 in ids3 and 12 in ids3 and tok.encode("(")[0] not in ids3
    assert abs(decoding.compression_ratio("aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa") - 40 / len(__import__("zlib").compress(b"a" * 40))) < 1e-12


def test_subword_tokenizer_with_vocabulary_file(fake_vocab):
    """--aligned_unit_type subword needs the BPE table: tiktoken-format file, GPT-2 pre-tokenisation
    pattern, rank-ordered merges; round trip and the word split (retokenize.py:22 -> tokenizer.split_to_word_tokens)."""
    import importlib
    tokmod = importlib.import_module("whisper-char-alignment_amd.tokenizer")
    retok = importlib.import_module("whisper-char-alignment_amd.retokenize")
    tok = tokmod.get_tokenizer(True, language="en", vocab_path=fake_vocab)
    assert tok.has_vocab and tok.n_vocab == 51865
    text = "hello world it's 42 o'clock"
    toks = retok.encode(text, tok, "subword")
    assert toks == tok.encode(text) and tok.decode(toks) == text
    # a merge that exists in the table is taken: 'aa' is not there, the 4-letter token 'aaaa' cannot be built from bytes
    assert len(tok.encode("aaaa")) == 4
    words, word_tokens = retok.split_tokens_on_spaces(toks + [tok.eot], tok, "subword")
    assert "".join(words) == text + "<|endoftext|>"
    assert [t for wt in word_tokens for t in wt] == toks + [tok.eot]
    assert words[0] == "hello" and words[1] == " world" and words[-1] == "<|endoftext|>"
    # non-ASCII text encodes to UTF-8 byte tokens and survives the unicode-aware split
    toks2 = tok.encode("café ñ")
    assert tok.decode(toks2) == "café ñ"
    pieces, piece_tokens = tok.split_tokens_on_unicode(toks2)
    assert "".join(pieces) == "café ñ" and all("�" not in p for p in pieces)
    # a wrong-sized vocabulary file is rejected
    import pytest
    bad = fake_vocab + ".bad"
    with open(fake_vocab, "rb") as f, open(bad, "wb") as g:
        g.write(b"".join(f.readlines()[:100]))
    with pytest.raises(ValueError):
        tokmod.get_tokenizer(True, language="en", vocab_path=bad)


def test_ami_dataset_wrapper(tmp_path):
    """ami_kaldi.pkl format of the reference's README (README.md:64-71) -> the (audio, mel, duration, text, starts, ends, fid)
    item tuple; RIFF WAV segment written by hand."""
    import importlib
    import pickle
    import struct
    import numpy as np
    ds = importlib.import_module("whisper-char-alignment_amd.dataset")
    pcm = (np.sin(np.arange(8000) * 0.05) * 8000).astype("<i2")
    wav = tmp_path / "seg.wav"
    data = pcm.tobytes()
    wav.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) +
                    b"data" + struct.pack("<I", len(data)) + data)
    sid = "AMI_TS3003d_H03_MTD012ME_0255148_0255515"
    pkl = tmp_path / "ami_kaldi.pkl"
    pkl.write_bytes(pickle.dumps({sid: [("hello", 0.05, 0.2), ("", 0.2, 0.25), ("world", 0.25, 0.45)]}))
    scp = tmp_path / "ami.scp"
    scp.write_text("%s %s\n" % (sid, wav))
    d = ds.AMI(str(scp), compute_mel=False, alignment_file=str(pkl))
    assert len(d) == 1
    audio, mel, duration, text, starts, ends, fid = d[0]
    assert duration == 8000 and audio.shape[-1] == 480000 and mel is None and fid == sid
    assert text == "hello world" and starts == [0.05, 0.25] and ends == [0.2, 0.45]
    import pytest
    scp.write_text("missing_segment %s\n" % wav)
    with pytest.raises(KeyError):
        ds.AMI(str(scp), compute_mel=False, alignment_file=str(pkl))


# ----------------------------------------------------------------------------- FLAC (LibriSpeech originals, dataset.py:104)
def test_flac_rfc9639_example_known_answer():
    """The 57-byte example stream of RFC 9639 (appendix D.1: one 1-sample stereo frame, 16 bit, 44.1 kHz, two VERBATIM
    subframes with 2 and 4 wasted bits). Hand decode: subframe 1 = 0b01100011111101 << 2 = 25588, subframe 2 =
    0b001010001011 << 4 = 10416; header CRC-8 0xbf and frame CRC-16 0xaa9a are verified by the decoder."""
    audio = importlib.import_module("whisper-char-alignment_amd.audio")
    b = bytes.fromhex("664c6143800000221000100000000f00000f0ac442f0000000013e84b41807dc690307586a3dad1a2e0ffff869180000bf0358fd03128baa9a")
    y, sr = audio._read_flac(b)
    assert sr == 44100 and y.shape == (2, 1)
    assert np.array_equal(np.round(y[:, 0] * 32768).astype(int), [25588, 10416])
    bad = bytearray(b)
    bad[-4] ^= 0x10  # corrupt one payload bit: the frame CRC-16 must catch it
    with pytest.raises(ValueError):
        audio._read_flac(bytes(bad))
    huge = bytearray(b)
    huge[21] |= 0x0f  # STREAMINFO total_samples = 2^35 + ...: must be rejected after COUNTING the stream, not by allocating 8 x 2^35 floats
    huge[22:26] = b"\xff\xff\xff\xff"
    with pytest.raises(ValueError):
        audio._read_flac(bytes(huge))


def test_flac_decoder_round_trip(tmp_path):
    """Every stream shape the decoder claims, encoded by the in-test encoder (tests/flac_fixture.py) and decoded bit-exactly:
    LPC / FIXED 0-4 / VERBATIM / CONSTANT subframes, Rice and Rice2 with escaped partitions, partition orders 0-3, wasted
    bits, the four stereo modes, 8 / 16 / 24 bit, a short last frame, STREAMINFO without a sample count."""
    import flac_fixture as ff
    audio = importlib.import_module("whisper-char-alignment_amd.audio")
    rng = np.random.default_rng(0)
    t = np.arange(9000)
    speechy = (6000 * np.sin(t * 0.03) * np.sin(t * 0.0007) + 200 * rng.standard_normal(len(t))).astype(np.int64)
    cases = []
    for kinds in [("lpc",), (("fixed", 0), ("fixed", 1), ("fixed", 2), ("fixed", 3), ("fixed", 4)), ("verbatim",), ("lpc", ("fixed", 2), "verbatim")]:
        for rice2 in (False, True):
            for esc in (False, True):
                for porder in (0, 3):
                    cases.append(dict(pcm=speechy, kinds=kinds, rice2=rice2, escape_first=esc, porder=porder, blocksize=1024))
    cases.append(dict(pcm=np.full(5000, -1234), kinds=("constant",), blocksize=4096))
    cases.append(dict(pcm=speechy * 8, kinds=("lpc",), blocksize=1152))                       # 3 wasted bits
    cases.append(dict(pcm=speechy, kinds=("lpc",), blocksize=4096, with_total=False))          # unknown length: counted first
    cases.append(dict(pcm=speechy // 64, kinds=(("fixed", 2),), bps=8, blocksize=576))
    cases.append(dict(pcm=speechy * 200 + 7, kinds=("lpc", ("fixed", 4)), bps=24, blocksize=2048, rice2=True))
    right = np.roll(speechy, 5) + (rng.integers(-50, 50, len(speechy)))
    for mode in ("independent", "left_side", "right_side", "mid_side"):
        cases.append(dict(pcm=np.stack([speechy, right]), kinds=("lpc", ("fixed", 1)), stereo=mode, blocksize=1024))
    for c in cases:
        pcm = np.asarray(c.pop("pcm"), dtype=np.int64)
        bps = c.get("bps", 16)
        y, sr = audio._read_flac(ff.encode(pcm, **c))
        assert sr == 16000
        got = np.round(np.asarray(y, dtype=np.float64) * (1 << (bps - 1))).astype(np.int64)
        assert got.shape == pcm.shape and np.array_equal(got, pcm), c
    # through the file loader (container sniffing) and a truncated file
    p = tmp_path / "x.flac"
    p.write_bytes(ff.encode(speechy))
    y, sr = audio.load_audio(str(p))
    assert sr == 16000 and np.array_equal(np.round(y * 32768).astype(np.int64), speechy)
    with pytest.raises(ValueError):
        audio._read_flac(ff.encode(speechy)[:3000])


def test_librispeech_dataset_flac_tree(tmp_path):
    """dataset.LibriSpeech (dataset.py:67-122) on a fake corpus tree: `<root>/<split>/<spk>/<chap>/<fid>.flac`,
    `*.trans.txt` transcripts, `ls_alignment_<split>.txt` word alignments (python-literal lists, empty words dropped),
    scp lines `<fid> <path>`; FLAC decoded in-tree; audio longer than 30 s trimmed by pad_or_trim, duration kept."""
    import flac_fixture as ff
    ds = importlib.import_module("whisper-char-alignment_amd.dataset")
    rng = np.random.default_rng(3)
    root = tmp_path / "LibriSpeech"
    split = "test-clean"
    scp_lines, ali_lines = [], []
    lens = {"1089-134686-0000": 52000, "1089-134686-0001": 33000, "121-127105-0003": 16000 * 31}
    for fid, n in lens.items():
        spk, chap, _ = fid.split("-")
        d = root / split / spk / chap
        d.mkdir(parents=True, exist_ok=True)
        pcm = (3000 * np.sin(np.arange(n) * 0.01) + 100 * rng.standard_normal(n)).astype(np.int64)
        (d / (fid + ".flac")).write_bytes(ff.encode(pcm, blocksize=4096, kinds=("lpc",)))
        with open(d / ("%s-%s.trans.txt" % (spk, chap)), "a") as f:
            f.write("%s HELLO BIG WORLD\n" % fid)
        ali_lines.append("%s [('', 0.0, 0.2), ('HELLO', 0.2, 0.7), ('BIG', 0.7, 1.0), ('', 1.0, 1.1), ('WORLD', 1.1, 1.9)]\n" % fid)
        scp_lines.append("%s %s\n" % (fid, d / (fid + ".flac")))
    scp = tmp_path / "test-clean.wav.scp"
    scp.write_text("".join(scp_lines))
    ali = tmp_path / ("ls_alignment_%s.txt" % split)
    ali.write_text("".join(ali_lines))
    data = ds.LibriSpeech(str(scp), n_mels=80, device="cpu", model=None, compute_mel=False, alignment_file=str(ali))
    assert len(data) == 3
    for i, (fid, n) in enumerate(lens.items()):
        audio, mel, duration, text, starts, ends, got_fid = data[i]
        assert got_fid == fid and mel is None and duration == n and tuple(audio.shape) == (480000,)
        assert text == "HELLO BIG WORLD" and starts == [0.2, 0.7, 1.1] and ends == [0.7, 1.0, 1.9]
        pcm, dur, text2, st2, en2, fid2 = data.read(i)   # the reader-pool entry: un-padded PCM, trimmed to 30 s
        assert dur == n and len(pcm) == min(n, 480000) and (text2, st2, en2, fid2) == (text, starts, ends, fid)
        assert np.array_equal(pcm, audio.numpy()[:len(pcm)])
    assert data.duration_hint(2) > data.duration_hint(1)
    audio, mel, duration, text, starts, ends, fid = ds.Collate()([data[0]])
    assert fid == "1089-134686-0000" and duration == 52000


def test_dropin_module_names_resolve():
    """whisper-char-alignment_amd/dropin on sys.path gives the reference's own import lines (timing.py:1-10,
    infer_ali.py:11-20, README.md:78-83) -- checked in a subprocess so that the stand-in `whisper` module does not leak
    into this test process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from timing import get_attentions, force_align, filter_attention, default_find_alignment\n"
        "from retokenize import encode, remove_punctuation, split_tokens_on_spaces\n"
        "from metrics import eval_n1, eval_n1_strict, get_seg_metrics, coverage_penalty\n"
        "from dataset import TIMIT, LibriSpeech, Collate\n"
        "import whisper\n"
        "from whisper.model import disable_sdpa\n"
        "from whisper.timing import median_filter, dtw\n"
        "from whisper.audio import HOP_LENGTH, SAMPLE_RATE, TOKENS_PER_SECOND\n"
        "from whisper.tokenizer import get_tokenizer\n"
        "assert (HOP_LENGTH, SAMPLE_RATE, TOKENS_PER_SECOND) == (160, 16000, 50)\n"
        "assert whisper.audio.HOP_LENGTH * 2 == 320\n"
        "tok = get_tokenizer(True, language='English'); assert len(tok.sot_sequence) == 3\n"
        "opt = whisper.DecodingOptions(language='en'); assert opt.language == 'en'\n"
        "try:\n"
        "    whisper.load_model('medium', download_root='/nonexistent')\n"
        "    raise SystemExit('load_model must not succeed without a local checkpoint')\n"
        "except RuntimeError as e:\n"
        "    assert 'never downloads' in str(e)\n"
        "print('ok')\n" % os.path.join(root, "whisper-char-alignment_amd", "dropin"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_bench_conditioning_probe_separates_sharp_from_flat_matrices():
    """bench.py's parity leg labels an utterance ill-conditioned when the ORACLE's own DTW path moves under relative noise on
    its aggregated matrix: a sharp diagonal map must be stable, a flat one (every path costs the same) must not."""
    import importlib.util
    import os
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from oracle import timing_ref, tokenizer_ref
    tok = tokenizer_ref.CharTokenizer()
    tt = tokenizer_ref.encode_char("one two three four", tok)
    _w, wt = tokenizer_ref.split_tokens_on_spaces(list(tt) + [tok.eot], tok, "char")
    n, F = len(tt) + 1, 240
    rows = torch.arange(n).float()[:, None]
    cols = torch.arange(F).float()[None, :]
    sharp = torch.exp(-0.5 * ((cols - (rows + 0.5) * F / n) / 3.0) ** 2) + 1e-3     # a clear diagonal ridge
    flat = torch.full((n, F), 0.5)
    for m, expect in ((sharp, False), (flat, True)):
        ti, tj = timing_ref.dtw(-m)
        st, en = timing_ref.jumps_to_times(ti, tj, wt)
        assert bench.oracle_is_ill_conditioned(m, tt, st, en) is expect


def test_alignment_head_tables_equal_the_published_masks():
    """engine.ALIGNMENT_HEADS (index lists used by from_checkpoint(name=) / --default_whisper_timing) against the upstream
    package's own base85 + gzip masks: every string must decompress (the gzip CRC-32 pins it byte for byte) to a
    [n_text_layer, n_text_head] boolean mask whose row-major true positions are exactly the list."""
    eng = importlib.import_module("whisper-char-alignment_amd.engine")
    assert set(eng.ALIGNMENT_HEADS_B85) | {"large"} == set(eng.ALIGNMENT_HEADS)
    for name, dump in eng.ALIGNMENT_HEADS_B85.items():
        d = eng.dims_for(name)
        assert eng.decode_alignment_heads(dump, d.n_text_layer, d.n_text_head) == eng.ALIGNMENT_HEADS[name], name
    assert eng.ALIGNMENT_HEADS["large"] == eng.ALIGNMENT_HEADS["large-v3"]


def test_bench_helpers_flop_accounting_and_core_count():
    """bench.py's executed-FLOP figure (SURVEY 8d formulas minus what the fused path elides) and its measured core count."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    eng = importlib.import_module("whisper-char-alignment_amd.engine")

    class A:
        chars = 64
    t = bench.executed_tflop_per_utt(eng.dims_for("medium"), A)
    # encoder 1.1381 + decoder 0.2103 TFLOP (logits 0.0073 not run); elided tail of the last decoder layer ~4.6 GFLOP
    assert 1.340 < t < 1.349, t
    assert abs((1.1381 + 0.2103) - t - 0.0046) < 0.0008
    used, measured, override = bench.host_cores(0)
    assert used == measured >= 1 and override is None
    used, measured, override = bench.host_cores(3)
    assert used == 3 and override == 3 and measured >= 1
