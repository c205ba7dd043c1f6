"""GPU tests of the callers around the hot path: the infer_ali / probe_oracle / eval_ali drivers on a
tiny TIMIT-shaped corpus (SPHERE audio + .wrd files generated from the sample PCM fixture), random-init
tiny model. Checks plumbing, schemas and the per-head probe kernel path against the drop-in API."""
import ctypes as C
import glob
import importlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
WORDS = "the quick brown fox jumps over the lazy dog and then it runs far away from all of them now".split()


def _m(n):
    return importlib.import_module("whisper-char-alignment_amd." + n)


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    root = tmp_path_factory.mktemp("timit")
    pcm = np.load(os.path.join(GOLD, "sample_pcm_int16.npy"))
    lines = []
    for u in range(5):
        x = np.roll(pcm, 1000 * u)[: len(pcm) - 2000 * u]
        head = ("NIST_1A\n   1024\nsample_count -i %d\nsample_rate -i 16000\nchannel_count -i 1\nsample_n_bytes -i 2\n"
                "sample_byte_format -s2 01\nsample_coding -s3 pcm\nend_head\n" % len(x)).encode()
        wav = root / ("utt%d.wav" % u)
        wav.write_bytes(head + b" " * (1024 - len(head)) + x.astype("<i2").tobytes())
        words = WORDS if u % 2 == 0 else WORDS[:6]
        step = len(x) // (len(words) + 1)
        (root / ("utt%d.wrd" % u)).write_text("".join("%d %d %s\n" % (i * step, (i + 1) * step, w) for i, w in enumerate(words)))
        lines.append("utt%d %s\n" % (u, wav))
    scp = root / "test.scp"
    scp.write_text("".join(lines))
    return root, scp


def test_infer_ali_and_eval_ali(corpus, capsys):
    root, scp = corpus
    infer = _m("infer_ali")
    out = root / "out"
    args = infer.parse_args(["--model", "tiny", "--random_init", "--dataset", "TIMIT", "--scp", str(scp), "--output_dir", str(out),
                             "--aggr", "topk", "--topk", "5", "--aligned_unit_type", "char", "--medfilt_width", "3", "--batch_size", "2",
                             "--save_prediction", "--strict", "--tolerance", "0.05", "--teacher", "text"])
    infer.infer_dataset(args)
    js = glob.glob(str(out / "*.json"))
    assert len(js) == 1
    res = json.load(open(js[0]))
    for k in ("precision", "recall", "f1", "r_value", "model", "aggr", "topk", "aligned_unit_type", "tolerance"):
        assert k in res
    assert res["teacher"] == "text" and "teacher_note" in res  # not the reference's protocol: said so in the results
    import joblib
    pk = glob.glob(str(out / "*-predictions.pkl"))
    preds = joblib.load(pk[0])
    assert sorted(preds) == [0, 1, 2, 3, 4]
    for p in preds.values():
        assert set(p) == {"starts", "ends", "texts", "starts_hat", "ends_hat", "predwords", "fids"}
        assert len(p["ends_hat"]) == len(p["texts"]) and p["predwords"][-1] == "<|endoftext|>"
        assert np.all(np.diff(p["ends_hat"]) >= 0) and np.all(np.asarray(p["starts_hat"])[1:] == np.asarray(p["ends_hat"])[:-1])
    ev = _m("eval_ali")
    r = ev.run_eval(ev.parse_args(["--pred", pk[0], "--tolerance", "0.05"]))
    assert abs(r["precision"] - res["precision"]) < 1e-9 and abs(r["recall"] - res["recall"]) < 1e-9


def test_infer_ali_precision_split(corpus):
    """--forward_precision split (the reference-precision mode) through the CLI: same schemas, and on this tiny random model the word
    times of the two modes agree to within a few frames (they differ by operand rounding only)."""
    root, scp = corpus
    infer = _m("infer_ali")
    import joblib
    preds = {}
    for mode in ("f16", "split", "reference"):
        out = root / ("out_" + mode)
        argv = ["--model", "tiny", "--random_init", "--dataset", "TIMIT", "--scp", str(scp), "--output_dir", str(out),
                "--aggr", "topk", "--topk", "5", "--aligned_unit_type", "char", "--medfilt_width", "3", "--batch_size", "2",
                "--save_prediction", "--teacher", "text"]
        args = infer.parse_args(argv + (["--forward_precision", mode] if mode != "reference" else []))   # reference = the CLI's default
        assert args.forward_precision == mode
        infer.infer_dataset(args)
        res = json.load(open(glob.glob(str(out / "*.json"))[0]))
        assert res["forward_precision"] == mode and 0.0 <= res["precision"] <= 1.0   # (`precision` = the P of P/R/F1, reference schema)
        preds[mode] = joblib.load(glob.glob(str(out / "*-predictions.pkl"))[0])
    assert sorted(preds["split"]) == sorted(preds["f16"]) == [0, 1, 2, 3, 4]
    same = sum(int(np.array_equal(preds["f16"][n]["ends_hat"], preds["split"][n]["ends_hat"])) for n in preds["f16"])
    assert same >= 3, same
    # "reference" (the default) and "split" name the same mode
    assert all(np.array_equal(preds["reference"][n]["ends_hat"], preds["split"][n]["ends_hat"]) for n in preds["split"])


def test_infer_ali_teacher_asr(corpus, fake_vocab, capsys):
    """The reference's own flow (infer_ali.py:60-68): greedy ASR pre-pass -> remove_punctuation -> char tokens -> alignment
    re-using the encoder state. Random weights give a meaningless hypothesis; this checks the plumbing end to end
    (every utterance is either aligned with monotone word times or reported as skipped)."""
    root, scp = corpus
    infer = _m("infer_ali")
    out = root / "out_asr"
    args = infer.parse_args(["--model", "tiny", "--random_init", "--dataset", "TIMIT", "--scp", str(scp), "--output_dir", str(out),
                             "--aggr", "topk", "--topk", "5", "--aligned_unit_type", "char", "--medfilt_width", "3", "--batch_size", "2",
                             "--save_prediction", "--tolerance", "0.05", "--teacher", "asr", "--vocab", fake_vocab])
    infer.infer_dataset(args)
    js = glob.glob(str(out / "*.json"))
    assert len(js) == 1
    res = json.load(open(js[0]))
    assert res["teacher"] == "asr" and "f1" in res
    import joblib
    preds = joblib.load(glob.glob(str(out / "*-predictions.pkl"))[0])
    printed = capsys.readouterr().out
    for u in range(5):
        assert u in preds or ("utt%d" % u) in printed
    for p in preds.values():
        assert np.all(np.diff(p["ends_hat"]) >= 0)


def test_infer_ali_default_teacher_is_asr_and_needs_vocab(corpus):
    """The reference always aligns the ASR hypothesis (infer_ali.py:60-68): that is the default; without a vocabulary
    file the CLI stops with a clear message instead of silently teacher-forcing the ground truth."""
    root, scp = corpus
    infer = _m("infer_ali")
    args = infer.parse_args(["--model", "tiny", "--weights", "/nonexistent/tiny.pt", "--scp", str(scp), "--output_dir", str(root / "x")])
    assert args.teacher is None
    with pytest.raises(SystemExit, match="--vocab"):
        infer.infer_dataset(args)


def test_infer_ali_default_whisper_timing_honours_strict_and_save(corpus):
    """--default_whisper_timing (timing.py:116-186) goes through the same scoring / saving code as the main path
    (infer_ali.py:114-132): --strict and --save_prediction are honoured, the model's official alignment heads are used."""
    root, scp = corpus
    infer = _m("infer_ali")
    engine = _m("engine")
    import joblib
    res = {}
    for strict in (False, True):
        out = root / ("out_dwt%d" % strict)
        argv = ["--model", "tiny", "--random_init", "--dataset", "TIMIT", "--scp", str(scp), "--output_dir", str(out),
                "--aligned_unit_type", "char", "--medfilt_width", "7", "--batch_size", "2", "--save_prediction", "--tolerance", "0.05",
                "--teacher", "text", "--default_whisper_timing"] + (["--strict"] if strict else [])
        infer.infer_dataset(infer.parse_args(argv))
        res[strict] = json.load(open(glob.glob(str(out / "*.json"))[0]))
        preds = joblib.load(glob.glob(str(out / "*-predictions.pkl"))[0])
        assert sorted(preds) == [0, 1, 2, 3, 4]
        for p in preds.values():
            assert len(p["ends_hat"]) == len(p["predwords"]) - 1 and np.all(np.diff(p["ends_hat"]) >= 0)
    assert res[True]["strict"] is True and res[False]["strict"] is False
    assert engine.ALIGNMENT_HEADS["tiny"] == [(2, 2), (3, 0), (3, 2), (3, 3), (3, 4), (3, 5)]
    assert len(engine.ALIGNMENT_HEADS["medium"]) == 6 and engine.model_name_from_dims(engine.dims_for("medium")) == "medium"
    assert engine.model_name_from_dims(engine.dims_for("large-v2")) is None and engine.model_name_from_dims(engine.dims_for("large-v3")) == "large-v3"


def test_infer_ali_pipeline_throughput(wca, tmp_path):
    """The CLI gets the engine's throughput: a 384-utterance TIMIT-shaped corpus (10 s SPHERE files, .wrd ground truth,
    64-char transcripts) through infer_ali --teacher text at whisper-medium dimensions, batch 64, must run within 20 %
    of the enqueue / fetch loop bench.py times on resident inputs at the same batch size (reader threads, pinned
    staging and the two-deep enqueue / fetch hide the host I/O)."""
    import time
    infer, syn, tk, rt, tm = _m("infer_ali"), _m("synthetic"), _m("tokenizer"), _m("retokenize"), _m("timing")
    B, n_utt = 64, 384
    lines = []
    for u in range(n_utt):
        x = (syn.synth_audio(5000 + u, 160000) * 32767.0).astype("<i2")
        head = ("NIST_1A\n   1024\nsample_count -i %d\nsample_rate -i 16000\nchannel_count -i 1\nsample_n_bytes -i 2\n"
                "sample_byte_format -s2 01\nsample_coding -s3 pcm\nend_head\n" % len(x)).encode()
        wav = tmp_path / ("s%d.wav" % u)
        wav.write_bytes(head + b" " * (1024 - len(head)) + x.tobytes())
        words = syn.synth_text(5000 + u, 64).split()
        step = len(x) // (len(words) + 1)
        (tmp_path / ("s%d.wrd" % u)).write_text("".join("%d %d %s\n" % (i * step, (i + 1) * step, w) for i, w in enumerate(words)))
        lines.append("s%d %s\n" % (u, wav))
    scp = tmp_path / "big.scp"
    scp.write_text("".join(lines))
    dims = wca.dims_for("medium")
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=B, precision="f16").load_state_dict(syn.random_state_dict(dims, seed=0))
    model.set_precision("reference")   # the CLI's default --forward_precision (the contract mode): both rates in the same arithmetic
    tok = tk.get_tokenizer(True, language="English")
    # reference rate: bench.py's loop (inputs resident in HBM, two batches in flight, host tail overlapped)
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    pcm = torch.from_numpy(np.stack([syn.synth_audio(5000 + u, 160000) for u in range(B)])).cuda()
    tts = [rt.encode(syn.synth_text(5000 + u, 64), tok, "char") for u in range(B)]
    toks = torch.tensor([[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts], dtype=torch.int64).cuda()

    def loop(steps):
        t0 = time.time()
        for i in range(steps):
            model.align_batch(pcm, [160000] * B, toks, [69] * B, [500] * B, opts, enqueue_only=True)
            if i > 0:
                jump, _ = model.fetch(B, 69, opts)
                for j in range(B):
                    tm.words_from_jump_frames(jump[j], tts[j], tok, "char")
        model.fetch(B, 69, opts)
        torch.cuda.synchronize()
        return steps * B / (time.time() - t0)

    loop(2)
    rate_bench = loop(6)
    argv = ["--model", "medium", "--random_init", "--dataset", "TIMIT", "--scp", str(scp), "--aggr", "topk", "--topk", "10",
            "--aligned_unit_type", "char", "--medfilt_width", "3", "--batch_size", str(B), "--teacher", "text", "--tolerance", "0.05",
            "--readers", "8"]
    infer.infer_dataset(infer.parse_args(argv + ["--output_dir", str(tmp_path / "warm")]), model=model)   # page cache + buffers warm
    r = infer.infer_dataset(infer.parse_args(argv + ["--output_dir", str(tmp_path / "timed")]), model=model)
    rate_cli = r["utterances"] / r["seconds"]
    line = "infer_ali CLI: %d utterances in %.2f s = %.0f utt/s; enqueue/fetch loop on resident inputs %.0f utt/s (ratio %.2f)" % (
        r["utterances"], r["seconds"], rate_cli, rate_bench, rate_cli / rate_bench)
    print(line)
    if os.path.isdir(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")):
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_cli_throughput.txt"), "a") as f:
            f.write(line + "\n")
    assert r["utterances"] == n_utt
    assert rate_cli >= 0.8 * rate_bench, line
    del model


def test_probe_heads_matches_per_head_force_align(wca):
    """wca_probe_heads == force_align(w[l,h][None,None], aggregation='mean') for every head (probe_oracle.py:88-90)."""
    tm, tk, probe = _m("timing"), _m("tokenizer"), _m("probe_oracle")
    from oracle import timing_ref
    tok = tk.get_tokenizer(True, language="English")
    g = torch.Generator().manual_seed(9)
    L, H, n, F = 3, 4, 40, 210
    w = torch.softmax(torch.randn(L, H, n, F, generator=g) * 4, -1).cuda()
    eng = _m("engine").default_engine(0)
    scores, jumps = probe.probe_heads(eng, w, 3)
    _, ref_scores = timing_ref.filter_attention(w.cpu(), L * H)
    for s, (l, h), _ in ref_scores:
        assert abs(scores[l * H + h] - s) <= 2e-5 * abs(s)
    tt = [64] * (n - 5)
    for l in range(L):
        for h in range(H):
            # the drop-in API on this single head: its (GPU-computed) matrix is within fp32 rounding of torch's, and the
            # probe's path for the head is BIT-EXACT w.r.t. the oracle DTW of that same matrix (north_star)
            words, st, en, matrix, _ = tm.force_align(w[l, h][None, None], tt, tok, "char", "mean", topk=1)
            m = (w[l, h] / w[l, h].norm(dim=-2, keepdim=True))[3:-1].cpu()
            np.testing.assert_allclose(matrix.numpy(), m.numpy(), rtol=2e-5, atol=1e-8)
            ti, tj = timing_ref.dtw(-matrix)
            jm = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
            assert np.array_equal(jumps[l * H + h], tj[jm]), (l, h)


def test_probe_strict_scoring_on_device_equals_host_loop(wca):
    """wca_probe_strict_tp == metrics.eval_n1_strict (golden-pinned to the reference's metrics.py:45-72) for every head:
    repeated words (the `used` set), reference boundaries exactly `tolerance` away (float64 comparison), more / fewer
    reference words than hypothesis words, no match at all."""
    probe, metrics = _m("probe_oracle"), _m("metrics")
    g = torch.Generator().manual_seed(21)
    L, H, n, F = 3, 4, 44, 260
    w = torch.softmax(torch.randn(L, H, n, F, generator=g) * 4, -1).cuda()
    eng = _m("engine").default_engine(0)
    scores, jumps = probe.probe_heads(eng, w, 3)
    N = n - 3 - 1
    rng = np.random.default_rng(4)
    hyp_words = ["the", "cat", "the", "dog", "The,", "cat", "sat", "the", "end"]
    cuts = np.sort(rng.choice(np.arange(1, N), size=len(hyp_words) - 1, replace=False))
    wb = np.concatenate([[0], cuts, [N]])          # word i covers token rows wb[i] .. wb[i+1]-1; its end boundary is row wb[i+1]
    wb_end = np.minimum(wb[1:], N - 1)
    for case, ref_words in enumerate((hyp_words, hyp_words[:-2], hyp_words + ["extra", "the"], ["none"] * 5)):
        # reference boundaries: some head's predictions shifted by exactly one tolerance, a little more, or far away
        base = (jumps[case % (L * H)] / 50.0)[wb_end]
        shifts = np.array([0.0, 0.02, -0.02, 0.021, 0.5, -0.019, 0.02, 0.0, 1.0, 0.0, 0.02])
        ends = np.array([base[min(j, len(base) - 1)] + shifts[j % len(shifts)] for j in range(len(ref_words))], dtype=np.float64)
        for tol in (0.02, 0.05, 0.0):
            tp = probe.probe_strict_tp(eng, L * H, wb_end, ends, ref_words, hyp_words, tol)
            for hd in range(L * H):
                ends_hat = (jumps[hd] / 50.0)[wb_end]
                rtp, rfp, rfn = metrics.eval_n1_strict(ends, ends_hat, ref_words, hyp_words, tol)
                assert tp[hd] == rtp, (case, tol, hd, int(tp[hd]), rtp)
            f1 = probe.strict_f1(tp, len(hyp_words), len(ref_words))
            for hd in range(L * H):
                t = int(tp[hd])
                assert f1[hd] == metrics.get_seg_metrics(t, t, len(hyp_words), len(ref_words))[2]


def test_probe_oracle_cli(corpus):
    root, scp = corpus
    probe = _m("probe_oracle")
    out = root / "probe"
    args = probe.parse_args(["--model", "tiny", "--random_init", "--scp", str(scp), "--output_dir", str(out), "--aligned_unit_type", "char",
                             "--medfilt_width", "3", "--hit_within", "5", "--strict", "--tolerance", "0.05"])
    probe.infer_dataset(args)
    res = json.load(open(glob.glob(str(out / "*.json"))[0]))
    assert 0.0 <= res["hit_rate"] <= 1.0 and "f1" in res


def test_reference_readme_snippet_runs_on_dropin_modules(tmp_path):
    """The reference's README example (README.md:78-131), line for line except torchaudio.load, with
    whisper-char-alignment_amd/dropin first on sys.path: `import whisper`, `from timing import get_attentions, force_align`,
    `from retokenize import encode, remove_punctuation` resolve to this engine. A tiny random checkpoint in openai format
    stands in for medium.pt (the known answer itself is tests/test_readme_kat_gpu.py). Runs in a subprocess."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, os, dataclasses
sys.path.insert(0, %(dropin)r)
sys.path.insert(0, %(root)r)
import numpy as np, torch, importlib
syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
eng = importlib.import_module("whisper-char-alignment_amd.engine")
dims = eng.dims_for("tiny")
torch.save({"dims": dataclasses.asdict(dims), "model_state_dict": syn.random_state_dict(dims, seed=1, cross_qk_std=0.1)}, %(ckpt)r)

# ---- README.md:78-131 ----
from timing import get_attentions, force_align
from retokenize import encode, remove_punctuation
import whisper
from whisper.tokenizer import get_tokenizer

AUDIO_SAMPLES_PER_TOKEN = whisper.audio.HOP_LENGTH * 2
AUDIO_TIME_PER_TOKEN = AUDIO_SAMPLES_PER_TOKEN / whisper.audio.SAMPLE_RATE
DEVICE = 'cuda:0'
model = whisper.load_model("tiny", download_root=%(wdir)r)
model.to(DEVICE)
assert model.precision == "split"   # the drop-in's load_model returns the CONTRACT mode (every site on operand pairs = the reference's fp32 forward)
options = whisper.DecodingOptions(language="en")
tokenizer = get_tokenizer(model.is_multilingual, language='English')
audio = torch.from_numpy(np.load(%(pcm)r).astype(np.float32) / 32768.0)   # torchaudio.load(sample_audio) stand-in
audio = audio.squeeze()
duration = len(audio.flatten())
audio = whisper.pad_or_trim(audio.flatten())
mel = whisper.log_mel_spectrogram(audio, 80)
mel = mel.to(DEVICE)
transcription = remove_punctuation("Artificial intelligence is for real.")    # (whisper.decode needs a vocabulary file)
text_tokens = encode(transcription, tokenizer, aligned_unit_type='char')
tokens = torch.tensor([*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot]).to(DEVICE)
max_frames = duration // AUDIO_SAMPLES_PER_TOKEN
attn_w, logits = get_attentions(mel, tokens, model, tokenizer, max_frames, medfilt_width=3, qk_scale=1.0)
words, start_times, end_times, ws, scores = force_align(attn_w, text_tokens, tokenizer, aligned_unit_type='char', aggregation='topk', topk=10)
for i, word in enumerate(words[:-1]):
    print(f"{start_times[i]:.2f} {end_times[i]:.2f} {word.strip()}")
assert [w.strip() for w in words[:-1]] == ["Artificial", "intelligence", "is", "for", "real"]
assert tuple(attn_w.shape) == (4, 6, len(tokens), 145) and tuple(ws.shape) == (len(text_tokens) + 1, 145) and len(scores) == 10
assert start_times[0] == 0.0 and all(e >= s for s, e in zip(start_times, end_times)) and end_times[-1] <= 2.9
print("snippet ok")
""" % dict(dropin=os.path.join(root, "whisper-char-alignment_amd", "dropin"), root=root, ckpt=str(tmp_path / "tiny.pt"), wdir=str(tmp_path),
           pcm=os.path.join(GOLD, "sample_pcm_int16.npy"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "snippet ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_line_contract_and_rccl_collation_path():
    """bench.py end to end on a tiny model: ONE JSON line with the contract's keys + roofline + cpu_baseline, the oracle parity
    leg at the timed configuration, and (WCA_FORCE_DIST=1) the RCCL process group / GPU all-gather / barrier path that the
    N > 1 runs take, with one rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WCA_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "tiny", "--batch", "4", "--steps", "4", "--warmup", "1",
                        "--distinct-batches", "2", "--seconds", "4", "--chars", "24", "--cpu-utts", "2", "--aligned-utts", "2"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak" and d["dtype"].startswith("f16x2") and d["value"] > 0
    assert d["config"]["precision"].startswith("reference") and "separate launches" in d["config"]["layernorm"]   # the contract mode, engine defaults
    f16 = d["f16_operating_point"]   # the fast mode is a secondary object of the same line, never `value`
    assert f16["value"] > 0 and f16["parity"]["precision"] == "f16" and f16["parity"]["word_boundaries"] > 0
    # the packed-record all-gathers (sizes + buffers) and the counter all-reduce really ran on RCCL with this one rank
    cfg = d["config"]
    assert cfg["dist_ranks"] == 1 and cfg["dist_backend"].startswith("nccl") and cfg["collated_utterances"] == 16
    assert cfg["collective_calls"]["all_gather"] == 2 and cfg["collective_calls"]["all_reduce"] == 1, cfg
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    par = cb["parity"]
    assert par["precision"] == "reference" and par["batch_invariant"] and par["within_one_frame"] == par["word_boundaries"], par   # no exclusions
    pal = cb["parity_alignment_like"]
    assert pal["utterances"] == 2 and pal["word_boundaries"] > 0 and pal["within_one_frame"] == pal["word_boundaries"], pal


def test_bench_collation_through_the_c_abi():
    """bench.py --collate abi: the timed region's collation runs wca_allgather_results / wca_allreduce_counters (one rank, WCA_FORCE_DIST=1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WCA_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "tiny", "--batch", "4", "--steps", "3", "--warmup", "1",
                        "--distinct-batches", "2", "--seconds", "4", "--chars", "24", "--no-cpu-baseline", "--collate", "abi"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    cfg = d["config"]
    assert "C ABI" in cfg["collation"] and cfg["collated_utterances"] == 12
    assert cfg["collective_calls"]["all_gather"] == 2 and cfg["collective_calls"]["all_reduce"] == 1


def test_rccl_collation_through_the_c_abi(wca):
    """wca_comm_* / wca_allgather_results / wca_allreduce_counters: the end-of-run collation straight from libwca.so over
    librccl, no torch.distributed. One rank on this one-GPU box (a communicator of one): packed records survive the size gather +
    padded all-gather, an EMPTY shard too, a shard above the 64 KB capacity floor (at one rank the capacity is the shard's own size, so
    nothing is retried here: the retry protocol between ranks with unequal shards is tests/test_distributed.py::
    test_abi_collation_retry_is_collective), counters are summed, and shard.allgather_results(..., engine=) takes this path."""
    shard = _m("shard")
    dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=1, _register=False, precision="f16")
    uid = wca.WhisperAMD.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    model.comm_init(uid, 0, 1)
    assert model.comm == (0, 1)
    res = {5: (np.array([0.0, 0.7]), np.array([0.7, 1.38])), 2: (np.zeros(0), np.zeros(0)), 9: (np.array([1.5]), np.array([1.52]))}
    before = dict(shard.COLLECTIVE_CALLS)
    back = shard.allgather_results(res, engine=model)
    assert sorted(back) == [2, 5, 9] and all(np.array_equal(back[k][0], res[k][0]) and np.array_equal(back[k][1], res[k][1]) for k in res)
    assert shard.allgather_results({}, engine=model) == {}
    rng = np.random.default_rng(0)
    big = {i: (np.sort(rng.random(20)), np.sort(rng.random(20)) + 1.0) for i in range(400)}   # 131 KB packed: above the 64 KB capacity floor
    back = shard.allgather_results(big, engine=model)
    assert sorted(back) == list(range(400)) and all(np.array_equal(back[i][1], big[i][1]) for i in big)
    assert shard.allreduce_counters(3, 5, 7, engine=model) == (3, 5, 7)
    assert shard.COLLECTIVE_CALLS["all_gather"] == before["all_gather"] + 6 and shard.COLLECTIVE_CALLS["all_reduce"] == before["all_reduce"] + 1
    with pytest.raises(wca._lib.WcaError, match="already has a communicator"):
        model.comm_init(uid, 0, 1)
    model.comm_destroy()
    assert model.comm is None
    with pytest.raises(wca._lib.WcaError, match="no communicator"):
        model.allreduce_counters(1)
