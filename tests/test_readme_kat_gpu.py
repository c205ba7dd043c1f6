"""Dormant known-answer test: the reference's README example (README.md:88-140) -- sample/test.wav, whisper-medium,
char alignment, aggregation='topk', topk=10, medfilt_width=3 -- must print

    0.00 0.70 Artificial / 0.70 1.38 intelligence / 1.38 1.52 is / 1.52 1.76 for / 1.76 2.06 real

It needs the real `medium.pt` (openai checkpoint format), which cannot be fetched offline: the test is skipped
unless WCA_MEDIUM_PT points at a local copy. With WCA_VOCAB (a local multilingual.tiktoken) the teacher text is
produced by the greedy ASR pre-pass exactly like the README (whisper.decode); without it the README's own
transcription is teacher-forced (char mode needs no vocabulary file for ASCII text). Tolerance: one 20 ms frame
(north_star), i.e. 0.02 s on every printed boundary."""
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

README_EXPECTED = [(0.00, 0.70, "Artificial"), (0.70, 1.38, "intelligence"), (1.38, 1.52, "is"), (1.52, 1.76, "for"),
                   (1.76, 2.06, "real")]
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.skipif(not os.environ.get("WCA_MEDIUM_PT") or not os.path.exists(os.environ.get("WCA_MEDIUM_PT", "")),
                    reason="needs a local whisper medium.pt (set WCA_MEDIUM_PT); no checkpoint exists offline")
def test_readme_known_answer(wca):
    m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
    tm, tk, rt, audio, decoding = m("timing"), m("tokenizer"), m("retokenize"), m("audio"), m("decoding")
    model = wca.WhisperAMD.from_checkpoint(os.environ["WCA_MEDIUM_PT"], device="cuda:0", max_batch=1)
    vocab = os.environ.get("WCA_VOCAB")
    tok = tk.get_tokenizer(model.is_multilingual, language="English", vocab_path=vocab if vocab and os.path.exists(vocab) else None)
    pcm = np.load(os.path.join(GOLD, "sample_pcm_int16.npy")).astype(np.float32) / 32768.0  # sample/test.wav (SPHERE, 46592 samples)
    duration = len(pcm)
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), 80, model=model)
    if vocab and os.path.exists(vocab):
        result = decoding.decode(model, mel, decoding.DecodingOptions(language="en", vocab_path=vocab))
        transcription = rt.remove_punctuation(result.text)
    else:
        transcription = "Artificial intelligence is for real"
    text_tokens = rt.encode(transcription, tok, aligned_unit_type="char")
    tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, *text_tokens, tok.eot]).cuda()
    max_frames = duration // 320
    assert max_frames == 145
    attn_w, _logits = tm.get_attentions(mel, tokens, model, tok, max_frames, medfilt_width=3, qk_scale=1.0)
    words, st, en, ws, scores = tm.force_align(attn_w, text_tokens, tok, aligned_unit_type="char", aggregation="topk", topk=10)
    got = [(float(st[i]), float(en[i]), w.strip()) for i, w in enumerate(words[:-1])]
    assert [g[2] for g in got] == [e[2] for e in README_EXPECTED]
    for (gs, ge, _), (es, ee, _) in zip(got, README_EXPECTED):
        assert abs(gs - es) <= 0.02 + 1e-9 and abs(ge - ee) <= 0.02 + 1e-9, (got, README_EXPECTED)
