#!/usr/bin/env python3
"""Parity at the north-star configuration: whisper-medium dims (seeded random weights), 10 s audio, 64-char
text, topk=10, medfilt 3 -- engine (f16 MFMA forward) vs the CPU oracle (fp32 forward)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
syn, tk, rt, tm, audio = m("synthetic"), m("tokenizer"), m("retokenize"), m("timing"), m("audio")
from oracle import timing_ref, whisper_ref, tokenizer_ref  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "medium"
qk_std = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
n_utts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dims = wca.dims_for(name)
t0 = time.time()
sd = syn.random_state_dict(dims, seed=0, cross_qk_std=qk_std)
model = wca.WhisperAMD(dims, max_batch=1, precision="f16").load_state_dict(sd)
ref = whisper_ref.WhisperRef(sd, dims)
tok, rtok = tk.get_tokenizer(True, language="English"), tokenizer_ref.CharTokenizer()
torch.set_num_threads(min(os.cpu_count(), 16))
print("setup %.1fs" % (time.time() - t0), flush=True)
tot = close = 0
for u in range(n_utts):
    pcm = syn.synth_audio(u, 160000)
    text = syn.synth_text(u, 64)
    tt = rt.encode(text, tok, "char")
    tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot])
    mel = audio.log_mel_spectrogram(audio.pad_or_trim(torch.from_numpy(pcm)), dims.n_mels, model=model)
    w, logits = tm.get_attentions(mel, tokens.cuda(), model, tok, 500, medfilt_width=3)
    words, st, en, matrix, scores = tm.force_align(w, tt, tok, "char", "topk", topk=10)
    t1 = time.time()
    rw, rlogits = timing_ref.get_attentions(mel.cpu(), tokens, ref, 500, 3, 1.0)
    rwords, rst, ren, rmatrix, rscores = timing_ref.force_align(rw, tt, rtok, "char", "topk", 10)
    dt = time.time() - t1
    dw = (w.cpu() - rw).abs()
    sel = [lh for _, lh, _ in scores]
    rsel = [lh for _, lh, _ in rscores]
    n = 2 * len(st)
    c = int((np.abs(st - rst) <= 0.0201).sum() + (np.abs(en - ren) <= 0.0201).sum())
    tot += n
    close += c
    print("utt %d: max|dW| %.2e mean|dW| %.2e  rel-err logits %.2e  max w %.3f  same top-10 heads %d/10  matrix max|d| %.2e  "
          "word times within 1 frame %d/%d  (cpu %.1fs)" % (u, dw.max().item(), dw.mean().item(),
          ((logits.cpu() - rlogits).abs().max() / rlogits.abs().max()).item(), rw.max().item(), len(set(sel) & set(rsel)),
          (matrix - rmatrix).abs().max().item(), c, n, dt), flush=True)
    print("   gpu starts", np.round(st, 2).tolist())
    print("   ref starts", np.round(rst, 2).tolist())
print("TOTAL within one frame: %d/%d" % (close, tot))
