#!/usr/bin/env python3
"""Timing of the greedy ASR pre-pass (wca_greedy_decode) at whisper-medium dims, random weights: per-step cost of the
KV-cached decode loop beside the encoder it shares with the alignment. Random weights never emit EOT on their own,
so the loop runs its full sample_len; usage: decode_bench.py [batch] [sample_len]."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
decoding = importlib.import_module("whisper-char-alignment_amd.decoding")
tokmod = importlib.import_module("whisper-char-alignment_amd.tokenizer")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dims = wca.dims_for("medium")
m = wca.WhisperAMD(dims, max_batch=B).load_state_dict(syn.random_state_dict(dims, seed=0))
tok = tokmod.get_tokenizer(True, language="en", task="transcribe")
sup, blank = decoding.filter_masks(tok, decoding.DecodingOptions(language="en"), dims.n_vocab)
pcm = torch.from_numpy(np.stack([syn.synth_audio(b) for b in range(B)])).cuda()
ns = [160000] * B


def run(sample_len):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.greedy_decode(None, pcm, ns, list(tok.sot_sequence), sup, blank, sample_len=sample_len, eot=tok.eot,
                    timestamp_begin=tok.timestamp_begin)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


run(4)
t_short = min(run(4) for _ in range(2))
t_long = min(run(S) for _ in range(2))
per_step = (t_long - t_short) / (S - 4)
print("B=%d: encoder+cross-KV+%d-step loop %.1f ms; %d-step loop %.1f ms -> %.3f ms per decode step (%.1f us per utterance-step)" %
      (B, 4, t_short, S, t_long, per_step, per_step * 1e3 / B), flush=True)

# ---- the whole ASR-teacher flow (infer_ali.py --teacher asr): encode -> greedy decode -> alignment re-using the encoder
# state, serial vs two-deep pipeline (next batch encoded on stream 1 while this one is decoded / aligned on stream 2)
retok = importlib.import_module("whisper-char-alignment_amd.retokenize")
row = [*tok.sot_sequence, tok.no_timestamps, *retok.encode(syn.synth_text(0, 64), tok, "char"), tok.eot]
tokens = torch.tensor([row] * B, dtype=torch.int64, device="cuda")
o = m.make_opts(aggregation="topk", topk=10, sot_len=len(tok.sot_sequence), medfilt_width=3)
kw = dict(sample_len=S, eot=tok.eot, timestamp_begin=tok.timestamp_begin)
NB = 4


def serial():
    for _ in range(NB):
        m.greedy_decode(None, pcm, ns, list(tok.sot_sequence), sup, blank, **kw)
        m.align_batch(None, None, tokens, [len(row)] * B, [500] * B, o)


def pipelined():
    m.encode_batch(pcm=pcm, n_samples=ns)
    for k in range(NB):
        if k + 1 < NB:
            m.encode_batch(pcm=pcm, n_samples=ns)
        m.greedy_decode(None, None, None, list(tok.sot_sequence), sup, blank, batch=B, **kw)
        m.align_batch(None, None, tokens, [len(row)] * B, [500] * B, o)


for name, fn in (("serial", serial), ("pipelined", pipelined)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / NB
    print("ASR-teacher flow, %s: %.1f ms per batch of %d (%d decode steps) = %.0f utt/s" % (name, dt * 1e3, B, S, B / dt), flush=True)
