#!/usr/bin/env python3
"""Experiment: P independent engines (each with its own arena and two streams) on ONE GPU, fed round-robin with
micro-batches of B utterances, against one engine with P*B -- does de-phasing the pipelines pay?
usage: two_engines.py P B steps"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
tk = importlib.import_module("whisper-char-alignment_amd.tokenizer")
retok = importlib.import_module("whisper-char-alignment_amd.retokenize")

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
dims = wca.dims_for("medium")
sd = syn.random_state_dict(dims, seed=0)
tok = tk.get_tokenizer(True, language="English")
engines = [wca.WhisperAMD(dims, max_batch=B, precision="f16").load_state_dict(sd) for _ in range(P)]
pcm = torch.from_numpy(np.stack([syn.synth_audio(u) for u in range(B)])).cuda()
rows = [[*tok.sot_sequence, tok.no_timestamps, *retok.encode(syn.synth_text(u, 64), tok, "char"), tok.eot] for u in range(B)]
n_max = max(len(r) for r in rows)
toks = np.full((B, n_max), tok.eot, dtype=np.int64)
for j, r in enumerate(rows):
    toks[j, :len(r)] = r
toks = torch.from_numpy(toks).cuda()
ns, nt, mf = [160000] * B, [len(r) for r in rows], [500] * B
opts = [e.make_opts(aggregation="topk", topk=10, sot_len=len(tok.sot_sequence), medfilt_width=3) for e in engines]


streams = [torch.cuda.Stream() for _ in range(P)]
torch.cuda.synchronize()


def run(n):
    pending = []
    for i in range(n):
        k = i % P
        if len(pending) >= 2 * P:  # two batches in flight per engine
            kk = pending.pop(0)
            engines[kk].fetch(B, n_max, opts[kk])
        with torch.cuda.stream(streams[k]):  # each engine's phase 1 on its own stream (the wrapper binds torch's current one)
            engines[k].align_batch(pcm, ns, toks, nt, mf, opts[k], enqueue_only=True)
        pending.append(k)
    for kk in pending:
        engines[kk].fetch(B, n_max, opts[kk])


run(2 * P)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d engine(s) x B=%d, %d micro-batches: %.1f utt/s (%.1f ms per micro-batch)" % (P, B, steps, steps * B / dt, 1e3 * dt / steps), flush=True)
