#!/usr/bin/env python3
"""Greedy ASR pre-pass (wca_greedy_decode) at the bench's model / batch: ms per autoregressive step with the encoder state
already queued (wca_encode_batch), so that only the KV-cached loop is timed.  usage: decode_bench.py [B] [sample_len] [rounds]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
tok_mod = importlib.import_module("whisper-char-alignment_amd.tokenizer")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sample_len = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
model_name = os.environ.get("WCA_MODEL", "medium")
dims = wca.dims_for(model_name)
m = wca.WhisperAMD(dims, max_batch=B, precision="f16")
m.load_state_dict(syn.random_state_dict(dims, seed=0, cross_qk_std=0.08))
tok = tok_mod.get_tokenizer(True, language="en")
initial = list(tok.sot_sequence)
sup = np.zeros(dims.n_vocab, np.uint8)
sup[tok.eot] = 1  # never finish early: every round runs exactly sample_len steps
sup[tok.no_timestamps] = 1
pcm = torch.from_numpy(np.stack([syn.synth_audio(b, 160000) for b in range(B)])).cuda()
ns = np.full(B, 160000, np.int32)
kw = dict(sample_len=sample_len, eot=tok.eot, timestamp_begin=tok.timestamp_begin, apply_timestamp_rules=True,
          max_initial_timestamp_index=50)
modes = [(bool(int(x.split(":")[0])), int(x.split(":")[1])) for x in os.environ.get("WCA_DEC_MODES", "0:1,0:2,1:1,1:2").split(",")]
for mode in modes:
  m.set_decode_mode(*mode)
  print("decode mode: fused=%s streams=%d" % mode, flush=True)
  for r in range(rounds + 1):
    m.encode_batch(pcm=pcm, n_samples=ns)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, n_tok, lp = m.greedy_decode(None, None, None, initial, sup, None, batch=B, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = len(initial) - 1 + sample_len  # the prompt is fed position by position through the same step
    print("  round %d: B=%d %s  %.1f ms for %d sampled tokens (+%d prompt positions)  -> %.3f ms per position, %.0f utt/s decode-only"
          % (r, B, model_name, dt * 1e3, sample_len, len(initial), dt * 1e3 / steps, B / dt), flush=True)
  print("  tokens[0][:12] =", toks[0][:12].tolist(), " n_tok[0] =", int(n_tok[0]))
