#!/usr/bin/env python3
"""Throughput of the reference's REAL per-utterance flow (infer_ali.py:57-101: whisper.decode pre-pass -> teacher text ->
get_attentions / force_align) at the bench configuration, through the product's pipeline: the encoder and the cross-K/V run ONCE
per micro-batch (wca_encode_batch), the greedy decode (wca_greedy_decode) and the alignment (wca_align_batch_enqueue with
pcm = NULL) both re-use them; the next batch's phase 1 is enqueued before the current batch is decoded, like the CLI does.

With seeded random weights the decoder never emits <|endoftext|> by itself, so the number of sampled tokens is FIXED per run
(EOT suppressed): a 64-character English sentence is 16-24 Whisper BPE tokens (+ 2 timestamp tokens); the teacher text that is
aligned afterwards is the synthetic 64-character text of bench.py (the decoded ids are noise). Reports utterances/s and the
split decode / align per batch.  usage: asr_flow_bench.py [B] [steps] [tokens,tokens,...]
Environment: WCA_PRECISION=f16|reference (default f16: round 3's record was taken in it); WCA_PART_CUS=32,64,... repeats the whole
measurement with the engine's CU partition (wca_set_cu_partition: phase 2 + decode loop on that many CUs, phase 1 on the rest)."""
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
retok = importlib.import_module("whisper-char-alignment_amd.retokenize")
timing = importlib.import_module("whisper-char-alignment_amd.timing")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
token_counts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16,24,32").split(",")]
dims = wca.dims_for(os.environ.get("WCA_MODEL", "medium"))
m = wca.WhisperAMD(dims, max_batch=B, precision="f16")
m.load_state_dict(syn.random_state_dict(dims, seed=0, cross_qk_std=0.08))
m.set_precision(os.environ.get("WCA_PRECISION", "f16"))
print("forward precision:", m.precision, flush=True)
tok = tok_mod.get_tokenizer(True, language="en")
initial = list(tok.sot_sequence)
sup = np.zeros(dims.n_vocab, np.uint8)
sup[tok.eot] = 1           # every row samples exactly `sample_len` tokens
sup[tok.no_timestamps] = 1
n_samples = 160000
n_batches = 4
batches = []
for bi in range(n_batches):
    pcm = np.stack([syn.synth_audio(bi * B + j, n_samples) for j in range(B)])
    tts = [retok.encode(syn.synth_text(bi * B + j, 64), tok, "char") for j in range(B)]
    rows = [[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts]
    n_max = max(len(r) for r in rows)
    tarr = np.full((B, n_max), tok.eot, dtype=np.int64)
    for j, r in enumerate(rows):
        tarr[j, :len(r)] = r
    batches.append(dict(pcm=torch.from_numpy(pcm).cuda(), tokens=torch.from_numpy(tarr).cuda(), n_tok=[len(r) for r in rows], tts=tts, n_max=n_max))
opts = m.make_opts(aggregation="topk", topk=10, sot_len=len(tok.sot_sequence), medfilt_width=3, qk_scale=1.0)
ns = [n_samples] * B
frames = [n_samples // 320] * B


def encode(i):
    m.encode_batch(pcm=batches[i % n_batches]["pcm"], n_samples=ns)


def decode(sample_len):
    return m.greedy_decode(None, None, None, initial, sup, None, batch=B, sample_len=sample_len, eot=tok.eot, timestamp_begin=tok.timestamp_begin,
                           apply_timestamp_rules=True, max_initial_timestamp_index=50)


def align(i):
    b = batches[i % n_batches]
    m.align_batch(None, None, b["tokens"], b["n_tok"], frames, opts, enqueue_only=True)


def fetch(i):
    b = batches[i % n_batches]
    jump, _sel = m.fetch(B, b["n_max"], opts)
    for j in range(B):
        timing.words_from_jump_frames(jump[j], b["tts"][j], tok, "char", want_words=False)


# reference point: alignment alone from resident PCM (bench.py's loop)
def text_teacher(n):
    for i in range(n):
        b = batches[i % n_batches]
        m.align_batch(b["pcm"], ns, b["tokens"], b["n_tok"], frames, opts, enqueue_only=True)
        if i > 0:
            fetch(i - 1)
    fetch(n - 1)


def measure():
    text_teacher(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    text_teacher(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("teacher = given text (bench.py's loop): %.1f ms per batch of %d -> %.0f utt/s" % (dt * 1e3 / steps, B, B * steps / dt), flush=True)

    for sample_len in token_counts:
        for rnd in range(2):   # round 0 = warm-up (buffer growth)
            n = 3 if rnd == 0 else steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t_dec = 0.0
            encode(0)
            for i in range(n):
                if i + 1 < n:
                    encode(i + 1)          # phase 1 of the next batch runs beside this batch's decode loop
                td = time.perf_counter()
                decode(sample_len)         # returns when the loop has finished (host reads the tokens)
                t_dec += time.perf_counter() - td
                align(i)
                fetch(i)                   # frees the K/V slot for batch i + 2
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("teacher = ASR pre-pass, %2d sampled tokens (+%d prompt positions): %.1f ms per batch of %d (decode loop %.1f ms = %.2f ms per position, "
              "rest %.1f ms) -> %.0f utt/s" % (sample_len, len(initial), dt * 1e3 / steps, B, t_dec * 1e3 / steps,
                                               t_dec * 1e3 / steps / (sample_len + len(initial) - 1), (dt - t_dec) * 1e3 / steps, B * steps / dt), flush=True)


for part in [0] + [int(x) for x in os.environ.get("WCA_PART_CUS", "").split(",") if x]:
    m.set_cu_partition(part)
    print("---- CU partition: %s" % ("off (every stream may use all %d CUs)" % 256 if part == 0 else "phase 2 / decode loop on %d CUs, phase 1 on the rest" % part), flush=True)
    measure()
m.set_cu_partition(0)
