#!/usr/bin/env python3
"""Diagnosis of the bench's parity leg: the utterances (ids 10000+u) whose GPU word times leave the one-frame band around the
fp32 CPU oracle. For each: do both select the same heads, how close are the swapped heads' oracle scores to the cut, how far
apart are the aggregated matrices, does the oracle DTW on the GPU's matrix reproduce the GPU's path (i.e. is the difference
upstream of the DTW), and how large a relative perturbation of the oracle's own matrix moves the oracle's path.
usage: parity_probe.py [n_utts] [first_id]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wca = importlib.import_module("whisper-char-alignment_amd")
syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
tok_mod = importlib.import_module("whisper-char-alignment_amd.tokenizer")
retok = importlib.import_module("whisper-char-alignment_amd.retokenize")
timing = importlib.import_module("whisper-char-alignment_amd.timing")
audio_mod = importlib.import_module("whisper-char-alignment_amd.audio")
from oracle import timing_ref, tokenizer_ref, whisper_ref  # noqa: E402

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 33
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
B, topk, medfilt, n_samples, chars = 64, 10, 3, 160000, 64
dims = wca.dims_for("medium")
sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
model = wca.WhisperAMD(dims, max_batch=B, precision="f16")
model.load_state_dict(sd)
tok = tok_mod.get_tokenizer(True, language="en")
opts = model.make_opts(aggregation="topk", topk=topk, sot_len=len(tok.sot_sequence), medfilt_width=medfilt, qk_scale=1.0)
ids = [first + u for u in range(n_utts)]
fill = (ids * ((B + len(ids) - 1) // len(ids)))[:B]
pcm = np.stack([syn.synth_audio(u, n_samples) for u in fill])
texts = [syn.synth_text(u, chars) for u in fill]
tts = [retok.encode(t, tok, "char") for t in texts]
rows = [[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts]
n_max = max(len(r) for r in rows)
toks = np.full((B, n_max), tok.eot, dtype=np.int64)
for j, r in enumerate(rows):
    toks[j, :len(r)] = r
pcm_d, toks_d = torch.from_numpy(pcm).cuda(), torch.from_numpy(toks).cuda()
jump, sel = model.align_batch(pcm_d, [n_samples] * B, toks_d, [len(r) for r in rows], [n_samples // 320] * B, opts)
mel_d = torch.stack([audio_mod.log_mel_spectrogram(audio_mod.pad_or_trim(torch.from_numpy(p)), dims.n_mels, model=model) for p in pcm]).cuda()
weights_gpu, _logits = model.get_attentions(mel_d, toks_d, [n_samples // 320] * B, medfilt, 1.0)
print("GPU batch done", flush=True)

torch.set_num_threads(min(16, os.cpu_count() or 1))
otok = tokenizer_ref.CharTokenizer()
ref = whisper_ref.WhisperRef({k: v.float() for k, v in sd.items()}, dims)
filt = audio_mod.mel_filters(dims.n_mels)
H = dims.n_text_head
for j, u in enumerate(ids):
    mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm[j])), filt)
    tt = tokenizer_ref.encode_char(texts[j], otok)
    tokens = torch.tensor([*otok.sot_sequence, otok.no_timestamps, *tt, otok.eot])
    w, _ = timing_ref.get_attentions(mel, tokens, ref, n_samples // 320, medfilt, 1.0)
    _words, rst, ren, rmat, rscores = timing_ref.force_align(w, tt, otok, "char", "topk", topk)
    _w, st, en = timing.words_from_jump_frames(jump[j], tts[j], tok, "char")
    st, en = np.asarray(st), np.asarray(en)
    off = int(np.sum(np.abs(st - rst) > 0.02 + 1e-9) + np.sum(np.abs(en - ren) > 0.02 + 1e-9))
    # the GPU's selection scores of its top-k heads against the oracle's scores of the same heads
    _gsel, gsc_sorted = timing.filter_attention(weights_gpu[j][:, :, :len(rows[j])], topk, 1, 1, 0)
    _oall, o_all = timing_ref.filter_attention(w, w.shape[0] * w.shape[1], 1, 1, 0)
    o_of = {(l, h): sc for sc, (l, h), _n in o_all}
    dev = max(abs(sc - o_of[lh]) / abs(o_of[lh]) for sc, lh, _n in gsc_sorted)
    print("utt %d: %d boundaries outside one frame; selection scores GPU vs oracle (top-%d heads): max rel deviation %.2e"
          % (u, off, topk, dev), flush=True)
    if off == 0:
        continue
    o_heads = [l * H + h for _s, (l, h), _n in rscores]
    g_heads = [int(x) for x in sel[j][:topk]]
    _sel_all, all_scores = timing_ref.filter_attention(w, w.shape[0] * w.shape[1], 1, 1, 0)
    score_of = {l * H + h: s for s, (l, h), _n in all_scores}
    ranked = sorted(score_of.values())
    kth, nxt = ranked[-topk], ranked[-topk - 1]
    print("   oracle heads %s\n   gpu    heads %s" % (sorted(o_heads), sorted(g_heads)))
    swapped = sorted(set(o_heads) ^ set(g_heads))
    print("   swapped heads %s oracle scores %s ; k-th score %.6f (k+1)-th %.6f rel gap %.2e"
          % (swapped, ["%.6f" % score_of[x] for x in swapped], kth, nxt, (kth - nxt) / kth))
    # the GPU's own aggregated matrix for this utterance: batched get_attentions (same GEMM kernels as align_batch)
    gw, gst, gen, gmat, gsc = timing.force_align(weights_gpu[j][:, :, :len(rows[j])], tts[j], tok, aligned_unit_type="char", aggregation="topk", topk=topk)
    gmat = torch.as_tensor(gmat).float().cpu()
    d = (gmat - rmat).abs()
    print("   GPU matrix (get_attentions + force_align) vs oracle: max abs diff %.3e (matrix max %.3e), rel fro %.3e; its boundaries outside: %d; equal to align_batch: %s"
          % (d.max(), rmat.abs().max(), d.norm() / rmat.norm(),
             int(np.sum(np.abs(np.asarray(gst) - rst) > 0.0200001) + np.sum(np.abs(np.asarray(gen) - ren) > 0.0200001)),
             np.array_equal(np.asarray(gst), st) and np.array_equal(np.asarray(gen), en)))
    ti, tj = timing_ref.dtw(-gmat)
    _ww, wt = tokenizer_ref.split_tokens_on_spaces(list(tt) + [otok.eot], otok, "char")
    s2, e2 = timing_ref.jumps_to_times(ti, tj, wt)
    print("   oracle DTW on the GPU matrix == GPU times: %s" % (np.array_equal(s2, np.asarray(gst)) and np.array_equal(e2, np.asarray(gen))))
    wgj = weights_gpu[j][:, :, :len(rows[j])].float().cpu()
    dw = (wgj - w).abs()
    print("   softmaxed weights GPU vs oracle: max abs diff %.3e, rel fro %.3e" % (dw.max(), dw.norm() / w.norm()))
    _ww, wt = tokenizer_ref.split_tokens_on_spaces(list(tt) + [otok.eot], otok, "char")
    rng = np.random.default_rng(1)
    for eps in (1e-4, 3e-4, 1e-3, 3e-3, 1e-2):
        moved = 0
        for _ in range(8):
            noisy = rmat * (1.0 + eps * torch.from_numpy(rng.standard_normal(tuple(rmat.shape)).astype(np.float32)))
            ti, tj = timing_ref.dtw(-noisy)
            s2, e2 = timing_ref.jumps_to_times(ti, tj, wt)
            moved += int(np.max(np.abs(s2 - rst)) > 0.0200001 or np.max(np.abs(e2 - ren)) > 0.0200001)
        print("   oracle path under %.0e relative noise on its matrix: moved in %d / 8 trials" % (eps, moved))
    print("   oracle starts %s\n   gpu    starts %s" % (np.round(rst, 2).tolist(), np.round(st, 2).tolist()), flush=True)
