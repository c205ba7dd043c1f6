#!/usr/bin/env python3
"""Parity legs on RAGGED utterances: the contract mode (and, for comparison, the f16 operating point) against the fp32 CPU oracle on
audio of 2-29.5 s and texts of 4-220 characters, whisper-medium dims, peaky seeded weights -- the shapes the fixed 10 s / 64-char legs
(tools/precision_ablation.py) never visit: different frame counts per utterance in one micro-batch, decoder lengths from 9 to 225 rows,
filter windows that touch the reflect padding at other places.

  leg A: the north-star settings (char units, aggr topk, topk 10, medfilt 3) on ragged lengths
  leg B: the reference CLI's defaults (infer_ali.py:160-162: medfilt 7, aggr mean) in char units on ragged lengths

  python tools/parity_ragged.py --leg A [--utts 128] [--first-id 20000] [--oracle-only] [--modes reference,f16]

The oracle's word times (2-5 s of CPU per utterance) are cached in tools/cache/ (git-ignored, travels to the GPU box): run once with
--oracle-only in the build container. Output: gpurun_out/r04_parity_ragged_<leg>.{txt,json}."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LEGS = {"A": dict(aggr="topk", topk=10, medfilt=3), "B": dict(aggr="mean", topk=15, medfilt=7)}


def spec(u):
    """(n_samples, n_chars) of utterance id u: 2-29.5 s, 3-9 characters per second (4..220)."""
    rng = np.random.default_rng(777 + int(u))
    seconds = float(rng.uniform(2.0, 29.5))
    n_samples = int(seconds * 16000)
    chars = int(min(max(seconds * rng.uniform(3.0, 9.0), 4), 220))
    return n_samples, chars


def oracle_word_times(leg, cfg, sd, dims, syn, audio_mod, ids, only_cached=False, model="medium"):
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    key = "oracle_ragged_%s_%s_peaky008_ids%d-%d" % (leg, model, ids[0], ids[-1])
    path = os.path.join(ROOT, "tools", "cache", key + ".npz")
    store = {}
    for p in (path, path + ".part.npz"):
        if os.path.exists(p):
            store = dict(np.load(p, allow_pickle=False))
            break
    if all("st_%d" % u in store for u in ids):
        print("oracle cache:", path, flush=True)
        return [(store["st_%d" % u], store["en_%d" % u]) for u in ids], path
    if only_cached:
        raise SystemExit("no oracle cache for leg %s ids %d-%d: run with --oracle-only first (CPU, no GPU needed)" % (leg, ids[0], ids[-1]))
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 64))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tok = tokenizer_ref.CharTokenizer()
    ref = whisper_ref.WhisperRef(sd, dims)
    filt = audio_mod.mel_filters(dims.n_mels)
    t00 = time.time()
    for i, u in enumerate(ids):
        if "st_%d" % u in store:
            continue
        n_samples, chars = spec(u)
        pcm = torch.from_numpy(syn.synth_audio(u, n_samples))
        text = syn.synth_text(u, chars)
        mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(pcm), filt)
        tt = tokenizer_ref.encode_char(text, tok)
        tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot])
        w, _ = timing_ref.get_attentions(mel, tokens, ref, n_samples // 320, cfg["medfilt"], 1.0)
        _words, st, en, _matrix, _s = timing_ref.force_align(w, tt, tok, "char", cfg["aggr"], cfg["topk"])
        store["st_%d" % u], store["en_%d" % u] = np.asarray(st, dtype=np.float64), np.asarray(en, dtype=np.float64)
        if i % 8 == 0:
            print("oracle utterance %d/%d (%.0f s)" % (i + 1, len(ids), time.time() - t00), flush=True)
            np.savez_compressed(path + ".part.npz", **store)
    np.savez_compressed(path, **store)
    if os.path.exists(path + ".part.npz"):
        os.remove(path + ".part.npz")
    return [(store["st_%d" % u], store["en_%d" % u]) for u in ids], path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", choices=sorted(LEGS), default="A")
    ap.add_argument("--utts", type=int, default=128)
    ap.add_argument("--first-id", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--modes", default="reference,f16")
    ap.add_argument("--model", default="medium", help="dimension family (wca.dims_for): medium | large-v2 | large-v3 ...")
    ap.add_argument("--oracle-only", action="store_true")
    args = ap.parse_args()
    cfg = LEGS[args.leg]

    wca = importlib.import_module("whisper-char-alignment_amd")
    m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
    syn, tok_mod, retok, timing, audio_mod = m("synthetic"), m("tokenizer"), m("retokenize"), m("timing"), m("audio")
    dims = wca.dims_for(args.model)
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
    ids = list(range(args.first_id, args.first_id + args.utts))
    oracle, cache_path = oracle_word_times(args.leg, cfg, sd, dims, syn, audio_mod, ids, only_cached=not args.oracle_only and not os.environ.get("WCA_ORACLE_ON_GPU_BOX"), model=args.model)
    if args.oracle_only:
        print("oracle word times:", cache_path)
        return

    device = torch.device("cuda", 0)
    model = wca.WhisperAMD(dims, device=str(device), max_batch=args.batch, precision="f16").load_state_dict(sd)
    tok = tok_mod.get_tokenizer(True, language="English")
    opts = model.make_opts(aggregation=cfg["aggr"], topk=cfg["topk"], sot_len=len(tok.sot_sequence), medfilt_width=cfg["medfilt"], qk_scale=1.0)
    batches = []
    for lo in range(0, len(ids), args.batch):
        chunk = ids[lo:lo + args.batch]
        sp = [spec(u) for u in chunk]
        smax = max(s for s, _ in sp)
        pcm = np.zeros((len(chunk), smax), dtype=np.float32)
        tts = []
        for j, (u, (ns, ch)) in enumerate(zip(chunk, sp)):
            pcm[j, :ns] = syn.synth_audio(u, ns)
            tts.append(retok.encode(syn.synth_text(u, ch), tok, "char"))
        rows = [[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts]
        n_max = max(len(r) for r in rows)
        toks = np.full((len(rows), n_max), tok.eot, dtype=np.int64)
        for j, r in enumerate(rows):
            toks[j, :len(r)] = r
        batches.append(dict(lo=lo, n=len(chunk), pcm=torch.from_numpy(pcm).to(device), tokens=torch.from_numpy(toks).to(device),
                            n_samples=[s for s, _ in sp], n_tok=[len(r) for r in rows], frames=[s // 320 for s, _ in sp], tts=tts))
    lines, record = [], dict(leg=args.leg, model=args.model, settings=cfg, utterances=len(ids), first_id=args.first_id, ids="%d-%d" % (ids[0], ids[-1]),
                             seconds_range=[min(spec(u)[0] for u in ids) / 16000.0, max(spec(u)[0] for u in ids) / 16000.0],
                             chars_range=[min(spec(u)[1] for u in ids), max(spec(u)[1] for u in ids)], modes={})
    head = ("ragged parity leg %s: %d utterances (ids %s), %.1f-%.1f s audio, %d-%d characters, whisper-%s dims, peaky seeded weights, char units, "
            "aggr %s%s, medfilt %d; fused wca_align_batch at B = %d against the fp32 CPU oracle"
            % (args.leg, len(ids), record["ids"], record["seconds_range"][0], record["seconds_range"][1], record["chars_range"][0], record["chars_range"][1], args.model,
               cfg["aggr"], " topk %d" % cfg["topk"] if cfg["aggr"] == "topk" else "", cfg["medfilt"], args.batch))
    print(head, flush=True)
    lines.append(head)
    for mode in args.modes.split(","):
        model.set_precision(mode)
        total = within = ident = 0
        worst, offenders = 0.0, []
        for b in batches:
            jump, _sel = model.align_batch(b["pcm"], b["n_samples"], b["tokens"], b["n_tok"], b["frames"], opts)
            for j in range(b["n"]):
                _w, st, en = timing.words_from_jump_frames(jump[j], b["tts"][j], tok, "char")
                ost, oen = oracle[b["lo"] + j]
                assert len(st) == len(ost), (ids[b["lo"] + j], len(st), len(ost))
                d = np.concatenate([np.abs(np.asarray(st) - ost), np.abs(np.asarray(en) - oen)])
                total += d.size
                within += int((d <= 0.02 + 1e-9).sum())
                ident += int((d == 0).sum())
                worst = max(worst, float(d.max()))
                if (d > 0.02 + 1e-9).any():
                    offenders.append((ids[b["lo"] + j], int((d > 0.02 + 1e-9).sum()), float(d.max())))
        line = ("  %-9s sites %-60s boundaries %6d  within one frame %6d (%.4f)  identical %6d (%.4f)  worst %.2f s  utterances with a miss %d %s"
                % (mode, "+".join(model.precision_sites[0]) or "-", total, within, within / total, ident, ident / total, worst, len(offenders),
                   [o[0] for o in offenders[:12]]))
        print(line, flush=True)
        lines.append(line)
        record["modes"][mode] = dict(boundaries=total, within_one_frame=within, identical=ident, worst_s=worst, offenders=offenders)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    base = os.path.join(ROOT, "gpurun_out", "r04_parity_ragged_%s%s" % (args.leg, "" if args.model == "medium" else "_" + args.model))
    open(base + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(record, open(base + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
