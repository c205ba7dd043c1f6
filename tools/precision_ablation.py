#!/usr/bin/env python3
"""Which stages of the forward need reference-precision operands?

For every row of a table of precision-site masks (wca_set_precision_sites) this aligns the SAME utterances -- bench.py's parity
leg: whisper-medium dims, peaky seeded weights, 10 s audio, 64-char text, topk 10, medfilt 3, fused wca_align_batch at B = 64 --
against the fp32 CPU oracle's word times and measures the throughput of the same loop bench.py times.

  python tools/precision_ablation.py [--utts 301] [--steps 24] [--rows NAME,NAME...] [--out gpurun_out/r04_precision_ablation.txt]

The oracle's word times (~4 s per utterance on 16 cores) come from the committed fixture tests/golden/oracle_word_times_medium_peaky.npz
(ids 100-131, 10000-10300 of the standard configuration) or are computed (with --oracle-only: on a machine without a GPU) into gpurun_out/.
"""
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

# name -> (sites, first encoder block of the ENC_* bits)
ROWS = [
    ("f16", "", 0),
    ("capture", "capture", 0),
    ("cross_kv+capture", "cross_kv,capture", 0),
    ("dec+cross_kv+capture", "dec,cross_kv,capture", 0),
    ("decoder side + logmel + conv", "logmel,conv,dec,cross_kv,capture", 0),
    ("decoder side + enc_attn (all 24)", "enc_attn,dec,cross_kv,capture", 0),
    ("decoder side + enc last 6", "enc_gemm,enc_attn,dec,cross_kv,capture", 18),
    ("decoder side + enc last 12", "enc_gemm,enc_attn,dec,cross_kv,capture", 12),
    ("decoder side + enc_gemm (all 24)", "enc_gemm,dec,cross_kv,capture", 0),
    ("decoder side + enc all 24 (no logmel/conv)", "enc_gemm,enc_attn,dec,cross_kv,capture", 0),
    ("encoder side only (logmel,conv,enc)", "logmel,conv,enc_gemm,enc_attn", 0),
    ("all but enc_attn", "logmel,conv,enc_gemm,cross_kv,dec,capture", 0),
    ("full split", "all", 0),
]


def oracle_word_times(args, sd, dims, syn, audio_mod, ids):
    from oracle import timing_ref, whisper_ref, tokenizer_ref
    key = "oracle_%s_peaky008_s%d_c%d_k%d_m%d_ids%d-%d" % (args.model, int(args.seconds), args.chars, args.topk, args.medfilt_width, ids[0], ids[-1])
    gold = os.path.join(ROOT, "tests", "golden", "oracle_word_times_medium_peaky.npz")   # tests/golden/make_oracle_word_times.py
    standard = (args.model, int(args.seconds), args.chars, args.topk, args.medfilt_width) == ("medium", 10, 64, 10, 3)
    # (tools/cache/ is git-ignored but travels to the GPU box: a cache computed with --oracle-only in the build container goes there)
    for path in ([gold] if standard else []) + [os.path.join(ROOT, "tools", "cache", key + ".npz"), os.path.join(ROOT, "gpurun_out", key + ".npz")]:
        if os.path.exists(path):
            z = np.load(path, allow_pickle=False)
            if all("st_%d" % u in z.files for u in ids):
                print("oracle cache:", path, flush=True)
                return [(z["st_%d" % u], z["en_%d" % u], z["sc_%d" % u].astype(np.float64)) for u in ids], path
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 64))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", key + ".npz")
    part = path + ".part.npz"   # progress survives a time-out: re-running continues from it
    store = dict(np.load(part, allow_pickle=False)) if os.path.exists(part) else {}
    tok = tokenizer_ref.CharTokenizer()
    ref = whisper_ref.WhisperRef(sd, dims)
    filt = audio_mod.mel_filters(dims.n_mels)
    n_samples = int(args.seconds * 16000)
    out = []
    t00 = time.time()
    for i, u in enumerate(ids):
        if "st_%d" % u in store:
            out.append((store["st_%d" % u], store["en_%d" % u], store["sc_%d" % u]))
            continue
        pcm = torch.from_numpy(syn.synth_audio(u, n_samples))
        text = syn.synth_text(u, args.chars)
        mel = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(pcm), filt)
        tt = tokenizer_ref.encode_char(text, tok)
        tokens = torch.tensor([*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot])
        w, _ = timing_ref.get_attentions(mel, tokens, ref, n_samples // 320, args.medfilt_width, 1.0)
        _words, st, en, _matrix, _s = timing_ref.force_align(w, tt, tok, "char", "topk", args.topk)
        _sel, all_scores = timing_ref.filter_attention(w, w.shape[0] * w.shape[1], 1, 1, 0)
        sc = np.zeros(w.shape[0] * w.shape[1], dtype=np.float64)
        for s_, (l, h), _n in all_scores:
            sc[l * w.shape[1] + h] = s_
        out.append((np.asarray(st), np.asarray(en), sc))
        store["st_%d" % u], store["en_%d" % u], store["sc_%d" % u] = out[-1]
        if i % 10 == 0:
            print("oracle utterance %d/%d (%.0f s)" % (i + 1, len(ids), time.time() - t00), flush=True)
            np.savez_compressed(part, **store)
    np.savez_compressed(path, **store)
    if os.path.exists(part):
        os.remove(part)
    return out, path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=301)
    ap.add_argument("--first-id", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--model", default="medium")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--chars", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--medfilt_width", type=int, default=3)
    ap.add_argument("--rows", default="", help="comma-separated row names (default: every row of the table); 'sites:a+b@first' adds a row")
    ap.add_argument("--attn-drop", default="", help="comma-separated pass masks of the encoder's pair attention (wca_test_set_attn_split_drop: 1 K_lo Q_hi, 2 K_hi Q_lo, "
                    "4 V_lo P_hi, 8 V_hi P_lo; 0, 1, 2, 3, 4, 8, 12, 15 exist): every mask becomes a row in the all-sites mode (VERDICT r4 item 2)")
    ap.add_argument("--oracle-only", action="store_true", help="compute / complete the oracle cache and stop (CPU only: runs without a GPU)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_precision_ablation.txt"))
    args = ap.parse_args()

    wca = importlib.import_module("whisper-char-alignment_amd")
    m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
    syn, tok_mod, retok, timing, audio_mod = m("synthetic"), m("tokenizer"), m("retokenize"), m("timing"), m("audio")
    dims = wca.dims_for(args.model)
    sd = syn.random_state_dict(dims, seed=0, cross_qk_std=0.08)
    ids = list(range(args.first_id, args.first_id + args.utts))
    oracle, cache_path = oracle_word_times(args, sd, dims, syn, audio_mod, ids)
    if args.oracle_only:
        print("oracle word times:", cache_path)
        return

    device = torch.device("cuda", 0)
    model = wca.WhisperAMD(dims, device=str(device), max_batch=args.batch, precision="f16").load_state_dict(sd)
    tok = tok_mod.get_tokenizer(True, language="English")
    opts = model.make_opts(aggregation="topk", topk=args.topk, sot_len=len(tok.sot_sequence), medfilt_width=args.medfilt_width, qk_scale=1.0)
    n_samples = int(args.seconds * 16000)

    # the parity batches: the utterances in order, the last batch filled by repetition
    batches = []
    for lo in range(0, len(ids), args.batch):
        chunk = ids[lo:lo + args.batch]
        fill = (chunk * ((args.batch + len(chunk) - 1) // len(chunk)))[:args.batch]
        pcm = np.stack([syn.synth_audio(u, n_samples) for u in fill])
        tts = [retok.encode(syn.synth_text(u, args.chars), tok, "char") for u in fill]
        rows = [[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts]
        n_max = max(len(r) for r in rows)
        toks = np.full((len(rows), n_max), tok.eot, dtype=np.int64)
        for j, r in enumerate(rows):
            toks[j, :len(r)] = r
        batches.append(dict(n=len(chunk), lo=lo, pcm=torch.from_numpy(pcm).to(device), tokens=torch.from_numpy(toks).to(device),
                            n_tok=[len(r) for r in rows], tts=tts, n_max=n_max))

    rows = list(ROWS)
    if args.rows:
        want = [r.strip() for r in args.rows.split(",") if r.strip()]
        sel = []
        for w in want:
            if w.startswith("sites:"):
                spec, _, first = w[6:].partition("@")
                sel.append((w, spec.replace("+", ","), int(first or 0)))
            else:
                sel += [r for r in ROWS if r[0] == w]
        rows = sel

    drop_names = {0: "all six passes (contract)", 1: "without K_lo Q_hi", 2: "without K_hi Q_lo", 3: "S from K_hi Q_hi alone", 4: "without V_lo P_hi",
                  8: "without V_hi P_lo", 9: "without K_lo Q_hi and V_hi P_lo", 12: "O from V_hi P_hi alone", 15: "one pass per product (pair operands elsewhere)"}
    if args.attn_drop:
        rows = [("attention passes: %s [mask %d]" % (drop_names.get(int(x), "?"), int(x)), "all", -1 - int(x)) for x in args.attn_drop.split(",")]

    lines = []
    hdr = "%-46s %9s %9s | %7s %7s %7s | %5s %5s | %s" % ("sites (encoder bits from block)", "utt/s", "ms/step", "bounds", "within", "ident", "utts", "clean", "head-score deviation from the oracle (relative, all L x H heads) | offenders (utt:boundaries off, same heads?, oracle k/k+1 gap)")
    lines.append(hdr)
    print(hdr, flush=True)
    records = []
    for name, sites, first in rows:
        if first < 0:   # an attention-pass row: all sites on pairs, the mask on the encoder's pair attention
            wca._lib.check(model._lib.wca_test_set_attn_split_drop(-1 - first))
            first = 0
        model.set_precision_sites(sites, first)
        # ---- parity
        total = within = ident = clean = 0
        offenders = []
        sc_dev_max, sc_dev_sq, sc_n = 0.0, 0.0, 0
        LH = dims.n_text_layer * dims.n_text_head
        for b in batches:
            jump, sel = model.align_batch(b["pcm"], [n_samples] * args.batch, b["tokens"], b["n_tok"], [n_samples // 320] * args.batch, opts)
            sc_gpu = np.zeros((args.batch, LH), dtype=np.float32)
            wca._lib.check(model._lib.wca_test_last_scores(model._h, args.batch, sc_gpu.ctypes.data_as(wca._lib._pf)))
            for j in range(b["n"]):
                rst, ren, rsc = oracle[b["lo"] + j]
                rel = np.abs(sc_gpu[j].astype(np.float64) - rsc) / np.abs(rsc)   # head selection scores (timing.py:13-43) against the oracle's, all L x H heads
                sc_dev_max = max(sc_dev_max, float(rel.max()))
                sc_dev_sq += float((rel ** 2).sum())
                sc_n += rel.size
                _w, st, en = timing.words_from_jump_frames(jump[j], b["tts"][j], tok, "char")
                off = 0
                for a_, r_ in ((np.asarray(st), rst), (np.asarray(en), ren)):
                    assert len(a_) == len(r_)
                    total += len(a_)
                    within += int(np.sum(np.abs(a_ - r_) <= 0.02 + 1e-9))
                    ident += int(np.sum(a_ == r_))
                    off += int(np.sum(np.abs(a_ - r_) > 0.02 + 1e-9))
                if off == 0:
                    clean += 1
                else:
                    ranked = np.sort(rsc)
                    kth, nxt = ranked[-args.topk], ranked[-args.topk - 1]
                    o_heads = set(np.nonzero(rsc >= kth)[0].tolist())
                    g_heads = {int(x) for x in sel[j][:args.topk]}
                    offenders.append("%d:%d,%s,%.1e" % (ids[b["lo"] + j], off, "same" if o_heads == g_heads else "swap", (kth - nxt) / abs(kth)))
        # ---- throughput: bench.py's loop (two batches in flight, host tail included)
        def enqueue(i):
            b = batches[i % max(1, len(batches) - 1)]  # (full batches only)
            model.align_batch(b["pcm"], [n_samples] * args.batch, b["tokens"], b["n_tok"], [n_samples // 320] * args.batch, opts, enqueue_only=True)

        def finish(i):
            b = batches[i % max(1, len(batches) - 1)]
            jump, _sel = model.fetch(args.batch, b["n_max"], opts)
            for j in range(args.batch):
                timing.words_from_jump_frames(jump[j], b["tts"][j], tok, "char", want_words=False)

        for i in range(2):
            enqueue(i)
            finish(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            enqueue(i)
            if i > 0:
                finish(i - 1)
        finish(args.steps - 1)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rate = args.batch * args.steps / el
        label = "%s%s" % (name, (" [from %d]" % first) if first else "")
        sc_rms = (sc_dev_sq / max(sc_n, 1)) ** 0.5
        line = "%-46s %9.1f %9.2f | %7d %7d %7d | %5d %5d | score dev max %.1e rms %.1e | %s" % (label, rate, 1e3 * el / args.steps, total, within, ident, len(ids), clean, sc_dev_max, sc_rms, " ".join(offenders[:12]))
        lines.append(line)
        print(line, flush=True)
        records.append(dict(row=name, sites=sites, enc_first_layer=first, utt_per_s=rate, ms_per_step=1e3 * el / args.steps, boundaries=total, within_one_frame=within,
                            identical=ident, utterances=len(ids), utterances_clean=clean, offenders=offenders,
                            score_rel_dev_max=sc_dev_max, score_rel_dev_rms=sc_rms))
    if args.attn_drop:
        wca._lib.check(model._lib.wca_test_set_attn_split_drop(0))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("# tools/precision_ablation.py --utts %d --steps %d  (whisper-%s dims, peaky seeded weights, %.0f s audio, %d chars, topk %d, medfilt %d, fused B = %d;\n"
                "# utterance ids %d..%d against the fp32 CPU oracle, tolerance = one 20 ms frame; oracle cache %s; commit %s)\n"
                % (args.utts, args.steps, args.model, args.seconds, args.chars, args.topk, args.medfilt_width, args.batch, ids[0], ids[-1],
                   os.path.basename(cache_path), os.environ.get("WCA_COMMIT", "?")))
        f.write("\n".join(lines) + "\n")
    with open(os.path.splitext(args.out)[0] + ".json", "w") as f:
        json.dump(records, f, indent=1)


if __name__ == "__main__":
    main()
