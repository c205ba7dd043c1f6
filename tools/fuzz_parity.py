#!/usr/bin/env python3
"""Randomised parity sweep of the integer / order-statistic kernels against the CPU oracle (one-off confidence run on the
MI355X; the fixed cases live in tests/). Bit-exact expectations: DTW paths (incl. tie-heavy matrices), median filter,
head selection order; tolerance on the scores / matrices.   usage: fuzz_parity.py [n_cases] [seed]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
tm = importlib.import_module("whisper-char-alignment_amd.timing")
tk = importlib.import_module("whisper-char-alignment_amd.tokenizer")
from oracle import timing_ref  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
tok = tk.get_tokenizer(True, language="English")
bad = 0

# ---- DTW: random / integer / constant / rank-1 matrices, N <= 448, M <= 1500
for i in range(n_cases):
    N = int(rng.integers(1, 449)) if i % 5 else int(rng.integers(1, 12))
    M = int(rng.integers(1, 1501)) if i % 7 else int(rng.integers(1, 12))
    kind = i % 4
    if kind == 0:
        x = rng.standard_normal((N, M))
    elif kind == 1:
        x = rng.integers(0, 3, (N, M)).astype(np.float64)
    elif kind == 2:
        x = np.full((N, M), float(rng.integers(-2, 3)))
    else:
        x = np.round(rng.standard_normal((N, 1)) @ rng.standard_normal((1, M)), 1)
    xt = torch.from_numpy(x.astype(np.float32))
    ti, tj = timing_ref.dtw(-xt)
    gi, gj = tm.dtw(-xt.cuda())
    if not (np.array_equal(ti, gi) and np.array_equal(tj, gj)):
        bad += 1
        print("DTW mismatch", N, M, kind)
print("dtw: %d cases, %d mismatches" % (n_cases, bad), flush=True)

# ---- median filter: random shapes / widths incl. F <= w // 2
mbad = 0
for i in range(n_cases):
    F = int(rng.integers(1, 1501)) if i % 6 else int(rng.integers(1, 8))
    w = int(rng.choice([1, 3, 5, 7, 9, 15, 33]))
    rows = int(rng.integers(1, 40))
    a = torch.from_numpy(rng.standard_normal((1, 1, rows, F)).astype(np.float32))
    if i % 3 == 0:
        a = torch.round(a * 2) / 2  # many equal values
    want = timing_ref.median_filter(a, w)
    got = tm.median_filter(a.cuda(), w).cpu()
    if not torch.equal(want, got):
        mbad += 1
        print("median mismatch", rows, F, w)
print("median_filter: %d cases, %d mismatches" % (n_cases, mbad), flush=True)

# ---- head scores + tuple-ordered top-k + aggregation + DTW through force_align
fbad = 0
nf = max(n_cases // 5, 10)
for i in range(nf):
    L, H = int(rng.integers(1, 5)), int(rng.integers(1, 7))
    n = int(rng.integers(6, 60))
    F = int(rng.integers(8, 400))
    w = torch.softmax(torch.from_numpy(rng.standard_normal((L, H, n, F)).astype(np.float32)) * float(rng.uniform(0.5, 6)), -1)
    if i % 4 == 0 and H > 1:
        w[:, 1] = w[:, 0]  # exact score ties between heads: the tuple order decides
    k = int(rng.integers(1, L * H + 1))
    wc, wr, wv = [(1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1)][i % 4]
    sel, scores = tm.filter_attention(w.cuda(), k, wc, wr, wv)
    rsel, rscores = timing_ref.filter_attention(w, k, wc, wr, wv)
    if [s[1] for s in scores] != [s[1] for s in rscores]:
        fbad += 1
        print("top-k order mismatch", L, H, n, F, k)
    tt = [64] * (n - len(tok.sot_sequence) - 2)
    out = tm.force_align(w.cuda(), tt, tok, "char", "topk" if i % 2 else "mean", topk=k, w_colnorm=wc, w_rownorm=wr, w_coverage=wv)
    ref = timing_ref.force_align(w, tt, tok, "char", "topk" if i % 2 else "mean", k, wc, wr, wv)
    if out[0] != ref[0] or (len(out[1]) and (np.max(np.abs(np.asarray(out[1]) - np.asarray(ref[1]))) > 0.02 + 1e-9)):
        fbad += 1
        print("force_align mismatch", L, H, n, F, k)
print("filter_attention + force_align: %d cases, %d mismatches" % (nf, fbad), flush=True)
print("FUZZ TOTAL mismatches:", bad + mbad + fbad)

# ---- greedy-decode filter / update kernel along random oracle trajectories (timestamp pairing states reached naturally)
import ctypes as C  # noqa: E402
dref = importlib.import_module("oracle.decoding_ref")
decoding = importlib.import_module("whisper-char-alignment_amd.decoding")
_lib = importlib.import_module("whisper-char-alignment_amd._lib")
tok2 = tk.get_tokenizer(True, language="en", task="transcribe")
sup, blank = decoding.filter_masks(tok2, decoding.DecodingOptions(language="en"), 51865)
V, Bd, T_max = 51865, 4, 40
initial = list(tok2.sot_sequence)
filters = dref.make_filters(len(initial), tok2.eot, tok2.timestamp_begin, tok2.no_timestamps,
                            [i for i in np.nonzero(sup)[0] if i != tok2.no_timestamps], tok2.encode(" "), True, 50)
supd, blankd = torch.from_numpy(sup).cuda(), torch.from_numpy(blank).cuda()
o = _lib.DecodeOpts(224, tok2.eot, tok2.timestamp_begin, 1, 50, -1)
eng._bind_stream()
dbad = steps = 0
g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
for traj in range(max(n_cases // 10, 5)):
    tokens = torch.tensor([initial] * Bd, dtype=torch.long)
    osum = torch.zeros(Bd)
    td = torch.full((Bd, T_max), tok2.eot, dtype=torch.int32)
    td[:, :len(initial)] = torch.tensor(initial, dtype=torch.int32)
    td = td.cuda()
    lpd = torch.zeros(Bd, device="cuda")
    nd = torch.zeros(T_max, dtype=torch.int32, device="cuda")
    for step in range(14):
        logits = torch.randn(Bd, V, generator=g) * 3
        mode = int(rng.integers(4))
        if mode == 0:
            logits[:, tok2.timestamp_begin:] += 6.0   # push towards timestamps
        elif mode == 1:
            logits[:, :tok2.eot] += 3.0               # push towards text
        elif mode == 2:
            logits[int(rng.integers(Bd)), tok2.eot] += 30.0  # end one row
        cur_len = tokens.shape[1]
        tokens, _done, _f = dref.select_step(logits, tokens, osum, filters, tok2.eot)
        ld = logits.cuda()
        _lib.check(eng._lib.wca_test_decode_select(eng._h, C.c_void_p(ld.data_ptr()), Bd, V, C.c_void_p(td.data_ptr()), T_max, cur_len,
                                                   len(initial), C.c_void_p(supd.data_ptr()), C.c_void_p(blankd.data_ptr()), C.byref(o),
                                                   C.c_void_p(lpd.data_ptr()), C.c_void_p(nd.data_ptr())))
        torch.cuda.synchronize()
        got = td[:, cur_len].cpu().long()
        steps += 1
        if not torch.equal(got, tokens[:, -1]):
            dbad += 1
            print("decode_select mismatch traj %d step %d" % (traj, step), got.tolist(), tokens[:, -1].tolist())
            td[:, cur_len] = tokens[:, -1].int().cuda()  # continue along the oracle's trajectory
    if not torch.allclose(lpd.cpu(), osum, rtol=1e-4, atol=1e-3):
        dbad += 1
        print("sum_logprob mismatch", lpd.cpu().tolist(), osum.tolist())
print("decode_select: %d steps on %d trajectories, %d mismatches" % (steps, max(n_cases // 10, 5), dbad), flush=True)
print("FUZZ TOTAL mismatches (all sections):", bad + mbad + fbad + dbad)
