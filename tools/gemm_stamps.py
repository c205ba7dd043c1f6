#!/usr/bin/env python3
"""Where a K step of the persistent 256x256 GEMM spends its cycles: s_memtime stamps of the CURRENT kernel
(diagnostic instantiation; read the SHARES, the stamps add fences) on the encoder's shapes, in the contract mode's SPLITW form and
the f16 form, twice each: the real tile walk, and with every tile coordinate taken modulo (2, 2) so that the operand footprint
(2 activation panels + 2 weight panels, 1.5-3 MB) is L2-resident for every XCD. Then plain timings (no stamps) of the same two walks.
  python tools/gemm_stamps.py [B=64] [t = timings only | s] [pair:fc1 | f16:qkv | ... = one shape]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
eng._bind_stream()
lib, chk = eng._lib, wca._lib.check
M = B * 1500
# name, N, K, out_mode bits (mode | gelu << 8), pair operands
SHAPES = [("fc1", 4096, 1024, 4 | 256, True), ("qkv", 3072, 1024, 4, True), ("fc2", 1024, 4096, 2, True), ("out", 1024, 1024, 2, True),
          ("fc1", 4096, 1024, 0 | 256, False), ("qkv", 3072, 1024, 0, False), ("fc2", 1024, 4096, 2, False)]
# (the stamp buffer holds 48 K steps per wave: the K = 4096 shapes' step stamps overlap the tile stamps -- read their timings, not their stamp rows)
only = sys.argv[3] if len(sys.argv) > 3 else ""
for name, n, k, mode, pairs in SHAPES:
    if only and (name != only.split(":")[-1] or (only.startswith("pair:") and not pairs) or (only.startswith("f16:") and pairs)):
        continue
    a = torch.randn(M, k, device="cuda") * 0.5
    hi = a.half()
    A = torch.cat([hi, (a - hi.float()).half()], dim=1).contiguous() if pairs else hi
    w = (torch.randn(n, k, device="cuda") * 0.05).half()
    om = mode & 0xff
    out = torch.zeros(M, 2 * n if om == 4 else n, device="cuda", dtype=torch.float16 if om in (0, 4) else torch.float32)
    flags = mode | (512 if pairs else 0)
    tag = "%s %s (M=%d N=%d K=%d)" % ("pair" if pairs else "f16 ", name, M, n, k)
    nk = (2 if pairs else 1) * k // 64
    for wrap in ((15, 2) if (len(sys.argv) <= 2 or sys.argv[2] != "t") else ()):   # 15 = no wrap (same diagnostic instantiation); second argument "t": timings only
        dbg = torch.zeros(4 * 8 * 64 * 8, dtype=torch.int64, device="cuda")
        fl = flags | (wrap << 12) | (wrap << 16)
        for _ in range(3):
            chk(lib.wca_test_gemm_stamped(eng._h, vp(A), vp(w), vp(out), M, n, k, fl, vp(dbg)))
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(4, 8, 64, 8)
        rows = []
        for blk in range(4):
            for wave in range(8):
                s = d[blk, wave, :nk, :5].astype(np.int64)
                if (s[:, 0] == 0).any():
                    continue
                seg = np.diff(s, axis=1)[2:-2]
                back = (s[1:, 0] - s[:-1, 4])[2:-2]
                step = np.diff(s[:, 0])[2:-2]
                ph = d[blk, wave, 48:60, :2].astype(np.int64)
                ph = ph[ph[:, 0] > 0]
                rows.append([seg[:, 0].mean(), seg[:, 1].mean(), seg[:, 2].mean(), seg[:, 3].mean(), back.mean(), step.mean(),
                             (ph[:, 1] - ph[:, 0]).mean(), np.diff(ph[:, 0]).mean() if len(ph) > 1 else 0.0])
        r = np.array(rows)
        print("%s %s: per K step (mean over %d waves): half0 %5.0f | wait %5.0f | barrier %5.0f | half1 %5.0f | loop-back %4.0f | step %5.0f (MFMA floor 2048)"
              " || tile: epilogue %6.0f, period %7.0f cycles (K loop %d steps)" %
              (tag, "wrapped 2x2" if wrap == 2 else "real walk  ", len(rows), *r.mean(axis=0)[:6], r[:, 6].mean(), r[:, 7].mean(), nk), flush=True)
        print("      per-wave step means: " + " ".join("%.0f" % x for x in r[:, 5]), flush=True)
        if len(rows) >= 8:   # workgroup 0, wave by wave (waves w and w + 4 share a SIMD): half0 / wait / barrier / half1
            print("      workgroup 0 by wave [half0 wait barrier half1]: " + "  ".join("w%d %.0f %.0f %.0f %.0f" % (i, *r[i, :4]) for i in range(8)), flush=True)
    # plain timings (no stamps): product kernel, diagnostic instantiation without wrap, wrapped 2x2, wrapped 1x1
    t = {}
    bias = None
    if pairs:
        t["product"] = min(timeit(lambda: chk(lib.wca_test_gemm_pairs(eng._h, vp(A), vp(w), vp(bias), vp(out), M, n, k, (mode >> 8) & 1, om))) for _ in range(3))
    else:
        t["product"] = min(timeit(lambda: chk(lib.wca_test_gemm(eng._h, vp(A), vp(w), vp(bias), vp(out), M, n, k, (mode >> 8) & 1, om))) for _ in range(3))
    for label, wr, kind in (("diag no-wrap", 15, 0), ("wrapped 2x2", 2, 0), ("operands wrapped 2x2, real outputs", 2, 1), ("real operands, outputs wrapped 2x2", 2, 2), ("wrapped 1x1", 1, 0),
                            ("real walk, packed DMA sources", 15, 4), ("wrapped 2x2, packed DMA sources", 2, 4)):
        fl = flags | (wr << 12) | (wr << 16) | (kind << 20)
        t[label] = min(timeit(lambda: chk(lib.wca_test_gemm_stamped(eng._h, vp(A), vp(w), vp(out), M, n, k, fl, None))) for _ in range(3))
    print("%s timing: " % tag + "  ".join("%s %.3f ms" % kv for kv in t.items()), flush=True)
    del a, hi, A, w, out
