#!/usr/bin/env python3
"""A/B of the two encoder self-attention kernels (16x16x32 vs 32x32x16 MFMA) at the bench's shape, interleaved rounds in
ONE process on random (gaussian) data, HIP events on the launch stream; prints median / best TFLOP/s per variant."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2]
H, S = 16, 1500
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
eng._bind_stream()
q = torch.randn(B, S, H * 64, device="cuda").half()
k = torch.randn(B, S, H * 64, device="cuda").half()
v = torch.randn(B, S, H * 64, device="cuda").half()
o = torch.empty_like(q)
flop = 4.0 * B * H * S * S * 64
res = {x: [] for x in variants}


def run(variant, iters):
    for _ in range(iters):
        wca._lib.check(eng._lib.wca_test_attention(eng._h, vp(q), vp(k), vp(v), vp(o), None, 0, 0, B, H, S, S, variant << 8))


for x in variants:
    run(x, 3)
torch.cuda.synchronize()
for r in range(rounds):
    for x in variants:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(x, 5)
        b.record()
        torch.cuda.synchronize()
        res[x].append(a.elapsed_time(b) / 5)
for x in variants:
    ms = np.array(res[x])
    print("variant %d (%s): median %.3f ms = %.0f TFLOP/s, best %.3f ms = %.0f TFLOP/s" % (
        x, {0: "auto", 1: "16x16x32", 2: "32x32x16", 3: "32x32x16, VALU row sums"}[x], np.median(ms), flop / np.median(ms) / 1e9, ms.min(), flop / ms.min() / 1e9), flush=True)
