#!/usr/bin/env python3
"""Per-segment cycle breakdown of the pipelined 256x256 GEMM main loop from s_memtime stamps
(diagnostic build; read the SHARES, not the absolute length -- the stamps add fences)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")


def vp(t):
    return C.c_void_p(t.data_ptr())


dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1)
eng._bind_stream()
M = 48000
for name, n, k, mode in [("qkv", 3072, 1024, 0), ("fc2", 1024, 4096, 2)]:
    a = (torch.randn(M, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * 0.05).half()
    out = torch.zeros(M, n, device="cuda", dtype=torch.float16 if mode == 0 else torch.float32)
    dbg = torch.zeros(4 * 8 * 64 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        wca._lib.check(eng._lib.wca_test_gemm_stamped(eng._h, vp(a), vp(w), vp(out), M, n, k, mode, vp(dbg)))
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(4, 8, 64, 8)
    nk = k // 64
    for blk in (0, 3):
        for wave in (0, 5):
            ph = d[blk, wave, 48:60, :2].astype(np.int64)
            ph = ph[ph[:, 0] > 0]
            print("%s blk %d wave %d: tiles %d | epilogue issue cycles %s | tile period (epilogue start to next) %s" %
                  (name, blk, wave, len(ph), (ph[:, 1] - ph[:, 0]).tolist(), np.diff(ph[:, 0]).tolist()))
            s = d[blk, wave, :nk, :5].astype(np.int64)
            seg = np.diff(s, axis=1)  # [half0 mfma+reads, wait vmcnt/lgkm, barrier, half1 (+dma, reads)]
            nxt = s[1:, 0] - s[:-1, 4]
            print("%s blk %d wave %d: per-tile cycles  half0 %5.0f | wait %5.0f | barrier %5.0f | half1 %5.0f | loop-back %4.0f | tile total %5.0f" %
                  (name, blk, wave, seg[1:-1, 0].mean(), seg[1:-1, 1].mean(), seg[1:-1, 2].mean(), seg[1:-1, 3].mean(), nxt[1:].mean(),
                   np.diff(s[:, 0])[1:-1].mean()))
