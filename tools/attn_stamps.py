#!/usr/bin/env python3
"""Per-segment cycle breakdown of the attention key-tile loop from s_memtime stamps (diagnostic build)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
eng._bind_stream()
B, H, S = 32, 16, 1500
q = torch.randn(B, S, H * 64, device="cuda").half()
k = torch.randn(B, S, H * 64, device="cuda").half()
v = torch.randn(B, S, H * 64, device="cuda").half()
o = torch.empty_like(q)
for name, var in (("16x16x32 kernel", 1), ("32x32x16 kernel", 2)):
    dbg = torch.zeros(4 * 4 * 32 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        wca._lib.check(eng._lib.wca_test_attention_stamped(eng._h, vp(q), vp(k), vp(v), vp(o), B, H, (S if var == 0 else -((var << 20) | S)), S, vp(dbg)))
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(4, 4, 32, 8)
    print(name)
    for blk in (0, 2):
        for wave in (0, 3):
            s = d[blk, wave, 2:22, :6].astype(np.int64)
            seg = np.diff(s, axis=1)
            print("  blk %d wave %d: vmcnt %5.0f | barrier %5.0f | dma-issue+QK %5.0f | softmax %5.0f | PV %5.0f | tile %5.0f" %
                  (blk, wave, seg[:, 0].mean(), seg[:, 1].mean(), seg[:, 2].mean(), seg[:, 3].mean(), seg[:, 4].mean(), np.diff(s[:, 0]).mean()))
