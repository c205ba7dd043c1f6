#!/usr/bin/env python3
"""Launches a few whisper-medium GEMM shapes with the 256x256 kernel (for rocprofv3 --pmc runs)."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
eng._bind_stream()
M = 48000
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name, n, k, gelu, mode in [("qkv", 3072, 1024, 0, 0), ("fc1", 4096, 1024, 1, 0), ("fc2", 1024, 4096, 0, 2)]:
    a = (torch.randn(M, k, device="cuda") * 0.5).half()
    w = (torch.randn(n, k, device="cuda") * 0.05).half()
    bias = torch.randn(n, device="cuda")
    out = torch.zeros(M, n, device="cuda", dtype=torch.float16 if mode == 0 else torch.float32)
    for _ in range(5):
        wca._lib.check(eng._lib.wca_test_gemm(eng._h, vp(a), vp(w), vp(bias), vp(out), M, n, k, gelu, mode | (tile << 8)))
    torch.cuda.synchronize()
