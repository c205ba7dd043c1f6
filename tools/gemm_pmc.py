#!/usr/bin/env python3
"""Launches the contract mode's pair GEMMs of one encoder block (product kernels, M = 64 x 1500) a few times: the workload of the
rocprofv3 --pmc passes in tools/session_r05_measure.sh.   python tools/gemm_pmc.py [launches=6]"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
eng._bind_stream()
M = 96000
# distinct kernel symbols: fc1 = <4, true, 1>, qkv = <4, false, 1>, fc2 = <2, false, 1> (the engine's fc2 is site 4: same code)
for name, n, k, gelu, mode in [("fc1", 4096, 1024, 1, 4), ("qkv", 3072, 1024, 0, 4), ("fc2", 1024, 4096, 0, 2)]:
    a = torch.randn(M, k, device="cuda") * 0.5
    hi = a.half()
    a2 = torch.cat([hi, (a - hi.float()).half()], dim=1).contiguous()
    w = (torch.randn(n, k, device="cuda") * 0.05).half()
    bias = torch.randn(n, device="cuda")
    out = torch.zeros(M, 2 * n, device="cuda", dtype=torch.float16) if mode == 4 else torch.zeros(M, n, device="cuda", dtype=torch.float32)
    for _ in range(n_launch):
        wca._lib.check(eng._lib.wca_test_gemm_pairs(eng._h, vp(a2), vp(w), vp(bias), vp(out), M, n, k, gelu, mode))
    torch.cuda.synchronize()
    del a, hi, a2, w, out
