#!/usr/bin/env python3
"""Epilogue cost of the persistent GEMM at the encoder's shapes (M = 96000): fc1 with / without GELU, f16 / f32 / RMW stores."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
eng._bind_stream()
M = 96000
for name, N, K, gelu, mode in (("fc1+gelu", 4096, 1024, 1, 0), ("fc1 no gelu", 4096, 1024, 0, 0), ("qkv", 3072, 1024, 0, 0), ("fc2 rmw", 1024, 4096, 0, 2),
                               ("fc2 f16 out", 1024, 4096, 0, 0), ("out rmw", 1024, 1024, 0, 2), ("out f16", 1024, 1024, 0, 0)):
    a = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float16 if mode == 0 else torch.float32)
    f = lambda: wca._lib.check(eng._lib.wca_test_gemm(eng._h, vp(a), vp(w), vp(bias), vp(out), M, N, K, gelu, mode))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-12s N=%d K=%d: %.3f ms = %.0f TFLOP/s" % (name, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del a, w, out
