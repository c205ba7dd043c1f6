#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels on the MI355X (development aid, not the contract bench):
times the GEMM shapes of the whisper-medium forward with both tile shapes, the encoder attention and the
LayerNorm, with HIP events on torch's current stream (the stream the engine launches on)."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
syn = importlib.import_module("whisper-char-alignment_amd.synthetic")


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, iters=20, warm=3):
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
    eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
    eng._bind_stream()
    lib = eng._lib
    M = B * 1500
    shapes = [("qkv", M, 3072, 1024, 0, 0), ("out", M, 1024, 1024, 0, 2), ("fc1", M, 4096, 1024, 1, 0), ("fc2", M, 1024, 4096, 0, 2),
              ("crosskv", M, 49152, 1024, 0, 0), ("conv2", M, 1024, 3072, 1, 1), ("dec_fc1", B * 69, 4096, 1024, 1, 0),
              ("dec_qkv", B * 69, 3072, 1024, 0, 0), ("dec_out", B * 69, 1024, 1024, 0, 2), ("dec_fc2", B * 69, 1024, 4096, 0, 2)]
    if os.environ.get("WCA_KB_ONLY"):
        shapes = [sh for sh in shapes if sh[0].startswith(os.environ["WCA_KB_ONLY"])]
    for name, m, n, k, gelu, mode in shapes:
        a = (torch.randn(m, k, device="cuda") * 0.5).half()
        w = (torch.randn(n, k, device="cuda") * 0.05).half()
        bias = torch.randn(n, device="cuda")
        out = torch.zeros(m, n, device="cuda", dtype=torch.float16 if mode == 0 else torch.float32)
        for tile in (128, 256, 258, 257):
            if tile >= 256 and m < 2048:
                continue
            ms = timeit(lambda: wca._lib.check(lib.wca_test_gemm(eng._h, vp(a), vp(w), vp(bias), vp(out), m, n, k, gelu, mode | (tile << 8))))
            print("gemm %-8s M=%6d N=%6d K=%5d tile=%3d  %8.3f ms  %7.1f TFLOP/s" % (name, m, n, k, tile, ms, 2.0 * m * n * k / ms / 1e9), flush=True)
        del a, w, out
    H, S = 16, 1500
    q = torch.randn(B, S, H * 64, device="cuda").half()
    k_ = torch.randn(B, S, H * 64, device="cuda").half()
    v = torch.randn(B, S, H * 64, device="cuda").half()
    o = torch.empty_like(q)
    ms = timeit(lambda: wca._lib.check(lib.wca_test_attention(eng._h, vp(q), vp(k_), vp(v), vp(o), None, 0, 0, B, H, S, S, 0)))
    print("attn enc  B=%d H=%d S=%d  %8.3f ms  %7.1f TFLOP/s" % (B, H, S, ms, 4.0 * B * H * S * S * 64 / ms / 1e9), flush=True)
    x = torch.randn(M, 1024, device="cuda")
    g, b_ = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
    y = torch.empty(M, 1024, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: wca._lib.check(lib.wca_test_layernorm(eng._h, vp(x), vp(g), vp(b_), vp(y), M, 1024)))
    print("layernorm rows=%d d=1024  %8.3f ms  %6.2f TB/s" % (M, ms, M * 1024 * 6 / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
