#!/usr/bin/env python3
"""Reference-precision kernels in isolation (development aid): the encoder's GEMM shapes as the K-doubled call
[A_hi | A_lo] [W | W]^T and in the SPLITW form (plain W, every W K-tile staged once), against the f16 call; the three-pass
attention against the f16 one. HIP events on the engine's stream, interleaved repeats.
  python tools/split_bench.py [B=64] [reps=5]"""
import ctypes as C
import importlib
import os
import sys

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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
    eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
    eng._bind_stream()
    lib, chk = eng._lib, wca._lib.check
    M = B * 1500
    shapes = [("qkv", 3072, 1024, 0, 4), ("out", 1024, 1024, 0, 2), ("fc1", 4096, 1024, 1, 4), ("fc2", 1024, 4096, 0, 2)]
    for name, n, k, gelu, mode in shapes:
        if only and only not in ("gemm", name):
            continue
        a = torch.randn(M, k, device="cuda") * 0.5
        hi = a.half()
        a2 = torch.cat([hi, (a - hi.float()).half()], dim=1).contiguous()
        w = (torch.randn(n, k, device="cuda") * 0.05).half()
        w2 = torch.cat([w, w], dim=1).contiguous()
        bias = torch.randn(n, device="cuda")
        out = torch.zeros(M, 2 * n, device="cuda", dtype=torch.float16) if mode == 4 else torch.zeros(M, n, device="cuda", dtype=torch.float32)
        mode16 = 0 if mode == 4 else mode
        t = {"f16": [], "k-doubled": [], "splitw": [], "splitw two-slot rings (r4)": []}
        for _ in range(reps):
            t["f16"].append(timeit(lambda: chk(lib.wca_test_gemm(eng._h, vp(hi), vp(w), vp(bias), vp(out), M, n, k, gelu, mode16))))
            t["k-doubled"].append(timeit(lambda: chk(lib.wca_test_gemm(eng._h, vp(a2), vp(w2), vp(bias), vp(out), M, n, 2 * k, gelu, mode))))
            t["splitw"].append(timeit(lambda: chk(lib.wca_test_gemm_pairs(eng._h, vp(a2), vp(w), vp(bias), vp(out), M, n, k, gelu, mode))))
            chk(lib.wca_test_set_switch(b"gemm_ring", 1))
            t["splitw two-slot rings (r4)"].append(timeit(lambda: chk(lib.wca_test_gemm_pairs(eng._h, vp(a2), vp(w), vp(bias), vp(out), M, n, k, gelu, mode))))
            chk(lib.wca_test_set_switch(b"gemm_ring", 0))
        fl = 2.0 * M * n * k
        print("gemm %-4s M=%d N=%d K=%d  " % (name, M, n, k) + "  ".join("%s %.3f ms (%.0f TF alg)" % (kk, min(v), fl / min(v) / 1e9) for kk, v in t.items()), flush=True)
        del a, hi, a2, w, w2, out
    if not only or only == "ln":
        # pair LayerNorm (f32 rows -> [hi | lo] f16 rows): HBM-bound, 8 bytes per element
        x = torch.randn(M, 1024, device="cuda")
        g_, b_ = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
        y2 = torch.empty(M, 2048, device="cuda", dtype=torch.float16)
        y1 = torch.empty(M, 1024, device="cuda", dtype=torch.float16)
        t2 = min(timeit(lambda: chk(lib.wca_test_layernorm_split(eng._h, vp(x), vp(g_), vp(b_), vp(y2), M, 1024))) for _ in range(reps))
        t1 = min(timeit(lambda: chk(lib.wca_test_layernorm(eng._h, vp(x), vp(g_), vp(b_), vp(y1), M, 1024))) for _ in range(reps))
        print("layernorm rows=%d d=1024  single %.3f ms (%.2f TB/s)  pair %.3f ms (%.2f TB/s)" % (M, t1, M * 1024 * 6 / t1 / 1e9, t2, M * 1024 * 8 / t2 / 1e9), flush=True)
        del x, y1, y2
    if only and only not in ("attn",):
        return
    H, S = 16, 1500
    q = torch.randn(B, S, H * 64, device="cuda")
    k_ = torch.randn(B, S, H * 64, device="cuda")
    v = torch.randn(B, S, H * 64, device="cuda")

    def pair(x):
        h = x.half()
        return torch.cat([h, (x - h.float()).half()], dim=-1).contiguous()

    q2, k2, v2 = pair(q), pair(k_), pair(v)
    qh, kh, vh = q.half(), k_.half(), v.half()
    o, o2 = torch.empty_like(qh), torch.empty_like(q2)
    t16, t3, t3o = [], [], []
    for _ in range(reps):   # interleaved: f16, pair 32x32x16 (the launcher's choice), pair 16x16x32 (switch attn_split_variant = 1)
        t16.append(timeit(lambda: chk(lib.wca_test_attention(eng._h, vp(qh), vp(kh), vp(vh), vp(o), None, 0, 0, B, H, S, S, 0))))
        chk(lib.wca_test_set_switch(b"attn_split_variant", 0))
        t3.append(timeit(lambda: chk(lib.wca_test_attention_split(eng._h, vp(q2), vp(k2), vp(v2), vp(o2), None, 0, 0, B, H, S, S, 0))))
        chk(lib.wca_test_set_switch(b"attn_split_variant", 1))
        t3o.append(timeit(lambda: chk(lib.wca_test_attention_split(eng._h, vp(q2), vp(k2), vp(v2), vp(o2), None, 0, 0, B, H, S, S, 0))))
        chk(lib.wca_test_set_switch(b"attn_split_variant", 0))
    fl = 4.0 * B * H * S * S * 64
    print("attn enc B=%d H=%d S=%d  f16 %.3f ms (%.0f TF)  pair 32x32x16 %.3f ms (%.0f TF alg)  pair 16x16x32 %.3f ms (%.0f TF alg)"
          % (B, H, S, min(t16), fl / min(t16) / 1e9, min(t3), fl / min(t3) / 1e9, min(t3o), fl / min(t3o) / 1e9), flush=True)

if __name__ == "__main__":
    main()
