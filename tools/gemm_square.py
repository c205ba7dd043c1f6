#!/usr/bin/env python3
"""The persistent 256x256 GEMM on square problems (4096^3, 8192^3, uniform random [-1, 1) operands), for comparison with the
figures the CDNA4 guide quotes for its 8-phase template on the same shapes and data."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, max_batch=1, precision="f16")
eng._bind_stream()
for n in (4096, 8192):
    a = (torch.rand(n, n, device="cuda") * 2 - 1).half()
    w = (torch.rand(n, n, device="cuda") * 2 - 1).half()
    out = torch.empty(n, n, device="cuda", dtype=torch.float16)
    for tile, sm in ((257, 0), (257, 4), (257, 8), (257, 16), (258, 8)):
        f = lambda: wca._lib.check(eng._lib.wca_test_gemm(eng._h, vp(a), vp(w), None, vp(out), n, n, n, 0, 0 | (tile << 8) | (sm << 20)))
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
        print("n=%d tile=%d supertile=%d: %.3f ms = %.0f TFLOP/s" % (n, tile, sm, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
