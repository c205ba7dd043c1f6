"""Does the f16 MFMA (v_mfma_f32_16x16x32_f16) keep f16 SUBNORMAL inputs? (split mode stores lo = f16(x - hi), which is subnormal
for |x| below ~0.1.) GEMM through the C ABI with subnormal activations (B operand) / subnormal weights (A operand), and the
split attention with subnormal V / subnormal expected outputs."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
lib = wca._lib.load()
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, device="cuda:0", max_batch=1, precision="f16")
eng._bind_stream()
M = N = 128
K = 64
tiny = 2.0 ** -16   # f16 subnormal (min normal 2^-14)
for tile in (128, 256, 257):
    for which in ("activation", "weight"):
        a = torch.full((M, K), tiny if which == "activation" else 1.0).half().cuda()
        w = torch.full((N, K), tiny if which == "weight" else 1.0).half().cuda()
        out = torch.zeros(M, N, device="cuda")
        wca._lib.check(lib.wca_test_gemm(eng._h, vp(a), vp(w), None, vp(out), M, N, K, 0, 1 | (tile << 8)))
        torch.cuda.synchronize()
        print("gemm tile %d subnormal %s: got %.6e expected %.6e" % (tile, which, out[0, 0].item(), K * tiny))
    # pair output of a value whose lo half is subnormal
    a = torch.full((M, K), 1.0).half().cuda()
    val = 0.1 + 2.0 ** -16
    w = torch.zeros(N, K).half()
    w[:, 0] = 1.0
    bias = torch.full((N,), val - 1.0).cuda()
    out2 = torch.zeros(M, 2 * N, dtype=torch.float16, device="cuda")
    wca._lib.check(lib.wca_test_gemm(eng._h, vp(a), vp(w.cuda()), vp(bias), vp(out2), M, N, K, 0, 4 | (tile << 8)))
    torch.cuda.synchronize()
    hi, lo = out2[0, 0].item(), out2[0, N].item()
    print("gemm tile %d pair store of %.9f: hi %.9f lo %.3e (hi + lo - v = %.3e)" % (tile, val, hi, lo, hi + lo - (val - 1.0 + 1.0)))
# attention: one key, so o = v exactly
B, H, nq, nk = 1, 1, 16, 1
d = 64
for vval in (0.1 + 2.0 ** -16, 2.0 ** -16):
    v = torch.full((B, nk, d), vval)
    vhi = v.half()
    vlo = (v - vhi.float()).half()
    q2 = torch.zeros(B, nq, 2 * d).half().cuda()
    k2 = torch.zeros(B, nk, 2 * d).half().cuda()
    v2 = torch.cat([vhi, vlo], -1).cuda()
    o2 = torch.zeros(B, nq, 2 * d, dtype=torch.float16, device="cuda")
    wca._lib.check(lib.wca_test_attention_split(eng._h, vp(q2), vp(k2), vp(v2), vp(o2), None, 0, 0, B, H, nq, nk, 0))
    torch.cuda.synchronize()
    print("attention split v = %.9e (hi %.6e lo %.3e): o hi %.6e lo %.3e sum %.9e" % (vval, vhi[0, 0, 0].item(), vlo[0, 0, 0].item(), o2[0, 0, 0].item(),
                                                                                    o2[0, 0, d].item(), o2[0, 0, 0].double().item() + o2[0, 0, d].double().item()))
