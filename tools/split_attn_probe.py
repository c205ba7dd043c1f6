"""Where does attn_split_kernel differ from a float64 attention? (development probe)"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
wca = importlib.import_module("whisper-char-alignment_amd")
lib = wca._lib.load()
import test_split_gpu as T  # noqa: E402

vp = T._vp
dims = wca.ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1)
eng = wca.WhisperAMD(dims, device="cuda:0", max_batch=1, precision="f16")
eng._bind_stream()
for (B, H, nq, nk, causal, spike) in [(2, 3, 150, 200, 0, True), (2, 3, 150, 200, 0, False), (1, 1, 16, 64, 0, False), (1, 1, 16, 2, 0, False), (1, 1, 16, 17, 0, False)]:
    g = torch.Generator().manual_seed(nq * 13 + nk)
    d = H * 64
    q = torch.randn(B, nq, d, generator=g)
    k = torch.randn(B, nk, d, generator=g)
    v = torch.randn(B, nk, d, generator=g)
    if spike:
        q[:, nq // 2, :64] *= 4.0
    q2, k2, v2 = T._split(q), T._split(k), T._split(v)
    o_ref, qk_ref = T._attn_ref64(T._join(q2), T._join(k2), T._join(v2), H, causal)
    out2 = torch.full((B, nq, 2 * d), float("nan"), dtype=torch.float16, device="cuda")
    qd, kd, vd = q2.cuda(), k2.cuda(), v2.cuda()
    wca._lib.check(lib.wca_test_attention_split(eng._h, vp(qd), vp(kd), vp(vd), vp(out2), None, 0, 0, B, H, nq, nk, causal))
    torch.cuda.synchronize()
    o2 = out2.cpu()
    got = T._join(o2)
    err = (got - o_ref).abs()
    # the same with the kernel's hi half only, and errors relative to the hi rounding
    hi_only = (o2[..., :d].double() - o_ref).abs()
    print("case", (B, H, nq, nk, causal, spike), "max err %.3e (hi alone %.3e), mean err %.3e" % (err.max().item(), hi_only.max().item(), err.mean().item()))
    flat = err.flatten().topk(5)
    for e_, idx in zip(flat.values.tolist(), flat.indices.tolist()):
        b, r, c = idx // (nq * d), (idx // d) % nq, idx % d
        print("   err %.3e at b %d row %d col %d: ref %.9f got hi %.9f lo %.3e" % (e_, b, r, c, o_ref[b, r, c].item(), o2[b, r, c].item(), o2[b, r, d + c].item()))
