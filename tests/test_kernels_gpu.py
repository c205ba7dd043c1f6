"""Kernel-level parity tests (run on the MI355X with `-m gpu`): every HIP kernel is called through the
C ABI (ctypes) and compared with a CPU reference -- the oracle for the integer/order-statistic kernels
(bit-exact), plain PyTorch fp32 for the floating-point kernels (tolerance stated per test)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


@pytest.fixture(scope="module")
def eng(wca):
    syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
    dims = wca.ModelDimensions(80, 1500, 384, 6, 2, 51865, 448, 384, 6, 2)
    m = wca.WhisperAMD(dims, device="cuda:0", max_batch=2, precision="f16")
    m.load_state_dict(syn.random_state_dict(dims, seed=1))
    m._bind_stream()
    return m


# ------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1500, 1152, 384), (77, 130, 64), (128, 128, 64), (3000, 384, 256)])
@pytest.mark.parametrize("mode", ["f16", "f16_gelu", "f32", "accum"])
@pytest.mark.parametrize("tile", [128, 256, 257])
def test_gemm(eng, lib, wca, M, N, K, mode, tile):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = (torch.randn(M, K, generator=g) * 0.5).half()
    w = (torch.randn(N, K, generator=g) * 0.1).half()
    bias = torch.randn(N, generator=g)
    ref = a.float() @ w.float().T + bias
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    if mode == "f16" or mode == "f16_gelu":
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        gelu = int(mode == "f16_gelu")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, gelu, 0 | (tile << 8)))
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        torch.cuda.synchronize()
        # f16 output rounding: 2^-11 relative + fp32 accumulation order
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-3, atol=2e-3)
    elif mode == "f32":
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, 0, 1 | (tile << 8)))
        torch.cuda.synchronize()
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    else:
        base = torch.randn(M, N, generator=g)
        out = base.clone().cuda()
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, 0, 2 | (tile << 8)))
        torch.cuda.synchronize()
        torch.testing.assert_close(out.cpu(), base + ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode", ["f16", "f16_gelu", "f32", "accum"])
@pytest.mark.parametrize("M,N,K", [(64, 1024, 1024), (37, 3072, 1024), (64, 1000, 4096), (1, 4096, 512), (3, 51865, 1024)])
def test_gemm_skinny(eng, lib, wca, M, N, K, mode):
    """M <= 64 weight-streaming kernel of the greedy-decode steps (ragged M, N not a multiple of 16, the logits shape)
    against an fp32 matmul of the same f16 operands; tolerances as in test_gemm."""
    g = torch.Generator().manual_seed(M * 11 + N + K)
    ad = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    wd = (torch.randn(N, K, generator=g) * 0.1).half().cuda()
    bd = torch.randn(N, generator=g).cuda()
    ref = ad.float() @ wd.float().T + bd
    tile = 64
    if mode in ("f16", "f16_gelu"):
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        gelu = int(mode == "f16_gelu")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, gelu, 0 | (tile << 8)))
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        torch.cuda.synchronize()
        torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)
    elif mode == "f32":
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, 0, 1 | (tile << 8)))
        torch.cuda.synchronize()
        torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-4)
    else:
        base = torch.randn(M, N, generator=g).cuda()
        out = base.clone()
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, 0, 2 | (tile << 8)))
        torch.cuda.synchronize()
        torch.testing.assert_close(out, base + ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("mode", ["f16", "f16_gelu", "f32", "accum"])
@pytest.mark.parametrize("tile", [257])
@pytest.mark.parametrize("M,N,K", [(10100, 2500, 256), (8192, 2560, 1024), (9500, 2560, 512), (11700, 1536, 1024),
                                   (8500, 2560, 320), (8300, 2560, 64)])  # odd number of K tiles / one K tile: one tile per workgroup
def test_gemm_persistent_many_tiles(eng, lib, wca, M, N, K, mode, tile):
    """More 256x256 tiles than CUs: a workgroup walks several tiles, its operand stream, bias buffers and ring-slot
    parity carry across tile boundaries (ragged M and N in the first shape; 38 and 46 m-panels in the last two, so the
    last supertile of the tile order is short). Reference: fp32 matmul of the same
    f16 operands on the GPU; tolerances = f16 output rounding (2^-11) resp. fp32 accumulation order."""
    g = torch.Generator().manual_seed(M + N + K)
    ad = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    wd = (torch.randn(N, K, generator=g) * 0.1).half().cuda()
    bd = torch.randn(N, generator=g).cuda()
    ref = ad.float() @ wd.float().T + bd
    if mode in ("f16", "f16_gelu"):
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        gelu = int(mode == "f16_gelu")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, gelu, 0 | (tile << 8)))
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        torch.cuda.synchronize()
        torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)
    elif mode == "f32":
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, 0, 1 | (tile << 8)))
        torch.cuda.synchronize()
        torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-4)
    else:
        base = torch.randn(M, N, generator=g).cuda()
        out = base.clone()
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, 0, 2 | (tile << 8)))
        torch.cuda.synchronize()
        torch.testing.assert_close(out, base + ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("name,N,K,mode", [("qkv", 3072, 1024, "f16"), ("fc1", 4096, 1024, "f16_gelu"), ("fc2", 1024, 4096, "accum")])
def test_gemm_bench_sized_encoder_shapes(eng, lib, wca, name, N, K, mode):
    """The encoder GEMMs at exactly the bench's size (64 utterances: M = 96000 rows, 375 m-panels = 46 supertiles of 8
    + one of 7, 17-23 tiles per workgroup) against an fp32 matmul of the same f16 operands on the GPU."""
    M = 96000
    g = torch.Generator(device="cuda").manual_seed(7)
    ad = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    wd = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bd = torch.randn(N, generator=g, device="cuda")
    if mode == "accum":
        out = torch.randn(M, N, generator=g, device="cuda")
        ref = out + (ad.float() @ wd.float().T + bd)
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, 0, 2))
        torch.cuda.synchronize()
        torch.testing.assert_close(out, ref, rtol=3e-4, atol=3e-4)
    else:
        gelu = int(mode == "f16_gelu")
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), _vp(bd), _vp(out), M, N, K, gelu, 0))
        ref = ad.float() @ wd.float().T + bd
        if gelu:
            ref = torch.nn.functional.gelu(ref)
        torch.cuda.synchronize()
        torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(2000, 768, 1024), (515, 1024, 4096), (4096, 256, 64)])
def test_gemm_256_tile_long_k_and_tails(eng, lib, wca, M, N, K):
    """The 256x256 kernel's DMA ring (prefetch distance 2, counted vmcnt) over many K tiles and ragged M."""
    g = torch.Generator().manual_seed(K + M)
    a = (torch.randn(M, K, generator=g) * 0.3).half()
    w = (torch.randn(N, K, generator=g) * 0.05).half()
    ref = a.float() @ w.float().T
    ad, wd = a.cuda(), w.cuda()
    for _ in range(3):  # repeated launches: a stale-ring race would show up as run-to-run differences
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        for tile in (256, 257):
            out.fill_(float('nan'))
            wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), None, _vp(out), M, N, K, 0, 1 | (tile << 8)))
            torch.cuda.synchronize()
            torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_asymmetric_identity(eng, lib, wca, tile):
    """A = I with an asymmetric W catches a transposed / permuted C write."""
    M = N = K = 128 if tile == 128 else 256
    a = torch.eye(M, K).half()
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125).half()
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    ad, wd = a.cuda(), w.cuda()  # keep the device copies alive across the launch
    wca._lib.check(lib.wca_test_gemm(eng._h, _vp(ad), _vp(wd), None, _vp(out), M, N, K, 0, 1 | (tile << 8)))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), w.float().T.contiguous())


@pytest.mark.parametrize("M,N,K,site", [(12288 + 100, 1024, 1024, 1), (12800, 1024, 4096, 4), (96000, 1024, 1024, 1),
                                         (10240 - 37, 1280, 1280, 1), (24576 + 13, 512, 2048, 4)])
def test_gemm_residual_layernorm_epilogue(eng, lib, wca, M, N, K, site):
    """out_mode 3: x += A W^T + bias and xn = LayerNorm(x) in ONE persistent launch, the row statistics exchanged between
    the N / 256 workgroups of a 256-row panel. Against fp32 torch on the GPU; rows with a large common offset (mean >> std)
    check the Chan-merged variance; ragged M checks the tail tile."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    gamma = torch.rand(N, generator=g, device="cuda") + 0.5
    beta = torch.randn(N, generator=g, device="cuda") * 0.3
    x0 = torch.randn(M, N, generator=g, device="cuda") * 2.0
    x0[5] += 300.0          # a row whose mean dwarfs its spread
    x0[M - 1] -= 1000.0     # ... in the tail tile
    x = x0.clone()
    xn = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
    wca._lib.check(lib.wca_test_gemm_ln(eng._h, _vp(a), _vp(w), _vp(bias), _vp(x), _vp(gamma), _vp(beta), _vp(xn), M, N, K, site))
    torch.cuda.synchronize()
    worst_x = worst_n = 0.0
    for r0 in range(0, M, 16384):
        r1 = min(M, r0 + 16384)
        ref_x = x0[r0:r1] + a[r0:r1].float() @ w.float().T + bias
        ref_n = torch.nn.functional.layer_norm(ref_x, (N,), gamma, beta, 1e-5)
        worst_x = max(worst_x, float(((x[r0:r1] - ref_x).abs() / (1.0 + ref_x.abs())).max()))
        worst_n = max(worst_n, float((xn[r0:r1].float() - ref_n).abs().max()))
    assert worst_x <= 2e-3, worst_x     # f16 operands, fp32 accumulation over K
    assert worst_n <= 1.5e-2, worst_n   # f16 output of O(1) normalised values (gamma up to 1.5, beta 0.3)
    # the separate LayerNorm kernel on the SAME updated x gives the same f16 values up to one rounding of the statistics
    ln = torch.empty(M, N, dtype=torch.float16, device="cuda")
    wca._lib.check(lib.wca_test_layernorm(eng._h, _vp(x), _vp(gamma), _vp(beta), _vp(ln), M, N))
    torch.cuda.synchronize()
    assert float((ln.float() - xn.float()).abs().max()) <= 4e-3


def test_gemm_residual_layernorm_rejects_unsupported_shapes(eng, lib, wca):
    t = torch.zeros(16, device="cuda")
    for M, N, K in ((4416, 1024, 1024), (96000, 1000, 1024), (96000, 1024, 1088)):   # too few tiles / ragged N / odd K-tile count
        with pytest.raises(wca._lib.WcaError):
            wca._lib.check(lib.wca_test_gemm_ln(eng._h, _vp(t), _vp(t), _vp(t), _vp(t), _vp(t), _vp(t), _vp(t), M, N, K, 1))


@pytest.mark.parametrize("M,N,K", [(1500, 1024, 4096), (1037, 1000, 2048), (3000, 512, 2048), (200, 1280, 5120)])
def test_gemm_residual_split_k_small_batches(eng, lib, wca, M, N, K):
    """out_mode 2 (x += A W^T + bias) with few 128 x 128 tiles and a long K -- fc2 of a one- or two-utterance batch: K is
    split over up to four workgroups per tile, partials added in order by a second kernel. Against fp32 torch; a second
    launch on the same inputs gives the same bits (ordered reduction, no atomics)."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    x0 = torch.randn(M, N, generator=g, device="cuda") * 2.0
    outs = []
    for _ in range(2):
        x = x0.clone()
        wca._lib.check(lib.wca_test_gemm(eng._h, _vp(a), _vp(w), _vp(bias), _vp(x), M, N, K, 0, 2))
        torch.cuda.synchronize()
        outs.append(x)
    ref = x0 + a.float() @ w.float().T + bias
    err = float(((outs[0] - ref).abs() / (1.0 + ref.abs())).max())
    assert err <= 2e-3, err
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------- few-row GEMM (decode steps)
def _rows_gemm(eng, lib, wca, a, x, gamma, beta, w, bias, M, N, K, gelu, out_mode, splitk=0, groups=0, c0=None, kv=None, T_max=0, t=0):
    if out_mode == 0:
        c = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
    elif out_mode == 1:
        c = torch.full((M, N), float("nan"), device="cuda")
    else:
        c = c0.clone()
    kk, kv_ = (kv if kv is not None else (None, None))
    wca._lib.check(lib.wca_test_gemm_rows(eng._h, _vp(a), _vp(x), _vp(gamma), _vp(beta), _vp(w), _vp(bias), _vp(c), M, N, K,
                                          gelu, out_mode, splitk, groups, _vp(kk), _vp(kv_), T_max, t))
    torch.cuda.synchronize()
    return c


@pytest.mark.parametrize("M,N,K,gelu,out_mode", [(64, 3072, 1024, 0, 0), (64, 4096, 1024, 1, 0), (64, 1024, 1024, 0, 2), (64, 1024, 4096, 0, 2),
                                                 (37, 1024, 4096, 0, 2), (1, 1024, 1024, 0, 0), (69, 3072, 1024, 0, 0), (128, 1024, 4096, 0, 2),
                                                 (64, 1280, 1280, 0, 2), (64, 1280, 5120, 0, 2), (33, 1536, 384, 1, 0), (64, 51865, 1024, 0, 1),
                                                 (5, 1000, 768, 0, 1)])
def test_gemm_rows_matches_fp32(eng, lib, wca, M, N, K, gelu, out_mode):
    """The few-row kernel on f16 rows (A_MODE 0) against fp32 torch: every decoder shape of a greedy step (medium and
    large widths: split-K 1, 2, 4, 5), ragged M / N, M > 64 (two row blocks), the vocabulary projection (13 column groups per
    workgroup through the register ring)."""
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    a = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    c0 = torch.randn(M, N, generator=g, device="cuda") * 2.0
    c = _rows_gemm(eng, lib, wca, a, None, None, None, w, bias, M, N, K, gelu, out_mode, c0=c0)
    ref = a.float() @ w.float().T + bias
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if out_mode == 2:
        ref = ref + c0
    tol = 2e-3 if out_mode else 6e-3
    err = float(((c.float() - ref).abs() / (1.0 + ref.abs())).max())
    assert err <= tol, err
    # deterministic: a second launch (split-K partials summed by whichever workgroup arrives last) gives the same bits
    c2 = _rows_gemm(eng, lib, wca, a, None, None, None, w, bias, M, N, K, gelu, out_mode, c0=c0)
    assert torch.equal(c, c2)


@pytest.mark.parametrize("M,N,K,gelu,out_mode", [(64, 3072, 1024, 0, 0), (64, 4096, 1024, 1, 0), (64, 1024, 1024, 0, 0), (1, 1024, 1024, 0, 0),
                                                 (50, 2304, 768, 0, 0), (64, 2048, 512, 1, 0), (64, 51865, 1024, 0, 1), (69, 3072, 1024, 0, 0)])
def test_gemm_rows_layernorm_prologue_is_bit_identical_to_separate_launches(eng, lib, wca, M, N, K, gelu, out_mode):
    """A_MODE 1: LayerNorm(x) computed in the GEMM's prologue. The prologue repeats the arithmetic of the LayerNorm kernel and
    the MFMA / cross-wave summation order of the f16-row path, so the result must equal LayerNorm kernel + few-row GEMM on
    its output BIT FOR BIT (and fp32 torch within f16 operand rounding)."""
    g = torch.Generator(device="cuda").manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, generator=g, device="cuda") * 2.0
    x[0] += 100.0
    gamma = torch.rand(K, generator=g, device="cuda") + 0.5
    beta = torch.randn(K, generator=g, device="cuda") * 0.3
    w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    fused = _rows_gemm(eng, lib, wca, None, x, gamma, beta, w, bias, M, N, K, gelu, out_mode)
    xn = torch.empty(M, K, dtype=torch.float16, device="cuda")
    wca._lib.check(lib.wca_test_layernorm(eng._h, _vp(x), _vp(gamma), _vp(beta), _vp(xn), M, K))
    torch.cuda.synchronize()
    sep = _rows_gemm(eng, lib, wca, xn, None, None, None, w, bias, M, N, K, gelu, out_mode)
    assert torch.equal(fused, sep)
    ref = torch.nn.functional.layer_norm(x, (K,), gamma, beta, 1e-5) @ w.float().T + bias
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    err = float(((fused.float() - ref).abs() / (1.0 + ref.abs())).max())
    assert err <= 8e-3, err


def test_gemm_rows_kv_append_routing(eng, lib, wca):
    """QKV projection of a decode step: q columns to C, k / v columns straight into the caches [B][T_max][d] at position t."""
    B, d, T_max, t = 48, 1024, 40, 17
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, d, generator=g, device="cuda")
    gamma = torch.rand(d, generator=g, device="cuda") + 0.5
    beta = torch.randn(d, generator=g, device="cuda") * 0.1
    w = (torch.randn(3 * d, d, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(3 * d, generator=g, device="cuda")
    plain = _rows_gemm(eng, lib, wca, None, x, gamma, beta, w, bias, B, 3 * d, d, 0, 0)
    kc = torch.full((B, T_max, d), 7.0, dtype=torch.float16, device="cuda")
    vc = torch.full((B, T_max, d), 9.0, dtype=torch.float16, device="cuda")
    routed = _rows_gemm(eng, lib, wca, None, x, gamma, beta, w, bias, B, 3 * d, d, 0, 0, kv=(kc, vc), T_max=T_max, t=t)
    assert torch.equal(routed[:, :d], plain[:, :d])
    assert torch.equal(kc[:, t], plain[:, d:2 * d]) and torch.equal(vc[:, t], plain[:, 2 * d:])
    keep = torch.ones(T_max, dtype=torch.bool)
    keep[t] = False
    assert bool((kc[:, keep] == 7.0).all()) and bool((vc[:, keep] == 9.0).all())   # nothing else touched
    assert bool(torch.isnan(routed[:, d:].float()).all())                             # the k / v columns of C are not written


# ------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, H, causal):
    B, nq, d = q.shape
    nk = k.shape[1]
    qh = q.float().view(B, nq, H, 64).permute(0, 2, 1, 3)
    kh = k.float().view(B, nk, H, 64).permute(0, 2, 1, 3)
    vh = v.float().view(B, nk, H, 64).permute(0, 2, 1, 3)
    qk = (qh @ kh.transpose(-1, -2)) * 0.125
    s = qk.clone()
    if causal:
        s = s + torch.full((nq, nk), float("-inf")).triu_(1)
    o = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B, nq, d)
    return o, qk


@pytest.mark.parametrize("B,H,nq,nk,causal,cap_cols", [
    (2, 3, 150, 200, 0, 0), (1, 2, 70, 70, 1, 0), (2, 6, 69, 1500, 0, 500), (1, 4, 448, 1500, 0, 1500),
    (1, 2, 300, 300, 1, 0), (1, 1, 5, 1500, 0, 145), (2, 6, 1500, 1500, 0, 0),
    # one query row per (utterance, head): the greedy-decode kernel (cross-attention length, short / ragged caches, one key)
    (3, 4, 1, 1500, 0, 0), (2, 6, 1, 37, 0, 0), (2, 2, 1, 1, 0, 0), (1, 3, 1, 227, 0, 0), (2, 2, 1, 1, 1, 0),
    # kernel variants forced (bits 8-9 of `causal`): 1 << 8 = the 16x16x32 kernel, 2 << 8 = the 32x32x16 encoder kernel, on
    # ragged shapes (query / key counts not multiples of the 128-row block / 64-key tile, a single key tile, one query row)
    (2, 3, 150, 200, 1 << 8, 0), (2, 3, 150, 200, 2 << 8, 0), (2, 6, 1500, 1500, 1 << 8, 0), (1, 2, 97, 33, 2 << 8, 0),
    (1, 2, 33, 1500, 2 << 8, 0), (3, 1, 129, 64, 2 << 8, 0), (1, 5, 2, 65, 2 << 8, 0), (1, 20, 1500, 1500, 2 << 8, 0),
    (1, 2, 300, 1500, 2 << 8, 0), (1, 3, 700, 129, 2 << 8, 0), (2, 3, 257, 200, 2 << 8, 0), (1, 2, 64, 192, 2 << 8, 0),
    # 3 << 8 = the 32x32x16 kernel with the row sums on the vector ALU (an experiment kept for A/B; auto = 2 << 8, sums on the matrix pipe)
    (2, 3, 150, 200, 3 << 8, 0), (1, 2, 97, 33, 3 << 8, 0), (1, 5, 2, 65, 3 << 8, 0), (1, 20, 1500, 1500, 3 << 8, 0), (2, 3, 257, 200, 3 << 8, 0)])
def test_attention(eng, lib, wca, B, H, nq, nk, causal, cap_cols):
    g = torch.Generator().manual_seed(nq * 13 + nk)
    d = H * 64
    q = torch.randn(B, nq, d, generator=g).half()
    k = torch.randn(B, nk, d, generator=g).half()
    v = torch.randn(B, nk, d, generator=g).half()
    # a few large logits so the online-softmax rescale path is exercised
    q[:, nq // 2, :64] *= 4.0
    o_ref, qk_ref = _attn_ref(q, k, v, H, causal & 1)
    out = torch.full((B, nq, d), float("nan"), dtype=torch.float16, device="cuda")
    cap_ld = (cap_cols + 3) & ~3
    cap = torch.full((B, H, nq, max(cap_ld, 4)), float("nan"), device="cuda") if cap_cols else None
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    wca._lib.check(lib.wca_test_attention(eng._h, _vp(qd), _vp(kd), _vp(vd), _vp(out), _vp(cap),
                                          cap_ld, cap_cols, B, H, nq, nk, causal))
    torch.cuda.synchronize()
    # P is rounded to f16 before P.V and the output is f16: ~1e-3 relative
    torch.testing.assert_close(out.float().cpu(), o_ref, rtol=4e-3, atol=4e-3)
    if cap_cols:
        got = cap.cpu()[..., :cap_cols]
        # f16 operands are exact, fp32 accumulation over 64 terms
        torch.testing.assert_close(got, qk_ref[..., :cap_cols], rtol=1e-4, atol=1e-4)


def test_attention_rescale_branch_forced(eng, lib, wca):
    """The lazy running-max scheme of both kernels takes its rescale branch only when a row's maximum grows: force it at a
    chosen LATE key tile (one key row spiked against one query row, far above everything before it) and compare the FULL
    output with an fp64 reference (a passing check on bounded random data never exercises that branch)."""
    B, H, S = 1, 2, 700
    d = H * 64
    g = torch.Generator().manual_seed(99)
    q = torch.randn(B, S, d, generator=g).half()
    k = torch.randn(B, S, d, generator=g).half()
    v = torch.randn(B, S, d, generator=g).half()
    for row, key in ((5, 650), (300, 333), (699, 64), (130, 699)):
        k[0, key, :64] = (q[0, row, :64].float() * 2.0).half()   # q.k*scale ~ 2*|q|^2/8 ~ 16: a jump of the row maximum
    # growth BELOW the deferral threshold (RESCALE_THR = 8 in log2 units) at one tile, then past it at a later one: the
    # probabilities of the first spike stay un-rescaled (up to 2^8) until the second one raises the maximum
    for row, key, gain in ((40, 100, 0.6), (40, 400, 0.95), (520, 70, 0.5), (520, 200, 0.7), (520, 600, 1.3)):
        k[0, key, 64:128] = (q[0, row, 64:128].float() * gain).half()
    qh = q.double().view(B, S, H, 64).permute(0, 2, 1, 3)
    kh = k.double().view(B, S, H, 64).permute(0, 2, 1, 3)
    vh = v.double().view(B, S, H, 64).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1) @ vh).permute(0, 2, 1, 3).reshape(B, S, d)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    for variant in (1, 2, 3):
        out = torch.full((B, S, d), float("nan"), dtype=torch.float16, device="cuda")
        wca._lib.check(lib.wca_test_attention(eng._h, _vp(qd), _vp(kd), _vp(vd), _vp(out), None, 0, 0, B, H, S, S, variant << 8))
        torch.cuda.synchronize()
        err = (out.double().cpu() - ref).abs().max().item()
        assert err <= 4e-3, (variant, err)


def test_attention_bench_sized_encoder(eng, lib, wca):
    """Encoder self-attention at the bench's size (64 utterances x 16 heads x 1500 x 1500: 12288 workgroups over the
    XCD-aware grid) against fp32 softmax(q k^T / 8) v computed on the GPU four utterances at a time."""
    B, H, S = 64, 16, 1500
    d = H * 64
    g = torch.Generator(device="cuda").manual_seed(21)
    q = torch.randn(B, S, d, generator=g, device="cuda").half()
    k = torch.randn(B, S, d, generator=g, device="cuda").half()
    v = torch.randn(B, S, d, generator=g, device="cuda").half()
    q[:, S // 3, :64] *= 4.0
    out = torch.full((B, S, d), float("nan"), dtype=torch.float16, device="cuda")
    wca._lib.check(lib.wca_test_attention(eng._h, _vp(q), _vp(k), _vp(v), _vp(out), None, 0, 0, B, H, S, S, 0))
    torch.cuda.synchronize()
    worst = 0.0
    for b0 in range(0, B, 4):
        qq = q[b0:b0 + 4].float().view(4, S, H, 64).transpose(1, 2)
        kk = k[b0:b0 + 4].float().view(4, S, H, 64).transpose(1, 2)
        vv = v[b0:b0 + 4].float().view(4, S, H, 64).transpose(1, 2)
        p = torch.softmax(qq @ kk.transpose(-1, -2) * 0.125, dim=-1)
        ref = (p @ vv).transpose(1, 2).reshape(4, S, d)
        worst = max(worst, float((out[b0:b0 + 4].float() - ref).abs().max()))
    assert worst <= 4e-3, worst  # P and the output are f16, Q is pre-scaled in f16: ~1e-3 relative


def test_layernorm(eng, lib, wca):
    g = torch.Generator().manual_seed(3)
    for d in (384, 1024, 1280):
        x = torch.randn(37, d, generator=g) * 3 + 1
        gm, bt = torch.randn(d, generator=g), torch.randn(d, generator=g)
        out = torch.empty(37, d, dtype=torch.float16, device="cuda")
        xd, gd, bd = x.cuda(), gm.cuda(), bt.cuda()
        wca._lib.check(lib.wca_test_layernorm(eng._h, _vp(xd), _vp(gd), _vp(bd), _vp(out), 37, d))
        torch.cuda.synchronize()
        ref = torch.nn.functional.layer_norm(x, (d,), gm, bt, 1e-5)
        torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-3, atol=2e-3)


# ------------------------------------------------------------------------------- DTW (bit-exact)
def _dtw_gpu(eng, lib, wca, m):
    m = np.ascontiguousarray(m, dtype=np.float32)
    N, M = m.shape
    ti = np.zeros(N + M, dtype=np.int32)
    tj = np.zeros(N + M, dtype=np.int32)
    n = C.c_int32(0)
    wca._lib.check(lib.wca_dtw(eng._h, m.ctypes.data_as(C.POINTER(C.c_float)), N, M, ti.ctypes.data_as(C.POINTER(C.c_int32)),
                               tj.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)))
    return ti[:n.value].astype(np.int64), tj[:n.value].astype(np.int64)


def _dtw_cases():
    rng = np.random.default_rng(0)
    cases = []
    for (N, M) in [(1, 1), (1, 7), (5, 1), (3, 3), (65, 500), (64, 500), (66, 145), (36, 145), (129, 700), (444, 1500), (200, 37),
                   (448, 100)]:
        cases.append(("rand%dx%d" % (N, M), rng.random((N, M), dtype=np.float32)))
    cases.append(("const", np.full((40, 90), 0.25, dtype=np.float32)))
    cases.append(("zeros", np.zeros((17, 33), dtype=np.float32)))
    cases.append(("ints", rng.integers(0, 3, size=(65, 300)).astype(np.float32)))
    cases.append(("ints_tall", rng.integers(0, 2, size=(130, 90)).astype(np.float32)))
    sm = torch.softmax(torch.from_numpy(rng.standard_normal((65, 500)).astype(np.float32)) * 3, -1).numpy()
    cases.append(("softmax", sm))
    diag = np.full((50, 400), 1e-4, dtype=np.float32)
    for i in range(50):
        diag[i, i * 8:(i + 1) * 8] = 0.5
    cases.append(("diagonal", diag))
    return cases


@pytest.mark.parametrize("name,m", _dtw_cases(), ids=[c[0] for c in _dtw_cases()])
def test_dtw_bit_exact(eng, lib, wca, name, m):
    from oracle import timing_ref
    ti, tj = _dtw_gpu(eng, lib, wca, m)
    ref = timing_ref.dtw(-torch.from_numpy(m))
    assert np.array_equal(ti, ref[0]) and np.array_equal(tj, ref[1])


def test_dtw_batch_jump_frames(eng, lib, wca):
    from oracle import timing_ref
    rng = np.random.default_rng(5)
    P, N, M = 7, 65, 333
    mats = rng.random((P, N, M), dtype=np.float32)
    jf = np.zeros((P, N), dtype=np.int32)
    md = torch.from_numpy(mats).cuda()
    wca._lib.check(lib.wca_dtw_batch_dev(eng._h, _vp(md), P, N, M, jf.ctypes.data_as(C.POINTER(C.c_int32))))
    for p in range(P):
        ti, tj = timing_ref.dtw(-torch.from_numpy(mats[p]))
        jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
        assert np.array_equal(jf[p], tj[jumps])


# ------------------------------------------------------------------------------- median filter (bit-exact)
@pytest.mark.parametrize("F,w", [(500, 3), (500, 7), (145, 5), (1500, 7), (10, 9), (3, 7), (64, 1), (777, 11), (40, 33)])
def test_median_filter_bit_exact(eng, lib, wca, F, w):
    from oracle import timing_ref
    g = torch.Generator().manual_seed(F + w)
    x = torch.randn(37, F, generator=g)
    x[3, : min(F, 20)] = 1.5  # ties
    out = torch.empty_like(x).cuda()
    xd = x.cuda()
    wca._lib.check(lib.wca_median_filter(eng._h, _vp(xd), _vp(out), 37, F, w))
    torch.cuda.synchronize()
    ref = timing_ref.median_filter(x.view(1, 1, 37, F), w).reshape(37, F)
    assert torch.equal(out.cpu(), ref)


# ------------------------------------------------------------------------------- filter_attention / force_align
def _fa_gpu(eng, lib, wca, A, topk, wc, wr, wv):
    L, H, n, F = A.shape
    sc = np.zeros(L * H, dtype=np.float32)
    keff = min(topk, L * H)
    idx = np.zeros(keff, dtype=np.int32)
    ss = np.zeros(keff, dtype=np.float32)
    Ad = A.cuda().contiguous()
    wca._lib.check(lib.wca_filter_attention(eng._h, _vp(Ad), L, H, n, F, topk, wc, wr, wv,
                                            sc.ctypes.data_as(C.POINTER(C.c_float)), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                            ss.ctypes.data_as(C.POINTER(C.c_float))))
    return sc, idx, ss


@pytest.mark.parametrize("shape", [(4, 6, 12, 50), (6, 8, 69, 500), (2, 3, 448, 130), (3, 2, 30, 1500)])
@pytest.mark.parametrize("wts", [(1.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 1.0, 1.0)])
def test_filter_attention(eng, lib, wca, shape, wts):
    from oracle import timing_ref
    g = torch.Generator().manual_seed(sum(shape))
    A = torch.softmax(torch.randn(*shape, generator=g) * 3, -1)
    topk = 5
    sc, idx, ss = _fa_gpu(eng, lib, wca, A, topk, *wts)
    _, ref = timing_ref.filter_attention(A, topk, *wts)
    H = shape[1]
    ref_scores_all = {(l, h): None for l in range(shape[0]) for h in range(H)}
    # scores: fp32 reductions in a different order -> 1e-5 relative
    full = timing_ref.filter_attention(A, shape[0] * H, *wts)[1]
    for s, (l, h), _ in full:
        assert abs(sc[l * H + h] - s) <= 2e-5 * max(1.0, abs(s))
    # selection, position by position: the reference's head, or one whose reference score is within the
    # reduction-order noise of it (no blanket skip: every position is asserted)
    allref = {l * H + h: s for s, (l, h), _ in full}
    assert len(set(idx.tolist())) == len(idx)
    for pos, (rs, (l, h), _) in enumerate(ref):
        if int(idx[pos]) != l * H + h:
            assert abs(allref[int(idx[pos])] - rs) <= 4e-5 * max(1.0, abs(rs)), (pos, int(idx[pos]), l * H + h)


@pytest.mark.parametrize("aggr,topk", [("mean", -1), ("topk", 10), ("topk", 3)])
def test_force_align_matrix_and_path(eng, lib, wca, aggr, topk):
    from oracle import timing_ref
    g = torch.Generator().manual_seed(11)
    L, H, n, F = 6, 8, 69, 500
    A = torch.softmax(torch.randn(L, H, n, F, generator=g) * 4, -1)
    opts = eng.make_opts(aggregation=aggr, topk=topk, sot_len=3)
    N = n - 3 - 1
    mat = np.zeros((N, F), dtype=np.float32)
    ti = np.zeros(N + F, dtype=np.int32)
    tj = np.zeros(N + F, dtype=np.int32)
    plen = C.c_int32(0)
    k = max(topk, 1)
    sel = np.zeros(k, dtype=np.int32)
    ssc = np.zeros(k, dtype=np.float32)
    Ad = A.cuda().contiguous()
    wca._lib.check(lib.wca_force_align(eng._h, _vp(Ad), L, H, n, F, C.byref(opts), mat.ctypes.data_as(C.POINTER(C.c_float)),
                                       ti.ctypes.data_as(C.POINTER(C.c_int32)), tj.ctypes.data_as(C.POINTER(C.c_int32)),
                                       C.byref(plen), sel.ctypes.data_as(C.POINTER(C.c_int32)), ssc.ctypes.data_as(C.POINTER(C.c_float))))
    ref_m, ref_scores = timing_ref.aggregate(A, aggr, topk)
    ref_m = ref_m[3:-1]
    # aggregated matrix: fp32, different reduction order for the column norms
    np.testing.assert_allclose(mat, ref_m.numpy(), rtol=2e-5, atol=1e-7)
    # DTW of the GPU matrix must be bit-exact w.r.t. the oracle DTW of the SAME matrix
    ref_path = timing_ref.dtw(-torch.from_numpy(mat))
    assert np.array_equal(ti[:plen.value], ref_path[0]) and np.array_equal(tj[:plen.value], ref_path[1])
    if aggr == "topk":
        assert list(sel[:topk]) == [l * H + h for _, (l, h), _ in ref_scores]


# ------------------------------------------------------------------------------- log-mel
def test_logmel_vs_oracle(eng, wca):
    from oracle import whisper_ref
    syn = importlib.import_module("whisper-char-alignment_amd.synthetic")
    audio_mod = importlib.import_module("whisper-char-alignment_amd.audio")
    filt = audio_mod.mel_filters(80)
    pcm = np.stack([syn.synth_audio(0, 160000), syn.synth_audio(1, 160000)])
    pcm[1, 100000:] = 0.0
    got = eng.log_mel(torch.from_numpy(pcm).cuda(), n_samples=[160000, 100000]).cpu()
    ref = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm)), filt)
    # fp32 DFT by direct summation vs FFT: values are log10/4-scaled, 1e-4 absolute
    assert got.shape == ref.shape == (2, 80, 3000)
    assert (got - ref).abs().max().item() < 2e-4


def test_logmel_full_length_and_sample(eng, wca):
    from oracle import whisper_ref
    import os
    audio_mod = importlib.import_module("whisper-char-alignment_amd.audio")
    filt = audio_mod.mel_filters(80)
    rng = np.random.default_rng(7)
    full = (0.05 * rng.standard_normal(480000)).astype(np.float32)  # exercises the reflect padding at 30 s
    sample = np.load(os.path.join(os.path.dirname(__file__), "golden", "sample_pcm_int16.npy")).astype(np.float32) / 32768.0
    for pcm in (full, sample):
        got = eng.log_mel(torch.from_numpy(pcm).cuda()).cpu()
        ref = whisper_ref.log_mel_spectrogram(whisper_ref.pad_or_trim(torch.from_numpy(pcm)), filt)
        assert (got - ref).abs().max().item() < 2e-4


# ------------------------------------------------------------------------------- head statistics: lean kernel == general kernel
@pytest.mark.parametrize("w", [1, 3, 5, 7, 9])
@pytest.mark.parametrize("n,S,F", [(1, 1, 1), (3, 2, 2), (5, 70, 3), (4, 64, 63), (7, 64, 64), (6, 130, 65), (70, 512, 500), (9, 777, 777),
                                   (11, 1500, 1024), (5, 1500, 1500)])
def test_head_stats_lean_kernel_bit_identical_to_general(eng, wca, switch, w, n, S, F):
    """`head_stats_fast_kernel` (postproc.hip: packed-f32 libm expf sequence, row-shared reciprocal refinement of the IEEE division, v_med3
    medians, no exec masking) must give the bits of the general kernel on every element: logits of wide dynamic range (rows whose smallest
    exponent argument lies below -68 take the plain-division branch, rows with -inf and with a 1e4 outlier included), every unrolled filter
    width, frame counts on both sides of the 64-lane groups, tight rows (S == F: clamped loads) and padded rows."""
    tm = importlib.import_module("whisper-char-alignment_amd.timing")
    g = torch.Generator().manual_seed(1000 * w + 10 * n + F)
    qk = torch.randn(2, 3, n, S, generator=g) * 4.0
    qk[0, 1] *= 8.0                       # x down to ~ -200: underflow arm, plain-division rows
    qk[1, 0] *= 0.05                      # near-uniform rows
    if S > 4:
        qk[1, 1, 0, 1] = float("-inf")
        qk[1, 2, n - 1, S // 2] = 1.0e4
    scale = 0.37
    if S >= 64 and n >= 3:                # exponent arguments sweeping through the branch threshold (-68) and the underflow bound (-103.3)
        qk[0, 2, 0] = torch.linspace(0, -110 / scale, S)
        qk[0, 2, 1] = torch.linspace(0, -67.9 / scale, S)
        qk[0, 2, 2] = torch.linspace(-72 / scale, 0, S)
    qk = qk.cuda()
    switch("head_stats_general", 1)
    ref = tm.attention_weights(qk, F, medfilt_width=w, qk_scale=scale).cpu()
    switch("head_stats_general", 0)
    out = tm.attention_weights(qk, F, medfilt_width=w, qk_scale=scale).cpu()
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (out - ref).abs().max().item()
    # and both are the softmax they claim to be
    x = qk.cpu()[..., :F]
    if w > 1 and F > w // 2:
        x = torch.nn.functional.pad(x.reshape(1, -1, F), (w // 2, w // 2), mode="reflect").unfold(-1, w, 1).sort(-1)[0][..., w // 2].reshape(x.shape)
    want = torch.softmax(x.double() * scale, -1)
    assert (out.double() - want).abs().max().item() < 2e-6
