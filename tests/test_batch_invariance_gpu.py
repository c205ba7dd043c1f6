"""`-m gpu`: does the micro-batch SIZE change the alignment? The reference is batch-1 (infer_ali.py:48); here `--batch_size` picks the
GEMM kernel by row count (few-row kernel <= 128 decoder rows, split-K for few tiles with K >= 2048, the persistent 256 x 256 kernel
for large batches), i.e. the fp32 summation ORDER -- never the operands. Bounded here at BASELINE configs[1] (medium dims, 10 s,
64 chars) and configs[3] (large-v2 dims, n = 448, F = 1500):
  * alignment-like checkpoint (synthetic.aligned_state_dict: what the method is used on): jump frames IDENTICAL at every batch size;
  * seeded random weights (maps without alignment, near-tied DTW steps): the number of differing frames is REPORTED for the default
    mode and must be zero in the reference-precision mode (summation-order noise 1e-6 instead of f16 operand noise... the operands
    are the same in both modes, but the default mode's f16 ROUNDING of intermediate activations amplifies a last-bit difference of
    one GEMM into the next one's operands).
Numbers go to gpurun_out/r03_batch_invariance.txt (copied to profiles/)."""
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _m(n):
    return importlib.import_module("whisper-char-alignment_amd." + n)


def _log(line):
    print(line, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "r03_batch_invariance.txt"), "a") as f:
            f.write(line + "\n")


def _frames_at_batch(model, tok, utts, B, n_samples, F, opts):
    """{utterance index: jump frames} with the utterances aligned in micro-batches of exactly B (the last one filled by repetition)."""
    out = {}
    for lo in range(0, len(utts), B):
        chunk = list(range(lo, min(lo + B, len(utts))))
        idx = (chunk * ((B + len(chunk) - 1) // len(chunk)))[:B]
        pcm = np.stack([utts[i][0] for i in idx])
        n_max = max(len(utts[i][1]) for i in idx)
        toks = np.full((B, n_max), tok.eot, dtype=np.int64)
        for j, i in enumerate(idx):
            toks[j, :len(utts[i][1])] = utts[i][1]
        jump, _sel = model.align_batch(torch.from_numpy(pcm).cuda(), [n_samples] * B, torch.from_numpy(toks).cuda(), [len(utts[i][1]) for i in idx],
                                       [F] * B, opts)
        for j, i in enumerate(idx[:len(chunk)]):
            out[i] = jump[j, :len(utts[i][1])].copy()
    return out


def _compare(frames, sizes, label):
    base = frames[sizes[0]]
    worst, n_diff_frames, n_diff_utts, total = 0, 0, 0, 0
    for B in sizes[1:]:
        for i in base:
            d = np.abs(base[i].astype(np.int64) - frames[B][i].astype(np.int64))
            total += d.size
            worst = max(worst, int(d.max()))
            n_diff_frames += int((d > 0).sum())
            n_diff_utts += int(d.max() > 0)
            if d.max() > 0:   # which frames, for the log
                w = np.nonzero(d)[0]
                _log("    %s: utterance %d at batch size %d: positions %s: batch-size-%d values %s, here %s"
                     % (label.split(",")[0], i, B, w[:8].tolist(), sizes[0], base[i][w[:8]].tolist(), frames[B][i][w[:8]].tolist()))
    _log("%s: batch sizes %s, %d utterances: %d of %d jump frames differ from batch size %d (%d utterance x size pairs), largest difference %d frame(s)"
         % (label, sizes, len(base), n_diff_frames, total, sizes[0], n_diff_utts, worst))
    return worst, n_diff_frames


def _utts(syn, rt, tok, ids, n_samples, chars):
    out = []
    for u in ids:
        tt = rt.encode(syn.synth_text(u, chars), tok, "char")
        out.append((syn.synth_audio(u, n_samples), [*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot]))
    return out


def test_config1_medium_batch_sizes_1_2_64(wca):
    syn, tk, rt = _m("synthetic"), _m("tokenizer"), _m("retokenize")
    dims = wca.dims_for("medium")
    tok = tk.get_tokenizer(True, language="English")
    sizes = [1, 2, 64]
    # ---- alignment-like checkpoint, default (f16-operand) mode: identical frames at every batch size
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=64, precision="f16").load_state_dict(syn.aligned_state_dict(dims, seed=0))
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    utts = _utts(syn, rt, tok, range(300, 308), 160000, 64)
    frames = {B: _frames_at_batch(model, tok, utts, B, 160000, 500, opts) for B in sizes}
    worst, ndiff = _compare(frames, sizes, "configs[1] medium dims, alignment-like checkpoint, f16 mode")
    assert worst == 0 and ndiff == 0
    # determinism at a fixed size: the same batch twice
    again = _frames_at_batch(model, tok, utts, 64, 160000, 500, opts)
    assert all(np.array_equal(again[i], frames[64][i]) for i in again)
    del model
    torch.cuda.empty_cache()
    # ---- seeded random weights (the bench's checkpoint): reported in the default mode, zero in the reference-precision mode
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=64, precision="f16").load_state_dict(syn.random_state_dict(dims, seed=0, cross_qk_std=0.08))
    utts = _utts(syn, rt, tok, list(range(100, 108)) + list(range(10000, 10008)), 160000, 64)
    frames = {B: _frames_at_batch(model, tok, utts, B, 160000, 500, opts) for B in sizes}
    _compare(frames, sizes, "configs[1] medium dims, random peaky weights, f16 mode")
    model.set_precision("split")
    frames = {B: _frames_at_batch(model, tok, utts, B, 160000, 500, opts) for B in sizes}
    worst, ndiff = _compare(frames, sizes, "configs[1] medium dims, random peaky weights, split (reference-precision) mode")
    assert worst <= 1, (worst, ndiff)
    del model
    torch.cuda.empty_cache()


def test_config3_large_v2_batch_sizes_1_2_8(wca):
    """configs[3] shape: large-v2 dims, n = 448 tokens, F = 1500 frames (a 64-utterance batch of this shape would need a 110 GB
    capture buffer: 8 is the large size here). Alignment-like checkpoint at 3.3 frames per token."""
    syn, tk, rt = _m("synthetic"), _m("tokenizer"), _m("retokenize")
    dims = wca.dims_for("large-v2")
    tok = tk.get_tokenizer(True, language="English")
    sizes = [1, 2, 8]
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=8, precision="f16").load_state_dict(syn.aligned_state_dict(dims, seed=0, frames_per_token=3.3))
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=7)
    utts = _utts(syn, rt, tok, range(900, 904), 480000, 443)
    assert all(len(u[1]) == 448 for u in utts)
    frames = {B: _frames_at_batch(model, tok, utts, B, 480000, 1500, opts) for B in sizes}
    worst, ndiff = _compare(frames, sizes, "configs[3] large-v2 dims n=448 F=1500, alignment-like checkpoint, f16 mode")
    assert worst == 0 and ndiff == 0
    del model
    torch.cuda.empty_cache()
