"""ORACLE -- test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product path).

CPU restatement (numpy / torch-CPU / plain C via ctypes) of the reference's alignment API:
  filter_attention   <- /root/reference/timing.py:13-43   (+ metrics.py:99-111 coverage_penalty)
  get_attentions     <- /root/reference/timing.py:45-67
  force_align        <- /root/reference/timing.py:69-114
and of the two upstream helpers it calls (whisper.timing.median_filter / dtw, absent third-party
dependency openai-whisper, restated from its published algorithm).

Pinning: filter_attention / the aggregation / the jump->time arithmetic are checked against the REAL
reference `timing.py` executed in the build container under stub modules (tests/golden/make_golden.py ->
tests/golden/*.npz). median_filter / dtw have no upstream fixture offline: PARITY UNPINNED for those two,
they are checked against hand-derived known answers, a brute-force path search and the independent ports of both
functions in HuggingFace transformers (tests/test_oracle.py::test_dtw_and_median_vs_hf_ports).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

TOKENS_PER_SECOND = 50
_HERE = os.path.dirname(os.path.abspath(__file__))
_clib = None


def _load_clib():
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.wca_oracle_dtw.restype = ctypes.c_int
            lib.wca_oracle_dtw.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            lib.wca_oracle_median_filter.restype = None
            lib.wca_oracle_median_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int]
            _clib = lib
        else:
            _clib = False
    return _clib


# ------------------------------------------------------------------ whisper.timing.median_filter
def median_filter(x, filter_width):
    """Reflect-pad by w//2 and take the sliding median along the last axis (CPU branch of upstream)."""
    x = torch.as_tensor(x)
    pad_width = filter_width // 2
    if x.shape[-1] <= pad_width:
        return x
    ndim = x.ndim
    if ndim <= 2:
        x = x[None, None, :]
    assert filter_width > 0 and filter_width % 2 == 1, "`filter_width` should be an odd number"
    x = F.pad(x, (pad_width, pad_width, 0, 0), mode="reflect")
    result = x.unfold(-1, filter_width, 1).sort()[0][..., filter_width // 2]
    if ndim <= 2:
        result = result[0, 0]
    return result


# ------------------------------------------------------------------ whisper.timing.dtw_cpu + backtrace
def dtw_py(x):
    """Pure-Python literal restatement (small cases only). x: [N, M] float64 (already negated)."""
    x = np.asarray(x, dtype=np.float64)
    N, M = x.shape
    cost = np.ones((N + 1, M + 1), dtype=np.float32) * np.inf
    trace = -np.ones((N + 1, M + 1), dtype=np.float32)
    cost[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0 = cost[i - 1, j - 1]
            c1 = cost[i - 1, j]
            c2 = cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = x[i - 1, j - 1] + c  # float64 sum stored into a float32 table
            trace[i, j] = t
    i, j = N, M
    trace[0, :] = 2
    trace[:, 0] = 1
    result = []
    while i > 0 or j > 0:
        result.append((i - 1, j - 1))
        if trace[i, j] == 0:
            i -= 1
            j -= 1
        elif trace[i, j] == 1:
            i -= 1
        elif trace[i, j] == 2:
            j -= 1
        else:
            raise ValueError("Unexpected trace[i, j]")
    result = np.array(result)
    return result[::-1, :].T


def dtw(x):
    """`whisper.timing.dtw` on a CPU tensor: dtw_cpu(x.double()). Returns (text_indices, time_indices)."""
    x = np.ascontiguousarray(torch.as_tensor(x).double().cpu().numpy())
    N, M = x.shape
    lib = _load_clib()
    if not lib:
        return dtw_py(x)
    ti = np.empty(N + M, dtype=np.int32)
    tj = np.empty(N + M, dtype=np.int32)
    n = lib.wca_oracle_dtw(x.ctypes.data, N, M, ti.ctypes.data, tj.ctypes.data)
    if n < 0:
        raise MemoryError("oracle dtw")
    return np.stack([ti[:n].astype(np.int64), tj[:n].astype(np.int64)])


# ------------------------------------------------------------------ metrics.py:99-111
def coverage_penalty(attn, threshold=0.5):
    coverage = torch.sum(attn, dim=0)
    penalty = torch.max(coverage, coverage.clone().fill_(threshold)).sum(-1)
    return penalty - coverage.size(-1) * threshold


# ------------------------------------------------------------------ timing.py:13-43
def filter_attention(attns, topk=20, w_colnorm=1, w_rownorm=1, w_coverage=0):
    n_layers, n_heads = attns.size(0), attns.size(1)
    score_matrix = torch.zeros(n_layers, n_heads)
    if w_colnorm > 0:
        score_matrix += w_colnorm * attns.norm(dim=-2).sum(-1)
    if w_rownorm > 0:
        score_matrix += w_rownorm * attns.norm(dim=-1).sum(-1)
    scores = []
    for l in range(n_layers):
        for h in range(n_heads):
            score = score_matrix[l, h]
            if w_coverage > 0:
                score -= w_coverage * coverage_penalty(attns[l, h])
            scores.append((score.item(), (l, h), f"sample_layer{l}_head{h}"))
    scores_sorted = sorted(scores)[-topk:]
    selected = [attns[l, h].unsqueeze(0) for _, (l, h), _ in scores_sorted]
    return selected, scores_sorted


# ------------------------------------------------------------------ timing.py:45-67
def get_attentions(mel, tokens, model_ref, max_frames, medfilt_width=7, qk_scale=1.0):
    """model_ref: oracle.whisper_ref.WhisperRef. mel [n_mels,3000], tokens [n] -> (weights [L,H,n,F], logits [n,V])."""
    logits, qks = model_ref.forward(mel.unsqueeze(0), tokens.unsqueeze(0))
    weights = torch.cat(qks)  # layers * heads * tokens * frames
    weights = weights[..., :max_frames]
    weights = median_filter(weights, medfilt_width)
    weights = (weights * qk_scale).softmax(dim=-1)
    return weights, logits[0]


def attention_weights(qks, max_frames, medfilt_width=7, qk_scale=1.0):
    """The post-capture half of get_attentions (timing.py:63-66) on given logits: qks = list of per-layer
    [1, H, n, S] tensors (what the forward hooks collected) or one [L, H, n, S] tensor."""
    weights = torch.cat(list(qks)) if isinstance(qks, (list, tuple)) else torch.as_tensor(qks)
    weights = weights[..., :max_frames]
    weights = median_filter(weights, medfilt_width)
    return (weights * qk_scale).softmax(dim=-1)


# ------------------------------------------------------------------ timing.py:69-114
def aggregate(ws, aggregation="mean", topk=-1, w_colnorm=1.0, w_rownorm=1.0, w_coverage=0.0):
    scores = None
    if aggregation == "mean":
        ws = ws / ws.norm(dim=-2, keepdim=True)
        n_layers = ws.size(0)
        ws = ws[n_layers // 2:]
        matrix = ws.mean(axis=(0, 1))
    elif aggregation == "topk":
        assert topk > 0
        sel, scores = filter_attention(ws, topk, w_colnorm, w_rownorm, w_coverage)
        matrix = torch.cat(sel, 0)
        col_norm = matrix.norm(dim=-2, keepdim=True)
        matrix = torch.mean(matrix / col_norm, 0)
    elif aggregation == "grad_norm":
        matrix = ws
    else:
        raise ValueError(aggregation)
    return matrix, scores


def jumps_to_times(text_indices, time_indices, word_tokens):
    """timing.py:108-113"""
    word_boundaries = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    jump_times = time_indices[jumps] / TOKENS_PER_SECOND
    return jump_times[word_boundaries[:-1]], jump_times[word_boundaries[1:]]


def force_align(ws, tokens, tokenizer, aligned_unit_type="subword", aggregation="mean", topk=-1, w_colnorm=1.0,
                w_rownorm=1.0, w_coverage=0.0, split_fn=None):
    """split_fn(tokens, tokenizer, unit) -> (words, word_tokens); defaults to oracle.tokenizer_ref."""
    if split_fn is None:
        from .tokenizer_ref import split_tokens_on_spaces as split_fn
    matrix, scores = aggregate(ws, aggregation, topk, w_colnorm, w_rownorm, w_coverage)
    matrix = matrix[len(tokenizer.sot_sequence):-1].cpu()
    text_indices, time_indices = dtw(-matrix)
    words, word_tokens = split_fn(tokens + [tokenizer.eot], tokenizer, aligned_unit_type)
    if len(word_tokens) <= 1:
        return [[], [], [], [], None]
    start_times, end_times = jumps_to_times(text_indices, time_indices, word_tokens)
    return words, start_times, end_times, matrix, scores


# ------------------------------------------------------------------ timing.py:157-165 (default_find_alignment, arithmetic part)
def default_find_alignment(qk, heads, text_tokens, tokenizer, max_frames, medfilt_width=7, qk_scale=1.0):
    """timing.py:116-186 after the forward: qk [L, H, n, S] captured logits (hooks keep outs[-1][0]), heads = list of
    (l, h) in the order of model.alignment_heads.indices().T (row-major over the (L, H) mask). Returns the reference's
    5-tuple (words, start_times, end_times, normalised weights [n_heads, n, F], None)."""
    sot_len = len(tokenizer.sot_sequence)
    qk = torch.as_tensor(qk)
    weights = torch.stack([qk[l][h] for l, h in heads])          # :155
    weights = weights[:, :, :max_frames]                          # :156
    weights = median_filter(weights, medfilt_width)               # :157
    weights = (weights * qk_scale).softmax(dim=-1)                # :158
    std, mean = torch.std_mean(weights, dim=-2, keepdim=True, unbiased=False)
    weights = (weights - mean) / std                              # :159-160
    matrix = weights.mean(axis=0)[sot_len:-1]                     # :162-163
    text_indices, time_indices = dtw(-matrix)                     # :165
    words, word_tokens = tokenizer.split_to_word_tokens(list(text_tokens) + [tokenizer.eot])
    if len(word_tokens) <= 1:
        return [[], [], [], [], None]
    start_times, end_times = jumps_to_times(text_indices, time_indices, word_tokens)
    return words, start_times, end_times, weights, None


def default_alignment_matrix(weights, heads, sot_len):
    """weights [L, H, n, F] already median filtered + softmaxed (timing.py:157-158); heads: list of (l, h).
    std/mean normalisation over the token axis, mean over heads, [sot:-1] slice (timing.py:159-163)."""
    w = torch.stack([weights[l][h] for l, h in heads])
    std, mean = torch.std_mean(w, dim=-2, keepdim=True, unbiased=False)
    w = (w - mean) / std
    matrix = w.mean(axis=0)
    return matrix[sot_len:-1]
