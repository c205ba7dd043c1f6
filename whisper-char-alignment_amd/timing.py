"""Drop-in for the reference's alignment API (timing.py:13-114): same function names, arguments, return
values and error behaviour; the arithmetic runs in libwca.so on the MI355X.

  get_attentions(mel, tokens, model, tokenizer, max_frames, medfilt_width=7, qk_scale=1.0) -> (weights, logits)
  filter_attention(attns, topk=20, w_colnorm=1, w_rownorm=1, w_coverage=0) -> (selected_attns, scores_sorted)
  force_align(ws, tokens, tokenizer, aligned_unit_type, aggregation, topk, w_colnorm, w_rownorm, w_coverage)
      -> (words, start_times, end_times, matrix, scores)

`model` is a whisper-char-alignment_amd WhisperAMD engine handle, `tokenizer` a tokenizer.Tokenizer.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .audio import HOP_LENGTH, SAMPLE_RATE, TOKENS_PER_SECOND  # noqa: F401  (re-exported like the reference)
from .retokenize import split_tokens_on_spaces

_pf = C.POINTER(C.c_float)
_pi = C.POINTER(C.c_int32)


def _engine_for(device):
    """An engine on `device` for the ops that need no weights (filter_attention / force_align / dtw)."""
    from .engine import default_engine
    return default_engine(device.index if device.index is not None else 0)


def _as_cuda_f32(t):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("the alignment engine needs an AMD GPU; there is no CPU fallback")
        t = t.cuda()
    return t.to(torch.float32).contiguous()


def get_attentions(mel, tokens, model, tokenizer, max_frames, medfilt_width=7, qk_scale=1.0):
    """Teacher-forced forward with every cross-attention head's QK logits captured, sliced to
    `max_frames`, median filtered, scaled and softmaxed (timing.py:45-67).
    mel (n_mels, 3000) f32, tokens 1-D int64 -> weights (L, H, n, F) f32 on the GPU, logits (n, V) f32."""
    max_frames = int(max_frames)
    weights, logits = model.get_attentions(mel.unsqueeze(0), tokens.unsqueeze(0), [max_frames], medfilt_width, qk_scale)
    return weights[0], logits[0]


def filter_attention(attns, topk=20, w_colnorm=1, w_rownorm=1, w_coverage=0):
    """attns (layers, heads, tokens, frames). Returns the top-k heads as a list of (1, T, F) tensors in
    ascending score order and the sorted (score, (l, h), name) tuples (timing.py:13-43)."""
    attns = _as_cuda_f32(attns)
    eng = _engine_for(attns.device)
    L, H, n, F = attns.shape
    keff = max(0, min(int(topk), L * H))
    scores = np.zeros(L * H, dtype=np.float32)
    idx = np.zeros(max(keff, 1), dtype=np.int32)
    ssc = np.zeros(max(keff, 1), dtype=np.float32)
    eng._bind_stream()
    if keff > 0:
        _lib.check(eng._lib.wca_filter_attention(eng._h, C.c_void_p(attns.data_ptr()), L, H, n, F, keff, float(w_colnorm),
                                                 float(w_rownorm), float(w_coverage), scores.ctypes.data_as(_pf),
                                                 idx.ctypes.data_as(_pi), ssc.ctypes.data_as(_pf)))
    scores_sorted = [(float(ssc[i]), (int(idx[i]) // H, int(idx[i]) % H), "sample_layer%d_head%d" % (idx[i] // H, idx[i] % H))
                     for i in range(keff)]
    selected = [attns[l, h].unsqueeze(0) for _, (l, h), _ in scores_sorted]
    return selected, scores_sorted


def _dtw_host(eng, matrix):
    """whisper.timing.dtw(-matrix) for a host matrix through the HIP kernel."""
    m = np.ascontiguousarray(matrix, dtype=np.float32)
    N, M = m.shape
    ti = np.zeros(N + M, dtype=np.int32)
    tj = np.zeros(N + M, dtype=np.int32)
    n = C.c_int32(0)
    eng._bind_stream()
    _lib.check(eng._lib.wca_dtw(eng._h, m.ctypes.data_as(_pf), N, M, ti.ctypes.data_as(_pi), tj.ctypes.data_as(_pi), C.byref(n)))
    return ti[:n.value].astype(np.int64), tj[:n.value].astype(np.int64)


def dtw(x):
    """whisper.timing.dtw drop-in: x is the ALREADY NEGATED cost matrix, as in `dtw(-matrix)`."""
    x = torch.as_tensor(x)
    dev = x.device if x.is_cuda else torch.device("cuda:0")
    return _dtw_host(_engine_for(dev), (-x).float().cpu().numpy())


def median_filter(x, filter_width):
    """whisper.timing.median_filter drop-in (reflect padding, last axis)."""
    x = _as_cuda_f32(x)
    if x.shape[-1] <= filter_width // 2:
        return x
    assert filter_width > 0 and filter_width % 2 == 1, "`filter_width` should be an odd number"
    eng = _engine_for(x.device)
    out = torch.empty_like(x)
    F = x.shape[-1]
    eng._bind_stream()
    _lib.check(eng._lib.wca_median_filter(eng._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel() // F, F,
                                          int(filter_width)))
    return out


def force_align(ws, tokens, tokenizer, aligned_unit_type="subword", aggregation="mean", topk=-1, w_colnorm=1.0,
                w_rownorm=1.0, w_coverage=0.0):
    """ws (layers, heads, tokens, frames) attention weights -> (words, start_times, end_times, matrix, scores)
    exactly as timing.py:69-114: aggregate heads, drop the sot rows and the last row, DTW on the negated
    matrix, merge token jumps into word start/end times. Returns [[], [], [], [], None] when the text has
    at most one word (timing.py:106-107)."""
    sot_len = len(tokenizer.sot_sequence)
    scores = None
    if aggregation == "grad_norm":  # passthrough branch (timing.py:99-100): ws is already a (tokens, frames) matrix
        matrix = torch.as_tensor(ws)[sot_len:-1].float().cpu()
        dev = ws.device if isinstance(ws, torch.Tensor) and ws.is_cuda else torch.device("cuda:0")
        text_indices, time_indices = _dtw_host(_engine_for(dev), matrix.numpy())
    else:
        if aggregation == "topk":
            assert topk > 0
        elif aggregation != "mean":
            raise ValueError("aggregation must be 'mean', 'topk' or 'grad_norm'")
        ws = _as_cuda_f32(ws)
        eng = _engine_for(ws.device)
        L, H, n, F = ws.shape
        N = n - sot_len - 1
        if N < 1:
            raise ValueError("ws has %d token rows; nothing is left after the [%d:-1] slice" % (n, sot_len))
        opts = eng.make_opts(aggregation=aggregation, topk=topk, w_colnorm=w_colnorm, w_rownorm=w_rownorm,
                             w_coverage=w_coverage, sot_len=sot_len)
        mat = np.zeros((N, F), dtype=np.float32)
        ti = np.zeros(N + F, dtype=np.int32)
        tj = np.zeros(N + F, dtype=np.int32)
        plen = C.c_int32(0)
        keff = max(1, min(int(topk), L * H)) if aggregation == "topk" else 1
        sel = np.zeros(keff, dtype=np.int32)
        ssc = np.zeros(keff, dtype=np.float32)
        eng._bind_stream()
        _lib.check(eng._lib.wca_force_align(eng._h, C.c_void_p(ws.data_ptr()), L, H, n, F, C.byref(opts), mat.ctypes.data_as(_pf),
                                            ti.ctypes.data_as(_pi), tj.ctypes.data_as(_pi), C.byref(plen), sel.ctypes.data_as(_pi),
                                            ssc.ctypes.data_as(_pf)))
        matrix = torch.from_numpy(mat)
        text_indices = ti[:plen.value].astype(np.int64)
        time_indices = tj[:plen.value].astype(np.int64)
        if aggregation == "topk":
            scores = [(float(ssc[i]), (int(sel[i]) // H, int(sel[i]) % H), "sample_layer%d_head%d" % (sel[i] // H, sel[i] % H))
                      for i in range(min(int(topk), L * H))]

    words, word_tokens = split_tokens_on_spaces(list(tokens) + [tokenizer.eot], tokenizer, aligned_unit_type)
    if len(word_tokens) <= 1:
        return [[], [], [], [], None]
    word_boundaries = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    jump_times = time_indices[jumps] / TOKENS_PER_SECOND
    start_times = jump_times[word_boundaries[:-1]]
    end_times = jump_times[word_boundaries[1:]]
    return words, start_times, end_times, matrix, scores


def attention_weights(qk, max_frames, medfilt_width=7, qk_scale=1.0):
    """The post-capture half of get_attentions (timing.py:63-66) on given logits: qk (L, H, n, S) f32 (the hooked
    cross-attention logits, concatenated over layers) -> [..., :max_frames] -> median filter -> * qk_scale -> softmax."""
    qk = _as_cuda_f32(qk)
    eng = _engine_for(qk.device)
    L, H, n, S = qk.shape
    out = torch.empty(L, H, n, int(max_frames), device=qk.device, dtype=torch.float32)
    eng._bind_stream()
    _lib.check(eng._lib.wca_attention_weights(eng._h, C.c_void_p(qk.data_ptr()), L, H, n, S, int(max_frames), int(medfilt_width),
                                              float(qk_scale), C.c_void_p(out.data_ptr())))
    return out


def _default_alignment_from_weights(eng, weights, heads, text_tokens, tokenizer):
    """timing.py:159-186 on filtered + softmaxed maps `weights` (L, H, n, F): returns the reference's 5-tuple."""
    sot_len = len(tokenizer.sot_sequence)
    L, H, n, F = weights.shape
    hd = np.asarray([l * H + h for l, h in heads], dtype=np.int32)
    N = n - sot_len - 1
    norm = torch.empty(len(hd), n, F, device=weights.device, dtype=torch.float32)
    ti = np.zeros(N + F, dtype=np.int32)
    tj = np.zeros(N + F, dtype=np.int32)
    plen = C.c_int32(0)
    eng._bind_stream()
    _lib.check(eng._lib.wca_default_find_alignment(eng._h, C.c_void_p(weights.data_ptr()), L, H, n, F, hd.ctypes.data_as(_pi), len(hd),
                                                   sot_len, C.c_void_p(norm.data_ptr()), None, ti.ctypes.data_as(_pi),
                                                   tj.ctypes.data_as(_pi), C.byref(plen)))
    text_indices = ti[:plen.value].astype(np.int64)
    time_indices = tj[:plen.value].astype(np.int64)
    words, word_tokens = tokenizer.split_to_word_tokens(list(text_tokens) + [tokenizer.eot])
    if len(word_tokens) <= 1:
        return [[], [], [], [], None]
    word_boundaries = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    jumps = np.pad(np.diff(text_indices), (1, 0), constant_values=1).astype(bool)
    jump_times = time_indices[jumps] / TOKENS_PER_SECOND
    return words, jump_times[word_boundaries[:-1]], jump_times[word_boundaries[1:]], norm, None


def default_find_alignment(model, tokenizer, text_tokens, mel, max_frames, *, medfilt_width=7, qk_scale=1.0):
    """openai-whisper's own aligner as restated by the reference (timing.py:116-186, `--default_whisper_timing`):
    the cross-attention maps of `model.alignment_heads` (row-major (layer, head) order, like `.indices().T` of the
    reference's sparse mask) are median filtered, softmaxed, normalised per head and frame over the token axis
    ((w - mean) / std, population std), averaged, sliced [sot:-1] and aligned with DTW; words come from
    tokenizer.split_to_word_tokens. Returns (words, start_times, end_times, weights, None) where `weights` are the
    NORMALISED maps of the alignment heads, (n_heads, n, F) on the GPU, exactly the reference's 4th value."""
    tokens = torch.tensor([*tokenizer.sot_sequence, tokenizer.no_timestamps, *text_tokens, tokenizer.eot]).to(model.device)
    weights, _logits = get_attentions(mel, tokens, model, tokenizer, max_frames, medfilt_width, qk_scale)
    return _default_alignment_from_weights(model, weights.contiguous(), model.alignment_heads, text_tokens, tokenizer)


def words_from_jump_frames(jump_frames, tokens, tokenizer, aligned_unit_type="char", want_words=True):
    """Host tail of the fused wca_align_batch path: `jump_frames[i]` is the frame at which the DTW path
    enters text row i (= time_indices[jumps], timing.py:110-111); returns (words, start_times, end_times)
    with the same meaning as force_align's first three outputs."""
    from .retokenize import char_word_starts
    toks = list(tokens) + [tokenizer.eot]
    starts = char_word_starts(toks, tokenizer) if aligned_unit_type == "char" else None
    if starts is None:
        words, word_tokens = split_tokens_on_spaces(toks, tokenizer, aligned_unit_type)
        if len(word_tokens) <= 1:
            return [], np.zeros(0), np.zeros(0)
        word_boundaries = np.pad(np.cumsum([len(t) for t in word_tokens[:-1]]), (1, 0))
    else:
        if len(starts) <= 1:
            return [], np.zeros(0), np.zeros(0)
        word_boundaries = starts
        words = None
        if want_words:
            ends = list(starts[1:]) + [len(toks)]
            words = [tokenizer.decode_with_timestamps(toks[a:b]) for a, b in zip(starts, ends)]
    jump_times = np.asarray(jump_frames[:len(toks)], dtype=np.int64) / TOKENS_PER_SECOND
    return words, jump_times[word_boundaries[:-1]], jump_times[word_boundaries[1:]]
