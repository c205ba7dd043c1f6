"""Greedy ASR pre-pass: `whisper.decode(model, mel, options)` as the reference calls it (infer_ali.py:40,60-61,
probe_oracle.py:37,59-60, README.md:107-108) on the MI355X engine (C ABI wca_greedy_decode).

Upstream openai-whisper `decoding.py` is an absent third-party dependency; its published algorithm is restated here
(host side: which tokens the filters suppress) and in csrc/decode.hip (the per-step filters and GreedyDecoder.update).
Only what DecodingOptions(language="en") exercises is built: a given language (no detection), temperature 0 without
beam search (greedy), no prompt / prefix. Anything else raises NotImplementedError instead of silently differing.
"""
import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Union

import numpy as np
import torch

from . import _lib
from .tokenizer import get_tokenizer

CHUNK_LENGTH = 30


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Union[str, List[int]]] = None
    prefix: Optional[Union[str, List[int]]] = None
    suppress_tokens: Optional[Union[str, tuple]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True
    vocab_path: Optional[str] = None  # engine-specific: local tiktoken file for the tokenizer


@dataclass(frozen=True)
class DecodingResult:
    language: str
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = np.nan
    no_speech_prob: float = np.nan
    temperature: float = np.nan
    compression_ratio: float = np.nan


def compression_ratio(text):
    text_bytes = text.encode("utf-8")
    return len(text_bytes) / len(zlib.compress(text_bytes))


def suppress_token_ids(tokenizer, options):
    """DecodingTask._get_suppress_tokens (upstream, restated): "-1" expands to the non-speech tokens; the task /
    sot / prev / lm / no-speech specials are always suppressed."""
    suppress = options.suppress_tokens
    if isinstance(suppress, str):
        suppress = [int(t) for t in suppress.split(",")]
    suppress = list(suppress) if suppress is not None else []
    if -1 in suppress:
        suppress = [t for t in suppress if t >= 0]
        suppress.extend(tokenizer.non_speech_tokens)
    elif len(suppress) == 0:
        suppress = []
    suppress.extend([tokenizer.transcribe, tokenizer.translate, tokenizer.sot, tokenizer.sot_prev, tokenizer.sot_lm])
    if tokenizer.no_speech is not None:
        suppress.append(tokenizer.no_speech)
    return tuple(sorted(set(suppress)))


def filter_masks(tokenizer, options, n_vocab):
    """(suppress_mask, blank_mask) byte arrays [n_vocab] for wca_greedy_decode."""
    sup = np.zeros(n_vocab, dtype=np.uint8)
    if options.suppress_tokens:
        ids = [t for t in suppress_token_ids(tokenizer, options) if t < n_vocab]
        sup[ids] = 1
    if not options.without_timestamps and tokenizer.no_timestamps is not None and tokenizer.no_timestamps < n_vocab:
        sup[tokenizer.no_timestamps] = 1  # ApplyTimestampRules suppresses <|notimestamps|>
    blank = None
    if options.suppress_blank:
        blank = np.zeros(n_vocab, dtype=np.uint8)
        blank[[t for t in tokenizer.encode(" ") + [tokenizer.eot] if t < n_vocab]] = 1
    return sup, blank


def _check_supported(options):
    if options.temperature != 0.0 or options.beam_size is not None or options.best_of is not None or options.patience is not None:
        raise NotImplementedError("only greedy decoding (temperature 0, no beam search / best_of) is built; the reference "
                                  "uses DecodingOptions(language='en') (infer_ali.py:40)")
    if options.prompt is not None or options.prefix is not None:
        raise NotImplementedError("prompt / prefix are not supported")
    if options.language is None:
        raise NotImplementedError("language detection is not built: pass DecodingOptions(language=...) as infer_ali.py:40 does")


@torch.no_grad()
def decode(model, mel, options=DecodingOptions(), pcm=None, n_samples=None, encoded_batch=None):
    """whisper.decode. mel: [n_mels, 3000] or [B, n_mels, 3000] f32 cuda tensor (or None with pcm [B, stride] f32 cuda +
    n_samples, the log-mel then runs on the device; or neither with encoded_batch=B: decode the state queued by
    model.encode_batch). Returns DecodingResult or a list of them."""
    _check_supported(options)
    single = mel is not None and mel.ndim == 2
    if single:
        mel = mel.unsqueeze(0)
    B = mel.shape[0] if mel is not None else (pcm.shape[0] if pcm is not None else int(encoded_batch))
    dims = model.dims
    tokenizer = get_tokenizer(model.is_multilingual, language=options.language, task=options.task, vocab_path=options.vocab_path)
    n_ctx = dims.n_text_ctx
    sample_len = options.sample_len or n_ctx // 2
    initial = list(tokenizer.sot_sequence_including_notimestamps if options.without_timestamps else tokenizer.sot_sequence)
    sup, blank = filter_masks(tokenizer, options, dims.n_vocab)
    max_init = -1
    if not options.without_timestamps and options.max_initial_timestamp is not None:
        precision = CHUNK_LENGTH / dims.n_audio_ctx
        max_init = round(options.max_initial_timestamp / precision)
    tokens, n_tokens, sum_logprobs = model.greedy_decode(
        mel, pcm, n_samples, initial, sup, blank, sample_len=sample_len, eot=tokenizer.eot, timestamp_begin=tokenizer.timestamp_begin,
        apply_timestamp_rules=not options.without_timestamps, max_initial_timestamp_index=max_init, batch=B,
        no_speech=tokenizer.no_speech if tokenizer.no_speech is not None else -1)
    no_speech_probs = model.last_no_speech_prob
    results = []
    for b in range(B):
        toks = [int(t) for t in tokens[b, len(initial):n_tokens[b]]]
        text = tokenizer.decode(toks).strip()
        results.append(DecodingResult(language=options.language, tokens=toks, text=text,
                                      avg_logprob=float(sum_logprobs[b]) / (len(toks) + 1), no_speech_prob=float(no_speech_probs[b]),
                                      temperature=options.temperature,
                                      compression_ratio=compression_ratio(text) if text else np.nan))
    return results[0] if single else results
