"""ORACLE -- test infrastructure only (imported by tests/; never by the product path).

CPU restatement of the greedy ASR pre-pass the reference runs before alignment:
    whisper.decode(model, mels, whisper.DecodingOptions(language="en"))     /root/reference/infer_ali.py:40,60-61
                                                                            /root/reference/probe_oracle.py:37,59-60
The arithmetic lives in the third-party dependency openai-whisper (`whisper/decoding.py`; unpinned in the reference,
>= v20240930), which is absent from /root/reference and from this container. PARITY UNPINNED:
the published algorithm is restated below (SuppressBlank, SuppressTokens, ApplyTimestampRules, GreedyDecoder.update,
DecodingTask._main_loop) and checked against hand-derived known answers and, for the timestamp rules, against the
independent port in HuggingFace transformers (WhisperTimeStampLogitsProcessor) in tests/test_oracle.py.
"""
import numpy as np
import torch
import torch.nn.functional as F


class SuppressBlank:
    def __init__(self, blank_tokens, eot, sample_begin):
        self.ids = list(blank_tokens) + [eot]
        self.sample_begin = sample_begin

    def apply(self, logits, tokens):
        if tokens.shape[1] == self.sample_begin:
            logits[:, self.ids] = -np.inf


class SuppressTokens:
    def __init__(self, suppress_tokens):
        self.suppress_tokens = list(suppress_tokens)

    def apply(self, logits, tokens):
        logits[:, self.suppress_tokens] = -np.inf


class ApplyTimestampRules:
    def __init__(self, timestamp_begin, eot, no_timestamps, sample_begin, max_initial_timestamp_index):
        self.timestamp_begin = timestamp_begin
        self.eot = eot
        self.no_timestamps = no_timestamps
        self.sample_begin = sample_begin
        self.max_initial_timestamp_index = max_initial_timestamp_index

    def apply(self, logits, tokens):
        if self.no_timestamps is not None:
            logits[:, self.no_timestamps] = -np.inf
        # timestamps have to appear in pairs, except directly before EOT
        for k in range(tokens.shape[0]):
            sampled_tokens = tokens[k, self.sample_begin:]
            seq = [t for t in sampled_tokens.tolist()]
            last_was_timestamp = len(seq) >= 1 and seq[-1] >= self.timestamp_begin
            penultimate_was_timestamp = len(seq) < 2 or seq[-2] >= self.timestamp_begin
            if last_was_timestamp:
                if penultimate_was_timestamp:  # has to be non-timestamp
                    logits[k, self.timestamp_begin:] = -np.inf
                else:  # cannot be normal text tokens
                    logits[k, :self.eot] = -np.inf
            timestamps = sampled_tokens[sampled_tokens.ge(self.timestamp_begin)]
            if timestamps.numel() > 0:
                # timestamps shouldn't decrease; each segment has a nonzero length
                if last_was_timestamp and not penultimate_was_timestamp:
                    timestamp_last = timestamps[-1]
                else:
                    timestamp_last = timestamps[-1] + 1
                logits[k, self.timestamp_begin:timestamp_last] = -np.inf
        if tokens.shape[1] == self.sample_begin:
            # suppress generating non-timestamp tokens at the beginning
            logits[:, :self.timestamp_begin] = -np.inf
            if self.max_initial_timestamp_index is not None:
                last_allowed = self.timestamp_begin + self.max_initial_timestamp_index
                logits[:, last_allowed + 1:] = -np.inf
        # if the probability mass over timestamps is above any other token, sample a timestamp
        logprobs = F.log_softmax(logits.float(), dim=-1)
        for k in range(tokens.shape[0]):
            timestamp_logprob = logprobs[k, self.timestamp_begin:].logsumexp(dim=-1)
            max_text_token_logprob = logprobs[k, :self.timestamp_begin].max()
            if timestamp_logprob > max_text_token_logprob:
                logits[k, :self.timestamp_begin] = -np.inf


def greedy_update(tokens, logits, sum_logprobs, eot):
    """GreedyDecoder.update at temperature 0."""
    next_tokens = logits.argmax(dim=-1)
    logprobs = F.log_softmax(logits.float(), dim=-1)
    current_logprobs = logprobs[torch.arange(logprobs.shape[0]), next_tokens]
    sum_logprobs += current_logprobs * (tokens[:, -1] != eot)
    next_tokens[tokens[:, -1] == eot] = eot
    tokens = torch.cat([tokens, next_tokens[:, None]], dim=-1)
    completed = bool((tokens[:, -1] == eot).all())
    return tokens, completed


def make_filters(initial_len, eot, timestamp_begin, no_timestamps, suppress_tokens, blank_tokens, apply_timestamp_rules=True,
                 max_initial_timestamp_index=50):
    filters = []
    if blank_tokens is not None:
        filters.append(SuppressBlank(blank_tokens, eot, initial_len))
    if suppress_tokens:
        filters.append(SuppressTokens(suppress_tokens))
    if apply_timestamp_rules:
        filters.append(ApplyTimestampRules(timestamp_begin, eot, no_timestamps, initial_len, max_initial_timestamp_index))
    return filters


def select_step(logits, tokens, sum_logprobs, filters, eot):
    """One loop iteration after the forward: filters (in place on a copy) + greedy update. Returns (tokens, completed, filtered)."""
    logits = logits.clone().float()
    for f in filters:
        f.apply(logits, tokens)
    tokens, completed = greedy_update(tokens, logits, sum_logprobs, eot)
    return tokens, completed, logits


def greedy_decode(model_ref, mel, initial_tokens, filters, eot, sample_len, n_ctx=448, forced=None):
    """DecodingTask._main_loop with the fp32 oracle model (model_ref: oracle.whisper_ref.WhisperRef; no KV cache: the
    whole prefix is re-run each step, fine at test sizes). mel [B, n_mels, 3000].
    forced: optional [B, T] token tensor -- teacher-force those tokens instead of the oracle's own argmax (used to score
    another implementation's choices step by step); the per-step filtered logits are returned either way."""
    xa = model_ref.encoder(mel)
    B = mel.shape[0]
    tokens = torch.tensor([list(initial_tokens)] * B, dtype=torch.long)
    sum_logprobs = torch.zeros(B)
    per_step = []
    for i in range(sample_len):
        logits = model_ref.decoder(tokens, xa)[0][:, -1]
        new_tokens, completed, filtered = select_step(logits, tokens, sum_logprobs, filters, eot)
        per_step.append(filtered)
        if forced is not None:
            if tokens.shape[1] >= forced.shape[1]:
                break
            nxt = forced[:, tokens.shape[1]].long()
            new_tokens = torch.cat([tokens, nxt[:, None]], dim=-1)
            completed = bool((nxt == eot).all())
        tokens = new_tokens
        if completed or tokens.shape[-1] > n_ctx:
            break
    return tokens, sum_logprobs, per_step
