"""Boundary-matching metrics with the reference's names and semantics (metrics.py:22-86, 99-111).
Host-side Python; only `coverage_penalty` touches tensors (the fused GPU version lives in
csrc/postproc.hip and is what timing.filter_attention uses)."""
import string

import numpy as np


def eval_n1(y, yhat, tolerance=1):
    """Greedy in-order matching of two sorted boundary lists; returns (hits, hits) (metrics.py:22-43)."""
    if len(yhat) == 0:
        return 0, 0
    hits = 0
    a = b = 0
    while a < len(y) and b < len(yhat):
        if abs(y[a] - yhat[b]) <= tolerance:
            hits += 1
            a += 1
            b += 1
        elif y[a] < yhat[b]:
            a += 1
        elif y[a] > yhat[b]:
            b += 1
        else:  # NaN: neither ordered nor within tolerance -- cannot advance meaningfully
            break
    return hits, hits


def eval_n1_strict(y, y_hat, words, words_hat, tolerance=1):
    """A prediction counts only if an unused reference boundary has the same (lower-cased,
    punctuation-stripped) word AND lies within tolerance; returns (tp, fp, fn) (metrics.py:45-72)."""
    ref_words = [w.lower().strip(string.punctuation) for w in words]
    hyp_words = [w.lower().strip(string.punctuation) for w in words_hat]
    used = set()
    tp = 0
    for i in range(len(y_hat)):
        for j in range(len(y)):
            if j not in used and ref_words[j] == hyp_words[i] and abs(y[j] - y_hat[i]) <= tolerance:
                used.add(j)
                tp += 1
                break
    return tp, len(y_hat) - tp, len(y) - len(used)


def get_seg_metrics(correct_predict, correct_retrieve, total_predict, total_gold):
    """precision, recall, F1, R-value, over-segmentation (metrics.py:74-86), EPS = 1e-7."""
    eps = 1e-7
    precision = correct_predict / (total_predict + eps)
    recall = correct_retrieve / (total_gold + eps)
    f1 = 2 * (precision * recall) / (precision + recall + eps)
    over_seg = recall / (precision + eps) - 1
    r1 = np.sqrt((1 - recall) ** 2 + over_seg ** 2)
    r2 = (-over_seg + recall - 1) / np.sqrt(2)
    r_value = 1 - (abs(r1) + abs(r2)) / 2
    return precision, recall, f1, r_value, over_seg


def coverage_penalty(attn, threshold=0.5):
    """sum_f max(sum_t attn[t, f], threshold) - F * threshold for a (tokens, frames) tensor (metrics.py:99-111)."""
    import torch
    coverage = attn.sum(dim=0)
    return torch.clamp(coverage, min=threshold).sum(-1) - coverage.size(-1) * threshold
