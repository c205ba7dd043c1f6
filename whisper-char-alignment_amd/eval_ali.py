#!/usr/bin/env python3
"""Offline re-scoring of a `*-predictions.pkl` written by infer_ali.py --save_prediction (same flags and
printed summary as the reference's eval_ali.py:9-65): strict word-matched precision / recall / F1 / R-value."""
import argparse
import os
import sys

if __package__ in (None, ""):
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _pkg = importlib.import_module("whisper-char-alignment_amd")
    __package__ = _pkg.__name__

from .metrics import eval_n1_strict, get_seg_metrics  # noqa: E402
from .retokenize import remove_punctuation  # noqa: E402


def run_eval(args):
    import joblib
    preds = joblib.load(args.pred)
    corrects = total_preds = total_gts = 0
    for key in sorted(preds):
        p = preds[key]
        if not p:
            continue
        gt_words = [remove_punctuation(w) for w in p["texts"]]
        hyp_words = [remove_punctuation(w) for w in p["predwords"]]
        print("gt: %s" % (p["ends"],))
        print("pred: %s" % (p["ends_hat"],))
        tp, fp, fn = eval_n1_strict(p["ends"], p["ends_hat"], gt_words, hyp_words, tolerance=args.tolerance)
        corrects += tp
        total_gts += tp + fn
        total_preds += tp + fp
    precision, recall, f1, r_value, _ = get_seg_metrics(corrects, corrects, total_preds, total_gts)
    print("-----------------")
    print("precision: %.2f" % precision)
    print("recall: %.2f" % recall)
    print("f1: %.2f" % f1)
    print("r value: %.2f" % r_value)
    print("-----------------")
    return dict(precision=precision, recall=recall, f1=f1, r_value=r_value)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="eval alignment")
    p.add_argument("--pred", type=str, required=True)  # /path/to/*-predictions.pkl
    p.add_argument("--tolerance", type=float, default=0.05)
    return p.parse_args(argv)


if __name__ == "__main__":
    run_eval(parse_args())
