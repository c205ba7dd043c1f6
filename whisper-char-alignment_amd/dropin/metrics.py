"""`import metrics` drop-in (metrics.py:22-111)."""
from _pkg import sub as _sub

_m = _sub("metrics")
eval_n1 = _m.eval_n1
eval_n1_strict = _m.eval_n1_strict
get_seg_metrics = _m.get_seg_metrics
coverage_penalty = _m.coverage_penalty
