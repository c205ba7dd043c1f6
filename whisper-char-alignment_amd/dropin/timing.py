"""`import timing` drop-in for the reference's timing.py (timing.py:13-186): the same four functions, running on the
MI355X engine. Put this directory on sys.path instead of the reference checkout."""
from _pkg import sub as _sub

_t = _sub("timing")
filter_attention = _t.filter_attention
get_attentions = _t.get_attentions
force_align = _t.force_align
default_find_alignment = _t.default_find_alignment
median_filter = _t.median_filter
dtw = _t.dtw
HOP_LENGTH, SAMPLE_RATE, TOKENS_PER_SECOND = _t.HOP_LENGTH, _t.SAMPLE_RATE, _t.TOKENS_PER_SECOND
