"""`import retokenize` drop-in (retokenize.py:5-50)."""
from _pkg import sub as _sub

_r = _sub("retokenize")
encode = _r.encode
split_tokens_on_spaces = _r.split_tokens_on_spaces
remove_punctuation = _r.remove_punctuation
