"""`import dataset` drop-in (dataset.py:14-122)."""
from _pkg import sub as _sub

_d = _sub("dataset")
TIMIT, LibriSpeech, AMI, Collate = _d.TIMIT, _d.LibriSpeech, _d.AMI, _d.Collate
