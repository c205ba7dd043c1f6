"""Resolves the engine package (its directory name has a hyphen, so it is imported by path)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
pkg = importlib.import_module("whisper-char-alignment_amd")


def sub(name):
    return importlib.import_module("whisper-char-alignment_amd." + name)
