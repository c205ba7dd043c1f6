"""MI355X-native forced-alignment engine (drop-in for the hot path of whisper-char-alignment).

The directory name contains a hyphen, so import it with
    wca = importlib.import_module("whisper-char-alignment_amd")
or put `whisper-char-alignment_amd/dropin` on sys.path to get the reference's module names unchanged: `timing`,
`retokenize`, `metrics`, `dataset` and a `whisper` stand-in with exactly the attributes the reference touches
(load_model from a LOCAL checkpoint, decode, DecodingOptions, pad_or_trim, log_mel_spectrogram, whisper.audio /
.model.disable_sdpa / .timing / .tokenizer), so the reference's README snippet and drivers run on this engine as written.

Nothing here falls back to the CPU: libwca.so (hand-written HIP for gfx950) must be built.
"""
from . import _lib  # noqa: F401
from .engine import WhisperAMD, ModelDimensions, dims_for  # noqa: F401

__all__ = ["WhisperAMD", "ModelDimensions", "dims_for"]
