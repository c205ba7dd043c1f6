"""`import whisper` stand-in covering exactly what the reference touches (timing.py:7-10, infer_ali.py:18-20,36-41,60,
dataset.py:4,47-48, README.md:93-108): load_model / decode / DecodingOptions / pad_or_trim / log_mel_spectrogram and the
`audio`, `model`, `timing`, `tokenizer` sub-modules. Nothing is downloaded: `load_model(name)` reads a LOCAL checkpoint,
`download_root/<name>.pt` or $WCA_WEIGHTS_DIR/<name>.pt (openai format)."""
import contextlib as _contextlib
import os as _os
import sys as _sys
import types as _types

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from _pkg import pkg as _pkg, sub as _sub  # noqa: E402

_audio, _decoding, _timing, _tok = _sub("audio"), _sub("decoding"), _sub("timing"), _sub("tokenizer")

pad_or_trim = _audio.pad_or_trim
log_mel_spectrogram = _audio.log_mel_spectrogram
DecodingOptions = _decoding.DecodingOptions
DecodingResult = _decoding.DecodingResult
decode = _decoding.decode


def load_model(name, device="cuda:0", download_root=None, in_memory=False, max_batch=8, precision=None):
    """whisper.load_model on the engine: the model runs the reference's fp32-equivalent forward (precision 'reference', the contract mode)
    unless precision='f16' (or $WCA_PRECISION=f16) asks for the fast mode."""
    root = download_root or _os.environ.get("WCA_WEIGHTS_DIR") or _os.path.join(_os.path.expanduser("~"), ".cache", "whisper")
    path = name if _os.path.isfile(name) else _os.path.join(root, name + ".pt")
    if not _os.path.isfile(path):
        raise RuntimeError("no local checkpoint %s: this engine never downloads by model name (set WCA_WEIGHTS_DIR or pass download_root)" % path)
    return _pkg.WhisperAMD.from_checkpoint(path, device=str(device), max_batch=max_batch, name=None if _os.path.isfile(name) else name,
                                           precision=precision or _os.environ.get("WCA_PRECISION", "reference"))


audio = _types.ModuleType("whisper.audio")
for _n in ("SAMPLE_RATE", "N_FFT", "HOP_LENGTH", "CHUNK_LENGTH", "N_SAMPLES", "N_FRAMES", "N_SAMPLES_PER_TOKEN", "FRAMES_PER_SECOND",
           "TOKENS_PER_SECOND", "pad_or_trim", "log_mel_spectrogram", "mel_filters", "load_audio"):
    setattr(audio, _n, getattr(_audio, _n))
model = _types.ModuleType("whisper.model")
model.disable_sdpa = _contextlib.nullcontext  # the engine always materialises qk for the hooked heads
model.ModelDimensions = _pkg.ModelDimensions
model.Whisper = _pkg.WhisperAMD
timing = _types.ModuleType("whisper.timing")
timing.median_filter, timing.dtw = _timing.median_filter, _timing.dtw
tokenizer = _types.ModuleType("whisper.tokenizer")
tokenizer.get_tokenizer, tokenizer.Tokenizer, tokenizer.LANGUAGES = _tok.get_tokenizer, _tok.Tokenizer, _tok.LANGUAGES
decoding = _types.ModuleType("whisper.decoding")
decoding.DecodingOptions, decoding.DecodingResult, decoding.decode = DecodingOptions, DecodingResult, decode
for _m in (audio, model, timing, tokenizer, decoding):
    _sys.modules[_m.__name__] = _m
