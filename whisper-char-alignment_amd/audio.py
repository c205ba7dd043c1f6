"""Audio front-end host side: constants, pad_or_trim, the Slaney mel filterbank table, audio file
readers (NIST SPHERE / RIFF WAV without torchaudio) and the `log_mel_spectrogram` drop-in whose
arithmetic runs in libwca.so (csrc/logmel.hip).

Reference call sites: dataset.py:31 (torchaudio.load), dataset.py:46-48 (pad_or_trim +
log_mel_spectrogram), whisper.audio constants used at timing.py:10,111 and infer_ali.py:179.
"""
import struct

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE       # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH           # 3000
N_SAMPLES_PER_TOKEN = HOP_LENGTH * 2         # 320
FRAMES_PER_SECOND = SAMPLE_RATE // HOP_LENGTH
TOKENS_PER_SECOND = SAMPLE_RATE // N_SAMPLES_PER_TOKEN  # 50


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filters(n_mels=80, sr=SAMPLE_RATE, n_fft=N_FFT):
    """Slaney-scale, Slaney-normalised triangular filterbank [n_mels, n_fft//2+1] f32 -- the table
    openai-whisper ships as assets/mel_filters.npz (librosa.filters.mel(sr=16000, n_fft=400, n_mels))."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_pts = np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2)
    mel_f = _mel_to_hz(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def pad_or_trim(array, length=N_SAMPLES, axis=-1):
    """whisper.pad_or_trim: trim or right-zero-pad along `axis` (numpy arrays or torch tensors)."""
    try:
        import torch
        if isinstance(array, torch.Tensor):
            if array.shape[axis] > length:
                array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
            if array.shape[axis] < length:
                pad = [(0, 0)] * array.ndim
                pad[axis] = (0, length - array.shape[axis])
                array = torch.nn.functional.pad(array, [p for sizes in pad[::-1] for p in sizes])
            return array
    except ImportError:
        pass
    array = np.asarray(array)
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        pad = [(0, 0)] * array.ndim
        pad[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad)
    return array


def log_mel_spectrogram(audio, n_mels=80, padding=0, device=None, model=None):
    """whisper.log_mel_spectrogram drop-in: audio f32 [n] or [B, n] (expected already pad_or_trim'ed to
    480000 like dataset.py:47; shorter input is treated as zero padded) -> [n_mels, 3000] f32 on the GPU."""
    import torch
    if model is None:
        from .engine import default_engine
        idx = 0
        if isinstance(audio, torch.Tensor) and audio.is_cuda and audio.device.index is not None:
            idx = audio.device.index
        model = default_engine(idx)
    eng = model
    if eng.dims.n_mels != n_mels:
        raise ValueError("engine was built for n_mels=%d, got %d" % (eng.dims.n_mels, n_mels))
    if not isinstance(audio, torch.Tensor):
        audio = torch.from_numpy(np.asarray(audio, dtype=np.float32))
    if padding > 0:
        audio = torch.nn.functional.pad(audio, (0, padding))
    return eng.log_mel(audio.to(eng.device))


# ------------------------------------------------------------------------------ file readers
def _read_sphere(buf):
    if buf[:7] != b"NIST_1A":
        raise ValueError("not a NIST SPHERE file")
    hdr_size = int(buf[8:16].split()[0])
    fields = {}
    for line in buf[16:hdr_size].decode("latin-1").split("\n"):
        parts = line.split()
        if len(parts) >= 3 and parts[0] != "end_head":
            fields[parts[0]] = parts[2]
        if parts and parts[0] == "end_head":
            break
    if fields.get("sample_coding", "pcm") not in ("pcm",):
        raise ValueError("unsupported SPHERE coding %r (only pcm)" % fields.get("sample_coding"))
    nbytes = int(fields.get("sample_n_bytes", 2))
    if nbytes != 2:
        raise ValueError("only 16-bit SPHERE supported")
    order = "<" if fields.get("sample_byte_format", "01") == "01" else ">"
    count = int(fields["sample_count"]) * int(fields.get("channel_count", 1))
    pcm = np.frombuffer(buf, dtype=order + "i2", count=count, offset=hdr_size)
    ch = int(fields.get("channel_count", 1))
    if ch > 1:
        pcm = pcm.reshape(-1, ch).T
    return pcm.astype(np.float32) / 32768.0, int(fields["sample_rate"])


def _read_riff(buf):
    if buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("malformed WAVE file")
    tag, ch, sr, _, _, bits = fmt
    if tag == 1 and bits == 16:
        pcm = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        pcm = np.frombuffer(data, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 3 and bits == 32:
        pcm = np.frombuffer(data, dtype="<f4").astype(np.float32)
    else:
        raise ValueError("unsupported WAVE format tag=%d bits=%d" % (tag, bits))
    if ch > 1:
        pcm = pcm.reshape(-1, ch).T
    return pcm, sr


def _read_flac(buf):
    """FLAC (LibriSpeech originals) through libwca.so's host-side decoder (csrc/flac.cpp; ctypes releases the GIL, so
    reader threads decode in parallel). Returns (f32 [n] or [channels, n] in [-1, 1), sample_rate)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    buf = bytes(buf)
    src = C.cast(C.c_char_p(buf), C.c_void_p)
    sr, ch, bits, total = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int64(0)
    if lib.wca_flac_info(src, len(buf), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(total)) != 0:
        raise ValueError("not a FLAC stream")
    n = C.c_int64(0)
    n_total = total.value
    if n_total <= 0 or n_total > len(buf) * 8192:
        # STREAMINFO without a sample count, or with one no stream of this size can hold (a 10-byte CONSTANT frame codes at most 65 536
        # samples): a corrupt or hostile header must not size the allocation below -- count first
        if lib.wca_flac_decode(src, len(buf), None, 0, C.byref(n)) != 0:
            raise ValueError("corrupt or unsupported FLAC stream")
        n_total = n.value
    out = np.empty((ch.value, max(n_total, 1)), dtype=np.float32)
    rc = lib.wca_flac_decode(src, len(buf), out.ctypes.data_as(C.c_void_p), out.shape[1], C.byref(n))
    if rc != 0:
        raise ValueError("corrupt or unsupported FLAC stream (wca_flac_decode returned %d)" % rc)
    out = out[:, :n.value]
    return (out[0] if ch.value == 1 else out), sr.value


def load_audio(path):
    """torchaudio.load stand-in for the formats the reference corpora use: returns (f32 array in
    [-1, 1) shaped [n] (mono) or [channels, n], sample_rate). TIMIT '.wav' files are NIST SPHERE, LibriSpeech is FLAC."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:7] == b"NIST_1A":
        return _read_sphere(buf)
    if buf[:4] == b"RIFF":
        return _read_riff(buf)
    if buf[:4] == b"fLaC":
        return _read_flac(buf)
    raise ValueError("%s: unknown audio container (SPHERE, RIFF/WAVE and FLAC supported)" % path)
