"""Model handle of the MI355X engine: the object callers pass as `model` to timing.get_attentions
(reference: `whisper.load_model(...)` at infer_ali.py:36 / README.md:93).

It exposes the attributes the reference's callers touch (`model.device`, `model.dims.n_text_layer`,
`model.is_multilingual`; infer_ali.py:41,46, timing.py:48) and owns one `wca_engine` (weights in HBM as
f16, activation arena, HIP stream). PyTorch-ROCm is used only to hold device buffers and to read
checkpoints; all arithmetic happens in libwca.so.
"""
import ctypes as C
from dataclasses import dataclass, asdict

import numpy as np
import torch

from . import _lib
from .audio import mel_filters

N_FRAMES = 3000
N_SAMPLES = 480000
MAX_FRAMES = 1500   # infer_ali.py:25
MAX_LENGTH = 448    # infer_ali.py:26


@dataclass
class ModelDimensions:
    """Same fields as whisper.model.ModelDimensions."""
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


# whisper model sizes
_SIZES = {
    "tiny": (384, 6, 4), "base": (512, 8, 6), "small": (768, 12, 12), "medium": (1024, 16, 24),
    "large": (1280, 20, 32), "large-v1": (1280, 20, 32), "large-v2": (1280, 20, 32), "large-v3": (1280, 20, 32),
}


def dims_for(name):
    """ModelDimensions for an official model name ('medium', 'tiny.en', 'large-v3', ...)."""
    base = name[:-3] if name.endswith(".en") else name
    if base not in _SIZES:
        raise ValueError("unknown whisper model %r" % name)
    d, h, l = _SIZES[base]
    multilingual = not name.endswith(".en")
    n_mels = 128 if base == "large-v3" else 80
    n_vocab = 51866 if base == "large-v3" else (51865 if multilingual else 51864)
    return ModelDimensions(n_mels, 1500, d, h, l, n_vocab, 448, d, h, l)


# openai-whisper's per-model alignment heads (`whisper._ALIGNMENT_HEADS`, applied by whisper.load_model(name) through
# set_alignment_heads): the (layer, head) pairs of the decoder cross-attention heads highly correlated with word timing.
# Published data of the upstream package, kept here in BOTH forms: the package's own base85 + gzip boolean masks
# (ALIGNMENT_HEADS_B85, decoded by decode_alignment_heads exactly like whisper.model.Whisper.set_alignment_heads:
# b85decode -> gzip -> bool[n_text_layer, n_text_head]) and the decoded index lists, row-major like
# `model.alignment_heads.indices().T` (timing.py:155). tests/test_host.py decodes every mask (the gzip CRC-32 validates each
# string byte for byte) and requires it to equal the list below.
ALIGNMENT_HEADS_B85 = {
    "tiny.en": b"ABzY8J1N>@0{>%R00Bk>$p{7v037`oCl~+#00",
    "tiny": b"ABzY8bu8Lr0{>%RKn9Fp%m@SkK7Kt=7ytkO",
    "base.en": b"ABzY8;40c<0{>%RzzG;p*o+Vo09|#PsxSZm00",
    "base": b"ABzY8KQ!870{>%RzyTQH3`Q^yNP!>##QT-<FaQ7m",
    "small.en": b"ABzY8>?_)10{>%RpeA61k&I|OI3I$65C{;;pbCHh0B{qLQ;+}v00",
    "small": b"ABzY8DmU8=0{>%Rpa?J`kvJ6qF(V^F86#Xh7JUGMK}P<N0000",
    "medium.en": b"ABzY8usPae0{>%R7<zz_OvQ{)4kMa0BMw6u5rT}kRKX;$NfYBv00*Hl@qhsU00",
    "medium": b"ABzY8B0Jh+0{>%R7}kK1fFL7w6%<-Pf*t^=N)Qr&0RR9",
    "large-v1": b"ABzY8r9j$a0{>%R7#4sLmoOs{s)o3~84-RPdcFk!JR<kSfC2yj",
    "large-v2": b"ABzY8zd+h!0{>%R7=D0pU<_bnWW*tkYAhobTNnu$jnkEkXqp)j;w1Tzk)UH3X%SZd&fFZ2fC2yj",
    "large-v3": b"ABzY8gWO1E0{>%R7(9S+Kn!D~%ngiGaR?*L!iJG9p-nab0JQ=-{D1-g00",
}


def decode_alignment_heads(dump, n_text_layer, n_text_head):
    """base85 + gzip boolean mask -> [(layer, head), ...] in row-major order (whisper.model.Whisper.set_alignment_heads)."""
    import base64
    import gzip
    mask = np.frombuffer(gzip.decompress(base64.b85decode(dump)), dtype=bool).reshape(n_text_layer, n_text_head)
    return [(int(l), int(h)) for l, h in np.argwhere(mask)]


ALIGNMENT_HEADS = {
    "tiny.en": [(1, 0), (2, 0), (2, 5), (3, 0), (3, 1), (3, 2), (3, 3), (3, 4)],
    "tiny": [(2, 2), (3, 0), (3, 2), (3, 3), (3, 4), (3, 5)],
    "base.en": [(3, 3), (4, 7), (5, 1), (5, 5), (5, 7)],
    "base": [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)],
    "small.en": [(6, 6), (7, 0), (7, 3), (7, 8), (8, 2), (8, 5), (8, 7), (9, 0), (9, 4), (9, 8), (9, 10), (10, 0), (10, 1), (10, 2),
                 (10, 3), (10, 6), (10, 11), (11, 2), (11, 4)],
    "small": [(5, 3), (5, 9), (8, 0), (8, 4), (8, 7), (8, 8), (9, 0), (9, 7), (9, 9), (10, 5)],
    "medium.en": [(11, 4), (14, 1), (14, 12), (14, 14), (15, 4), (16, 0), (16, 4), (16, 9), (17, 12), (17, 14), (18, 7), (18, 10),
                  (18, 15), (20, 0), (20, 3), (20, 9), (20, 14), (21, 12)],
    "medium": [(13, 15), (15, 4), (15, 15), (16, 1), (20, 0), (23, 4)],
    "large-v1": [(9, 19), (11, 2), (11, 4), (11, 17), (22, 7), (22, 11), (22, 17), (23, 2), (23, 15)],
    "large-v2": [(10, 12), (13, 17), (16, 11), (16, 12), (16, 13), (17, 15), (17, 16), (18, 4), (18, 11), (18, 19), (19, 11), (21, 2),
                 (21, 3), (22, 3), (22, 9), (22, 12), (23, 5), (23, 7), (23, 13), (25, 5), (26, 1), (26, 12), (27, 15)],
    "large-v3": [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)],
}
ALIGNMENT_HEADS["large"] = ALIGNMENT_HEADS["large-v3"]


def model_name_from_dims(dims):
    """The official model name a set of dimensions identifies, or None when it is ambiguous (large-v1 / large-v2 share
    every dimension) or not an official size."""
    for base, (d, h, l) in _SIZES.items():
        if (dims.n_text_state, dims.n_text_head, dims.n_text_layer) != (d, h, l) or base in ("large", "large-v1", "large-v2", "large-v3"):
            continue
        return base if dims.n_vocab >= 51865 else base + ".en"
    if (dims.n_text_state, dims.n_text_head, dims.n_text_layer) == _SIZES["large"] and dims.n_mels == 128:
        return "large-v3"
    return None


_registry = {}   # device index -> weakref of the most recently constructed engine
_utility = {}    # device index -> weight-less engine created on demand for filter_attention / force_align / dtw


def default_engine(device_index=0, need_weights=False):
    """Most recent live engine on the device; if none exists a tiny weight-less utility engine is
    created (enough for the ops that take no model argument in the reference API)."""
    ref = _registry.get(device_index)
    eng = ref() if ref is not None else None
    if eng is not None and (eng._finalized or not need_weights):
        return eng
    if need_weights:
        raise RuntimeError("no WhisperAMD engine with weights exists on cuda:%d" % device_index)
    if device_index not in _utility:
        _utility[device_index] = WhisperAMD(ModelDimensions(80, 1500, 128, 2, 1, 51865, 448, 128, 2, 1),
                                            device="cuda:%d" % device_index, max_batch=1, _register=False, precision="f16")   # (runs no forward)
    return _utility[device_index]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class WhisperAMD:
    def __init__(self, dims, device="cuda:0", max_batch=8, _register=True, precision="reference"):
        """precision: 'reference' (default since round 5, like wca_engine_create): the CONTRACT mode -- the reference's fp32 forward
        (/root/reference/timing.py:58) to fp32 summation noise; 'f16': the fast single-f16-operand mode (98.5 % of the word boundaries
        within one frame on the parity legs) as an explicit opt-in. set_precision() switches later."""
        if precision not in ("reference", "split", "f16"):
            raise ValueError("precision must be 'reference' (= 'split') or 'f16'")
        if not torch.cuda.is_available():
            raise RuntimeError("WhisperAMD needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self._lib = _lib.load()
        self.dims = dims
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        self._h = C.c_void_p(0)
        cd = _lib.ModelDims(**asdict(dims))
        index = self.device.index if self.device.index is not None else 0
        _lib.check(self._lib.wca_engine_create_ex(C.byref(cd), index, self.max_batch, 0 if precision == "f16" else 1, C.byref(self._h)))
        self._stream_bound = None
        self._finalized = False
        # whisper.model.Whisper default: every head of the upper half of the decoder layers; whisper.load_model(name)
        # replaces it with the model's curated heads (ALIGNMENT_HEADS) -- from_checkpoint / use_official_alignment_heads
        self.alignment_heads = [(l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer) for h in range(dims.n_text_head)]
        self.alignment_heads_source = "default (upper half of the decoder layers)"
        filt = np.ascontiguousarray(mel_filters(dims.n_mels), dtype=np.float32)
        self._load_one("mel_filters", filt)
        if _register:
            import weakref
            _registry[index] = weakref.ref(self)

    # ---- whisper.model.Whisper-like attributes
    @property
    def is_multilingual(self):
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self):
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    def set_alignment_heads(self, heads):
        """heads: iterable of (layer, head) pairs used by timing.default_find_alignment."""
        heads = sorted({(int(l), int(h)) for l, h in heads})  # row-major like alignment_heads.indices().T (timing.py:155)
        for l, h in heads:
            if not (0 <= l < self.dims.n_text_layer and 0 <= h < self.dims.n_text_head):
                raise ValueError("alignment head (%d, %d) out of range" % (l, h))
        self.alignment_heads = heads
        self.alignment_heads_source = "set_alignment_heads"

    def use_official_alignment_heads(self, name=None):
        """What whisper.load_model(name) does after loading the weights: install the model's curated alignment heads.
        `name` defaults to the official model the dimensions identify; returns the name used, or None (with a warning)
        when there is no table for it -- default_find_alignment then averages the generic default set, which is NOT what
        the reference's baseline uses."""
        name = name or model_name_from_dims(self.dims)
        heads = ALIGNMENT_HEADS.get(name)
        if heads is not None and all(l < self.dims.n_text_layer and h < self.dims.n_text_head for l, h in heads):
            self.set_alignment_heads(heads)
            self.alignment_heads_source = "openai-whisper _ALIGNMENT_HEADS[%r]" % name
            return name
        import warnings
        warnings.warn("no official alignment heads for %r with these dimensions: default_find_alignment will use the generic "
                      "default (every head of the upper decoder half); pass the model name (e.g. 'large-v2') or call "
                      "set_alignment_heads" % (name,))
        return None

    def to(self, device):
        if torch.device(device) != self.device:
            raise ValueError("a WhisperAMD engine is bound to %s at construction" % self.device)
        return self

    def eval(self):
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.wca_engine_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass

    # ---- weights
    def _load_one(self, name, arr):
        if isinstance(arr, torch.Tensor):
            arr = arr.detach().cpu()
            if arr.dtype == torch.float16:
                arr = arr.contiguous().numpy()
            else:
                arr = arr.float().contiguous().numpy()
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.float16:
            dt = _lib.DTYPE_F16
        else:
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            dt = _lib.DTYPE_F32
        shape = (C.c_int64 * max(arr.ndim, 1))(*(arr.shape if arr.ndim else (1,)))
        _lib.check(self._lib.wca_load_weight(self._h, name.encode(), arr.ctypes.data_as(C.c_void_p), dt, shape, max(arr.ndim, 1)))

    def load_state_dict(self, sd, allow_rounded_weights=False):
        """sd: openai-whisper naming (encoder.blocks.0.attn.query.weight ...), torch tensors or numpy arrays.
        Weight matrices are stored f16 (every openai checkpoint is f16 at rest; whisper.load_model upcasts those values,
        /root/reference/infer_ali.py:36-37). An fp32 state dict whose values are NOT f16-representable (a fine-tuned fp32 checkpoint, which the
        reference runs in true fp32) keeps the remainders f16(w - f16(w)) in a second slab and the contract mode multiplies the extra term
        A_hi W_lo^T -- the same hi + lo representation the activations travel in; slower (one more f16 pass per affected GEMM), never narrower.
        allow_rounded_weights=True skips that term (the engine then computes what the f16-ROUNDED checkpoint computes). `weights_inexact` reports
        what was not exact; the f16 mode ignores the remainders (it is approximate by definition)."""
        _lib.check(self._lib.wca_set_allow_rounded_weights(self._h, 1 if allow_rounded_weights else 0))
        for name, t in sd.items():
            self._load_one(name, t)
        _lib.check(self._lib.wca_finalize_weights(self._h))
        self._finalized = True
        return self

    @property
    def weights_inexact(self):
        """(tensors, values, name of the first tensor) whose fp32 source values the f16 weight storage rounded; (0, 0, '') for f16 checkpoints."""
        nt, nv = C.c_longlong(0), C.c_longlong(0)
        buf = C.create_string_buffer(160)
        _lib.check(self._lib.wca_weights_inexact(self._h, C.byref(nt), C.byref(nv), buf, 160))
        return int(nt.value), int(nv.value), buf.value.decode()

    @classmethod
    def from_checkpoint(cls, path, device="cuda:0", max_batch=8, name=None, precision="reference", allow_rounded_weights=False):
        """Loads an openai-format checkpoint ({'dims':..., 'model_state_dict':...}) from a LOCAL path and, like
        whisper.load_model(name), installs the official alignment heads of `name` (inferred from the dimensions when
        they identify the model; large-v1 / large-v2 need the name). This is the drop-in's `whisper.load_model`: the model comes
        back in the CONTRACT precision mode ('reference': the fp32 forward of timing.py:58 to fp32 summation noise), like a bare
        WhisperAMD(); precision='f16' opts out. A checkpoint whose fp32 weights are not f16-representable runs with its remainder slab
        (load_state_dict); allow_rounded_weights=True drops the remainders."""
        ck = torch.load(path, map_location="cpu")
        dims = ModelDimensions(**ck["dims"])
        m = cls(dims, device=device, max_batch=max_batch, precision=precision)
        m.load_state_dict(ck["model_state_dict"], allow_rounded_weights=allow_rounded_weights)
        m.use_official_alignment_heads(name)
        return m

    # ---- stream handling: always run on torch's current stream so torch tensors stay ordered
    def _bind_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        if getattr(self, "_partitioned", False):
            # CU-partitioned engine: its masked streams are not ordered with torch's stream -- complete the caller's work first (and every
            # entry point that returns device tensors synchronises the engine afterwards: _after_call)
            torch.cuda.current_stream(self.device).synchronize()
        if s != self._stream_bound:
            rc = self._lib.wca_engine_set_stream(self._h, C.c_void_p(s))
            if rc != 1:   # WCA_STATUS_PARTITIONED: recorded only
                _lib.check(rc)
            self._stream_bound = s

    def synchronize(self):
        _lib.check(self._lib.wca_engine_synchronize(self._h))

    def _after_call(self):
        """Entry points that leave their results in device tensors are ordered with torch's stream by running ON it -- except on a
        CU-partitioned engine, whose masked streams torch knows nothing about: wait for them here."""
        if getattr(self, "_partitioned", False):
            self.synchronize()

    # ---- hot path entry points (thin wrappers; shapes documented in include/wca.h)
    def log_mel(self, pcm, n_samples=None):
        """pcm: f32 cuda tensor [n] or [B, n] (n <= 480000). Returns [n_mels, 3000] or [B, n_mels, 3000]."""
        single = pcm.dim() == 1
        p = pcm[None] if single else pcm
        p = p.to(self.device, torch.float32).contiguous()
        B, n = p.shape
        if n > N_SAMPLES:
            p = p[:, :N_SAMPLES].contiguous()
            n = N_SAMPLES
        ns = [n] * B if n_samples is None else [min(int(v), n) for v in n_samples]
        out = torch.empty(B, self.dims.n_mels, N_FRAMES, device=self.device, dtype=torch.float32)
        self._bind_stream()
        for b0 in range(0, B, self.max_batch):
            b1 = min(B, b0 + self.max_batch)
            _lib.check(self._lib.wca_log_mel(self._h, _ptr(p[b0:b1]), n, _lib.i32_array(ns[b0:b1]), b1 - b0, _ptr(out[b0:b1])))
        self._after_call()
        return out[0] if single else out

    def get_attentions(self, mel, tokens, max_frames, medfilt_width=7, qk_scale=1.0, n_tok=None, want_logits=True):
        """mel [B,n_mels,3000] f32, tokens [B,n] int64, max_frames list[int] -> (weights [B,L,H,n,F], logits [B,n,V])."""
        mel = mel.to(self.device, torch.float32).contiguous()
        tokens = tokens.to(self.device, torch.int64).contiguous()
        B, n = tokens.shape
        if B > self.max_batch:
            raise ValueError("batch %d > engine max_batch %d" % (B, self.max_batch))
        mf = [int(v) for v in max_frames]
        F = max(mf)
        L, H = self.dims.n_text_layer, self.dims.n_text_head
        weights = torch.empty(B, L, H, n, max(F, 1), device=self.device, dtype=torch.float32)
        logits = torch.empty(B, n, self.dims.n_vocab, device=self.device, dtype=torch.float32) if want_logits else None
        self._bind_stream()
        nt = _lib.i32_array(n_tok) if n_tok is not None else None
        _lib.check(self._lib.wca_get_attentions(self._h, _ptr(mel), _ptr(tokens), B, n, nt, _lib.i32_array(mf),
                                                int(medfilt_width), float(qk_scale), _ptr(weights), _ptr(logits)))
        self._after_call()
        return weights, logits

    def encode(self, mel):
        """Encoder only (tests): mel [B,n_mels,3000] -> ln_post output [B,1500,d] f32."""
        mel = mel.to(self.device, torch.float32).contiguous()
        B = mel.shape[0]
        out = torch.empty(B, 1500, self.dims.n_audio_state, device=self.device, dtype=torch.float32)
        self._bind_stream()
        _lib.check(self._lib.wca_test_encoder(self._h, _ptr(mel), B, _ptr(out)))
        self._after_call()
        return out

    def make_opts(self, aggregation="mean", topk=-1, w_colnorm=1.0, w_rownorm=1.0, w_coverage=0.0, sot_len=3,
                  medfilt_width=7, qk_scale=1.0):
        if aggregation not in ("mean", "topk"):
            raise ValueError("aggregation must be 'mean' or 'topk'")
        if aggregation == "topk":
            assert topk > 0  # timing.py:92
        return _lib.AlignOpts(_lib.AGGR_TOPK if aggregation == "topk" else _lib.AGGR_MEAN, int(topk), float(w_colnorm),
                              float(w_rownorm), float(w_coverage), int(sot_len), int(medfilt_width), float(qk_scale))

    def align_batch(self, pcm, n_samples, tokens, n_tok, max_frames, opts, enqueue_only=False):
        """Fused pipeline. pcm [B,stride] f32 cuda, tokens [B,n_max] int64 cuda. Returns (jump_frames [B,n_max] int32, sel [B,topk])."""
        B, n_max = tokens.shape
        self._bind_stream()
        # pcm=None re-uses the encoder state left by the preceding greedy_decode of the same batch
        args = (self._h, _ptr(pcm) if pcm is not None else None, pcm.shape[1] if pcm is not None else 0,
                _lib.i32_array(n_samples) if n_samples is not None else None, _ptr(tokens), n_max, _lib.i32_array(n_tok),
                _lib.i32_array(max_frames), B, C.byref(opts))
        _lib.check(self._lib.wca_align_batch_enqueue(*args))
        if enqueue_only:
            return None
        return self.fetch(B, n_max, opts)

    def encode_batch(self, mel=None, pcm=None, n_samples=None):
        """C ABI wca_encode_batch: enqueue log-mel/encoder/cross-K/V of a micro-batch (no host sync). The state is picked up
        by greedy_decode(None, None, None, ..., batch=B) and by align_batch(pcm=None, ...)."""
        self._bind_stream()
        B = mel.shape[0] if mel is not None else pcm.shape[0]
        if mel is not None:
            mel = mel.contiguous().float()
        _lib.check(self._lib.wca_encode_batch(self._h, _ptr(mel) if mel is not None else None, _ptr(pcm) if pcm is not None else None,
                                              pcm.shape[1] if pcm is not None else 0,
                                              _lib.i32_array(n_samples) if n_samples is not None else None, B))
        if not hasattr(self, "_keep"):
            import collections
            self._keep = collections.deque(maxlen=3)
        self._keep.append((mel, pcm))  # the kernels read these buffers asynchronously; up to two encodes are in flight
        return B

    def greedy_decode(self, mel, pcm, n_samples, initial_tokens, suppress_mask, blank_mask, sample_len, eot, timestamp_begin,
                      apply_timestamp_rules=True, max_initial_timestamp_index=50, batch=None, no_speech=-1):
        """C ABI wca_greedy_decode. mel [B,n_mels,3000] f32 cuda XOR pcm [B,stride] f32 cuda (+ n_samples).
        Returns (tokens [B, n_initial+sample_len] int32, n_tokens [B] int32, sum_logprobs [B] f32) as numpy arrays; the
        encoder state stays in the engine for a following align_batch(pcm=None, ...). With no_speech >= 0 (the
        <|nospeech|> token id) self.last_no_speech_prob [B] holds DecodingResult.no_speech_prob."""
        self._bind_stream()
        B = mel.shape[0] if mel is not None else (pcm.shape[0] if pcm is not None else int(batch))  # batch=: decode a state queued by encode_batch
        n_init = len(initial_tokens)
        T = n_init + int(sample_len)
        tokens = np.zeros((B, T), dtype=np.int32)
        n_tok = np.zeros(B, dtype=np.int32)
        lp = np.zeros(B, dtype=np.float32)
        sup = np.ascontiguousarray(suppress_mask, dtype=np.uint8)
        blank = np.ascontiguousarray(blank_mask, dtype=np.uint8) if blank_mask is not None else None
        if sup.shape[0] != self.dims.n_vocab or (blank is not None and blank.shape[0] != self.dims.n_vocab):
            raise ValueError("filter masks must have n_vocab entries")
        opts = _lib.DecodeOpts(int(sample_len), int(eot), int(timestamp_begin), 1 if apply_timestamp_rules else 0,
                               int(max_initial_timestamp_index), int(no_speech))
        nsp = np.full(B, np.nan, dtype=np.float32)
        if mel is not None:
            mel = mel.contiguous().float()
        _lib.check(self._lib.wca_greedy_decode(
            self._h, _ptr(mel) if mel is not None else None, _ptr(pcm) if pcm is not None else None,
            pcm.shape[1] if pcm is not None else 0, _lib.i32_array(n_samples) if n_samples is not None else None, B,
            _lib.i32_array(initial_tokens), n_init, sup.ctypes.data_as(C.c_void_p),
            blank.ctypes.data_as(C.c_void_p) if blank is not None else None, C.byref(opts),
            tokens.ctypes.data_as(_lib._pi32), n_tok.ctypes.data_as(_lib._pi32), lp.ctypes.data_as(_lib._pf),
            nsp.ctypes.data_as(_lib._pf) if no_speech >= 0 else None))
        self.last_no_speech_prob = nsp
        return tokens, n_tok, lp

    def decode(self, mel, options=None):
        """whisper.decode(model, mel, options) (infer_ali.py:60)."""
        from . import decoding
        return decoding.decode(self, mel, options if options is not None else decoding.DecodingOptions())

    def fetch(self, B, n_max, opts):
        k = opts.topk if opts.aggregation == _lib.AGGR_TOPK else 0
        jump = np.zeros((B, n_max), dtype=np.int32)
        sel = np.zeros((B, max(k, 1)), dtype=np.int32)
        _lib.check(self._lib.wca_align_batch_fetch(self._h, B, n_max, k, jump.ctypes.data_as(_lib._pi32),
                                                   sel.ctypes.data_as(_lib._pi32)))
        return jump, (sel if k > 0 else None)

    def set_profiling(self, on):
        _lib.check(self._lib.wca_set_profiling(self._h, 1 if on else 0))

    def kernel_ms(self, site):
        """(launches, summed ms, algorithmic flops per launch, algorithmic bytes per launch) of one encoder kernel site
        ('qkv' | 'attention' | 'out_proj' | 'fc1' | 'fc2' | 'ln1' | 'ln2') in the last align_batch call (profiling on)."""
        n = C.c_int(0)
        ms = C.c_float(0)
        fl = C.c_double(0)
        by = C.c_double(0)
        _lib.check(self._lib.wca_last_kernel_ms(self._h, _lib.SITES[site], C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        return n.value, ms.value, fl.value, by.value

    # ---- collation over RCCL through the C ABI (no torch.distributed involved; shard.allgather_results(..., engine=self))
    @staticmethod
    def comm_unique_id():
        """128-byte id of a new communicator (rank 0 creates it; hand it to the other ranks by any side channel)."""
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().wca_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self._lib.wca_comm_init(self._h, buf, int(rank), int(world)))
        self._comm = (int(rank), int(world))
        return self

    def comm_destroy(self):
        _lib.check(self._lib.wca_comm_destroy(self._h))
        self._comm = None

    @property
    def comm(self):
        """(rank, world) of this engine's RCCL communicator, or None."""
        return getattr(self, "_comm", None)

    def allgather_packed(self, packed):
        """packed: uint8 numpy array of this rank -> list of every rank's uint8 array (wca_allgather_results)."""
        rank, world = self._comm
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        sizes = (C.c_int64 * world)()
        cap = max(int(packed.size), 1 << 16)
        self._bind_stream()
        while True:
            out = np.zeros((world, cap), dtype=np.uint8)
            rc = self._lib.wca_allgather_results(self._h, packed.ctypes.data_as(C.c_void_p), int(packed.size), out.ctypes.data_as(C.c_void_p), cap, sizes)
            if rc == _lib.ERR_TOO_LONG:   # some rank packed more than the smallest `cap` of any rank: EVERY rank gets this verdict
                cap = max(int(v) for v in sizes)   # (decided on the gathered {size, capacity} pairs), and retries with the same room
                continue
            _lib.check(rc)
            return [out[r, :int(sizes[r])].copy() for r in range(world)]

    def allreduce_counters(self, *counters):
        arr = (C.c_int64 * len(counters))(*[int(v) for v in counters])
        self._bind_stream()
        _lib.check(self._lib.wca_allreduce_counters(self._h, arr, len(counters)))
        return tuple(int(v) for v in arr)

    def set_precision(self, mode):
        """'f16' (the engine's construction default): operands rounded to f16 once, fp32 accumulation. 'split' = 'reference' (the
        contract mode: what bench.py's `value` and the CLI default use): every operand of every stage as an f16 (hi, lo) pair against
        the exact f16 weights, three-pass attention (wca.h: wca_set_precision) -- the reference's fp32 forward to fp32 summation noise.
        No batch may be in flight; the activation arena is re-created when the engine leaves or enters f16."""
        modes = {"f16": 0, "split": 1, "reference": 1}
        if mode not in modes:
            raise ValueError("precision must be 'f16', 'reference' or 'split'")
        _lib.check(self._lib.wca_set_precision(self._h, modes[mode]))
        return self

    def set_precision_sites(self, sites, enc_first_layer=0):
        """Per-stage precision (wca.h: wca_set_precision_sites). `sites`: an iterable of names from _lib.PRECISION_SITES
        ('logmel', 'conv', 'enc_gemm', 'enc_attn', 'cross_kv', 'dec', 'capture'), the string 'all', or the integer mask; the
        named stages compute on (hi, lo) operand pairs, the others on single f16 operands. The encoder bits apply to blocks
        >= enc_first_layer."""
        if isinstance(sites, str):
            sites = list(_lib.PRECISION_SITES) if sites == "all" else [x for x in sites.replace("+", ",").split(",") if x]
        if isinstance(sites, int):
            mask = sites
        else:
            unknown = [x for x in sites if x not in _lib.PRECISION_SITES]
            if unknown:
                raise ValueError("unknown precision site(s) %s (known: %s)" % (unknown, sorted(_lib.PRECISION_SITES)))
            mask = 0
            for x in sites:
                mask |= _lib.PRECISION_SITES[x]
        _lib.check(self._lib.wca_set_precision_sites(self._h, mask, int(enc_first_layer)))
        return self

    @property
    def precision_sites(self):
        mask, first = C.c_uint(0), C.c_int(0)
        _lib.check(self._lib.wca_get_precision_sites(self._h, C.byref(mask), C.byref(first)))
        return sorted((n for n, b in _lib.PRECISION_SITES.items() if mask.value & b), key=lambda n: _lib.PRECISION_SITES[n]), first.value

    @property
    def precision(self):
        return {0: "f16", 1: "split", 2: "mixed"}[self._lib.wca_get_precision(self._h)]   # ("reference" is an alias of "split")

    def set_fuse_ln(self, on):
        """LayerNorm in the residual GEMMs' epilogue (needs the GPU to itself: wca.h) or as separate launches (the default)."""
        _lib.check(self._lib.wca_set_fuse_ln(self._h, 1 if on else 0))

    def set_overlap(self, on):
        """Phase 2 on its own stream (default) or everything on one stream (clean per-kernel profiles)."""
        _lib.check(self._lib.wca_set_overlap(self._h, 1 if on else 0))

    def set_cu_partition(self, phase2_cus):
        """Experiment (wca.h: wca_set_cu_partition): phase 2 / the decode loop on CU-masked streams owning `phase2_cus` compute units,
        phase 1 on the rest; 0 lifts it. Inputs must be complete (synchronised) before the engine is called while it is active."""
        torch.cuda.synchronize()
        _lib.check(self._lib.wca_set_cu_partition(self._h, int(phase2_cus)))
        self._partitioned = int(phase2_cus) > 0
        self._stream_bound = None   # re-bind torch's current stream at the next call (after a lift the engine must not stay on a private stream)

    def set_decode_mode(self, fused=True, streams=1):
        """Few-row decoder GEMMs fused with LayerNorm / KV append / split-K (default) or separate launches; greedy decode as
        one stream (default) or two interleaved half-batches."""
        _lib.check(self._lib.wca_set_decode_mode(self._h, 1 if fused else 0, int(streams)))

    def last_stage_ms(self):
        ms = (C.c_float * 8)()
        _lib.check(self._lib.wca_last_stage_ms(self._h, ms))
        return list(ms)
