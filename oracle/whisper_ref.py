"""ORACLE -- test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product path).

CPU (PyTorch fp32) restatement of the parts of the un-vendored third-party dependency `openai-whisper`
(unpinned, README.md:8 of the reference; >= v20240930 because timing.py:8 imports disable_sdpa) that the
reference's hot path calls:
  * whisper.audio.log_mel_spectrogram / pad_or_trim      <- dataset.py:46-48, README.md:101-103
  * whisper.model.Whisper.forward with the non-SDPA attention branch whose qk output the reference's
    forward hooks capture                                   <- timing.py:50-58
The upstream source is NOT in /root/reference; this follows its published algorithm as recorded in
SURVEY.md Appendix A.1/A.2. PARITY UNPINNED against upstream itself; cross-checked against the
independent HuggingFace `transformers` Whisper implementation (random-init, tests/test_oracle.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE, N_FFT, HOP_LENGTH, N_SAMPLES, N_FRAMES = 16000, 400, 160, 480000, 3000


def pad_or_trim(audio, length=N_SAMPLES):
    if audio.shape[-1] > length:
        audio = audio[..., :length]
    if audio.shape[-1] < length:
        audio = F.pad(audio, (0, length - audio.shape[-1]))
    return audio


def log_mel_spectrogram(audio, filters):
    """audio: f32 tensor [..., 480000]; filters: [n_mels, 201] f32 -> [..., n_mels, 3000]."""
    audio = torch.as_tensor(audio, dtype=torch.float32)
    filters = torch.as_tensor(filters, dtype=torch.float32)
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    mel_spec = filters @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    if log_spec.dim() == 2:
        log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    else:  # batched: the max is per utterance (the reference processes one utterance at a time)
        mx = log_spec.amax(dim=(-2, -1), keepdim=True)
        log_spec = torch.maximum(log_spec, mx - 8.0)
    return (log_spec + 4.0) / 4.0


class WhisperRef:
    """Functional fp32 Whisper forward over an openai-named state dict (values upcast to f32, as
    whisper.load_model does with its f16 checkpoints)."""

    def __init__(self, state_dict, dims):
        self.dims = dims
        self.sd = {k: torch.as_tensor(v).to(torch.float32) for k, v in state_dict.items()}

    def _lin(self, x, prefix, bias=True):
        return F.linear(x, self.sd[prefix + ".weight"], self.sd[prefix + ".bias"] if bias else None)

    def _ln(self, x, prefix):
        return F.layer_norm(x.float(), (x.shape[-1],), self.sd[prefix + ".weight"], self.sd[prefix + ".bias"], 1e-5)

    def _mha(self, x, src, prefix, n_head, mask=None):
        q = self._lin(x, prefix + ".query")
        k = self._lin(src, prefix + ".key", bias=False)
        v = self._lin(src, prefix + ".value")
        B, n, d = q.shape
        scale = (d // n_head) ** -0.25
        q = q.view(B, n, n_head, -1).permute(0, 2, 1, 3)
        k = k.view(B, k.shape[1], n_head, -1).permute(0, 2, 1, 3)
        v = v.view(B, v.shape[1], n_head, -1).permute(0, 2, 1, 3)
        qk = (q * scale) @ (k * scale).transpose(-1, -2)
        if mask is not None:
            qk = qk + mask[:n, :n]
        qk = qk.float()
        w = F.softmax(qk, dim=-1)
        out = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
        return self._lin(out, prefix + ".out"), qk

    def _block(self, x, prefix, n_head, xa=None, mask=None):
        x = x + self._mha(self._ln(x, prefix + ".attn_ln"), self._ln(x, prefix + ".attn_ln"), prefix + ".attn", n_head, mask)[0]
        qk = None
        if xa is not None:
            y, qk = self._mha(self._ln(x, prefix + ".cross_attn_ln"), xa, prefix + ".cross_attn", n_head)
            x = x + y
        h = self._ln(x, prefix + ".mlp_ln")
        h = F.gelu(self._lin(h, prefix + ".mlp.0"))
        x = x + self._lin(h, prefix + ".mlp.2")
        return x, qk

    @torch.no_grad()
    def encoder(self, mel):
        """mel [B, n_mels, 3000] -> [B, 1500, d]"""
        sd, D = self.sd, self.dims
        x = F.gelu(F.conv1d(mel, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1)
        assert x.shape[1:] == sd["encoder.positional_embedding"].shape, "incorrect audio shape"
        x = x + sd["encoder.positional_embedding"]
        for i in range(D.n_audio_layer):
            x, _ = self._block(x, f"encoder.blocks.{i}", D.n_audio_head)
        return self._ln(x, "encoder.ln_post")

    @torch.no_grad()
    def decoder(self, tokens, xa):
        """tokens [B, n] int64, xa [B, 1500, d] -> (logits [B, n, V], [qk_l [B, H, n, 1500]] per layer)"""
        sd, D = self.sd, self.dims
        n = tokens.shape[-1]
        x = sd["decoder.token_embedding.weight"][tokens] + sd["decoder.positional_embedding"][:n]
        mask = torch.full((D.n_text_ctx, D.n_text_ctx), float("-inf")).triu_(1)
        qks = []
        for i in range(D.n_text_layer):
            x, qk = self._block(x, f"decoder.blocks.{i}", D.n_text_head, xa=xa, mask=mask)
            qks.append(qk)
        x = self._ln(x, "decoder.ln")
        logits = (x @ sd["decoder.token_embedding.weight"].T).float()
        return logits, qks

    @torch.no_grad()
    def forward(self, mel, tokens):
        return self.decoder(tokens, self.encoder(mel))
