"""Seeded synthetic inputs of the benchmark / parity configuration: random-init
weights with a Whisper architecture (there is no checkpoint offline), gated-noise audio and
[a-z ] teacher text. Data generation only -- no part of the alignment algorithm lives here."""
import numpy as np
import torch


def sinusoids(length, channels, max_timescale=10000):
    """whisper.model.sinusoids (encoder positional embedding)."""
    assert channels % 2 == 0
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    scaled = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1)


def random_state_dict(dims, seed=0, std=0.02, cross_qk_std=None, dtype=torch.float16):
    """openai-whisper-named state dict with N(0, std) linears/convs, LayerNorm gamma 1 / beta 0,
    sinusoidal encoder positions, N(0, 0.01) decoder positions. Stored in f16 like real checkpoints
    (values are therefore exactly representable as GEMM operands). `cross_qk_std` optionally widens the
    cross-attention query/key weights so the attention maps are peaky enough for DTW parity tests."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rnd(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    d, dt = dims.n_audio_state, dims.n_text_state
    sd["encoder.conv1.weight"] = rnd(d, dims.n_mels, 3)
    sd["encoder.conv1.bias"] = rnd(d)
    sd["encoder.conv2.weight"] = rnd(d, d, 3)
    sd["encoder.conv2.bias"] = rnd(d)
    sd["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, d).to(torch.float32)

    def block(prefix, n, cross):
        for att in (["attn", "cross_attn"] if cross else ["attn"]):
            s_qk = cross_qk_std if (att == "cross_attn" and cross_qk_std) else std
            sd[f"{prefix}.{att}.query.weight"] = rnd(n, n, s=s_qk)
            sd[f"{prefix}.{att}.query.bias"] = rnd(n)
            sd[f"{prefix}.{att}.key.weight"] = rnd(n, n, s=s_qk)
            sd[f"{prefix}.{att}.value.weight"] = rnd(n, n)
            sd[f"{prefix}.{att}.value.bias"] = rnd(n)
            sd[f"{prefix}.{att}.out.weight"] = rnd(n, n)
            sd[f"{prefix}.{att}.out.bias"] = rnd(n)
            sd[f"{prefix}.{att}_ln.weight"] = torch.ones(n, dtype=dtype)
            sd[f"{prefix}.{att}_ln.bias"] = torch.zeros(n, dtype=dtype)
        sd[f"{prefix}.mlp.0.weight"] = rnd(4 * n, n)
        sd[f"{prefix}.mlp.0.bias"] = rnd(4 * n)
        sd[f"{prefix}.mlp.2.weight"] = rnd(n, 4 * n)
        sd[f"{prefix}.mlp.2.bias"] = rnd(n)
        sd[f"{prefix}.mlp_ln.weight"] = torch.ones(n, dtype=dtype)
        sd[f"{prefix}.mlp_ln.bias"] = torch.zeros(n, dtype=dtype)

    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", d, False)
    sd["encoder.ln_post.weight"] = torch.ones(d, dtype=dtype)
    sd["encoder.ln_post.bias"] = torch.zeros(d, dtype=dtype)
    sd["decoder.token_embedding.weight"] = rnd(dims.n_vocab, dt)
    sd["decoder.positional_embedding"] = rnd(dims.n_text_ctx, dt, s=0.01)
    for i in range(dims.n_text_layer):
        block(f"decoder.blocks.{i}", dt, True)
    sd["decoder.ln.weight"] = torch.ones(dt, dtype=dtype)
    sd["decoder.ln.bias"] = torch.zeros(dt, dtype=dtype)
    return sd


def aligned_state_dict(dims, seed=0, frames_per_token=7.0, text_row0=3, branch_scale=0.05, dtype=torch.float16):
    """random_state_dict with ALIGNMENT-LIKE cross-attention planted into head 0 of the upper half of the decoder layers (what a
    trained Whisper's alignment heads look like to the alignment pipeline: a sharp monotonic ridge, well separated head scores),
    for parity tests on maps that resemble the real use. Construction: the residual branches (attention out-projections, mlp.2)
    are damped by `branch_scale` so that positions survive the stack; decoder position p carries the ENCODER's sinusoid code
    of time t = frames_per_token * (p - text_row0); the planted heads' cross-attention query / key projections select 32
    (sin, cos) channel pairs of that code, so q.k = g^2 * sum_c cos(w_c (t_p - f)) peaks at frame f = t_p. Gains differ per
    layer so that the planted heads' selection scores are not tied. Data generation only."""
    sd = random_state_dict(dims, seed=seed, dtype=dtype)
    d, dt = dims.n_audio_state, dims.n_text_state
    assert d == dt and d >= 128
    for k in list(sd):
        if k.endswith(".out.weight") or k.endswith(".out.bias") or k.endswith(".mlp.2.weight") or k.endswith(".mlp.2.bias"):
            sd[k] = (sd[k].float() * branch_scale).to(dtype)
    half = d // 2
    inc = np.log(10000) / (half - 1)
    inv = torch.exp(-inc * torch.arange(half))
    t = frames_per_token * (torch.arange(dims.n_text_ctx).float() - text_row0)
    sd["decoder.positional_embedding"] = torch.cat([torch.sin(t[:, None] * inv[None, :]), torch.cos(t[:, None] * inv[None, :])], dim=1).to(dtype)
    chans = [min(half - 1, (half // 64) * 2 * i) for i in range(32)]   # wavelengths from ~6 to ~600 frames at d = 1024
    L = dims.n_text_layer
    for li in range(L // 2, L):
        gain = 1.2 + 0.6 * (li - L // 2) / max(1, L - L // 2 - 1)
        for name in ("query", "key"):
            w = sd[f"decoder.blocks.{li}.cross_attn.{name}.weight"].float()
            w[:64] = 0.0
            for r, c in enumerate(chans):
                w[r, c] = gain
                w[32 + r, half + c] = gain
            sd[f"decoder.blocks.{li}.cross_attn.{name}.weight"] = w.to(dtype)
        b = sd[f"decoder.blocks.{li}.cross_attn.query.bias"].float()
        b[:64] = 0.0
        sd[f"decoder.blocks.{li}.cross_attn.query.bias"] = b.to(dtype)
    return sd


def synth_audio(utt_id, n_samples=160000):
    """0.1*N(0,1) gated by a 4 Hz square envelope, clipped to [-1,1] (exercises the max-8 dB floor)."""
    rng = np.random.default_rng(1234 + int(utt_id))
    x = 0.1 * rng.standard_normal(n_samples)
    t = np.arange(n_samples) / 16000.0
    gate = (np.floor(t * 8.0).astype(np.int64) % 2 == 0).astype(np.float64)  # 4 Hz square wave
    return np.clip(x * gate, -1.0, 1.0).astype(np.float32)


def synth_text(utt_id, n_chars=64):
    """n_chars characters from [a-z ]: words of 2-9 letters, single spaces, no leading/trailing space."""
    rng = np.random.default_rng(4321 + int(utt_id))
    letters = "abcdefghijklmnopqrstuvwxyz"
    out = ""
    while len(out) < n_chars:
        remaining = n_chars - len(out)
        if out:
            if remaining < 3:  # cannot fit ' ' + 2 letters: extend the last word instead
                out += "".join(rng.choice(list(letters), size=remaining))
                break
            out += " "
            remaining -= 1
        wl = int(rng.integers(2, 10))
        wl = min(wl, remaining)
        if remaining - wl in (1, 2):  # do not strand a tail too short for another word
            wl = remaining
        out += "".join(rng.choice(list(letters), size=wl))
    return out[:n_chars]
