#!/usr/bin/env python3
"""Maximum-size run (BASELINE configs[3]/[4] shapes): large-v2 dimensions (d=1280, 20 heads, 32+32 layers),
30 s audio (max_frames 1500), 448 decoder tokens (443 characters), mean and top-k aggregation.
Checks the structural invariants of the result and reports timing."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wca = importlib.import_module("whisper-char-alignment_amd")
m = lambda n: importlib.import_module("whisper-char-alignment_amd." + n)  # noqa: E731
syn, tk, rt, tm = m("synthetic"), m("tokenizer"), m("retokenize"), m("timing")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dims = wca.dims_for("large-v2")
t0 = time.time()
model = wca.WhisperAMD(dims, max_batch=B, precision="f16").load_state_dict(syn.random_state_dict(dims, seed=0, cross_qk_std=0.05))
print("setup %.1fs" % (time.time() - t0), flush=True)
tok = tk.get_tokenizer(True, language="English")
pcm = np.stack([syn.synth_audio(u, 480000) for u in range(B)])
texts = [syn.synth_text(u, 443) for u in range(B)]
tts = [rt.encode(t, tok, "char") for t in texts]
toks = np.array([[*tok.sot_sequence, tok.no_timestamps, *tt, tok.eot] for tt in tts], dtype=np.int64)
assert toks.shape == (B, 448)
pcm_d, tok_d = torch.from_numpy(pcm).cuda(), torch.from_numpy(toks).cuda()
for aggr, topk in (("topk", 10), ("mean", -1)):
    opts = model.make_opts(aggregation=aggr, topk=topk, sot_len=3, medfilt_width=7)
    for it in range(2):
        torch.cuda.synchronize()
        t1 = time.time()
        jump, sel = model.align_batch(pcm_d, [480000] * B, tok_d, [448] * B, [1500] * B, opts)
        dt = time.time() - t1
    for b in range(B):
        j = jump[b, :444]
        assert j[0] == 0 and np.all(np.diff(j) >= 0) and j[-1] <= 1499, (b, j[:5], j[-5:])
        words, st, en = tm.words_from_jump_frames(jump[b], tts[b], tok, "char")
        assert len(st) == len(texts[b].split()) and np.all(en >= st)
    print("large-v2 dims, B=%d, n=448, F=1500, aggr=%s: %.1f ms per batch (%.1f utt/s); DTW 444x1500 rows ok" %
          (B, aggr, dt * 1e3, B / dt), flush=True)
