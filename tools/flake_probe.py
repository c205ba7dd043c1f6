#!/usr/bin/env python3
"""Development probe for an order-dependent mismatch seen once in tests/test_batch_invariance_gpu.py after test_split_gpu / test_kernels_gpu in one process."""
import importlib, os, sys
import numpy as np
import pytest
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
rc = pytest.main(["-q", "-x", "tests/test_split_gpu.py", "tests/test_kernels_gpu.py", "-p", "no:cacheprovider"])
print("pre-tests rc", rc, flush=True)
import test_batch_invariance_gpu as tb
wca = importlib.import_module("whisper-char-alignment_amd")
syn, tk, rt = tb._m("synthetic"), tb._m("tokenizer"), tb._m("retokenize")
dims = wca.dims_for("medium")
tok = tk.get_tokenizer(True, language="English")
utts = tb._utts(syn, rt, tok, range(300, 308), 160000, 64)
sd = syn.aligned_state_dict(dims, seed=0)
ref = None
for it in range(8):
    model = wca.WhisperAMD(dims, device="cuda:0", max_batch=64).load_state_dict(sd)
    opts = model.make_opts(aggregation="topk", topk=10, sot_len=3, medfilt_width=3)
    for B in (1, 2, 64):
        fr = tb._frames_at_batch(model, tok, utts, B, 160000, 500, opts)
        if ref is None:
            ref = fr
        bad = [(i, np.nonzero(fr[i] != ref[i])[0][:6].tolist(), fr[i][fr[i] != ref[i]][:6].tolist(), ref[i][fr[i] != ref[i]][:6].tolist()) for i in fr if not np.array_equal(fr[i], ref[i])]
        print("iteration", it, "B", B, "differs from the first result in", bad, flush=True)
    del model
    torch.cuda.empty_cache()
