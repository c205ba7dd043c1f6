#!/usr/bin/env python3
"""Generates tests/golden/oracle_word_times_medium_peaky.npz: the fp32 CPU ORACLE's word start / end times and all 384 head
selection scores for the north-star configuration (whisper-medium dims, synthetic.random_state_dict(seed=0, cross_qk_std=0.08),
10 s audio synth_audio(id), 64-char text synth_text(id), char units, topk 10, medfilt 3) on utterance ids 100-131 and
10000-10300 -- what tests/test_e2e_gpu.py's contract gate and bench-sized parity legs compare the GPU path with, so that the GPU
suite does not spend minutes of CPU time per run re-deriving them (the oracle is ~4 s per utterance on 16 cores).

    python tests/golden/make_oracle_word_times.py          (~25 min on 8 cores; no GPU; nothing from /root/reference is needed)
    python tests/golden/make_oracle_word_times.py --leg 700   (round 5: the second leg, ids 10301-11000 -> oracle_word_times_medium_peaky_700.npz,
                                                               ~50 min on 8 cores; tests/test_e2e_gpu.py::test_contract_mode_parity_1033_fixture_utterances)

It runs oracle/ (timing_ref / whisper_ref / tokenizer_ref) through tools/precision_ablation.py --oracle-only for both id ranges and
merges the two caches. The GPU tests also run the LIVE oracle on a few of these utterances and require it to reproduce the fixture
(which ties the fixture to the oracle code of the commit under test)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LEGS = {"333": ([(100, 32), (10000, 301)], "oracle_word_times_medium_peaky.npz"), "700": ([(10301, 700)], "oracle_word_times_medium_peaky_700.npz")}


def main():
    leg = sys.argv[sys.argv.index("--leg") + 1] if "--leg" in sys.argv else "333"
    RANGES, out_name = LEGS[leg]
    merged = {}
    for first, n in RANGES:
        key = "oracle_medium_peaky008_s10_c64_k10_m3_ids%d-%d.npz" % (first, first + n - 1)
        path = next((p for p in (os.path.join(ROOT, "gpurun_out", key), os.path.join(ROOT, "tools", "cache", key)) if os.path.exists(p)), None)   # (a previous run's cache)
        if path is None:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "precision_ablation.py"), "--oracle-only", "--first-id", str(first), "--utts", str(n)])
            path = os.path.join(ROOT, "gpurun_out", key)
        z = np.load(path, allow_pickle=False)
        for u in range(first, first + n):
            merged["st_%d" % u] = z["st_%d" % u].astype(np.float64)
            merged["en_%d" % u] = z["en_%d" % u].astype(np.float64)
            merged["sc_%d" % u] = z["sc_%d" % u].astype(np.float32)   # (selection scores: float32 is what the oracle computes in)
    out = os.path.join(ROOT, "tests", "golden", out_name)
    np.savez_compressed(out, **merged)
    print(out, os.path.getsize(out), "bytes,", len(merged) // 3, "utterances")


if __name__ == "__main__":
    main()
