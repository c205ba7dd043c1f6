#!/usr/bin/env python3
"""Generates tests/golden/oracle_word_times_ragged_A.npz: the fp32 CPU ORACLE's word start / end times of tools/parity_ragged.py's leg A
(128 ragged utterances ids 20000-20127: 2.0-29.3 s audio, 9-220 characters; whisper-medium dims, synthetic.random_state_dict(seed=0,
cross_qk_std=0.08); char units, aggr topk, topk 10, medfilt 3) -- what tests/test_e2e_gpu.py::test_contract_mode_parity_ragged_lengths and
tests/test_oracle.py::test_ragged_oracle_fixture_structure_and_one_live_utterance read.

    python tests/golden/make_oracle_word_times_ragged.py        (~13 min on 8 cores; no GPU; nothing from /root/reference is needed)
    python tests/golden/make_oracle_word_times_ragged.py B      (round 5: leg B = the reference CLI's defaults, infer_ali.py:160-162: medfilt 7, aggr mean,
                                                                 same 128 utterances -> oracle_word_times_ragged_B.npz)
    python tests/golden/make_oracle_word_times_ragged.py A large-v3   (round 5: 24 utterances ids 21000-21023 at whisper-large-v3 DIMENSIONS, 128 mel bins, 32 + 32
                                                                 layers, 640 captured heads -> oracle_word_times_ragged_A_large_v3.npz; ~70 min on 8 cores)

It runs oracle/ (timing_ref / whisper_ref / tokenizer_ref) through `tools/parity_ragged.py --leg A --oracle-only` and copies the cache."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    leg = sys.argv[1] if len(sys.argv) > 1 else "A"
    model = sys.argv[2] if len(sys.argv) > 2 else "medium"
    first, n = (20000, 128) if model == "medium" else (21000, 24)
    cache = os.path.join(ROOT, "tools", "cache", "oracle_ragged_%s_%s_peaky008_ids%d-%d.npz" % (leg, model, first, first + n - 1))
    if not os.path.exists(cache):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "parity_ragged.py"), "--leg", leg, "--model", model, "--utts", str(n), "--first-id", str(first), "--oracle-only"])
    out = os.path.join(ROOT, "tests", "golden", "oracle_word_times_ragged_%s%s.npz" % (leg, "" if model == "medium" else "_" + model.replace("-", "_")))
    shutil.copyfile(cache, out)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
