#!/usr/bin/env python3
"""Generates tests/golden/oracle_word_times_ragged_A.npz: the fp32 CPU ORACLE's word start / end times of tools/parity_ragged.py's leg A
(128 ragged utterances ids 20000-20127: 2.0-29.3 s audio, 9-220 characters; whisper-medium dims, synthetic.random_state_dict(seed=0,
cross_qk_std=0.08); char units, aggr topk, topk 10, medfilt 3) -- what tests/test_e2e_gpu.py::test_contract_mode_parity_ragged_lengths and
tests/test_oracle.py::test_ragged_oracle_fixture_structure_and_one_live_utterance read.

    python tests/golden/make_oracle_word_times_ragged.py        (~13 min on 8 cores; no GPU; nothing from /root/reference is needed)

It runs oracle/ (timing_ref / whisper_ref / tokenizer_ref) through `tools/parity_ragged.py --leg A --oracle-only` and copies the cache."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    cache = os.path.join(ROOT, "tools", "cache", "oracle_ragged_A_medium_peaky008_ids20000-20127.npz")
    if not os.path.exists(cache):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "parity_ragged.py"), "--leg", "A", "--utts", "128", "--first-id", "20000", "--oracle-only"])
    out = os.path.join(ROOT, "tests", "golden", "oracle_word_times_ragged_A.npz")
    shutil.copyfile(cache, out)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
