"""CPU sanitizer + fuzz build of the host-side FLAC decoder (SURVEY section 5 "sanitizers": CPU build only).
csrc/flac.cpp parses untrusted files in place of torchaudio.load (/root/reference/dataset.py:31,104). This test compiles it with
g++ -fsanitize=address,undefined (non-recoverable) together with tools/flac_fuzz.cpp, twice -- as shipped, and with the frame CRCs not
enforced so that mutations reach the subframe / residual decoders -- and runs 10^4 seeded mutations (bit flips, overwrites,
truncations, deletions, duplications, insertions, header tampering) of streams written by the in-test encoder (tests/flac_fixture.py) and
of the RFC 9639 example: every call must return a status code. No GPU, no libwca.so."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17", "-Wall"]


def _seeds(d):
    import flac_fixture as ff
    rng = np.random.default_rng(5)
    n = 3000
    t = np.arange(n)
    speech = (6000 * np.sin(t * 0.05) * (1 + 0.3 * np.sin(t * 0.003)) + rng.normal(0, 300, n)).astype(np.int64)
    stereo = np.stack([speech, np.roll(speech, 7) // 2 + rng.integers(-50, 50, n)])
    streams = {
        "lpc": ff.encode(speech, blocksize=1024, kinds=("lpc",)),
        "fixed_rice2": ff.encode(speech, blocksize=576, kinds=(("fixed", 2), ("fixed", 4), ("fixed", 0)), rice2=True, porder=1),
        "verbatim_constant": ff.encode(np.concatenate([np.full(512, 77), speech[:700]]), blocksize=512, kinds=("constant", "verbatim", ("fixed", 1))),
        "escape": ff.encode(speech, blocksize=1024, kinds=("lpc", ("fixed", 3)), escape_first=True),
        "wasted": ff.encode(speech * 8, blocksize=1152, kinds=("lpc",), bps=24),
        "mid_side": ff.encode(stereo, blocksize=1024, kinds=("lpc",), stereo="mid_side"),
        "left_side": ff.encode(stereo, blocksize=256, kinds=(("fixed", 2),), stereo="left_side", with_total=False),
        "right_side_8bit": ff.encode(np.clip(stereo // 64, -120, 120), blocksize=4096, kinds=("lpc",), stereo="right_side", bps=8, padding_block=False),
        "rfc9639": bytes.fromhex("664c6143800000221000100000000f00000f0ac442f0000000013e84b41807dc690307586a3dad1a2e0ffff869180000bf0358fd03128baa9a"),
    }
    for k, b in streams.items():
        with open(os.path.join(d, k + ".flac"), "wb") as f:
            f.write(b)
    return len(streams)


@pytest.mark.parametrize("skip_crc", [False, True])
def test_flac_decoder_asan_ubsan_mutation_fuzz(tmp_path, skip_crc):
    seeds = tmp_path / "seeds"
    seeds.mkdir()
    n_seeds = _seeds(str(seeds))
    exe = str(tmp_path / ("flac_fuzz_nocrc" if skip_crc else "flac_fuzz"))
    cmd = ["g++"] + SAN + (["-DWCA_FLAC_FUZZ_SKIP_CRC"] if skip_crc else []) + [
        os.path.join(ROOT, "tools", "flac_fuzz.cpp"), os.path.join(ROOT, "whisper-char-alignment_amd", "csrc", "flac.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe, str(seeds), "10000", "20261005"], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    line = run.stdout.strip().splitlines()[-1]
    print(line)
    f = dict(zip(["ok", "invalid", "nomem"], [int(x) for x in line.split("mutated cases 10000:")[1].replace("ok", "").replace("invalid", "").replace("nomem", "").split()]))
    assert "seeds %d decoded %d" % (n_seeds, n_seeds) in line
    assert f["ok"] + f["invalid"] + f["nomem"] == 10000
    assert f["invalid"] > 5000                 # most damage is detected ...
    if skip_crc:
        assert f["ok"] > 500                   # ... and without the checksums a good share of the mutations decodes all the way through
    else:
        assert f["ok"] < f["invalid"]          # with them, only mutations outside the audio frames (padding, unused header bits) survive
