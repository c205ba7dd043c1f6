import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def wca():
    """The product package (hyphenated directory name => importlib)."""
    return importlib.import_module("whisper-char-alignment_amd")


@pytest.fixture(scope="session")
def lib(wca):
    return wca._lib.load()


@pytest.fixture
def switch(wca, lib):
    """Sets process-wide A/B / test switches of libwca.so (wca_test_set_switch) and puts every one back to 0 after the test."""
    used = set()

    def set_switch(name, value):
        used.add(name)
        wca._lib.check(lib.wca_test_set_switch(name.encode(), int(value)))

    yield set_switch
    for name in used:
        lib.wca_test_set_switch(name.encode(), 0)


@pytest.fixture(scope="session")
def fake_vocab(tmp_path_factory):
    """A tiktoken-format vocabulary with the right SIZE (50257 ranks: the 256 bytes + synthetic 4-letter tokens) so that
    any id a random-weight model emits can be decoded; the real multilingual.tiktoken is not in the container."""
    import base64
    tokmod = importlib.import_module("whisper-char-alignment_amd.tokenizer")
    path = tmp_path_factory.mktemp("vocab") / "fake.tiktoken"
    ranks = dict(tokmod._byte_ranks())
    i = 0
    while len(ranks) < 50257:
        w = bytes([97 + (i % 26), 97 + (i // 26) % 26, 97 + (i // 676) % 26, 97 + (i // 17576) % 26])
        i += 1
        if w not in ranks:
            ranks[w] = len(ranks)
    with open(path, "wb") as f:
        for tokb, r in sorted(ranks.items(), key=lambda kv: kv[1]):
            f.write(base64.b64encode(tokb) + b" " + str(r).encode() + b"\n")
    return str(path)
