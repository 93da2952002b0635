"""Shared test plumbing.

* ``gpu`` marker: tests that need a real MI355X (run with ``-m gpu`` on the GPU box).  Everything else runs on CPU.
* ``oracle/`` is test infrastructure: it is put on sys.path here (and only here / smoke / bench's cpu_baseline leg).
"""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    import latok_oracle
    latok_oracle.lib()
    return latok_oracle


def has_gpu() -> bool:
    try:
        from latok_amd import _lib
        return _lib.load().latok_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The initialised product library; GPU tests must not silently pass without it."""
    from latok_amd import _lib
    lib = _lib.ensure_init()
    return lib


# ---- adversarial string material shared by CPU and GPU tests -------------------------------------------------------
ALPHABETS = {
    "mixed": list("abcXYZ  \t.,:/@#$^!1 9éあ日🤓́Ⅷ"),
    "starts": list("ab@:/ .#"),
    "nospace_at": list("abcdefgh@"),
    "rare_space_at": list("abcdefgh@") + [" "],
    "words": list("abc ") + ["#x", "http://a", ".@u", "a@b"],
}


def random_strings(rng: random.Random, n, lo, hi, alphabet):
    return ["".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(n)]


def pack(texts):
    cps = [np.frombuffer(t.encode("utf-32-le", "surrogatepass"), dtype="<u4") for t in texts]
    row = np.zeros(len(texts) + 1, np.int64)
    np.cumsum([len(c) for c in cps], out=row[1:])
    flat = np.concatenate(cps).astype(np.uint32) if cps else np.zeros(0, np.uint32)
    return np.ascontiguousarray(flat), row


def bits_to_bool(bits: np.ndarray, total: int) -> np.ndarray:
    return np.unpackbits(bits.view(np.uint8), bitorder="little")[:total].astype(bool)
