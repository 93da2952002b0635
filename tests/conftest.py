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
    # a fresh checkout has no built library (binaries are not committed): build it once, as __graft_entry__.build() does
    if not os.path.exists(os.path.join(ROOT, "latok_amd", "liblatok_hip.so")) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "latok_amd", "csrc")])


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
    # PEP 393 kind 1 (every char <= U+00FF, with Latin-1 letters, symbols and the no-break space) and kind 2 (BMP only,
    # with a lone surrogate, a noncharacter, a combining mark and the ideographic space)
    "latin1": list("abcXYZ  \t.,:/@#$1 9\xe9\xc9\xd7\xb5\xa0\xff\xbf\xaa") + ["http://\xe9", "a@\xf1"],
    "bmp": list("abcXYZ  \t.,:/@#1 9\xe9\u3042\u65e5\u0301\u2167\u3000\ud800\uffff\u0416\u03b4") + ["http://\u65e5", "a@\u0436"],
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


# ---- runtime rule tables (reference extension point: default_tokenizer.py:9-30) ---------------------------------------
DEFAULT_RULES = (
    np.array([[5, -1], [6, -1], [20, -1], [4, 17], [4, 16]], np.int8),                               # C_SPLIT :49-55
    np.array([[7, 18, 13, -1], [11, 18, 21, 23], [8, 14, 15, -1], [9, 22, 24, 12]], np.int8),     # C_MASK  :80-91
    np.array([[6, 19]], np.int8),                                                                   # C_SYM   :100-102
)

# hand-written variations that break assumptions the built-in tables happen to satisfy
RULE_SETS = {
    "default": DEFAULT_RULES,
    # every symbol keeps its own boundary even inside a masked block; digits split; starts may sit ON a space
    "sym_everywhere": (np.array([[5, -1], [2, -1], [4, 17]], np.int8), np.array([[5, 13], [8, -1]], np.int8),
                       np.array([[6]], np.int8)),
    # no masking at all, split on case changes only
    "no_mask": (np.array([[4, 16, -1], [3, 14, 15]], np.int8), np.zeros((0, 1), np.int8), np.zeros((0, 1), np.int8)),
    # everything is a start (maximum pressure on the pending-start queue), split everywhere
    "all_starts": (np.array([[0], [5], [6], [2]], np.int8), np.array([[1], [6]], np.int8), np.array([[6, 19]], np.int8)),
    # uses every context column at least once, 1-D table for C_SYM (sum of rows, latok.c:340-353)
    "all_columns": (np.array([[12, 13], [14, 15], [16, 17], [18, 19], [20, 21], [22, 23], [24, 0], [1, 2], [3, 4], [5, 6],
                              [7, 8], [9, -1], [10, -1], [11, -1]], np.int8),
                    np.array([[10, 22, 24], [9, 12, 22], [7, 18, -1]], np.int8), np.array([11, 9, -1], np.int8)),
}


def random_rule_tables(rng: random.Random):
    def table(max_rows, max_cols, empty_ok):
        rows = rng.randint(0 if empty_ok else 1, max_rows)
        cols = rng.randint(1, max_cols)
        t = np.full((rows, cols), -1, np.int8)
        for r in range(rows):
            k = rng.randint(1, cols)
            t[r, :k] = [rng.randrange(25) for _ in range(k)]
        return t
    return table(8, 3, False), table(5, 4, True), table(3, 3, True)


MAX_RULE_ROWS = 32     # LK_MAX_RULE_ROWS (latok_amd/csrc/split_code.h)


def rule_row_sets(tables):
    """(C_SPLIT, C_MASK, C_SYM) -> (uint32[3 * MAX_RULE_ROWS] column sets, int32[3] row counts): the packing the kernel interprets."""
    rows = np.zeros(3 * MAX_RULE_ROWS, np.uint32)
    n_rows = np.zeros(3, np.int32)
    for t, tab in enumerate(tables):
        a = np.asarray(tab, np.int8)
        if a.ndim == 1:
            a = a[a != -1].reshape(-1, 1)
        n_rows[t] = a.shape[0]
        for r in range(a.shape[0]):
            for v in a[r]:
                if v != -1:
                    rows[t * MAX_RULE_ROWS + r] |= np.uint32(1) << np.uint32(v)
    return rows, n_rows


def oracle_rule_bits(oracle, texts, tables):
    """boundary bitmask of a batch under custom tables, from the reference-shaped oracle, string by string"""
    total = sum(len(t) for t in texts)
    flags = np.zeros(total, bool)
    k = 0
    for t in texts:
        if len(t):
            flags[k:k + len(t)] = oracle.split_values_rules(t, *tables) != 0
        k += len(t)
    bits = np.packbits(np.concatenate([flags, np.zeros((-total) % 64, bool)]), bitorder="little").view(np.uint64)
    return bits
