"""GPU parity: the HIP path (through the C ABI) against the oracle on identical inputs.  Bit-exact everywhere:
this is integer / byte / index work."""
import ctypes as C
import random

import numpy as np
import pytest

from conftest import ALPHABETS, bits_to_bool, pack, random_strings

pytestmark = pytest.mark.gpu

G1 = "This is a #test! Testing, Testing, 1 2 3"


def _check_batch(oracle, texts):
    from latok_amd import batch
    cps, row = pack(texts)
    total = int(row[-1])
    ov, ob = oracle.split_batch(cps, row)
    gv = batch.split_values_batch(cps, row)
    gb = batch.split_mask_batch(cps, row)
    if not np.array_equal(ov, gv):
        bad = np.nonzero(ov != gv)[0]
        i = int(bad[0])
        s = int(np.searchsorted(row, i, side="right") - 1)
        raise AssertionError(f"values differ at packed char {i} (string {s}, pos {i - row[s]}, {len(bad)} diffs): "
                             f"oracle {ov[max(row[s], i - 8):i + 8]} gpu {gv[max(row[s], i - 8):i + 8]} "
                             f"text {texts[s][max(0, i - row[s] - 8):i - row[s] + 8]!r}")
    assert np.array_equal(ob, gb), "bitmask differs although values agree"
    assert np.array_equal(bits_to_bool(gb, total), gv != 0)
    # offsets API against np.nonzero of the oracle values, string by string
    counts, offs = batch.split_offsets_csr(cps, row)
    exp = [np.nonzero(ov[row[s]:row[s + 1]])[0] for s in range(len(texts))]
    assert np.array_equal(counts, [len(e) for e in exp])
    assert np.array_equal(offs, np.concatenate(exp) if exp else np.zeros(0, np.int64))


def test_reference_main_sentence(gpu, oracle):
    """reference default_tokenizer.py:194-209 (__main__) / notebook golden G1."""
    from latok_amd.core import default_tokenizer as dt
    assert list(dt.tokenize(G1)) == ['This', 'is', 'a', '#test', '!', 'Testing', ',', 'Testing', ',', '1', '2', '3']
    assert list(dt.tokenize(G1)) == oracle.tokenize(G1)
    from latok_amd import batch
    nz = batch.split_offsets_batch([G1])[0]
    assert nz.tolist() == [0, 4, 7, 9, 15, 16, 17, 24, 25, 26, 33, 34, 36, 38]


@pytest.mark.parametrize("kind,n,lo,hi", [
    ("mixed", 400, 0, 40), ("starts", 60, 0, 300), ("mixed", 4, 3000, 20000), ("nospace_at", 3, 5000, 30000),
    ("rare_space_at", 3, 5000, 30000), ("words", 200, 0, 200), ("mixed", 3000, 0, 12), ("starts", 2000, 1, 3),
])
def test_random_adversarial(gpu, oracle, kind, n, lo, hi):
    rng = random.Random(hash((kind, n, lo, hi)) & 0xFFFF)
    for _ in range(6):
        _check_batch(oracle, random_strings(rng, rng.randint(1, n), lo, hi, ALPHABETS[kind]))


def test_edge_lengths_and_empties(gpu, oracle):
    rng = random.Random(5)
    texts = []
    for n in (0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 0, 0, 8191, 8192, 8193):
        texts.append("".join(rng.choice(ALPHABETS["mixed"]) for _ in range(n)))
    _check_batch(oracle, texts)
    _check_batch(oracle, ["", "", ""])
    _check_batch(oracle, ["a"])
    _check_batch(oracle, [" "])


def test_block_mask_stress_documents(gpu, oracle):
    """SURVEY 8d C5 stress docs, scaled to sizes the oracle finishes quickly."""
    n = 200_000
    base = "abcdefghij" * (n // 10)
    docs = [
        base,                                                  # (i) no whitespace at all
        "http://" + base,                                      # (ii) URL start at char 4: whole doc masked
        ("word, " * (n // 6)) + "see http://x.y/z",            # (iii) a start in the final block
        "a@b@c@d x,y p,q r,s t,u " * 2000,                     # (iv) k=3 starts in one block + spill-over
        "a " * (n // 2),                                       # (v) alternating space / non-space
        ("x" * 5000 + "@" + "y" * 5000 + " ") * 10,            # long blocks with a start in the middle
    ]
    _check_batch(oracle, docs)


def test_corpus_device_matches_host_and_oracle(gpu, oracle):
    from latok_amd import _lib
    lib = gpu
    for model, seed, lo, hi in ((_lib.CORPUS_ASCII, 0x1A70C0DE, 64, 192), (_lib.CORPUS_UNICODE, 0x1A70C0DF, 128, 384)):
        n_str = 20000
        row = np.zeros(n_str + 1, np.int64)
        _lib.check(lib.latok_corpus_offsets(seed, 0, n_str, lo, hi, row.ctypes.data))
        total = int(row[-1])
        host = np.zeros(total, np.uint32)
        _lib.check(lib.latok_corpus_fill_host(seed, model, 0, n_str, row.ctypes.data, host.ctypes.data))
        d_row = lib.latok_dev_alloc(row.nbytes)
        d_cps = lib.latok_dev_alloc(host.nbytes)
        d_bits = lib.latok_dev_alloc(((total + 63) // 64) * 8)
        try:
            _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
            _lib.check(lib.latok_corpus_fill_device(seed, model, 0, n_str, d_row, d_cps, None))
            dev = np.zeros(total, np.uint32)
            _lib.check(lib.latok_memcpy_d2h(dev.ctypes.data, d_cps, dev.nbytes))
            assert np.array_equal(host, dev), "device corpus generator differs from host generator"
            # device-pointer API on the device-generated corpus
            _lib.check(lib.latok_split_mask_batch(d_cps, d_row, n_str, total, d_bits, _lib.DEVICE_PTRS, None))
            _lib.check(lib.latok_sync())
            bits = np.zeros((total + 63) // 64, np.uint64)
            _lib.check(lib.latok_memcpy_d2h(bits.ctypes.data, d_bits, bits.nbytes))
            n8 = C.c_int64(0)
            _lib.check(lib.latok_utf8_bytes(d_cps, total, C.byref(n8), _lib.DEVICE_PTRS))
            assert n8.value == sum(1 if c < 0x80 else 2 if c < 0x800 else 3 if c < 0x10000 else 4 for c in host.tolist())
        finally:
            for p in (d_row, d_cps, d_bits):
                lib.latok_dev_free(p)
        _, ob = oracle.split_batch(host, row, want_values=False)
        assert np.array_equal(ob, bits)


def test_compat_native_functions(gpu, oracle):
    """_gen_parse_matrix / _combine_matrix_rows through the compat kernels."""
    from latok_amd import latok as ext
    rng = random.Random(11)
    for text in [G1, "$#@^:a./", "can’t wait to get my glasses back 🤓", "a", "ab", "x\t\ny", "①②Ⅷ 五 ½"] + \
            random_strings(rng, 50, 1, 300, ALPHABETS["mixed"]):
        m = ext._gen_parse_matrix(text)
        assert m.dtype == np.int8 and m.shape == (len(text), 25)
        assert np.array_equal(m, oracle.gen_parse_matrix(text))
        for idx in (np.array([[5, -1], [6, -1], [20, -1], [4, 17], [4, 16]], np.int8),
                    np.array([[7, 18, 13, -1], [11, 18, 21, 23], [8, 14, 15, -1], [9, 22, 24, 12]], np.int8),
                    np.array([[6, 19]], np.int8)):
            assert np.array_equal(ext._combine_matrix_rows(m.T, idx), oracle.combine_matrix_rows(m.T, idx))
        rows = np.arange(0, min(len(text), 100), dtype=np.int8)
        assert np.array_equal(ext._combine_matrix_rows(m, rows), oracle.combine_matrix_rows(m, rows))
    assert ext._gen_parse_matrix("").shape == (0, 25)
    with pytest.raises(ValueError):
        ext._gen_parse_matrix()
    with pytest.raises(ValueError):
        ext._combine_matrix_rows(np.zeros((2, 2), np.int8))
