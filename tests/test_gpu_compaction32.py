"""Compaction entry points (np.nonzero per string, token spans, featurize spans: reference default_tokenizer.py:148-191)
in their int32 form (LATOK_OUT_INT32), the single-launch counts + scan, and the capacity protocol of the device-pointer
mode (records are written only if they fit; one synchronisation per call)."""
import ctypes as C
import random

import numpy as np
import pytest

from conftest import ALPHABETS, bits_to_bool, pack, random_strings

pytestmark = pytest.mark.gpu


def _batches():
    rng = random.Random(3232)
    yield random_strings(rng, 500, 0, 60, ALPHABETS["mixed"]) + ["", "x", " "]
    yield random_strings(rng, 6, 3000, 30000, ALPHABETS["words"]) + random_strings(rng, 50, 0, 9, ALPHABETS["starts"])
    yield random_strings(rng, 3, 20000, 70000, ALPHABETS["rare_space_at"])
    yield random_strings(rng, 30000, 0, 40, ALPHABETS["words"])          # > 512 strings: leaves the pinned small-batch path
    yield [""] * 7


def test_int32_records_equal_int64_records(gpu, oracle):
    from latok_amd import batch
    for texts in _batches():
        cps, row = pack(texts)
        c64, o64 = batch.split_offsets_csr(cps, row)
        c32, o32 = batch.split_offsets_csr(cps, row, dtype=np.int32)
        assert c32.dtype == o32.dtype == np.int32 and np.array_equal(c64, c32) and np.array_equal(o64, o32)
        vals, _ = oracle.split_batch(cps, row, want_bits=False)
        want = [np.nonzero(vals[row[s]:row[s + 1]])[0] for s in range(len(texts))]
        assert np.array_equal(o32, np.concatenate(want) if want else np.zeros(0))
        s64, s32 = batch.token_spans_csr(cps, row), batch.token_spans_csr(cps, row, dtype=np.int32)
        assert all(np.array_equal(a, b) for a, b in zip(s64, s32)) and s32[1].dtype == np.int32
        f64, f32 = batch.token_features_csr(cps, row), batch.token_features_csr(cps, row, dtype=np.int32)
        assert all(np.array_equal(a, b) for a, b in zip(f64, f32)) and f32[1].dtype == np.int32 and f32[2].dtype == np.int8
        # the other input forms
        units, krow = batch.pack_kind(texts)
        for fn in (batch.split_offsets_kind_csr, batch.token_spans_kind_csr, batch.token_features_kind_csr):
            assert all(np.array_equal(a, b) for a, b in zip(fn(units, krow), fn(units, krow, dtype=np.int32)))
        utf8, boff = batch.pack_utf8([t.encode("utf-8", "surrogatepass") for t in texts])
        for fn in (batch.split_offsets_utf8_csr, batch.token_spans_utf8_csr, batch.split_offsets_utf8_bytes_csr,
                   batch.token_spans_utf8_bytes_csr):
            assert all(np.array_equal(a, b) for a, b in zip(fn(utf8, boff), fn(utf8, boff, dtype=np.int32)))
        assert batch.tokenize_batch(texts) == [oracle.tokenize(t) if t else [] for t in texts]


def _dev(lib, a):
    from latok_amd import _lib
    p = lib.latok_dev_alloc(max(a.nbytes, 16))
    assert p
    _lib.check(lib.latok_memcpy_h2d(p, a.ctypes.data, a.nbytes))
    return p


def test_device_mode_capacity_protocol(gpu, oracle):
    """device pointers: everything is enqueued without waiting for the total; a too small buffer is left untouched and
    the call reports the needed size (include/latok_hip.h: latok_split_offsets_batch)"""
    from latok_amd import _lib, batch
    rng = random.Random(77)
    texts = random_strings(rng, 4000, 0, 80, ALPHABETS["mixed"])
    cps, row = pack(texts)
    total = int(row[-1])
    wc, wo = batch.split_offsets_csr(cps, row)
    n = len(wo)
    d_cps, d_row = _dev(gpu, cps), _dev(gpu, row)
    for flags, dt in ((_lib.DEVICE_PTRS, np.int64), (_lib.DEVICE_PTRS | _lib.OUT_INT32, np.int32)):
        sentinel = np.full(n + 8, -7, dt)
        d_counts, d_out = _dev(gpu, np.zeros(len(texts), dt)), _dev(gpu, sentinel)
        n_out = C.c_int64(0)
        rc = gpu.latok_split_offsets_batch(d_cps, d_row, len(texts), total, d_counts, d_out, n - 1, C.byref(n_out), flags, None)
        assert rc == _lib.ERR_INVALID and n_out.value == n and b"capacity" in gpu.latok_last_error()
        back = np.empty_like(sentinel)
        _lib.check(gpu.latok_memcpy_d2h(back.ctypes.data, d_out, back.nbytes))
        assert (back == -7).all(), "records were written although they did not fit"
        counts = np.empty(len(texts), dt)
        _lib.check(gpu.latok_memcpy_d2h(counts.ctypes.data, d_counts, counts.nbytes))
        assert np.array_equal(counts, wc)                     # the counts are always delivered
        _lib.check(gpu.latok_split_offsets_batch(d_cps, d_row, len(texts), total, d_counts, d_out, n, C.byref(n_out), flags, None))
        _lib.check(gpu.latok_memcpy_d2h(back.ctypes.data, d_out, back.nbytes))
        assert n_out.value == n and np.array_equal(back[:n], wo) and (back[n:] == -7).all()
        # counts only: NULL item buffer with capacity 0
        rc = gpu.latok_token_spans_batch(d_cps, d_row, len(texts), total, d_counts, None, 0, C.byref(n_out), flags, None)
        assert rc == _lib.ERR_INVALID and n_out.value == len(batch.token_spans_csr(cps, row)[1])
        for p in (d_counts, d_out):
            gpu.latok_dev_free(p)
    # host pointers: the same protocol
    counts = np.zeros(len(texts), np.int64)
    small = np.full(n - 1, -7, np.int64)
    n_out = C.c_int64(0)
    rc = gpu.latok_split_offsets_batch(cps.ctypes.data, row.ctypes.data, len(texts), total, counts.ctypes.data, small.ctypes.data,
                                       n - 1, C.byref(n_out), 0, None)
    assert rc == _lib.ERR_INVALID and n_out.value == n and (small == -7).all() and np.array_equal(counts, wc)
    for p in (d_cps, d_row):
        gpu.latok_dev_free(p)


def test_scan_epoch_wrap(gpu, oracle):
    """the look-back state of k_word_counts_scan is tagged with an 18-bit launch epoch instead of being cleared; at the
    wrap it is cleared once"""
    from latok_amd import _lib, batch
    lib = C.CDLL(_lib.LIB_PATH)
    rng = random.Random(1)
    texts = random_strings(rng, 9000, 0, 90, ALPHABETS["mixed"])        # ~100 tiles: several scan workgroups
    cps, row = pack(texts)
    want = batch.split_offsets_csr(cps, row)
    assert lib.latok_debug_set_scan_epoch(0x3FFFC) == 0
    for _ in range(8):
        got = batch.split_offsets_csr(cps, row)
        assert all(np.array_equal(a, b) for a, b in zip(want, got))
        got = batch.token_spans_csr(cps, row, dtype=np.int32)
        assert np.array_equal(got[0], batch.token_spans_csr(cps, row)[0])


def test_string_too_long_for_int32(gpu):
    """a string of 2^31 chars or more cannot be reported in int32 records: LATOK_ERR_INVALID, the 64-bit form works"""
    from latok_amd import _lib
    n = (1 << 31) + 70
    row = np.array([0, 10, 10 + n], np.int64)
    # host pointers: refused from the row offsets alone (no data is touched)
    n_out = C.c_int64(0)
    rc = gpu.latok_split_offsets_kind_batch(1, 1, row.ctypes.data, 2, int(row[-1]), 1, 1, 1, C.byref(n_out), _lib.OUT_INT32, None)
    assert rc == _lib.ERR_INVALID and b"too long" in gpu.latok_last_error()
    # device pointers: the kernel raises the flag
    total = int(row[-1])
    d_units = gpu.latok_dev_alloc(total + 64)
    assert d_units
    _lib.check(gpu.latok_memset_dev(d_units, ord("a"), total + 64))
    d_row = _dev(gpu, row)
    d_counts = gpu.latok_dev_alloc(64)
    d_out = gpu.latok_dev_alloc(1024)
    rc = gpu.latok_split_offsets_kind_batch(d_units, 1, d_row, 2, total, d_counts, d_out, 16, C.byref(n_out),
                                            _lib.DEVICE_PTRS | _lib.OUT_INT32, None)
    assert rc == _lib.ERR_INVALID and b"too long" in gpu.latok_last_error()
    _lib.check(gpu.latok_split_offsets_kind_batch(d_units, 1, d_row, 2, total, d_counts, d_out, 16, C.byref(n_out), _lib.DEVICE_PTRS, None))
    out = np.empty(2, np.int64)
    _lib.check(gpu.latok_memcpy_d2h(out.ctypes.data, d_out, 16))
    assert n_out.value == 2 and out.tolist() == [0, 0]      # "aaa...": one boundary per string, at its start
    for p in (d_units, d_row, d_counts, d_out):
        gpu.latok_dev_free(p)


def test_one_tile_batches_single_launch(gpu, oracle):
    """host batches of at most one tile (4096 chars) -- what tokenize(text) sends -- are served by ONE single-wave launch
    (k_small_batch: tile function + counts + ranks + scatter); everything around the 4096-char and 512-string limits of
    that path, offsets and spans, both record widths, built-in and run-time rule tables, against the oracle."""
    from conftest import RULE_SETS, oracle_rule_bits
    from latok_amd import batch
    rng = random.Random(4096)

    def check(texts, rules=None):
        cps, row = pack(texts)
        if rules is None:
            vals, _ = oracle.split_batch(cps, row, want_bits=False)
            flags = vals != 0
        else:
            flags = bits_to_bool(oracle_rule_bits(oracle, texts, rules), int(row[-1]))
        want = [np.nonzero(flags[row[s]:row[s + 1]])[0] for s in range(len(texts))]
        for dt in (np.int64, np.int32):
            c, o = batch.split_offsets_csr(cps, row, dtype=dt)
            assert np.array_equal(c, [len(w) for w in want]) and np.array_equal(o, np.concatenate(want) if want else np.zeros(0))
        if rules is None:
            toks = batch.tokenize_batch(texts)
            assert toks == [oracle.tokenize(t) if t else [] for t in texts]
            c64, s64 = batch.token_spans_csr(cps, row)
            c32, s32 = batch.token_spans_csr(cps, row, dtype=np.int32)
            assert np.array_equal(c64, c32) and np.array_equal(s64, s32)
            assert [[t[a:b] for a, b in s64[i0:i0 + n].tolist()] for t, i0, n in
                    zip(texts, np.concatenate([[0], np.cumsum(c64)[:-1]]).tolist(), c64.tolist())] == toks

    for n in (1, 2, 63, 64, 65, 127, 128, 129, 4094, 4095, 4096):
        check(["".join(rng.choice(ALPHABETS["mixed"]) for _ in range(n))])
        check(["x" * (n - 1) + " "])
    check(["http://" + "a" * 4089])                                  # one masked token that fills the tile exactly
    check(["a" * 4090 + " #tag"])
    check([" " * 4096])
    for _ in range(30):
        n_str = rng.choice([1, 2, 5, 64, 65, 200, 511, 512])
        budget = rng.choice([50, 700, 4096])
        texts, used = [], 0
        for _ in range(n_str):
            ln = rng.randint(0, max(0, min(60, budget - used)))
            texts.append("".join(rng.choice(ALPHABETS[rng.choice(["mixed", "starts", "words"])]) for _ in range(ln)))
            used += len(texts[-1])
        texts = [t[:max(0, 4096)] for t in texts]
        while sum(len(t) for t in texts) > 4096:
            texts.pop()
        check(texts)
    for name in ("sym_everywhere", "all_starts", "no_mask"):
        batch.set_rules(*RULE_SETS[name])
        try:
            for _ in range(5):
                check(random_strings(rng, rng.randint(1, 40), 0, 90, ALPHABETS["mixed"]), RULE_SETS[name])
        finally:
            batch.reset_rules()
