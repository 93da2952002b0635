"""GPU sweep of the Unicode classification: every code point through every input form.

The reference classifies with ``gettyperecord`` (latok/core/src/latok/latok.c:15-29) over ``index1`` / ``index2`` / the
record table (latok.h:1814-4173); the device uses its own two-stage table (latok_amd/csrc/unicode_tables.inc) that is
uploaded and transcoded by latok_init, copied into LDS by every workgroup and read by four different front ends
(UTF-32, PEP 393 kind 1 / kind 2, UTF-8 in byte space and in code-point units).  The CPU model only shares the .inc
file with that chain, so here all 0x110000 code points (plus out-of-range values) go through the real thing and are
pinned against ``tests/golden/unicode_classes.json`` (SHA-256 of the reference's own sweep) and against the oracle,
each char also placed next to a space / a letter / a URL trigger so that its class shows in every context column.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits_to_bool

pytestmark = pytest.mark.gpu

N_CP = 0x110000
OUT_OF_RANGE = np.array([0x110000, 0x110001, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF], np.uint32)


def _golden():
    with open(os.path.join(GOLDEN, "unicode_classes.json")) as f:
        return json.load(f)


def _interleave(cps, *fillers):
    """cp f0 f1 .. cp f0 f1 ..: every cp followed by the filler chars"""
    k = 1 + len(fillers)
    out = np.empty(cps.size * k, np.uint32)
    out[0::k] = cps
    for j, f in enumerate(fillers):
        out[1 + j::k] = f
    return out


def _variants(cps):
    """the swept code points alone, and next to each kind of neighbour the rules look at (space, letter, '@', '/', ':',
    '.', '#', upper/lower case) so that PREV_* / NEXT_* / AFTER_NEXT_* of every class are exercised"""
    yield "plain", cps
    yield "spaces", _interleave(cps, ord(" "))
    yield "letters", _interleave(cps, ord("a"))
    yield "upper-lower", _interleave(cps, ord("A"), ord("b"))
    yield "at", _interleave(cps, ord("@"), ord("x"), ord(" "))
    yield "url", _interleave(cps, ord(":"), ord("/"), ord("/"), ord("x"), ord(" "))
    yield "period-at", _interleave(cps, ord(" "), ord("."), ord("@"))
    yield "hash", _interleave(cps, ord(" "), ord("#"))


def _one_string(cps):
    return np.array([0, cps.size], np.int64)


def _many_strings(cps, every):
    """the same chars cut into strings of `every` chars: each swept char also sits at string starts / ends"""
    return np.concatenate([np.arange(0, cps.size, every, dtype=np.int64), [cps.size]])


def test_parse_matrix_all_code_points_matches_reference_sweep(gpu, oracle):
    """latok_parse_matrix (compat _gen_parse_matrix, latok.c:31-138) over all code points: the 12 base columns hash to
    what the reference itself produced; values >= 0x110000 classify as record 0 (latok.c:20-21)."""
    from latok_amd import _lib
    g = _golden()
    cps = np.concatenate([np.arange(N_CP, dtype=np.uint32), OUT_OF_RANGE])
    m = np.empty((cps.size, 25), np.int8)
    _lib.check(gpu.latok_parse_matrix(cps.ctypes.data, cps.size, m.ctypes.data, 0, None))
    assert set(np.unique(m).tolist()) <= {0, 1}
    words = (m[:, :12].astype(np.uint16) << np.arange(12, dtype=np.uint16)).sum(axis=1).astype("<u2")
    assert hashlib.sha256(words[:N_CP].tobytes()).hexdigest() == g["sha256_uint16le_words"]
    assert len(np.unique(words[:N_CP])) == g["n_classes"]
    assert (words[N_CP:] == 0).all()
    for key, info in g["classes"].items():
        assert (words[np.array(info["code_points"], np.int64)] == int(key, 16)).all()
    # the context columns of the same matrix against the oracle (whole matrix, one string)
    assert np.array_equal(m, oracle.gen_parse_matrix(cps))


def test_utf32_all_code_points(gpu, oracle):
    """fused kernel, UTF-32 front end: split VALUES and bitmask for every code point in every neighbourhood"""
    from latok_amd import batch
    cps0 = np.concatenate([np.arange(N_CP, dtype=np.uint32), OUT_OF_RANGE])
    for name, cps in _variants(cps0):
        for row in (_one_string(cps), _many_strings(cps, 7)):
            ov, ob = oracle.split_batch(cps, row)
            assert np.array_equal(batch.split_values_batch(cps, row), ov), name
            assert np.array_equal(batch.split_mask_batch(cps, row), ob), name


def test_ucs2_all_bmp_code_points(gpu, oracle):
    """PEP 393 kind 2 (latok.c:53-55,79): U+0000..U+FFFF incl. the surrogate range and noncharacters as UCS-2 units"""
    from latok_amd import batch
    for name, cps in _variants(np.arange(0x10000, dtype=np.uint32)):
        for row in (_one_string(cps), _many_strings(cps, 5)):
            _, ob = oracle.split_batch(cps, row, want_values=False)
            units = cps.astype(np.uint16)
            assert np.array_equal(batch.split_mask_kind_csr(units, row), ob), name
            counts, offs = batch.split_offsets_kind_csr(units, row)
            flags = bits_to_bool(ob, cps.size)
            glob = offs + np.repeat(row[:-1], counts)
            assert np.array_equal(np.nonzero(flags)[0], glob), name


def test_latin1_all_code_points(gpu, oracle):
    """PEP 393 kind 1: U+0000..U+00FF, every pair (a, b) of Latin-1 chars adjacent once, and in the usual neighbourhoods"""
    from latok_amd import batch
    a = np.arange(256, dtype=np.uint32)
    pairs = np.stack([np.repeat(a, 256), np.tile(a, 256)], axis=1).reshape(-1)
    cases = [("pairs", pairs)] + list(_variants(np.tile(a, 40)))
    for name, cps in cases:
        for row in (_one_string(cps), _many_strings(cps, 3)):
            _, ob = oracle.split_batch(cps, row, want_values=False)
            assert np.array_equal(batch.split_mask_kind_csr(cps.astype(np.uint8), row), ob), name


def _utf8_encode(cps):
    """vectorised UTF-8 encoder (surrogates encode like any 3-byte value: 'surrogatepass') -> (bytes, byte offset of each cp)"""
    cps = cps.astype(np.int64)
    n = 1 + (cps >= 0x80) + (cps >= 0x800) + (cps >= 0x10000)
    start = np.zeros(cps.size + 1, np.int64)
    np.cumsum(n, out=start[1:])
    out = np.zeros(int(start[-1]), np.uint8)
    s = start[:-1]
    m1, m2, m3, m4 = n == 1, n == 2, n == 3, n == 4
    out[s[m1]] = cps[m1]
    out[s[m2]] = 0xC0 | (cps[m2] >> 6)
    out[s[m2] + 1] = 0x80 | (cps[m2] & 0x3F)
    out[s[m3]] = 0xE0 | (cps[m3] >> 12)
    out[s[m3] + 1] = 0x80 | ((cps[m3] >> 6) & 0x3F)
    out[s[m3] + 2] = 0x80 | (cps[m3] & 0x3F)
    out[s[m4]] = 0xF0 | (cps[m4] >> 18)
    out[s[m4] + 1] = 0x80 | ((cps[m4] >> 12) & 0x3F)
    out[s[m4] + 2] = 0x80 | ((cps[m4] >> 6) & 0x3F)
    out[s[m4] + 3] = 0x80 | (cps[m4] & 0x3F)
    return out, start


def test_utf8_encoder_of_this_test_is_pythons():
    cps = np.array([0, 0x41, 0x7F, 0x80, 0x7FF, 0x800, 0xD7FF, 0xD800, 0xDFFF, 0xFFFF, 0x10000, 0x10FFFF], np.uint32)
    got, start = _utf8_encode(cps)
    want = "".join(chr(c) for c in cps).encode("utf-8", "surrogatepass")
    assert got.tobytes() == want and start[-1] == len(want)


def test_utf8_all_scalar_values(gpu, oracle):
    """UTF-8 front ends: every code point (surrogates as 3-byte sequences, like CPython's surrogatepass) in byte space
    (kModeBytes: positions are bytes) and in code-point units (device decode + UTF-32 kernel)"""
    from latok_amd import batch
    cps0 = np.arange(N_CP, dtype=np.uint32)
    for name, cps in _variants(cps0):
        for row in (_one_string(cps), _many_strings(cps, 6)):
            _, ob = oracle.split_batch(cps, row, want_values=False)
            flags = bits_to_bool(ob, cps.size)
            utf8, start = _utf8_encode(cps)
            boff = start[row]
            # byte space: a boundary char shows at its lead byte
            want = np.zeros(utf8.size, bool)
            want[start[:-1][flags]] = True
            got = bits_to_bool(batch.split_mask_utf8_bytes_csr(utf8, boff), utf8.size)
            if not np.array_equal(got, want):
                bad = int(np.nonzero(got != want)[0][0])
                k = int(np.searchsorted(start, bad, side="right") - 1)
                raise AssertionError(f"{name}: byte mask differs at byte {bad} = char {k} (U+{int(cps[k]):04X})")
            counts, offs = batch.split_offsets_utf8_bytes_csr(utf8, boff)
            assert np.array_equal(offs + np.repeat(boff[:-1], counts), np.nonzero(want)[0]), name
            # code-point units
            bits, cp_row = batch.split_mask_utf8_csr(utf8, boff)
            assert np.array_equal(cp_row, row) and np.array_equal(bits, ob), name
        if name == "spaces":
            dec, dec_row = batch.utf8_decode_csr(utf8, boff)
            assert np.array_equal(dec, cps) and np.array_equal(dec_row, row)


@pytest.mark.parametrize("n_tiles", [1537, 1600, 1679, 3328, 685])
def test_first_call_of_a_process_with_mid_size_batch(n_tiles):
    """the segment-aggregate workspace was sized for 16-tile segments while plan_segments makes
    12-tile ones -> a batch of 685..3328 tiles that SETS the workspace capacity overran it.  Every size runs as the
    first call of a fresh process (fresh workspace) and must equal the oracle."""
    import subprocess
    import sys
    code = f"""
import sys, numpy as np
sys.path[:0] = [{os.path.dirname(GOLDEN)!r} + '/..', {os.path.dirname(GOLDEN)!r} + '/../oracle']
from latok_amd import _lib, batch
import latok_oracle as orc
lib = _lib.ensure_init()
n_chars = {n_tiles} * 4096 - 100
n_str = n_chars // 97
row = np.zeros(n_str + 1, np.int64)
_lib.check(lib.latok_corpus_offsets(0x1A70C0DE, 0, n_str, 64, 130, row.ctypes.data))
cps = np.zeros(int(row[-1]), np.uint32)
_lib.check(lib.latok_corpus_fill_host(0x1A70C0DE, 0, 0, n_str, row.ctypes.data, cps.ctypes.data))
tiles = (int(row[-1]) + 4095) // 4096
bits = batch.split_mask_batch(cps, row)
_, ob = orc.split_batch(cps, row, want_values=False)
assert np.array_equal(bits, ob), 'mask differs'
c, o = batch.split_offsets_csr(cps, row)
assert int(c.sum()) == int(np.unpackbits(ob.view(np.uint8)).sum())
print('ok', tiles)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-2000:]
