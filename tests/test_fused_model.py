"""The CPU model of the GPU pipeline (oracle/fused_model.cpp: same tiling, same lane math header, same four stages)
against the reference-shaped oracle.  This is where the algorithm -- bit tricks, tile summaries, scan, fix-up,
spill-over slow path -- is validated without a GPU.  CPU only."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import ALPHABETS, ROOT, pack, random_strings


@pytest.fixture(scope="module")
def model():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "fused"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "libfused_model.so"))
    L.fused_split_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fused_block_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return L


def run_model(L, cps, row):
    total = int(row[-1])
    vals = np.zeros(total, np.uint8)
    bits = np.zeros((total + 63) // 64, np.uint64)
    nfix = C.c_int64(0)
    assert L.fused_split_batch(cps.ctypes.data, row.ctypes.data, len(row) - 1, vals.ctypes.data, bits.ctypes.data,
                               C.byref(nfix)) == 0
    return vals, bits, nfix.value


@pytest.mark.parametrize("kind,n,lo,hi", [
    ("mixed", 400, 0, 40), ("starts", 60, 0, 300), ("mixed", 4, 3000, 20000), ("nospace_at", 3, 5000, 30000),
    ("rare_space_at", 3, 5000, 30000), ("words", 200, 0, 200),
])
def test_model_matches_oracle(model, oracle, kind, n, lo, hi):
    rng = random.Random(hash((kind, n)) & 0xFFFF)
    fixed = 0
    for _ in range(12):
        cps, row = pack(random_strings(rng, rng.randint(1, n), lo, hi, ALPHABETS[kind]))
        ov, ob = oracle.split_batch(cps, row)
        mv, mb, nf = run_model(model, cps, row)
        fixed += nf
        assert np.array_equal(ov, mv) and np.array_equal(ob, mb)
        # bitmask-only mode takes the patch-in-place path of the scan stage for the common fix-up cases
        pb = np.zeros_like(mb)
        assert model.fused_split_batch(cps.ctypes.data, row.ctypes.data, len(row) - 1, None, pb.ctypes.data, None) == 0
        assert np.array_equal(ob, pb)
    if kind in ("nospace_at", "rare_space_at"):
        assert fixed > 0, "the fix-up stage was never exercised"


def test_model_all_code_points(model, oracle):
    """The device table layout (unicode_tables.inc, two-stage + split codes) classifies every code point like the
    oracle's run-length table: one string holding all of 0..0x10FFFF plus out-of-range values."""
    cps = np.concatenate([np.arange(0x110000, dtype=np.uint32), np.array([0x110000, 0x7FFFFFFF, 0xFFFFFFFF], np.uint32)])
    row = np.array([0, cps.size], np.int64)
    ov, _ = oracle.split_batch(cps, row, want_bits=False)
    mv, _, _ = run_model(model, cps, row)
    assert np.array_equal(ov, mv)
    # and with a letter between consecutive code points so every char's own class shows in its neighbours' context
    inter = np.empty(2 * 0x110000, np.uint32)
    inter[0::2] = np.arange(0x110000)
    inter[1::2] = ord("a")
    row = np.array([0, inter.size], np.int64)
    assert np.array_equal(oracle.split_batch(inter, row, want_bits=False)[0], run_model(model, inter, row)[0])


def test_model_block_mask(model, oracle):
    rng = random.Random(17)
    for _ in range(1500):
        n = rng.choice([1, 2, 3, 7, 63, 64, 65, 200, 4096, 4097, 9000])
        p1, p2 = rng.choice([0, .01, .1, .4]), rng.choice([0, .01, .1, .4])
        a1 = np.array([rng.random() < p1 for _ in range(n)], np.int8)
        a2 = np.array([rng.random() < p2 for _ in range(n)], np.int8)
        out = np.zeros(n, np.int8)
        model.fused_block_mask(a1.ctypes.data, a2.ctypes.data, n, out.ctypes.data)
        assert np.array_equal(out, oracle.gen_block_mask(a1, a2)), (n, p1, p2)


@pytest.mark.parametrize("name", ["default", "sym_everywhere", "no_mask", "all_starts", "all_columns", "random"])
def test_model_runtime_rule_tables(model, oracle, name):
    """kModeRules: caller-supplied C_SPLIT / C_MASK / C_SYM interpreted over the 25 feature planes (lane_math.h:
    lk_rules_generic) against the reference recipe run on the same tables by the oracle."""
    from conftest import RULE_SETS, oracle_rule_bits, random_rule_tables, rule_row_sets
    model.fused_split_batch_rules.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    rng = random.Random(hash(name) & 0xFFFF)
    for rep in range(6 if name == "random" else 3):
        tables = random_rule_tables(rng) if name == "random" else RULE_SETS[name]
        rows, n_rows = rule_row_sets(tables)
        for kind, n, lo, hi in [("mixed", 120, 0, 60), ("starts", 20, 0, 400), ("mixed", 2, 4000, 9000),
                                ("rare_space_at", 2, 5000, 9000)]:
            texts = random_strings(rng, rng.randint(1, n), lo, hi, ALPHABETS[kind])
            cps, row = pack(texts)
            bits = np.zeros((int(row[-1]) + 63) // 64, np.uint64)
            assert model.fused_split_batch_rules(cps.ctypes.data, row.ctypes.data, len(row) - 1, rows.ctypes.data,
                                                 n_rows.ctypes.data, bits.ctypes.data, None) == 0
            assert np.array_equal(bits, oracle_rule_bits(oracle, texts, tables)), (name, rep, kind)
            if name == "default":
                assert np.array_equal(bits, oracle.split_batch(cps, row, want_values=False)[1])


def test_model_utf8_byte_space(model, oracle):
    """Byte-space rule algebra (lane_math.h: lk_rules_bytes -- smeared code planes, continuation plane, "next lead"
    shifts) in the CPU model: boundary bits and the smeared SPACE plane at BYTE positions of the UTF-8 encoding against
    the oracle's code-point results mapped to bytes."""
    model.fused_split_batch_utf8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = random.Random(31)
    alpha = ALPHABETS["mixed"] + list("é日🤓ü　 ") + ["http://é", "a@日", ".@ü", "#日"]
    for it in range(400):
        kind = rng.choice([0, 0, 0, 1, 2])
        if kind == 0:
            texts = random_strings(rng, rng.randint(1, 6), 0, 14, alpha)
        elif kind == 1:
            texts = random_strings(rng, rng.randint(1, 30), 0, 300, alpha)
        else:
            texts = random_strings(rng, rng.randint(1, 3), 2000, 9000, alpha)
        blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
        boff = np.zeros(len(texts) + 1, np.int64)
        np.cumsum([len(b) for b in blobs], out=boff[1:])
        total = int(boff[-1])
        if total == 0:
            continue
        flags = np.zeros(total, bool)
        space = np.zeros(total, bool)
        for t, b0 in zip(texts, boff[:-1]):
            if not t:
                continue
            v = oracle.split_values(t)
            m = oracle.gen_parse_matrix(t)
            pos = 0
            for i, ch in enumerate(t):
                n = len(ch.encode("utf-8", "surrogatepass"))
                flags[b0 + pos] = v[i] != 0
                space[b0 + pos:b0 + pos + n] = m[i, 5] != 0
                pos += n
        u8 = np.frombuffer(b"".join(blobs), np.uint8)
        bits = np.zeros((total + 63) // 64, np.uint64)
        sp = np.zeros_like(bits)
        assert model.fused_split_batch_utf8(u8.ctypes.data, boff.ctypes.data, len(texts), bits.ctypes.data, sp.ctypes.data,
                                            None) == 0
        got = np.unpackbits(bits.view(np.uint8), bitorder="little")[:total].astype(bool)
        gsp = np.unpackbits(sp.view(np.uint8), bitorder="little")[:total].astype(bool)
        assert np.array_equal(got, flags), (it, texts if total < 200 else total)
        assert np.array_equal(gsp, space), (it, "space plane")


def test_model_smear_arithmetic_on_malformed_utf8(model):
    """lane_math.h lk_owner_before / lk_smear_planes (the tile kernel's phase 2 in byte space: codes at lead bytes only, the
    continuation bytes filled by mask arithmetic) against the per-byte definition "owner = nearest non-continuation byte at
    most 3 back" on RANDOM bytes: runs of > 3 continuation bytes, truncated sequences, lone leads, 0xF8..0xFF, at every
    word and tile phase.  The model checks both forms against each other internally and aborts on a difference."""
    model.fused_split_batch_utf8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    pools = [np.arange(256, dtype=np.uint8),
             np.array([0x20, 0x61, 0x80, 0x80, 0xBF, 0xC3, 0xE3, 0xF0, 0xFF, 0x2E, 0x40], np.uint8),
             np.array([0x80, 0x81, 0xE3, 0x61], np.uint8)]
    for it in range(60):
        n_str = int(rng.integers(1, 40))
        lens = rng.integers(0, [20, 300, 9000][it % 3], n_str)
        boff = np.zeros(n_str + 1, np.int64)
        np.cumsum(lens, out=boff[1:])
        total = int(boff[-1])
        if total == 0:
            continue
        u8 = np.ascontiguousarray(rng.choice(pools[it % len(pools)], total))
        bits = np.zeros((total + 63) // 64, np.uint64)
        sp = np.zeros_like(bits)
        assert model.fused_split_batch_utf8(u8.ctypes.data, boff.ctypes.data, n_str, bits.ctypes.data, sp.ctypes.data, None) == 0


def test_table_free_ascii_classification_is_the_table(model):
    """lane_math.h lk_ascii_code_planes (the narrow-input tile kernels classify ASCII words AFTER bit-slicing, as boolean
    functions of the raw bit planes, instead of 128 LDS lookups per word) against the class table for all 128 ASCII values,
    each in every position of the 64-byte word."""
    model.fused_ascii_codes.argtypes = [C.c_void_p] * 3
    for rot in range(64):
        for base in (0, 64):
            b = np.array([base + (i + rot) % 64 for i in range(64)], np.uint8)
            got, want = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
            model.fused_ascii_codes(b.ctypes.data, got.ctypes.data, want.ctypes.data)
            assert np.array_equal(got, want), (rot, base)
    rng = np.random.default_rng(3)
    for _ in range(200):
        b = rng.integers(0, 128, 64).astype(np.uint8)
        got, want = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        model.fused_ascii_codes(b.ctypes.data, got.ctypes.data, want.ctypes.data)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name", ["default", "sym_everywhere", "no_mask", "all_starts", "all_columns", "random"])
def test_model_runtime_rule_tables_in_byte_space(model, oracle, name):
    """Run-time rule tables on UTF-8 input in byte space (lane_math.h: lk_feature_planes_bytes / lk_rules_generic_bytes --
    PREV_* columns from the smeared planes, NEXT_* / AFTER_NEXT_* through the next-lead operator): boundary bits at the
    lead bytes against the reference recipe run on the same tables, string by string, in char space."""
    from conftest import RULE_SETS, random_rule_tables, rule_row_sets
    model.fused_split_batch_utf8_rules.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = random.Random(hash(name) & 0xFFF)
    alpha = ALPHABETS["mixed"] + list("é日🤓ü　Жδ") + ["http://é", "a@日", ".@ü", "#日", "Ünï", "９"]
    for rep in range(6 if name == "random" else 3):
        tables = random_rule_tables(rng) if name == "random" else RULE_SETS[name]
        rows, n_rows = rule_row_sets(tables)
        for n, lo, hi in [(120, 0, 60), (20, 0, 400), (2, 4000, 9000)]:
            texts = random_strings(rng, rng.randint(1, n), lo, hi, alpha)
            blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
            boff = np.zeros(len(texts) + 1, np.int64)
            np.cumsum([len(b) for b in blobs], out=boff[1:])
            total = int(boff[-1])
            if total == 0:
                continue
            want = np.zeros(total, bool)
            for t, b0 in zip(texts, boff[:-1]):
                if t:
                    v = oracle.split_values_rules(t, *tables)
                    pos = np.cumsum([0] + [len(ch.encode("utf-8", "surrogatepass")) for ch in t])[:-1]
                    want[b0 + pos] = v != 0
            u8 = np.frombuffer(b"".join(blobs), np.uint8)
            bits = np.zeros((total + 63) // 64, np.uint64)
            assert model.fused_split_batch_utf8_rules(u8.ctypes.data, boff.ctypes.data, len(texts), rows.ctypes.data, n_rows.ctypes.data,
                                                      bits.ctypes.data, None) == 0
            got = np.unpackbits(bits.view(np.uint8), bitorder="little")[:total].astype(bool)
            assert np.array_equal(got, want), (name, rep, int(np.nonzero(got != want)[0][0]))


def test_pext_of_the_lane_math_is_bit_extraction(model):
    """lane_math.h lk_pext64 (five-round compress on 32-bit halves) against the definition, on random words, sparse / dense
    masks, and the lead-byte masks UTF-8 text produces."""
    model.fused_pext64.restype = C.c_uint64
    model.fused_pext64.argtypes = [C.c_uint64, C.c_uint64]

    def want(x, m):
        out, k = 0, 0
        for i in range(64):
            if (m >> i) & 1:
                out |= ((x >> i) & 1) << k
                k += 1
        return out

    rng = random.Random(77)
    cases = [(0, 0), (2**64 - 1, 2**64 - 1), (2**64 - 1, 0), (0x8000000000000001, 0x8000000000000001), (2**64 - 1, 0xFFFFFFFF00000000),
             (2**64 - 1, 0x00000000FFFFFFFF), (0x123456789ABCDEF0, 0x5555555555555555)]
    for _ in range(3000):
        m = rng.getrandbits(64)
        if rng.random() < 0.3:
            m &= rng.getrandbits(64)
        if rng.random() < 0.3:
            m |= rng.getrandbits(64)
        cases.append((rng.getrandbits(64), m))
    for _ in range(500):   # lead masks of UTF-8: every lead followed by 0..3 continuation bytes
        m, i = 0, rng.randint(0, 3)
        while i < 64:
            m |= 1 << i
            i += rng.choice([1, 1, 2, 3, 3, 4])
        cases.append((rng.getrandbits(64), m))
    for x, m in cases:
        assert model.fused_pext64(x, m) == want(x, m), (hex(x), hex(m))


def _decode_per_lead(u8):
    """the device decoder's rule (utf8_decode.h): one code point per lead byte (any byte that is not 10xxxxxx), read from the lead and
    the continuation bytes right behind it; a sequence that is cut short gives U+FFFD; 0xF8..0xFF count as 4-byte leads.
    Returns (cps, byte position of every cp)."""
    b = u8.astype(np.int64)
    n = b.size
    is_cont = (b & 0xC0) == 0x80
    lead = np.nonzero(~is_cont)[0]
    b0 = b[lead]
    extra = (b0 >= 0xC0).astype(np.int64) + (b0 >= 0xE0) + (b0 >= 0xF0)
    cp = np.where(b0 < 0x80, b0, np.where(b0 >= 0xF0, b0 & 7, np.where(b0 >= 0xE0, b0 & 15, b0 & 31)))
    bad = np.zeros(lead.size, bool)
    for j in (1, 2, 3):
        idx = lead + j
        ok = idx < n
        nxt = np.where(ok, b[np.minimum(idx, n - 1)], 0xFF)
        need = extra >= j
        good = need & ((nxt & 0xC0) == 0x80)
        bad |= need & ~good
        cp = np.where(good & ~bad, (cp << 6) | (nxt & 0x3F), cp)
    cp = np.where(bad, 0xFFFD, cp)
    return cp.astype(np.uint32), lead


def test_model_code_point_mask_is_the_byte_space_mask_at_lead_bytes(model):
    """The claim behind the code-point UTF-8 route (api.cpp mask_utf8_via_bytes): on ANY bytes -- well formed or not -- in which
    every continuation byte has a lead byte within the 3 bytes before it inside its string, the boundaries of the decoded
    code points (one per lead byte) are the byte-space boundaries read at the lead bytes.  Both sides in the CPU model.  Batches
    that break the condition are what the kernel flags as `odd` (they go to the decoder)."""
    model.fused_split_batch_utf8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(17)
    # strings made of chunks: chars of every length, truncated sequences, lone leads, stray continuation bytes behind ASCII rule
    # chars and behind complete sequences, 0xF8..0xFF leads; a few chunks that break the condition
    chunks = [b"a", b"b", b"Z", b" ", b" ", b".", b"@", b"#", b":", b"/", b"\t", b"1", b"\xc3\xa9", b"\xe3\x81\x82", b"\xe6\x97\xa5",
              b"\xf0\x9f\xa4\x93", b"\xe3\x80\x80", b"\xe3\x81", b"\xc3", b"\xf0\x9f", b"\xf0\x9f\xa4", b"a\x80", b".\x80", b"@\xbf\x80",
              b"\xc3\xa9\x80", b"\xff", b"\xf8\x80\x80\x80", b"\xe3\x81\x82\x80", b"\xd0\x96", b"\xce\xb4", b"http://", b"\xed\xa0\x80"]
    odd_chunks = [b"\x80\x80\x80\x80", b"\xf0\x9f\xa4\x93\x80", b"a\x80\x80\x80\x80\x80"]
    checked = odd_seen = 0
    for it in range(400):
        n_str = int(rng.integers(1, 12))
        blobs = []
        for _ in range(n_str):
            k = int(rng.integers(0, [6, 40, 400][it % 3]))
            parts = [chunks[i] for i in rng.integers(0, len(chunks), k)]
            if it % 7 == 0 and k:
                parts[int(rng.integers(0, k))] = odd_chunks[int(rng.integers(0, len(odd_chunks)))]
            if it % 11 == 0:
                parts = [b"\x80"] + parts                  # a string that begins with a continuation byte
            blobs.append(b"".join(parts))
        lens = np.array([len(x) for x in blobs], np.int64)
        boff = np.zeros(n_str + 1, np.int64)
        np.cumsum(lens, out=boff[1:])
        total = int(boff[-1])
        if total == 0:
            continue
        u8 = np.frombuffer(b"".join(blobs), np.uint8).copy()
        is_cont = (u8 & 0xC0) == 0x80
        # odd: a continuation byte at a string start, or behind three other continuation bytes
        odd = bool(is_cont[boff[:-1][lens > 0]].any())
        run = is_cont.copy()
        for k in (1, 2, 3):
            run[k:] &= is_cont[:-k]
            run[:k] = False
        odd = odd or bool(run.any())
        if odd:
            odd_seen += 1
            continue
        bits = np.zeros((total + 63) // 64, np.uint64)
        sp = np.zeros_like(bits)
        assert model.fused_split_batch_utf8(u8.ctypes.data, boff.ctypes.data, n_str, bits.ctypes.data, sp.ctypes.data, None) == 0
        by_byte = np.unpackbits(bits.view(np.uint8), bitorder="little")[:total].astype(bool)
        cps, lead = _decode_per_lead(u8)
        row = np.searchsorted(lead, boff).astype(np.int64)          # leads before every string start
        vals, cbits, _ = run_model(model, np.ascontiguousarray(cps), row)
        by_cp = np.unpackbits(cbits.view(np.uint8), bitorder="little")[:cps.size].astype(bool)
        assert np.array_equal(by_byte[lead], by_cp), (it, u8.tolist() if total < 100 else total)
        assert not by_byte[is_cont].any()
        checked += 1
    assert checked > 250 and odd_seen > 40


def test_table_driven_lead_decode_is_the_decoder_rule(model):
    """lane_math.h lk_lead_index (the byte-space kernel classifies a multi-byte char from cp >> 6 and the last byte's payload,
    computed straight from its bytes with one table entry per byte value) against the decoder's rule (utf8_decode.h): every lead
    byte 0xC0..0xFF with every second byte, third / fourth bytes over a set that holds both ends of the continuation range and
    non-continuation bytes; a byte below 0xC0 starts nothing (stage 1's last entry, never "cut short")."""
    model.fused_lead_decode.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
    tail = [0x00, 0x41, 0x7F, 0x80, 0x81, 0x9F, 0xA0, 0xBE, 0xBF, 0xC0, 0xE3, 0xFF]
    cp = C.c_uint32(0)
    for b0 in range(0xC0):
        for rest in (0, 0x808080, 0xBFBFBF, 0xFFFFFF, 0x41E380):
            assert model.fused_lead_decode(b0 | rest << 8, C.byref(cp)) == 0
            assert cp.value == 0x110000, (b0, hex(cp.value))
    n = 0
    for b0 in range(0xC0, 0x100):
        k = 1 + (b0 >= 0xE0) + (b0 >= 0xF0)          # continuation bytes the lead asks for
        for b1 in range(256):
            for b2 in tail:
                for b3 in tail:
                    bs = (b0, b1, b2, b3)
                    bad = model.fused_lead_decode(b0 | b1 << 8 | b2 << 16 | b3 << 24, C.byref(cp))
                    ok = all((bs[j] & 0xC0) == 0x80 for j in range(1, k + 1))
                    assert bad == (0 if ok else 1), bs
                    if ok:
                        want = b0 & (0x3F >> k)
                        for j in range(1, k + 1):
                            want = (want << 6) | (bs[j] & 0x3F)
                        assert cp.value == want, (bs, hex(cp.value), hex(want))
                    n += 1
    assert n == 64 * 256 * 144
