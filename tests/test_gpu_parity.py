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


def test_compat_block_mask(gpu, oracle):
    """_gen_block_mask (reference latok.c:140-258) on arbitrary aligned arrays, incl. the probed SURVEY vectors,
    coincident start/space positions, multi-tile lengths and the element-0 quirk."""
    from latok_amd import latok as ext
    cases = [
        ([0, 0, 1, 0, 0, 0, 0], [0, 1, 0, 0, 1, 0, 0], [1, 1, 0, 0, 1, 1, 1]),
        ([0, 0, 1, 1, 0, 0, 0, 0], [0, 1, 0, 0, 1, 0, 1, 0], [1, 1, 0, 0, 1, 0, 1, 1]),
        ([0, 0, 0, 0, 0, 1, 0], [0, 1, 0, 0, 1, 0, 0], [1, 1, 1, 1, 1, 0, 0]),
        ([0, 0, 1, 0], [0, 0, 0, 0], [0, 0, 0, 0]),
    ]
    for a1, a2, want in cases:
        got = ext._gen_block_mask(np.array(a1, np.int8), np.array(a2, np.int8))
        assert got.dtype == np.int8 and got.tolist() == want
    rng = random.Random(3)
    for n, p1, p2 in [(1, .5, .5), (2, .5, .5), (7, .3, .3), (64, .1, .2), (65, .1, .2), (1000, .02, .15),
                      (4096, .01, .1), (4097, .01, .001), (20000, .001, .0), (20000, .0, .2), (30000, .3, .01),
                      (50000, .002, .0005)]:
        for _ in range(4):
            a1 = np.array([rng.random() < p1 for _ in range(n)], np.int8) * rng.choice([1, 3, -1])
            a2 = np.array([rng.random() < p2 for _ in range(n)], np.int8)
            assert np.array_equal(ext._gen_block_mask(a1, a2), oracle.gen_block_mask(a1, a2)), (n, p1, p2)
    with pytest.raises(ValueError):
        ext._gen_block_mask(np.zeros(3, np.int8))
    with pytest.raises(ValueError):
        ext._gen_block_mask(np.zeros((2, 2), np.int8), np.zeros(4, np.int8))
    with pytest.raises(ValueError):
        ext._gen_block_mask(np.zeros(3, np.int8), np.zeros(4, np.int8))
    assert ext._gen_block_mask(np.zeros(0, np.int8), np.zeros(0, np.int8)).shape == (0,)


def test_gen_split_mask_recipe_and_tokenize(gpu, oracle):
    """The user-editable recipe gen_split_mask(m) over the three compat kernels equals the fused kernel's values."""
    from latok_amd.core import default_tokenizer as dt
    from latok_amd import batch
    rng = random.Random(21)
    texts = [G1, "$#@^:a./", "camelCaseXMLParser", "foo@bar.com, .@user hi", "http://a@b X,y z", " ", "x"] + \
        random_strings(rng, 40, 1, 200, ALPHABETS["mixed"]) + random_strings(rng, 10, 1, 400, ALPHABETS["words"])
    for t in texts:
        sp = dt.gen_split_mask(dt._gen_parse_matrix(t))
        assert sp.dtype == np.int8
        assert np.array_equal(sp, oracle.split_values(t)), t
        assert list(dt.tokenize(t)) == oracle.tokenize(t)
        feats = list(dt.featurize(t))
        assert [f.text for f in feats] == oracle.tokenize(t)
    with pytest.raises(IndexError):
        list(dt.tokenize(""))
    assert batch.tokenize_batch(["", "a b", G1]) == [[], ["a", "b"], oracle.tokenize(G1)]


# ---- committed golden fixtures through the GPU path ------------------------------------------------------------------
def _golden(name):
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _text_of(cps):
    return np.array(cps, dtype="<u4").tobytes().decode("utf-32-le", "surrogatepass")


def test_golden_fixtures_on_gpu(gpu):
    """tests/golden was produced by the REAL reference (make_golden.py); no oracle involved here."""
    import hashlib
    from latok_amd import batch, latok as ext
    from latok_amd.core import default_tokenizer as dt
    g1 = _golden("g1_notebook.json")
    assert ext._gen_parse_matrix(g1["text"]).tolist() == g1["matrix"]
    assert dt.gen_split_mask(ext._gen_parse_matrix(g1["text"])).tolist() == g1["splits"]
    items = _golden("ref_strings.json")["items"]
    texts = [_text_of(it["cps"]) for it in items]
    cps, row = batch.pack(texts)
    vals = batch.split_values_batch(cps, row)
    offs = batch.split_offsets_batch(texts)
    toks = batch.tokenize_batch(texts)
    for i, it in enumerate(items):
        assert vals[row[i]:row[i + 1]].tolist() == it["splits"], texts[i]
        assert offs[i].tolist() == it["offsets"]
        assert toks[i] == [_text_of(t) for t in it["tokens"]]
        m = ext._gen_parse_matrix(texts[i])
        assert hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest() == it["matrix_sha256"]
    c1 = _golden("c1_paragraph.json")
    assert list(dt.tokenize(c1["text"])) == c1["tokens"]
    assert batch.split_offsets_batch([c1["text"]])[0].tolist() == c1["offsets"]
    nv = _golden("native_vectors.json")
    for v in nv["block_mask"]:
        assert ext._gen_block_mask(np.array(v["a1"], np.int8), np.array(v["a2"], np.int8)).tolist() == v["mask"]
    for v in nv["combine"]:
        assert ext._combine_matrix_rows(np.array(v["m"], np.int8), np.array(v["idx"], np.int8)).tolist() == v["out"]


def test_golden_rule_tables_on_gpu(gpu):
    """rules_strings.json: the REAL reference's gen_split_mask with other combo matrices installed; the fused kernel's
    rule-table interpreter must reproduce its boundaries AND its values (no oracle involved) -- through every input form:
    UTF-32, UTF-8 in byte space, PEP 393 kind 1 / kind 2 units, each as a small batch (pinned path) and as a batch large
    enough for the tile kernels of that form (k_tiles_main<kModeRules / BytesRules / Latin1Rules / Ucs2Rules / ValuesRules>)."""
    from latok_amd import batch
    try:
        for rs in _golden("rules_strings.json")["sets"]:
            batch.set_rules(np.array(rs["c_split"], np.int8), np.array(rs["c_mask"], np.int8), np.array(rs["c_sym"], np.int8))
            texts = [_text_of(it["cps"]) for it in rs["items"]]
            splits = [np.array(it["splits"], np.uint8) for it in rs["items"]]
            nz = [np.nonzero(v)[0] for v in splits]
            offs = batch.split_offsets_batch(texts)
            for o, e in zip(offs, nz):
                assert o.tolist() == e.tolist(), rs["name"]
            reps = max(1, 300_000 // max(1, sum(len(t) for t in texts))) + 1
            for k in (1, reps):          # small (pinned path) and large (tile kernels)
                tx, sp, zz = texts * k, splits * k, nz * k
                cps, row = pack(tx)
                # the VALUES gen_split_mask returns under these tables (default_tokenizer.py:121-132)
                assert np.array_equal(batch.split_values_batch(cps, row), np.concatenate(sp)), (rs["name"], k, "values")
                c, o = batch.split_offsets_csr(cps, row)
                assert np.array_equal(o, np.concatenate(zz)) and c.tolist() == [len(z) for z in zz], (rs["name"], k, "utf32")
                # UTF-8 in byte space: the same boundaries at the byte positions of their chars
                blobs = [t.encode("utf-8", "surrogatepass") for t in tx]
                u8, boff = batch.pack_utf8(blobs)
                bpos = [np.cumsum([0] + [len(ch.encode("utf-8", "surrogatepass")) for ch in t]) for t in texts] * k
                c, o = batch.split_offsets_utf8_bytes_csr(u8, boff)
                assert o.tolist() == [int(bp[v]) for z, bp in zip(zz, bpos) for v in z], (rs["name"], k, "utf8 bytes")
                bits = batch.split_mask_utf8_bytes_csr(u8, boff)
                assert int(sum(bin(int(w)).count("1") for w in bits)) == sum(len(z) for z in zz)
                # PEP 393 units: the strings that fit the kind
                for kind, limit, enc, dt in ((1, 0x100, "latin-1", np.uint8), (2, 0x10000, "utf-16-le", np.uint16)):
                    sel = [i for i, t in enumerate(tx) if all(ord(ch) < limit for ch in t)]
                    if not sel:
                        continue
                    units = np.frombuffer("".join(tx[i] for i in sel).encode(enc, "surrogatepass"), dt)
                    urow = np.zeros(len(sel) + 1, np.int64)
                    np.cumsum([len(tx[i]) for i in sel], out=urow[1:])
                    c, o = batch.split_offsets_kind_csr(units, urow)
                    assert o.tolist() == [int(v) for i in sel for v in zz[i]], (rs["name"], k, "kind", kind)
    finally:
        batch.reset_rules()


def _device_corpus(lib, seed, model, n_str, lo, hi, sid0=0, prefix=None):
    """Generate a corpus on the device; optionally prepend one host-supplied string (shifts every tile boundary)."""
    from latok_amd import _lib
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, sid0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    d_row = lib.latok_dev_alloc(row.nbytes)
    d_cps = lib.latok_dev_alloc(total * 4)
    _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
    _lib.check(lib.latok_corpus_fill_device(seed, model, sid0, n_str, d_row, d_cps, None))
    cps = np.zeros(total, np.uint32)
    _lib.check(lib.latok_memcpy_d2h(cps.ctypes.data, d_cps, cps.nbytes))
    lib.latok_dev_free(d_row)
    lib.latok_dev_free(d_cps)
    if prefix is not None:
        cps = np.concatenate([prefix, cps])
        row = np.concatenate([[0], row + len(prefix)])
    return cps, row


def test_golden_corpus_hashes_device_generator_and_kernel(gpu):
    """F7 end to end without the reference: device-generated corpus and GPU offsets hash like the reference's."""
    import hashlib
    from latok_amd import batch
    for name, c in _golden("corpus_samples.json")["corpora"].items():
        cps, row = _device_corpus(gpu, c["seed"], c["model"], c["n_str"], c["len_lo"], c["len_hi"])
        assert hashlib.sha256(cps.astype("<u4").tobytes()).hexdigest() == c["sha256_cps_u32le"], name
        counts, offs = batch.split_offsets_csr(cps, row)
        assert int(counts.sum()) == c["n_boundaries"]
        assert hashlib.sha256(offs.astype("<i8").tobytes()).hexdigest() == c["sha256_offsets_i64le"], name


@pytest.mark.parametrize("workload", ["C2", "C3"])
def test_full_size_properties(gpu, oracle, workload):
    """BASELINE full sizes (1 M strings): size-independent properties + oracle parity on samples."""
    import hashlib
    from latok_amd import _lib, batch
    seed, model, lo, hi = {"C2": (0x1A70C0DE, 0, 64, 192), "C3": (0x1A70C0DF, 1, 128, 384)}[workload]
    n_str = 1_000_000
    cps, row = _device_corpus(gpu, seed, model, n_str, lo, hi)
    total = int(row[-1])
    bits = batch.split_mask_batch(cps, row)
    flags = bits_to_bool(bits, total)
    # (1) every string start is a boundary; nothing beyond the last char
    assert flags[row[:-1]].all()
    # (2) idempotent / deterministic, and host-pointer path == device-pointer path
    assert np.array_equal(bits, batch.split_mask_batch(cps, row))
    counts, offs = batch.split_offsets_csr(cps, row)
    assert int(counts.sum()) == int(flags.sum()) == offs.size
    # (3) offsets are per-string ascending, start with 0, and re-create the bitmask
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    assert (offs[starts] == 0).all()
    glob = offs + np.repeat(row[:-1], counts)
    assert (np.diff(glob) > 0).all()
    rebuilt = np.zeros(total, bool)
    rebuilt[glob] = True
    assert np.array_equal(rebuilt, flags)
    # (4) translation invariance: a string's boundaries do not depend on where it sits in the packed buffer
    #     (prepending 1777 chars moves every tile / word boundary); compared through a checksum of checksums
    pre = np.frombuffer(("x" * 1776 + " ").encode("utf-32-le"), dtype="<u4").astype(np.uint32)
    cps2, row2 = _device_corpus(gpu, seed, model, n_str, lo, hi, prefix=pre)
    counts2, offs2 = batch.split_offsets_csr(cps2, row2)
    assert np.array_equal(counts2[1:], counts)
    assert hashlib.sha256(offs2[counts2[0]:].tobytes()).hexdigest() == hashlib.sha256(offs.tobytes()).hexdigest()
    # (5) oracle parity on a head, a middle and a tail sample (the oracle needs seconds for these)
    for lo_s, hi_s in ((0, 20000), (n_str // 2, n_str // 2 + 5000), (n_str - 5000, n_str)):
        sub_row = row[lo_s:hi_s + 1] - row[lo_s]
        sub = cps[row[lo_s]:row[hi_s]]
        ov, _ = oracle.split_batch(sub, sub_row, want_bits=False)
        assert np.array_equal(ov != 0, flags[row[lo_s]:row[hi_s]])


def test_long_documents(gpu, oracle):
    """BASELINE configs[4] shape at a reduced count: 12 documents x 1 M chars, oracle parity on all of them."""
    cps, row = _device_corpus(gpu, 0x1A70C0E0, 0, 12, 1_000_000, 1_000_000)
    from latok_amd import batch
    bits = batch.split_mask_batch(cps, row)
    _, ob = oracle.split_batch(cps, row, want_values=False)
    assert np.array_equal(ob, bits)


def test_one_huge_document_between_small_strings(gpu, oracle):
    """One 24 M-char document (5 860 tiles: several segments per workgroup, its entry of the tile index is written by a
    whole wave) between tiny strings and empty ones: mask, offsets and token spans against the oracle / the small-batch
    forms of the same strings."""
    from latok_amd import batch
    big_cps, _ = _device_corpus(gpu, 0x1A70C0E1, 0, 1, 24_000_000, 24_000_000)
    small = ["ab c", "", "x@y.z", "", "#tag http://a.b"]
    head, _ = pack(small[:3])
    tail, _ = pack(small[3:])
    cps = np.concatenate([head, big_cps, tail])
    lens = [len(t) for t in small[:3]] + [big_cps.size] + [len(t) for t in small[3:]]
    row = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=row[1:])
    ov, ob = oracle.split_batch(cps, row)
    assert np.array_equal(batch.split_mask_batch(cps, row), ob)
    counts, offs = batch.split_offsets_csr(cps, row)
    exp = [np.nonzero(ov[row[s]:row[s + 1]])[0] for s in range(len(lens))]
    assert np.array_equal(counts, [len(e) for e in exp]) and np.array_equal(offs, np.concatenate(exp))
    # token spans of the document alone == its spans inside the batch (string relative)
    c_all, s_all = batch.token_spans_csr(cps, row)
    c_doc, s_doc = batch.token_spans_csr(big_cps, np.array([0, big_cps.size], np.int64))
    k0 = int(c_all[:3].sum())
    assert c_all[3] == c_doc[0] and np.array_equal(s_all[k0:k0 + int(c_doc[0])], s_doc)


def test_token_spans_on_device(gpu, oracle):
    """latok_token_spans_batch == the reference's slice/strip/drop-empty loop (default_tokenizer.py:149-158)."""
    from latok_amd import batch
    rng = random.Random(77)
    texts = [G1, "", " ", "  lead and trail  ", "a", "x\t\ny", "tab\tsep nbsp ls　ideo  ", "   ", "a  b   c",
             "foo@bar.com, .@user hi", "http://a@b X,y z", "日本語のテキスト、です。 🤓 ok "] + \
        random_strings(rng, 300, 0, 120, ALPHABETS["mixed"]) + random_strings(rng, 50, 0, 400, ALPHABETS["words"]) + \
        random_strings(rng, 3, 5000, 20000, ALPHABETS["mixed"]) + random_strings(rng, 100, 0, 30, list("ab \t\n"))
    got = batch.tokenize_batch(texts)
    # the span form of the same call (no per-token Python objects): identical tokens, lazily sliced
    ts = batch.token_spans_batch(texts)
    assert len(ts) == len(texts) and int(ts.counts.sum()) == len(ts.spans) and list(ts) == got
    from latok_amd.core import default_tokenizer as dtk
    assert list(dtk.tokenize_spans(texts[:20])) == got[:20]
    for t, g in zip(texts, got):
        want = oracle.tokenize(t) if t else []
        assert g == want, (t[:80], g[:10], want[:10])
    cps, row = pack(texts)
    counts, spans = batch.token_spans_csr(cps, row)
    assert counts.sum() == len(spans) and (spans[:, 1] > spans[:, 0]).all()
    assert counts.tolist() == [len(x) for x in got]


def test_featurize_on_device(gpu, oracle):
    """featurize(): per-token sum of matrix rows over the unstripped span (default_tokenizer.py:163-191), against the
    oracle's matrix; includes tokens beyond char 127 (where the reference's int8 row index overflows) and > 255 chars
    (uint8 wrap-around of the sums)."""
    from latok_amd import batch
    from latok_amd.core import default_tokenizer as dt
    rng = random.Random(5)
    texts = [G1, "a", " x ", "http://" + "a" * 300 + " tail", "foo@bar.com, .@user hi #tag"] + \
        random_strings(rng, 120, 1, 200, ALPHABETS["mixed"]) + random_strings(rng, 30, 1, 300, ALPHABETS["words"])
    got = batch.featurize_batch(texts)
    for t, toks in zip(texts, got):
        m = oracle.gen_parse_matrix(t).astype(np.uint8)
        nz = oracle.split_offsets(t).tolist() + [len(t)]
        want = []
        for a, b in zip(nz[:-1], nz[1:]):
            if t[a:b].strip():
                want.append((t[a:b].strip(), a, b, m[a:b].sum(axis=0, dtype=np.uint64).astype(np.uint8).astype(np.int8)))
        assert len(toks) == len(want), t
        for tok, (txt, a, b, f) in zip(toks, want):
            assert (tok.text, tok.start_idx, tok.end_idx) == (txt, a, b)
            assert tok.features.dtype == np.int8 and np.array_equal(tok.features, f), (t[a:b], tok.features, f)
    one = list(dt.featurize(G1))
    assert [x.text for x in one] == oracle.tokenize(G1)
    assert one[3].feature_weights()["Twitter"] == 1 and one[3].weight() > 0


def test_beyond_4GiB_batch(gpu, oracle):
    """One GPU's share of BASELINE configs[3]: 12.5 M strings, 1.6e9 chars = 6.4 GB of code points, device resident
    (byte offsets exceed 32 bits).  Device-pointer API; oracle parity on slices from the head, the middle and the tail."""
    from latok_amd import _lib
    lib = gpu
    n_str, seed, model, lo, hi = 12_500_000, 0x1A70C0DE, 0, 64, 192
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    assert total * 4 > 2 ** 32
    n_words = (total + 63) // 64
    d_row, d_cps, d_bits = lib.latok_dev_alloc(row.nbytes), lib.latok_dev_alloc(total * 4), lib.latok_dev_alloc(n_words * 8)
    assert d_row and d_cps and d_bits
    try:
        _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
        _lib.check(lib.latok_corpus_fill_device(seed, model, 0, n_str, d_row, d_cps, None))
        _lib.check(lib.latok_split_mask_batch(d_cps, d_row, n_str, total, d_bits, _lib.DEVICE_PTRS, None))
        _lib.check(lib.latok_sync())
        bits = np.zeros(n_words, np.uint64)
        _lib.check(lib.latok_memcpy_d2h(bits.ctypes.data, d_bits, bits.nbytes))
        for s0 in (0, n_str // 2, n_str - 4000):
            s1 = s0 + 4000
            c0, c1 = int(row[s0]), int(row[s1])
            cps = np.zeros(c1 - c0, np.uint32)
            _lib.check(lib.latok_memcpy_d2h(cps.ctypes.data, d_cps + 4 * c0, cps.nbytes))
            ov, _ = oracle.split_batch(cps, row[s0:s1 + 1] - c0, want_bits=False)
            w0, w1 = c0 // 64, (c1 + 63) // 64
            got = bits_to_bool(bits[w0:w1], (w1 - w0) * 64)[c0 - w0 * 64:c1 - w0 * 64]
            assert np.array_equal(got, ov != 0), s0
        # every string start is a boundary, nothing set beyond the last char
        starts = row[:-1]
        assert ((bits[starts >> 6] >> (starts & 63).astype(np.uint64)) & np.uint64(1)).all()
        if total % 64:
            assert int(bits[-1]) >> (total % 64) == 0
    finally:
        for p in (d_row, d_cps, d_bits):
            lib.latok_dev_free(p)


def test_positions_beyond_int32(gpu, oracle):
    """2.3e9 chars in ONE batch (9.5 GB of code points): character positions, bitmask word indices times 64 and token
    ranks near the tail no longer fit 32 bits.  Mask against the oracle on slices from the head, the 2^31 crossing and
    the tail; offsets / spans / featurize of the tail against the same strings run as a small batch of their own (the
    results are per string, so they must not depend on where the string sits); byte space on the same text as UTF-8
    (ASCII corpus: byte positions == char positions, so the two bitmasks must be identical)."""
    from latok_amd import _lib, batch
    lib = gpu
    n_str, seed, model, lo, hi = 18_000_000, 0x1A70C0DE, 0, 64, 192
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    assert total > 2 ** 31 + 10_000_000
    n_words = (total + 63) // 64
    D = _lib.DEVICE_PTRS
    ptrs = []

    def alloc(nbytes):
        p = lib.latok_dev_alloc(nbytes)
        assert p, _lib.last_error()
        ptrs.append(p)
        return p

    def back(p, shape, dtype, skip=0):
        out = np.empty(shape, dtype)
        if out.nbytes:
            _lib.check(lib.latok_memcpy_d2h(out.ctypes.data, p + skip, out.nbytes))
        return out

    try:
        d_row, d_cps, d_bits = alloc(row.nbytes), alloc(total * 4), alloc(n_words * 8 + 8)
        _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
        _lib.check(lib.latok_corpus_fill_device(seed, model, 0, n_str, d_row, d_cps, None))
        _lib.check(lib.latok_split_mask_batch(d_cps, d_row, n_str, total, d_bits, D, None))
        _lib.check(lib.latok_sync())
        bits = back(d_bits, n_words, np.uint64)
        s_cross = int(np.searchsorted(row, 2 ** 31)) - 2000
        n_tail = 3000
        for s0 in (0, s_cross, n_str - n_tail):
            s1 = s0 + (4000 if s0 != n_str - n_tail else n_tail)
            c0, c1 = int(row[s0]), int(row[s1])
            cps = back(d_cps, c1 - c0, np.uint32, skip=4 * c0)
            ov, _ = oracle.split_batch(cps, row[s0:s1 + 1] - c0, want_bits=False)
            w0, w1 = c0 // 64, (c1 + 63) // 64
            got = bits_to_bool(bits[w0:w1], (w1 - w0) * 64)[c0 - w0 * 64:c1 - w0 * 64]
            assert np.array_equal(got, ov != 0), s0
        starts = row[:-1]
        assert ((bits[starts >> 6] >> (starts & 63).astype(np.uint64)) & np.uint64(1)).all()
        if total % 64:
            assert int(bits[-1]) >> (total % 64) == 0
        n_bound = int(np.bitwise_count(bits).sum())

        # the tail strings as a batch of their own (host pointers, positions < 2^20)
        s0 = n_str - n_tail
        c0 = int(row[s0])
        t_cps = back(d_cps, total - c0, np.uint32, skip=4 * c0)
        t_row = np.ascontiguousarray(row[s0:] - c0)
        d_counts = alloc(n_str * 8)
        nout = C.c_int64(0)
        cap = n_bound + 8
        d_items = alloc(cap * 32)
        d_feat = alloc(cap * 25)

        def tail(width, dtype=np.int64, p=None):
            counts = back(d_counts, n_str, np.int64)
            assert int(counts.sum()) == nout.value
            k0 = int(counts[:s0].sum())
            item = width * np.dtype(dtype).itemsize
            return counts[s0:], back(p or d_items, (nout.value - k0, width), dtype, skip=k0 * item)

        _lib.check(lib.latok_split_offsets_batch(d_cps, d_row, n_str, total, d_counts, d_items, cap, C.byref(nout), D, None))
        assert nout.value == n_bound
        cnt, items = tail(1)
        w_cnt, w_items = batch.split_offsets_csr(t_cps, t_row)
        assert np.array_equal(cnt, w_cnt) and np.array_equal(items.ravel(), w_items)
        _lib.check(lib.latok_token_spans_batch(d_cps, d_row, n_str, total, d_counts, d_items, cap, C.byref(nout), D, None))
        cnt, items = tail(2)
        w_cnt, w_items = batch.token_spans_csr(t_cps, t_row)
        assert np.array_equal(cnt, w_cnt) and np.array_equal(items, w_items)
        n_tok = nout.value
        _lib.check(lib.latok_token_features_batch(d_cps, d_row, n_str, total, d_counts, d_items, d_feat, cap, C.byref(nout), D, None))
        assert nout.value == n_tok
        cnt, items = tail(4)
        _, feats = tail(25, np.int8, d_feat)
        w_cnt, w_items, w_feats = batch.token_features_csr(t_cps, t_row)
        assert np.array_equal(cnt, w_cnt) and np.array_equal(items, w_items) and np.array_equal(feats, w_feats)

        # byte space over the same text (ASCII: one byte per char, identical positions)
        lib.latok_dev_free(ptrs.pop())   # d_feat
        u8 = np.empty(total, np.uint8)
        step = 1 << 28
        for a in range(0, total, step):
            b = min(total, a + step)
            chunk = back(d_cps, b - a, np.uint32, skip=4 * a)
            assert int(chunk.max()) < 0x80
            u8[a:b] = chunk
        d_u8 = alloc(total + 64)
        _lib.check(lib.latok_memcpy_h2d(d_u8, u8.ctypes.data, total))
        _lib.check(lib.latok_memset_dev(d_bits, 0, n_words * 8))
        _lib.check(lib.latok_split_mask_utf8_bytes_batch(d_u8, d_row, n_str, total, d_bits, D, None))
        _lib.check(lib.latok_sync())
        assert np.array_equal(back(d_bits, n_words, np.uint64), bits)
        _lib.check(lib.latok_token_spans_utf8_bytes_batch(d_u8, d_row, n_str, total, d_counts, d_items, cap, C.byref(nout), D, None))
        assert nout.value == n_tok
        cnt, items = tail(2)
        assert np.array_equal(cnt, w_cnt) and np.array_equal(items, w_items[:, 2:4] if w_items.shape[1] == 4 else w_items)
    finally:
        for p in ptrs:
            lib.latok_dev_free(p)


@pytest.mark.parametrize("alpha,dtype", [("latin1", np.uint8), ("bmp", np.uint16), ("words", np.uint8), ("mixed", np.uint32)])
def test_pep393_kinds(gpu, oracle, alpha, dtype):
    """The reference's own input format (PyUnicode_KIND 1 / 2 / 4, latok.c:53-55,79): the narrow code units go to the
    device as they are.  Mask against the oracle on the same text; offsets / spans / featurize against the UTF-32 entry
    points; run-time rule tables; tiny, ragged, empty and multi-tile batches (tile tails, halo chars across tiles)."""
    from latok_amd import batch
    from conftest import RULE_SETS, oracle_rule_bits
    rng = random.Random(393 + len(alpha))
    A = ALPHABETS[alpha]
    batches = [
        [G1 if alpha != "bmp" else G1 + "\u65e5\u672c"],
        random_strings(rng, 300, 0, 60, A) + ["", ""],
        random_strings(rng, 700, 0, 90, A) + random_strings(rng, 3, 3000, 9000, A) + ["", "x"],
        random_strings(rng, 2, 4096, 4096, A) + random_strings(rng, 1, 4095, 4095, A) + random_strings(rng, 3, 1, 2, A),
        random_strings(rng, 2000, 1, 3, A),
    ]
    for texts in batches:
        units, row = batch.pack_kind(texts)
        if sum(map(len, texts)) > 200:
            assert units.dtype == dtype
        cps, row32 = pack(texts)
        assert np.array_equal(row, row32) and np.array_equal(units.astype(np.uint32), cps)
        ov, ob = oracle.split_batch(cps, row)
        assert np.array_equal(batch.split_mask_kind_csr(units, row), ob)
        c1, o1 = batch.split_offsets_kind_csr(units, row)
        exp = [np.nonzero(ov[row[s]:row[s + 1]])[0] for s in range(len(texts))]
        assert np.array_equal(c1, [len(e) for e in exp]) and np.array_equal(o1, np.concatenate(exp))
        c2, s2 = batch.token_spans_kind_csr(units, row)
        w2 = batch.token_spans_csr(cps, row)
        assert np.array_equal(c2, w2[0]) and np.array_equal(s2, w2[1])
        c3, s3, f3 = batch.token_features_kind_csr(units, row)
        w3 = batch.token_features_csr(cps, row)
        assert np.array_equal(c3, w3[0]) and np.array_equal(s3, w3[1]) and np.array_equal(f3, w3[2])
        if batch._narrow_pays(texts) and len(texts) < 1000:
            # the host API ships such a batch as narrow units by itself: tokens == the reference's, string by string
            assert batch.tokenize_batch(texts) == [oracle.tokenize(t) if t else [] for t in texts]
        if int(row[-1]) < 40000:
            tables = RULE_SETS["sym_everywhere"]
            batch.set_rules(*tables)
            try:
                assert np.array_equal(batch.split_mask_kind_csr(units, row), oracle_rule_bits(oracle, texts, tables))
                c4, o4 = batch.split_offsets_kind_csr(units, row)
                w4 = batch.split_offsets_csr(cps, row)
                assert np.array_equal(c4, w4[0]) and np.array_equal(o4, w4[1])
            finally:
                batch.reset_rules()
    # all-empty batch, no strings, bad kind
    u0 = np.zeros(0, dtype)
    assert batch.split_mask_kind_csr(u0, np.zeros(4, np.int64)).size == 0
    c, o = batch.split_offsets_kind_csr(u0, np.zeros(4, np.int64))
    assert c.tolist() == [0, 0, 0] and o.size == 0
    assert batch.token_spans_kind_csr(u0, np.zeros(1, np.int64))[0].size == 0
    from latok_amd import _lib
    with pytest.raises(ValueError):
        _lib.check(gpu.latok_split_mask_kind_batch(None, 3, None, 0, 0, None, 0, None))


def test_pep393_kinds_device_pointers(gpu, oracle):
    """Kind 1 / 2 units resident in HBM (LATOK_DEVICE_PTRS), outputs in HBM: same results as the host-pointer forms."""
    from latok_amd import _lib, batch
    lib = gpu
    D = _lib.DEVICE_PTRS
    rng = random.Random(3932)
    for alpha in ("latin1", "bmp"):
        A = ALPHABETS[alpha]
        texts = random_strings(rng, 900, 0, 120, A) + random_strings(rng, 4, 5000, 9000, A) + ["", "y"]
        units, row = batch.pack_kind(texts)
        kind = units.dtype.itemsize
        n, total = len(texts), int(row[-1])
        ptrs = []

        def dev(a=None, nbytes=0):
            p = lib.latok_dev_alloc((a.nbytes if a is not None else nbytes) + 64)
            assert p
            ptrs.append(p)
            if a is not None:
                _lib.check(lib.latok_memcpy_h2d(p, a.ctypes.data, a.nbytes))
            return p

        def back(p, shape, dtype):
            out = np.empty(shape, dtype)
            if out.nbytes:
                _lib.check(lib.latok_memcpy_d2h(out.ctypes.data, p, out.nbytes))
            return out

        try:
            d_units, d_row = dev(units), dev(row)
            words = (total + 63) // 64
            d_bits, d_counts, d_items, d_feat = dev(nbytes=words * 8), dev(nbytes=n * 8), dev(nbytes=total * 32), dev(nbytes=total * 25)
            nout = C.c_int64(0)
            _lib.check(lib.latok_split_mask_kind_batch(d_units, kind, d_row, n, total, d_bits, D, None))
            _lib.check(lib.latok_sync())
            assert np.array_equal(back(d_bits, words, np.uint64), batch.split_mask_kind_csr(units, row))
            _lib.check(lib.latok_split_offsets_kind_batch(d_units, kind, d_row, n, -1, d_counts, d_items, total, C.byref(nout), D, None))
            hc, ho = batch.split_offsets_kind_csr(units, row)
            assert nout.value == ho.size and np.array_equal(back(d_counts, n, np.int64), hc)
            assert np.array_equal(back(d_items, nout.value, np.int64), ho)
            _lib.check(lib.latok_token_spans_kind_batch(d_units, kind, d_row, n, total, d_counts, d_items, total, C.byref(nout), D, None))
            hc, hs = batch.token_spans_kind_csr(units, row)
            assert nout.value == len(hs) and np.array_equal(back(d_items, (nout.value, 2), np.int64), hs)
            _lib.check(lib.latok_token_features_kind_batch(d_units, kind, d_row, n, total, d_counts, d_items, d_feat, total, C.byref(nout), D, None))
            hc, hs, hf = batch.token_features_kind_csr(units, row)
            assert nout.value == len(hs) and np.array_equal(back(d_items, (nout.value, 4), np.int64), hs)
            assert np.array_equal(back(d_feat, (nout.value, 25), np.int8), hf)
        finally:
            for p in ptrs:
                lib.latok_dev_free(p)


def test_utf8_ingest(gpu, oracle):
    """UTF-8 CSR input: device decode == Python's decoder, and offsets / spans equal the UTF-32 path."""
    from latok_amd import batch
    rng = random.Random(123)
    texts = [G1, "", "\u65e5\u672c\u8a9e\u306e\u30c6\u30ad\u30b9\u30c8\u3001\u3067\u3059\u3002 \U0001f913 ok", "\u00e9\u00e0 x",
             "\U0010ffff max  \u07ff \u0800 \uffff", "a"] + \
        random_strings(rng, 400, 0, 150, ALPHABETS["mixed"]) + random_strings(rng, 5, 3000, 9000, ALPHABETS["mixed"])
    blobs = [t.encode("utf-8") for t in texts]
    u8, boff = batch.pack_utf8(blobs)
    cps, row = batch.utf8_decode_csr(u8, boff)
    want_cps, want_row = pack(texts)
    assert np.array_equal(row, want_row) and np.array_equal(cps, want_cps)
    c1, o1 = batch.split_offsets_utf8_csr(u8, boff)
    c2, o2 = batch.split_offsets_csr(want_cps, want_row)
    assert np.array_equal(c1, c2) and np.array_equal(o1, o2)
    exp = [oracle.split_offsets(t) if t else np.zeros(0, np.int64) for t in texts]
    assert np.array_equal(o1, np.concatenate(exp))
    mb, mrow = batch.split_mask_utf8_csr(u8, boff)
    assert np.array_equal(mrow, want_row) and np.array_equal(mb, batch.split_mask_batch(want_cps, want_row))
    t1, s1 = batch.token_spans_utf8_csr(u8, boff)
    t2, s2 = batch.token_spans_csr(want_cps, want_row)
    assert np.array_equal(t1, t2) and np.array_equal(s1, s2)
    # lone surrogates survive the round trip ("surrogatepass"), empty batch is fine
    odd = ["\ud800 x \udfff"]
    cps, row = batch.utf8_decode_csr(*batch.pack_utf8([t.encode("utf-8", "surrogatepass") for t in odd]))
    assert np.array_equal(cps, pack(odd)[0])
    assert batch.split_offsets_utf8_csr(np.zeros(0, np.uint8), np.zeros(1, np.int64))[1].size == 0


def test_utf8_code_point_mask_without_a_utf32_copy(gpu, oracle):
    """latok_split_mask_utf8_batch on batches beyond the small-batch size: the byte-space tile kernel + the mask packed at the
    lead bytes (api.cpp mask_utf8_via_bytes, compact_kernels.hip k_lead_compress) must give what the reference gives for the
    decoded str (latok.c:53-55,79 reads code points): mask and code-point row offsets against the oracle through the UTF-32 path,
    host and device pointers, ASCII-only text, empty strings at both ends, a mask buffer that is too small, run-time rule tables.
    Malformed bytes: whatever the staged decoder's rule (one char per lead byte) gives -- batches the byte-space model treats
    differently are detected on the device and take the decoder."""
    from conftest import RULE_SETS, oracle_rule_bits
    from latok_amd import _lib, batch
    rng = random.Random(2718)
    alpha = ALPHABETS["mixed"] + list("é日🤓ü\u3000Жδ") + ["http://é", "a@日", ".@ü", "#日"]
    sets = [
        ["", ""] + random_strings(rng, 6000, 0, 120, alpha) + ["", "x", ""],
        random_strings(rng, 3, 60000, 90000, alpha) + random_strings(rng, 2000, 0, 60, alpha),
        random_strings(rng, 5000, 0, 150, list("abc XYZ,.#@:/ 19\t")),                     # ASCII only
        ["日" * 100000, "é" * 70000, "🤓" * 50000, "a" * 3 + "日" * 4093 + " b"],          # long runs, every sequence length
    ]
    for i, texts in enumerate(sets):
        blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
        u8, boff = batch.pack_utf8(blobs)
        assert u8.size > 262144
        cps, row = pack(texts)
        want = oracle.split_batch(cps, row, want_values=False)[1]
        got, got_row = batch.split_mask_utf8_csr(u8, boff)
        assert np.array_equal(got_row, row), i
        assert np.array_equal(got, want), i
        # offsets and token spans in code-point units take the same road (the compaction runs on the packed masks)
        for dtype in (np.int64, np.int32):
            c1, o1 = batch.split_offsets_utf8_csr(u8, boff, dtype=dtype)
            c2, o2 = batch.split_offsets_csr(cps, row, dtype=dtype)
            assert np.array_equal(c1, c2) and np.array_equal(o1, o2), (i, dtype)
            t1, s1 = batch.token_spans_utf8_csr(u8, boff, dtype=dtype)
            t2, s2 = batch.token_spans_csr(cps, row, dtype=dtype)
            assert np.array_equal(t1, t2) and np.array_equal(s1, s2), (i, dtype)
        assert np.array_equal(o1, np.concatenate([np.nonzero(bits_to_bool(want, int(row[-1]))[row[k]:row[k + 1]])[0] for k in range(len(texts))]))
        # device pointers, the mask buffer exactly as large as the code points need (fewer words than the bytes have)
        words = (int(row[-1]) + 63) // 64
        d = {k: gpu.latok_dev_alloc(n + 64) for k, n in (("u8", u8.nbytes), ("boff", boff.nbytes), ("mask", words * 8), ("row", row.nbytes))}
        try:
            _lib.check(gpu.latok_memcpy_h2d(d["u8"], u8.ctypes.data, u8.nbytes))
            _lib.check(gpu.latok_memcpy_h2d(d["boff"], boff.ctypes.data, boff.nbytes))
            _lib.check(gpu.latok_memset_dev(d["mask"], 0xEE, words * 8))
            tcp = C.c_int64(0)
            _lib.check(gpu.latok_split_mask_utf8_batch(d["u8"], d["boff"], len(texts), -1, d["mask"], words, d["row"], C.byref(tcp),
                                                       _lib.DEVICE_PTRS, None))
            assert tcp.value == int(row[-1])
            m2, r2 = np.empty(words, np.uint64), np.empty_like(row)
            _lib.check(gpu.latok_memcpy_d2h(m2.ctypes.data, d["mask"], m2.nbytes))
            _lib.check(gpu.latok_memcpy_d2h(r2.ctypes.data, d["row"], r2.nbytes))
            assert np.array_equal(m2, want) and np.array_equal(r2, row), i
            if words > 1:      # too small by one word: refused, the needed size is reported
                rc = gpu.latok_split_mask_utf8_batch(d["u8"], d["boff"], len(texts), -1, d["mask"], words - 1, d["row"], C.byref(tcp),
                                                     _lib.DEVICE_PTRS, None)
                assert rc == _lib.ERR_INVALID and b"mask_cap_words" in gpu.latok_last_error() and tcp.value == int(row[-1])
        finally:
            for p_ in d.values():
                gpu.latok_dev_free(p_)
    # run-time rule tables
    texts = sets[0]
    u8, boff = batch.pack_utf8([t.encode("utf-8", "surrogatepass") for t in texts])
    for name in ("sym_everywhere", "all_columns"):
        batch.set_rules(*RULE_SETS[name])
        try:
            got, _ = batch.split_mask_utf8_csr(u8, boff)
        finally:
            batch.reset_rules()
        assert np.array_equal(got, oracle_rule_bits(oracle, texts, RULE_SETS[name])), name
    # malformed bytes: strings of chunks (truncated sequences, lone leads, stray continuation bytes ...); every third batch also
    # holds what the byte-space model treats differently (a run of > 3 continuation bytes, a string that starts with one)
    chunks = [b"a", b"b", b"Z", b" ", b" ", b".", b"@", b"#", b":", b"/", b"\t", b"1", b"\xc3\xa9", b"\xe3\x81\x82", b"\xe6\x97\xa5",
              b"\xf0\x9f\xa4\x93", b"\xe3\x80\x80", b"\xe3\x81", b"\xc3", b"\xf0\x9f", b"\xf0\x9f\xa4", b"a\x80", b".\x80", b"@\xbf\x80",
              b"\xc3\xa9\x80", b"\xff", b"\xf8\x80\x80\x80", b"\xe3\x81\x82\x80", b"\xd0\x96", b"http://", b"\xed\xa0\x80"]
    odd_chunks = [b"\x80\x80\x80\x80", b"\xf0\x9f\xa4\x93\x80", b"\x80"]
    for it in range(6):
        blobs = []
        for s_ in range(6000):
            parts = [rng.choice(chunks) for _ in range(rng.randint(0, 60))]
            if it % 3 == 2 and s_ % 500 == 7:
                parts.insert(0 if s_ % 1000 == 7 else len(parts) // 2, rng.choice(odd_chunks))
            blobs.append(b"".join(parts))
        u8, boff = batch.pack_utf8(blobs)
        assert u8.size > 262144
        cps, row = batch.utf8_decode_csr(u8, boff)               # the staged decoder: one code point per lead byte
        is_lead = (u8 & 0xC0) != 0x80
        assert cps.size == int(is_lead.sum())
        got, got_row = batch.split_mask_utf8_csr(u8, boff)
        assert np.array_equal(got_row, row), it
        assert np.array_equal(got, batch.split_mask_batch(cps, row)), it
        c1, o1 = batch.split_offsets_utf8_csr(u8, boff, dtype=np.int32)
        c2, o2 = batch.split_offsets_csr(cps, row, dtype=np.int32)
        assert np.array_equal(c1, c2) and np.array_equal(o1, o2), it
        t1, s1 = batch.token_spans_utf8_csr(u8, boff)
        t2, s2 = batch.token_spans_csr(cps, row)
        assert np.array_equal(t1, t2) and np.array_equal(s1, s2), it
    # a sequence cut short by the END OF THE BATCH is U+FFFD too (the staged decoder used to complete it with its own padding)
    for tail in (b"\xc3", b"\xe6\x97", b"\xf0\x9f\xa4", b"x\xf0"):
        blobs = [b"filler words, a #tag and http://x.y/z "] * 8000 + [b"end " + tail]
        u8, boff = batch.pack_utf8(blobs)
        cps, row = batch.utf8_decode_csr(u8, boff)
        assert cps[-1] == 0xFFFD and row[-1] == cps.size == int(((u8 & 0xC0) != 0x80).sum()), tail
        got, got_row = batch.split_mask_utf8_csr(u8, boff)
        assert np.array_equal(got_row, row) and np.array_equal(got, batch.split_mask_batch(cps, row)), tail


def test_edge_aligned_patterns(gpu, oracle):
    """Rule triggers, spaces, string starts / ends and empty strings placed exactly around word (64) and tile (4096)
    boundaries of the packed buffer, where the cross-lane / cross-tile carries of the kernel live."""
    rng = random.Random(4096)
    specials = ["http://x.y/z", " #tag", "a@b.c", " .@user", " ", "  ", "", "X", "aB", "Ab", "a,", ", ", "://", "@", "#a #b",
                "a@b@c@d e,f g,h", "\t", "x" * 70, "日本", "🤓"]
    for trial in range(12):
        texts, pos = [], 0
        # filler strings whose lengths steer `pos` onto interesting offsets, followed by a special
        targets = sorted(set([64 * k + d for k in (1, 2, 63, 64, 65, 127, 128) for d in (-3, -2, -1, 0, 1, 2)] +
                             [4096 * k + d for k in (1, 2, 3) for d in (-66, -65, -64, -2, -1, 0, 1, 2, 63, 64, 65)]))
        for tgt in targets:
            gap = tgt - pos
            if gap < 0:
                continue
            n_fill = rng.randint(1, 3)
            cuts = sorted(rng.randint(0, gap) for _ in range(n_fill - 1))
            prev = 0
            for c in cuts + [gap]:
                texts.append("".join(rng.choice("abc de,F") for _ in range(c - prev)))
                prev = c
            pos = tgt
            sp = rng.choice(specials)
            if rng.random() < 0.5:
                texts.append(sp)                      # special as its own string, starting exactly at the target
            else:
                texts[-1] = texts[-1] + sp            # or glued to the filler, straddling the target
            pos += len(sp)
        _check_batch(oracle, texts)


@pytest.mark.parametrize("name", ["default", "sym_everywhere", "no_mask", "all_starts", "all_columns", "random"])
def test_runtime_rule_tables(gpu, oracle, name):
    """latok_set_rules: the reference's extension point (other C_SPLIT / C_MASK / C_SYM combo matrices,
    default_tokenizer.py:9-30,108-134) evaluated inside the fused kernel, against the reference recipe run on the same
    tables by the oracle; bitmask, offsets and token spans."""
    from conftest import RULE_SETS, oracle_rule_bits, random_rule_tables
    from latok_amd import batch
    rng = random.Random(hash(name) & 0xFFFF)
    try:
        for rep in range(5 if name == "random" else 2):
            tables = random_rule_tables(rng) if name == "random" else RULE_SETS[name]
            batch.set_rules(*tables)
            assert batch.rules_active()
            for kind, n, lo, hi in [("mixed", 300, 0, 60), ("starts", 40, 0, 400), ("mixed", 3, 4000, 20000),
                                    ("rare_space_at", 3, 5000, 30000), ("words", 100, 0, 200)]:
                texts = random_strings(rng, rng.randint(1, n), lo, hi, ALPHABETS[kind])
                cps, row = pack(texts)
                total = int(row[-1])
                want = oracle_rule_bits(oracle, texts, tables)
                got = batch.split_mask_batch(cps, row)
                assert np.array_equal(got, want), (name, rep, kind)
                counts, offs = batch.split_offsets_csr(cps, row)
                flags = bits_to_bool(want, total)
                exp = [np.nonzero(flags[row[s]:row[s + 1]])[0] for s in range(len(texts))]
                assert np.array_equal(counts, [len(e) for e in exp])
                assert np.array_equal(offs, np.concatenate(exp))
                # token spans = the reference's slice / strip / drop-empty loop over those offsets
                toks = batch.tokenize_batch(texts[:50])
                for t, e, g in zip(texts[:50], exp, toks):
                    cut = [int(x) for x in e] + [len(t)]
                    assert g == [w for w in (t[a:b].strip() for a, b in zip(cut, cut[1:])) if w]
                if name == "default":
                    batch.reset_rules()
                    assert np.array_equal(batch.split_mask_batch(cps, row), want)
                    batch.set_rules(*tables)
            # the VALUES under custom tables (rows that hold, times the mask, plus C_SYM rows), small and large batches
            for texts in (["abc", "a #b c@d.e http://x/y Z"], random_strings(rng, 400, 500, 900, ALPHABETS["mixed"])):
                cps, row = pack(texts)
                want_v = np.concatenate([oracle.split_values_rules(t, *tables).astype(np.uint8) for t in texts])
                assert np.array_equal(batch.split_values_batch(cps, row), want_v), (name, rep, "values")
            # byte space and PEP 393 units, batches beyond the small-batch path
            texts = random_strings(rng, 500, 300, 900, ALPHABETS["mixed"] + list("éЖ日🤓"))
            flags = [oracle.split_values_rules(t, *tables) != 0 for t in texts]
            blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
            u8, boff = batch.pack_utf8(blobs)
            c, o = batch.split_offsets_utf8_bytes_csr(u8, boff)
            want_o = []
            for t, f in zip(texts, flags):
                bp = np.cumsum([0] + [len(ch.encode("utf-8", "surrogatepass")) for ch in t])
                want_o += [int(bp[i]) for i in np.nonzero(f)[0]]
            assert o.tolist() == want_o, (name, rep, "utf8 bytes")
            for kind_alpha, enc, dt in (("latin1", "latin-1", np.uint8), ("bmp", "utf-16-le", np.uint16)):
                texts = random_strings(rng, 500, 300, 900, ALPHABETS[kind_alpha])
                units = np.frombuffer("".join(texts).encode(enc, "surrogatepass"), dt)
                urow = np.zeros(len(texts) + 1, np.int64)
                np.cumsum([len(t) for t in texts], out=urow[1:])
                c, o = batch.split_offsets_kind_csr(units, urow)
                assert o.tolist() == [int(i) for t in texts for i in np.nonzero(oracle.split_values_rules(t, *tables))[0]], (name, rep, kind_alpha)
    finally:
        batch.reset_rules()
    assert not batch.rules_active()
    cps, row = pack([G1])
    assert np.array_equal(batch.split_mask_batch(cps, row), oracle.split_batch(cps, row, want_values=False)[1])


def test_runtime_rule_tables_are_validated(gpu):
    from conftest import DEFAULT_RULES
    from latok_amd import batch
    s, m, y = DEFAULT_RULES
    for bad in [(np.array([[25]], np.int8), m, y), (s, np.array([[-1, 3]], np.int8), y), (s, m, np.array([[-2]], np.int8)),
                (np.zeros((33, 1), np.int8), m, y)]:
        with pytest.raises(ValueError):
            batch.set_rules(*bad)
        assert not batch.rules_active()


def test_small_batch_path_thresholds(gpu, oracle):
    """Host-pointer calls up to 16 384 chars / 512 strings run on pinned mapped memory in place (api.cpp: kSmallChars,
    kSmallStrings); the results must not depend on which side of the thresholds a batch falls."""
    from latok_amd import batch
    rng = random.Random(4242)
    for total, n_str in [(16384, 1), (16385, 1), (16384, 512), (16384, 513), (16000, 512), (100, 513), (1, 1), (4096, 7)]:
        cuts = sorted(rng.sample(range(1, total), min(n_str - 1, total - 1))) if n_str > 1 else []
        cuts += [total] * (n_str - 1 - len(cuts))   # not enough room for distinct cuts: the rest are empty strings
        blob = "".join(rng.choice(ALPHABETS["mixed"]) for _ in range(total))
        texts = [blob[a:b] for a, b in zip([0] + cuts, cuts + [total])]
        assert len(texts) == n_str and sum(map(len, texts)) == total
        _check_batch(oracle, texts)
        got = batch.tokenize_batch(texts)
        assert got == [oracle.tokenize(t) if t else [] for t in texts]
        feats = batch.featurize_batch(texts[:40])
        for t, toks in zip(texts[:40], feats):
            if not t:
                assert toks == []
                continue
            m = oracle.gen_parse_matrix(t)
            for tok in toks:
                assert np.array_equal(tok.features, m[tok.start_idx:tok.end_idx].sum(axis=0, dtype=np.int8))


def test_featurize_long_tokens_across_words_and_tiles(gpu, oracle):
    """Token sums are popcounts of the feature planes per 64-char word; a token that leaves its word collects the
    following words' head sums, one that leaves its 4096-char tile is finished char by char.  Masked URL / e-mail
    blocks of 60 ... 10 000 chars placed so that they straddle word and tile boundaries, ASCII and non-ASCII, with the
    uint8 wrap-around of sums beyond 255."""
    from latok_amd import batch
    rng = random.Random(99)
    texts = []
    for n in [60, 64, 65, 127, 128, 129, 200, 256, 300, 1000, 4090, 4096, 4100, 8192, 10000]:
        for lead in [0, 1, 40, 63, 4000, 4095]:
            body = "".join(rng.choice("abcXYZ9_/.:é日") for _ in range(n))
            texts.append("x" * lead + " http://" + body + " end #tag a@" + body[:70] + ".org")
    texts += ["a" * 5000, "a" * 4096 + " " + "b" * 4096, "@" + "q" * 9000, "é" * 700 + " " + "Ⅷ" * 300]
    texts += random_strings(rng, 40, 3000, 9000, ALPHABETS["rare_space_at"])
    got = batch.featurize_batch(texts)
    for t, toks in zip(texts, got):
        m = oracle.gen_parse_matrix(t).astype(np.uint8)
        nz = oracle.split_offsets(t).tolist() + [len(t)]
        want = [(a, b) for a, b in zip(nz[:-1], nz[1:]) if t[a:b].strip()]
        assert [(x.start_idx, x.end_idx) for x in toks] == want, t[:60]
        for tok, (a, b) in zip(toks, want):
            f = m[a:b].sum(axis=0, dtype=np.uint64).astype(np.uint8).astype(np.int8)
            assert np.array_equal(tok.features, f), (t[:40], a, b, tok.features, f)


def test_featurize_million_char_tokens(gpu, oracle):
    """BASELINE configs[4] stress documents (SURVEY 8d C5 i, ii): 1 M chars without whitespace, and the same with a URL
    start at char 4 (the whole document is one masked token).  The token that leaves its tile is continued by the whole
    wave, 4096 chars per step (it used to be finished char by char on one lane: 10^6 dependent steps)."""
    from latok_amd import batch
    rng = random.Random(4242)
    n = 1_000_000
    body = "".join(rng.choice("abcdefghXYZ019_") for _ in range(n))
    docs = [body,                                         # (i) no whitespace at all: camelCase humps split it
            "see http://" + body[:n - 11],                # (ii)-like: a masked block of ~1 M chars behind a short word
            "http://" + body[:n - 7],                     # (ii) the whole document is one token
            "x " * 10 + "a@" + body[:300000] + "/.:" + body[:200000] + " tail",   # long masked e-mail between short tokens
            "é" * 5000 + "@" + "日" * 200000 + " end"]
    got = batch.featurize_batch(docs)
    for t, toks in zip(docs, got):
        m = oracle.gen_parse_matrix(t).astype(np.uint8)
        nz = oracle.split_offsets(t).tolist() + [len(t)]
        want = [(a, b) for a, b in zip(nz[:-1], nz[1:]) if t[a:b].strip()]
        assert [(x.start_idx, x.end_idx) for x in toks] == want, t[:40]
        csum = np.zeros((len(t) + 1, 25), np.uint64)
        np.cumsum(m, axis=0, dtype=np.uint64, out=csum[1:])
        for tok, (a, b) in zip(toks, want):
            f = (csum[b] - csum[a]).astype(np.uint8).astype(np.int8)
            assert np.array_equal(tok.features, f), (t[:30], a, b, tok.features, f)
    # the same documents in one batch with short strings around them, 32-bit records
    texts = ["short one", docs[2], "", docs[0][:5000] + " x", docs[1], "tail #tag"]
    cps, row = pack(texts)
    a = batch.token_features_csr(cps, row)
    b = batch.token_features_csr(cps, row, dtype=np.int32)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def _byte_expect(oracle, texts):
    """oracle boundaries / SPACE flags mapped from code-point to byte positions of the UTF-8 encoding"""
    blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
    boff = np.zeros(len(texts) + 1, np.int64)
    np.cumsum([len(b) for b in blobs], out=boff[1:])
    flags = np.zeros(int(boff[-1]), bool)
    per_string = []
    for t, b0 in zip(texts, boff[:-1]):
        lens = np.array([len(ch.encode("utf-8", "surrogatepass")) for ch in t], np.int64)
        start = np.zeros(len(t) + 1, np.int64)
        np.cumsum(lens, out=start[1:])
        nz = oracle.split_offsets(t) if t else np.zeros(0, np.int64)
        flags[b0 + start[nz]] = True
        per_string.append(start[nz])
    return blobs, boff, flags, per_string


def test_utf8_byte_space(gpu, oracle):
    """Fused UTF-8 ingest: the tile kernel works on BYTE positions (kModeBytes); boundaries, offsets and token spans are
    the reference's, mapped to byte positions of the UTF-8 buffer.  ASCII tiles, mixed tiles, multi-byte chars across word
    and tile edges, multi-byte spaces, lone surrogates (surrogatepass), empty strings."""
    from latok_amd import batch
    rng = random.Random(2024)
    alpha = ALPHABETS["mixed"] + list("é日🤓ü　 ") + ["http://é", "a@日", ".@ü", "#日"]
    cases = [[G1, "", "日本語のテキスト、です。 🤓 ok ", "é", "🤓", "a　b c", "x" * 4095 + "é" + "y" * 10, "\ud800 lone \udfff"],
             random_strings(rng, 300, 0, 60, alpha), random_strings(rng, 40, 0, 400, ALPHABETS["words"]),
             random_strings(rng, 4, 3000, 20000, alpha), random_strings(rng, 3, 5000, 30000, ALPHABETS["rare_space_at"] + ["é"]),
             random_strings(rng, 2000, 0, 6, alpha), ["é" * 5000, "🤓" * 3000 + " a", "日" * 4096 + "@" + "語" * 4096]]
    for texts in cases:
        blobs, boff, flags, per_string = _byte_expect(oracle, texts)
        utf8 = np.frombuffer(b"".join(blobs), np.uint8)
        total = int(boff[-1])
        bits = batch.split_mask_utf8_bytes_csr(utf8, boff)
        got = bits_to_bool(bits, total)
        if not np.array_equal(got, flags):
            bad = int(np.nonzero(got != flags)[0][0])
            s = int(np.searchsorted(boff, bad, side="right") - 1)
            raise AssertionError(f"byte mask differs at byte {bad} (string {s}, byte {bad - boff[s]} of {boff[s + 1] - boff[s]})")
        counts, offs = batch.split_offsets_utf8_bytes_csr(utf8, boff)
        assert np.array_equal(counts, [len(x) for x in per_string])
        assert np.array_equal(offs, np.concatenate(per_string) if per_string else np.zeros(0, np.int64))
        toks = batch.tokenize_utf8_batch(blobs)
        for t, g in zip(texts, toks):
            want = [w.encode("utf-8", "surrogatepass") for w in (oracle.tokenize(t) if t else [])]
            assert g == want, (t[:60], g[:8], want[:8])


def test_utf8_byte_space_tables_arrive_when_the_first_multibyte_char_does(gpu, oracle):
    """Byte space copies its class table to LDS on demand (split_kernels.hip: tables_ensure_bytes): the launch brings in the 128
    ASCII codes, the first wave of a workgroup that meets a multi-byte char fetches the rest, waves that arrive meanwhile take a
    share or wait.  Large ASCII batches (hundreds of tiles per workgroup) whose ONLY multi-byte chars sit at chosen places -- the
    very first tile, the very last, one tile somewhere in the middle, the last byte of a tile, every 50th tile -- so that one wave
    fetches alone, early or late in its workgroup's life, or many workgroups do at once; results against the oracle, mapped to
    byte positions; the same call repeated (every launch starts without the table)."""
    from latok_amd import batch
    rng = random.Random(77)
    words = ALPHABETS["words"]
    base = random_strings(rng, 9000, 100, 300, words)                    # ~1.8 MB of ASCII: ~440 tiles
    variants = []
    v = list(base); v[0] = "é" + v[0]; variants.append(v)                                   # first tile
    v = list(base); v[-1] = v[-1] + " 日本語"; variants.append(v)                            # last tile
    v = list(base); v[len(v) // 2] = v[len(v) // 2][:50] + "🤓 ü" + v[len(v) // 2][50:]; variants.append(v)   # the middle
    v = list(base)
    head = sum(len(t) for t in v[:20])
    pad = (4096 - (head % 4096) - 1) % 4096                              # a 2-byte char whose lead is a tile's last byte
    v[20] = "x" * pad + "é" + v[20]
    variants.append(v)
    v = list(base)
    for i in range(0, len(v), 1000): v[i] = v[i] + " Привет ①"
    variants.append(v)
    variants.append(list(base))                                          # and none at all
    for texts in variants:
        blobs, boff, flags, per_string = _byte_expect(oracle, texts)
        utf8 = np.frombuffer(b"".join(blobs), np.uint8)
        total = int(boff[-1])
        for _ in range(2):
            got = bits_to_bool(batch.split_mask_utf8_bytes_csr(utf8, boff), total)
            if not np.array_equal(got, flags):
                bad = int(np.nonzero(got != flags)[0][0])
                s_ = int(np.searchsorted(boff, bad, side="right") - 1)
                raise AssertionError(f"byte mask differs at byte {bad} (string {s_}, byte {bad - boff[s_]} of {boff[s_ + 1] - boff[s_]})")
        counts, offs = batch.split_offsets_utf8_bytes_csr(utf8, boff)
        assert np.array_equal(counts, [len(x) for x in per_string])
        assert np.array_equal(offs, np.concatenate(per_string))


def test_utf8_byte_space_blocks_that_end_in_a_multibyte_symbol_near_tile_edges(gpu, oracle):
    """The resolve stage patches a tile in place when one pending start enters it or its open tail block turns out to be masked;
    what stays of a cleared block is the C_SYM bit of its LAST char (default_tokenizer.py:128-131: splits = raw * mask + sym).  In
    byte space that bit sits at the char's lead byte, up to 3 bytes before the block's last position, also when the tile holds
    multi-byte chars (the stage finds the lead in the bytes).  URLs (a masked block from 'http://' to the next space) made of
    2-, 3- and 4-byte chars, closed by a multi-byte SYMBOL + space, with the closing char at every byte offset around a tile edge
    -- in the tile's head block, in its tail block, straddling the edge."""
    from latok_amd import batch
    fillers = ["日", "é", "🤓", "aé日"]
    closers = ["。", "、", "🤓", "¡", "."]
    for fill in fillers:
        fb = len(fill.encode())
        for closer in closers:
            texts = []
            for shift in range(0, 7):
                n = (2 * 4096 - 7 - shift) // fb
                for dn in (-2, -1, 0, 1):
                    texts.append("x" * shift)                                   # moves the URL's bytes against the tile grid
                    texts.append("http://" + fill * (n + dn) + closer + " tail " + fill + closer)
                    texts.append("a " + "http://" + fill * (n + dn) + closer)  # ... and the block that ends with the string
            blobs, boff, flags, per_string = _byte_expect(oracle, texts)
            utf8 = np.frombuffer(b"".join(blobs), np.uint8)
            total = int(boff[-1])
            got = bits_to_bool(batch.split_mask_utf8_bytes_csr(utf8, boff), total)
            if not np.array_equal(got, flags):
                bad = int(np.nonzero(got != flags)[0][0])
                s_ = int(np.searchsorted(boff, bad, side="right") - 1)
                raise AssertionError(f"{fill!r} {closer!r}: byte mask differs at byte {bad} (string {s_}, byte {bad - boff[s_]} of {boff[s_ + 1] - boff[s_]}; tile offset {bad % 4096})")
            counts, offs = batch.split_offsets_utf8_bytes_csr(utf8, boff)
            assert np.array_equal(offs, np.concatenate(per_string))


def test_utf8_byte_space_malformed_bytes_against_the_cpu_model(gpu):
    """Byte space on bytes that are NOT well-formed UTF-8: stray continuation bytes (also right behind '#' '@' ':' '/' '.',
    which sends the tile through the general rule form), runs of more than three of them, truncated sequences, lone leads,
    0xF8..0xFF, at every word and tile phase and across tile edges.  The definition of the result is oracle/fused_model.cpp
    (one char per lead byte, a continuation byte belongs to the nearest lead at most 3 back -- checked there against the
    per-byte statement of that rule and, on well-formed text, against the reference-shaped oracle): the kernel must give
    the same boundary bits and the same smeared SPACE plane (token spans), large batches and small ones."""
    import ctypes as C
    import os
    from conftest import ROOT
    from latok_amd import batch
    model = C.CDLL(os.path.join(ROOT, "oracle", "libfused_model.so"))
    model.fused_split_batch_utf8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(77)
    pools = [np.arange(256, dtype=np.uint8),
             np.array([0x20, 0x61, 0x41, 0x80, 0x80, 0xBF, 0xC3, 0xE3, 0xF0, 0xFF, 0x2E, 0x40, 0x23, 0x3A, 0x2F], np.uint8),
             np.array([0x80, 0x81, 0xE3, 0x61, 0x20, 0x23], np.uint8),
             np.frombuffer("a# @b .@c http://é 日本 🤓 Ünï ".encode(), np.uint8)]
    for it in range(40):
        big = it % 4 == 3
        n_str = int(rng.integers(1, 400 if big else 40))
        lens = rng.integers(0, [24, 300, 9000, 3000][it % 4], n_str)
        boff = np.zeros(n_str + 1, np.int64)
        np.cumsum(lens, out=boff[1:])
        total = int(boff[-1])
        if total == 0:
            continue
        u8 = np.ascontiguousarray(rng.choice(pools[(it // 4) % len(pools)], total))
        if it % 5 == 0:   # well-formed stretches with single damaged bytes
            good = np.frombuffer(("word #tag a@b.c http://x.y/z 日本語のテキスト、です。 🤓 Ünï Привет " * (total // 60 + 1)).encode(), np.uint8)[:total].copy()
            hit = rng.integers(0, total, max(1, total // 97))
            good[hit] = rng.choice(pools[1], hit.size)
            u8 = good
        want = np.zeros((total + 63) // 64, np.uint64)
        want_sp = np.zeros_like(want)
        assert model.fused_split_batch_utf8(u8.ctypes.data, boff.ctypes.data, n_str, want.ctypes.data, want_sp.ctypes.data, None) == 0
        got = batch.split_mask_utf8_bytes_csr(u8, boff)
        if not np.array_equal(got, want):
            bad = int(np.nonzero(bits_to_bool(got, total) != bits_to_bool(want, total))[0][0])
            raise AssertionError(f"case {it}: byte mask differs at byte {bad}: ...{bytes(u8[max(0, bad - 8):bad + 8])!r}")
        # token spans read the smeared SPACE plane: same spans as the model's planes give
        counts, spans = batch.token_spans_utf8_bytes_csr(u8, boff)
        b, sp = bits_to_bool(want, total), bits_to_bool(want_sp, total)
        exp = []
        for s_ in range(n_str):
            lo, hi = int(boff[s_]), int(boff[s_ + 1])
            cuts = [lo + int(x) for x in np.nonzero(b[lo:hi])[0]] + [hi]
            for a_, z_ in zip(cuts[:-1], cuts[1:]):
                keep = np.nonzero(~sp[a_:z_])[0]
                if keep.size:
                    exp.append((a_ - lo + int(keep[0]), a_ - lo + int(keep[-1]) + 1))
        assert spans.reshape(-1, 2).tolist() == [list(e) for e in exp], it


def test_c_example_program(gpu, oracle, tmp_path):
    """examples/tokenize_utf8.c: a plain C caller of the C ABI (byte-space token ranges) prints the reference's tokens."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "tokenize_utf8")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "tokenize_utf8.c"),
                           "-L" + os.path.join(ROOT, "latok_amd"), "-llatok_hip", "-Wl,-rpath," + os.path.join(ROOT, "latok_amd"),
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, timeout=120, check=True).stdout.decode("utf-8").splitlines()
    texts = [G1, "see http://a.b/c or mail me@x.org", "camelCase 日本語 🤓"]
    assert out == [f"{i}:" + "".join(f" [{t}]" for t in oracle.tokenize(s)) for i, s in enumerate(texts)]


def test_c_example_sharding_over_contexts(gpu, oracle, tmp_path):
    """examples/shard_contexts.c: a C caller cuts one batch over several contexts (one pthread each, two and three contexts
    on the one device of this box) and prints every string's boundary offsets: np.nonzero of the oracle's split values."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "shard_contexts")
    subprocess.check_call(["gcc", "-std=c99", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "shard_contexts.c"), "-L" + os.path.join(ROOT, "latok_amd"), "-llatok_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "latok_amd"), "-o", exe])
    texts = [G1, "see http://a.b/c or mail me@x.org", "camelCaseXMLParser", "foo@bar.com, .@user hi", "x\t\ny", "$#@^:a./"]
    want = [f"{i}:" + "".join(f" {o}" for o in oracle.split_offsets(t).tolist()) for i, t in enumerate(texts)]
    for args in ([], ["0", "0", "0"], ["resident"], ["resident", "0", "0", "0"]):
        out = subprocess.run([exe] + args, capture_output=True, timeout=120, check=True).stdout.decode().splitlines()
        assert out == want, (args, out)


def test_device_pointer_forms_match_host_pointer_forms(gpu, oracle):
    """Every batch entry point with LATOK_DEVICE_PTRS (inputs and outputs in HBM, caller-owned) against the same call with
    host pointers, on a mixed batch that is larger than the small-batch path and contains multi-byte chars."""
    from latok_amd import _lib, batch
    lib = gpu
    rng = random.Random(808)
    alpha = ALPHABETS["mixed"] + list("é日🤓") + ["http://é", "a@日"]
    texts = random_strings(rng, 700, 0, 120, alpha) + random_strings(rng, 3, 3000, 9000, alpha) + ["", "x"]
    cps, row = pack(texts)
    n, total = len(texts), int(row[-1])
    blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
    u8, boff = batch.pack_utf8(blobs)
    nb = int(boff[-1])
    D = _lib.DEVICE_PTRS

    def dev(a):
        a = np.ascontiguousarray(a)
        p = lib.latok_dev_alloc(max(a.nbytes, 16) + 64)
        assert p
        _lib.check(lib.latok_memcpy_h2d(p, a.ctypes.data, a.nbytes))
        return p

    def back(p, shape, dtype):
        out = np.empty(shape, dtype)
        if out.nbytes:
            _lib.check(lib.latok_memcpy_d2h(out.ctypes.data, p, out.nbytes))
        return out

    d_cps, d_row, d_u8, d_boff = dev(cps), dev(row), dev(u8), dev(boff)
    cap = max(total, nb) + 8
    d_bits = lib.latok_dev_alloc(((cap + 63) // 64) * 8 + 8)
    d_counts = lib.latok_dev_alloc(n * 8 + 8)
    d_items = lib.latok_dev_alloc(cap * 32)
    d_feat = lib.latok_dev_alloc(cap * 25)
    d_cprow = lib.latok_dev_alloc((n + 1) * 8)
    nout, tcp = C.c_int64(0), C.c_int64(0)
    try:
        _lib.check(lib.latok_split_mask_batch(d_cps, d_row, n, total, d_bits, D, None))
        _lib.check(lib.latok_sync())
        assert np.array_equal(back(d_bits, (total + 63) // 64, np.uint64), batch.split_mask_batch(cps, row))
        for name, fn, host, width in [
            ("offsets", lambda: lib.latok_split_offsets_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None),
             batch.split_offsets_csr(cps, row), 1),
            ("spans", lambda: lib.latok_token_spans_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None),
             batch.token_spans_csr(cps, row), 2),
            ("utf8 offsets", lambda: lib.latok_split_offsets_utf8_batch(d_u8, d_boff, n, nb, d_counts, d_items, cap, C.byref(nout), D, None),
             batch.split_offsets_utf8_csr(u8, boff), 1),
            ("utf8 spans", lambda: lib.latok_token_spans_utf8_batch(d_u8, d_boff, n, nb, d_counts, d_items, cap, C.byref(nout), D, None),
             batch.token_spans_utf8_csr(u8, boff), 2),
            ("byte offsets", lambda: lib.latok_split_offsets_utf8_bytes_batch(d_u8, d_boff, n, nb, d_counts, d_items, cap, C.byref(nout), D, None),
             batch.split_offsets_utf8_bytes_csr(u8, boff), 1),
            ("byte spans", lambda: lib.latok_token_spans_utf8_bytes_batch(d_u8, d_boff, n, nb, d_counts, d_items, cap, C.byref(nout), D, None),
             batch.token_spans_utf8_bytes_csr(u8, boff), 2),
        ]:
            _lib.check(fn())
            _lib.check(lib.latok_sync())
            h_counts, h_items = host
            assert nout.value == len(h_items), name
            assert np.array_equal(back(d_counts, n, np.int64), h_counts), name
            assert np.array_equal(back(d_items, nout.value * width, np.int64).reshape(h_items.shape), h_items), name
        _lib.check(lib.latok_token_features_batch(d_cps, d_row, n, total, d_counts, d_items, d_feat, cap, C.byref(nout), D, None))
        _lib.check(lib.latok_sync())
        h_counts, h_spans, h_feats = batch.token_features_csr(cps, row)
        assert nout.value == len(h_spans) and np.array_equal(back(d_counts, n, np.int64), h_counts)
        assert np.array_equal(back(d_items, (nout.value, 4), np.int64), h_spans)
        assert np.array_equal(back(d_feat, (nout.value, 25), np.int8), h_feats)
        _lib.check(lib.latok_split_mask_utf8_batch(d_u8, d_boff, n, nb, d_bits, (cap + 63) // 64, d_cprow, C.byref(tcp), D, None))
        _lib.check(lib.latok_sync())
        assert tcp.value == total and np.array_equal(back(d_cprow, n + 1, np.int64), row)
        assert np.array_equal(back(d_bits, (total + 63) // 64, np.uint64), batch.split_mask_batch(cps, row))
        _lib.check(lib.latok_split_mask_utf8_bytes_batch(d_u8, d_boff, n, nb, d_bits, D, None))
        _lib.check(lib.latok_sync())
        assert np.array_equal(back(d_bits, (nb + 63) // 64, np.uint64), batch.split_mask_utf8_bytes_csr(u8, boff))
    finally:
        for p in (d_cps, d_row, d_u8, d_boff, d_bits, d_counts, d_items, d_feat, d_cprow):
            lib.latok_dev_free(p)


def test_calls_on_different_streams_are_ordered(gpu, oracle):
    """Two device-pointer calls on two caller streams share the library's workspaces: the second must wait for the first on
    the device.  Alternate big batches between two streams without synchronising in between and check both results."""
    import ctypes as C2
    from latok_amd import _lib, batch
    lib = gpu
    hip = C2.CDLL("libamdhip64.so")
    s1, s2 = C2.c_void_p(), C2.c_void_p()
    assert hip.hipStreamCreate(C2.byref(s1)) == 0 and hip.hipStreamCreate(C2.byref(s2)) == 0
    rng = random.Random(5150)
    batches = []
    try:
        for i in range(2):
            texts = random_strings(rng, 20000 + 5000 * i, 0, 90, ALPHABETS["mixed"])
            cps, row = pack(texts)
            d_cps = lib.latok_dev_alloc(cps.nbytes + 64)
            d_row = lib.latok_dev_alloc(row.nbytes)
            d_bits = lib.latok_dev_alloc(((cps.size + 63) // 64) * 8 + 8)
            _lib.check(lib.latok_memcpy_h2d(d_cps, cps.ctypes.data, cps.nbytes))
            _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
            batches.append((cps, row, d_cps, d_row, d_bits))
        _lib.check(lib.latok_sync())
        for rep in range(6):
            for i, st in ((0, s1), (1, s2)):
                cps, row, d_cps, d_row, d_bits = batches[i]
                _lib.check(lib.latok_split_mask_batch(d_cps, d_row, len(row) - 1, cps.size, d_bits, _lib.DEVICE_PTRS, st))
        assert hip.hipStreamSynchronize(s1) == 0 and hip.hipStreamSynchronize(s2) == 0
        for cps, row, d_cps, d_row, d_bits in batches:
            got = np.empty((cps.size + 63) // 64, np.uint64)
            _lib.check(lib.latok_memcpy_d2h(got.ctypes.data, d_bits, got.nbytes))
            assert np.array_equal(got, oracle.split_batch(cps, row, want_values=False)[1])
    finally:
        for b in batches:
            for p in b[2:]:
                lib.latok_dev_free(p)
        hip.hipStreamDestroy(s1)
        hip.hipStreamDestroy(s2)


def test_very_uneven_string_lengths(gpu, oracle):
    """The segment prologue guesses where a segment's strings sit in row_off from an even spacing and falls back to a
    search when the guess window misses: batches whose string lengths are as uneven as possible (hundreds of thousands of
    1-3-char strings next to 100 K-char documents, both orders, and interleaved) must take that fallback and stay exact."""
    from latok_amd import batch
    rng = random.Random(271828)
    tiny = random_strings(rng, 200000, 1, 3, ALPHABETS["starts"])
    huge = random_strings(rng, 40, 90000, 110000, ALPHABETS["words"])
    mixed = []
    for i, h in enumerate(huge[:10]):
        mixed += tiny[i * 5000:(i + 1) * 5000] + [h]
    for texts in (tiny + huge, huge + tiny, mixed):
        cps, row = pack(texts)
        _, want = oracle.split_batch(cps, row, want_values=False)
        assert np.array_equal(batch.split_mask_batch(cps, row), want)
        counts, offs = batch.split_offsets_csr(cps, row)
        assert int(counts.sum()) == len(offs) == int(np.unpackbits(want.view(np.uint8)).sum())
