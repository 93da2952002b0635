"""The oracle (oracle/latok_oracle.c) pinned against the reference's own golden material and reference-generated
fixtures (tests/golden/, made by make_golden.py from the REAL reference); plus, when the reference is present (build
container), a live differential run against it.  CPU only."""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

from conftest import ALPHABETS, GOLDEN, random_strings


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def text_of(cps):
    return np.array(cps, dtype="<u4").tobytes().decode("utf-32-le", "surrogatepass")


def test_notebook_golden_matrix_and_splits(oracle):
    """G1: the only known-answer data in the reference repo (notebooks/scratch/LaTokenizer.ipynb, cell 0 output)."""
    g = load("g1_notebook.json")
    assert g["text"] == "This is a #test! Testing, Testing, 1 2 3"
    m = oracle.gen_parse_matrix(g["text"])
    assert m.tolist() == g["matrix"]
    assert oracle.gen_split_mask(m).tolist() == g["splits"]
    assert oracle.split_offsets(g["text"]).tolist() == [0, 4, 7, 9, 15, 16, 17, 24, 25, 26, 33, 34, 36, 38]


def test_reference_generated_strings(oracle):
    for it in load("ref_strings.json")["items"]:
        text = text_of(it["cps"])
        m = oracle.gen_parse_matrix(text)
        assert hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest() == it["matrix_sha256"]
        assert oracle.split_values(text).tolist() == it["splits"]
        assert oracle.split_offsets(text).tolist() == it["offsets"]
        assert oracle.tokenize(text) == [text_of(t) for t in it["tokens"]]


def test_reference_native_vectors(oracle):
    g = load("native_vectors.json")
    for v in g["block_mask"]:
        got = oracle.gen_block_mask(np.array(v["a1"], np.int8), np.array(v["a2"], np.int8))
        assert got.tolist() == v["mask"]
    for v in g["combine"]:
        got = oracle.combine_matrix_rows(np.array(v["m"], np.int8), np.array(v["idx"], np.int8))
        assert got.tolist() == v["out"]


def test_config1_paragraph(oracle):
    """BASELINE configs[0]: single 1 KB ASCII paragraph, bit-exact boundary check on CPU."""
    g = load("c1_paragraph.json")
    assert g["n_chars"] == 1024 and len(g["text"]) == 1024
    assert oracle.split_values(g["text"]).tolist() == g["splits"]
    assert oracle.split_offsets(g["text"]).tolist() == g["offsets"]
    assert oracle.tokenize(g["text"]) == g["tokens"]


def test_unicode_class_table_pin(oracle):
    """Every code point's 12 base features, hashed, equals what the reference produced (tools/gen_unicode_tables.py)."""
    g = load("unicode_classes.json")
    words = np.fromiter((oracle.base_word(cp) for cp in range(g["n_code_points"])), dtype="<u2", count=g["n_code_points"])
    assert hashlib.sha256(words.tobytes()).hexdigest() == g["sha256_uint16le_words"]
    assert len(np.unique(words)) == g["n_classes"] == 17
    for key, info in g["classes"].items():
        for cp in info["code_points"]:
            assert oracle.base_word(cp) == int(key, 16)
    assert oracle.base_word(0x110000) == 0 and oracle.base_word(0xFFFFFFFF) == 0   # reference latok.c:20-21


def _host_corpus(seed, model, n_str, lo, hi):
    from latok_amd import _lib
    lib = _lib.load()
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n_str, lo, hi, row.ctypes.data))
    cps = np.zeros(int(row[-1]), np.uint32)
    _lib.check(lib.latok_corpus_fill_host(seed, model, 0, n_str, row.ctypes.data, cps.ctypes.data))
    return cps, row


def test_corpus_generator_and_offsets_hashes(oracle):
    """F7: host corpus generator is pinned by hash, and the oracle's offsets on it equal the reference's."""
    for name, c in load("corpus_samples.json")["corpora"].items():
        cps, row = _host_corpus(c["seed"], c["model"], c["n_str"], c["len_lo"], c["len_hi"])
        assert int(row[-1]) == c["total_chars"]
        assert hashlib.sha256(cps.astype("<u4").tobytes()).hexdigest() == c["sha256_cps_u32le"], name
        assert hashlib.sha256(row.astype("<i8").tobytes()).hexdigest() == c["sha256_row_off_i64le"]
        assert cps[row[0]:row[1]].tolist() == c["first_string"]
        vals, _ = oracle.split_batch(cps, row, want_bits=False)
        h, n = hashlib.sha256(), 0
        for s in range(c["n_str"]):
            nz = np.nonzero(vals[row[s]:row[s + 1]])[0].astype("<i8")
            h.update(nz.tobytes())
            n += len(nz)
        assert n == c["n_boundaries"] and h.hexdigest() == c["sha256_offsets_i64le"], name


def test_empty_string_conventions(oracle):
    with pytest.raises(IndexError):      # reference: splits[0] = 1 on an empty array (default_tokenizer.py:132)
        oracle.split_values("")
    vals, bits = oracle.split_batch(np.zeros(0, np.uint32), np.array([0, 0, 0], np.int64))
    assert vals.size == 0 and bits.size == 0
    assert oracle.tokenize(" ") == []


def test_refglue_matches_oracle_when_built(oracle):
    """oracle/_ref (the reference's own C, compiled from its own source) under the restated glue == the restatement."""
    import ref_loader
    if not ref_loader.ref_ext_available():
        pytest.skip("oracle/_ref not built")
    glue = oracle.RefGlue()
    rng = random.Random(2)
    for text in random_strings(rng, 300, 1, 120, ALPHABETS["mixed"]) + random_strings(rng, 100, 1, 300, ALPHABETS["words"]):
        assert np.array_equal(glue.gen_parse_matrix(text), oracle.gen_parse_matrix(text))
        assert np.array_equal(glue.split_values(text), oracle.split_values(text))


def test_live_differential_against_reference(oracle):
    """Build container only: the real reference (its C + its own Python glue) vs the restatement on random input."""
    import ref_loader
    if not ref_loader.ref_python_available():
        pytest.skip("/root/reference not present (expected on the GPU box)")
    import subprocess
    import sys
    code = r'''
import sys, random, numpy as np
sys.path.insert(0, %r)
import ref_loader, latok_oracle as orc
dt = ref_loader.load_ref_python()
rng = random.Random(99)
alpha = list("abcXYZ  \t.,:/@#$^!1 9éあ日\U0001f913́Ⅷ")
for it in range(20000):
    s = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 60)))
    r = dt.gen_split_mask(dt._gen_parse_matrix(s))
    assert np.array_equal(r, orc.split_values(s)), s
    assert list(dt.tokenize(s)) == orc.tokenize(s), s
print("ok")
''' % os.path.join(os.path.dirname(GOLDEN), "..", "oracle")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_reference_generated_rule_tables(oracle):
    """Other C_SPLIT / C_MASK / C_SYM combo matrices: split values produced by the REAL reference's gen_split_mask with
    those matrices installed (rules_strings.json) against the oracle's recipe, and the CPU model of the GPU's
    rule-table interpreter on the non-zero pattern."""
    import subprocess
    from conftest import ROOT, pack, rule_row_sets
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "fused"])
    model = C.CDLL(os.path.join(ROOT, "oracle", "libfused_model.so"))
    model.fused_split_batch_rules.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
    sets = load("rules_strings.json")["sets"]
    assert len(sets) >= 10
    for rs in sets:
        tables = (np.array(rs["c_split"], np.int8), np.array(rs["c_mask"], np.int8), np.array(rs["c_sym"], np.int8))
        texts = [text_of(it["cps"]) for it in rs["items"]]
        for text, it in zip(texts, rs["items"]):
            assert oracle.split_values_rules(text, *tables).tolist() == it["splits"], (rs["name"], text)
        rows, n_rows = rule_row_sets(tables)
        cps, row = pack(texts)
        bits = np.zeros((int(row[-1]) + 63) // 64, np.uint64)
        assert model.fused_split_batch_rules(cps.ctypes.data, row.ctypes.data, len(texts), rows.ctypes.data,
                                             n_rows.ctypes.data, bits.ctypes.data, None) == 0
        want = np.concatenate([np.array(it["splits"]) != 0 for it in rs["items"]])
        got = np.unpackbits(bits.view(np.uint8), bitorder="little")[:want.size].astype(bool)
        assert np.array_equal(got, want), rs["name"]
