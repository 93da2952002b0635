#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (build container only: needs /root/reference).

Everything here is DATA: inputs and the outputs the reference itself produced for them.
  * g1_notebook.json      -- the reference's own stored golden: the 40 x 25 feature matrix and the Splits column of
                             "This is a #test! Testing, Testing, 1 2 3", parsed out of the saved cell output of
                             notebooks/scratch/LaTokenizer.ipynb (the only known-answer material in the reference).
  * ref_strings.json      -- sample strings (reference default_tokenizer.py:195-198 + SURVEY probes + edge lengths) with the
                             matrix checksum, split values, boundary offsets and tokens computed by the REAL reference
                             (its latok.c compiled into oracle/_ref + its own latok/core/*.py imported from /root/reference).
  * native_vectors.json   -- input/output vectors of the reference's _gen_block_mask and _combine_matrix_rows.
  * c1_paragraph.json     -- BASELINE config 1: one 1 KB ASCII paragraph with reference offsets / tokens.
  * rules_strings.json    -- the reference's extension point: other C_SPLIT / C_MASK / C_SYM combo matrices installed as
                             module globals of the REAL default_tokenizer, its own gen_split_mask run on sample strings.
  * corpus_samples.json   -- first 10 000 strings of the C2 / C3 synthetic corpora: SHA-256 of the code points, of the
                             offsets table and of the reference's boundary offsets, so a GPU box without the reference can
                             check generator + kernel end to end.

Run:  make -C oracle ref && make -C latok_amd/csrc && python3 tests/golden/make_golden.py
"""
import hashlib
import html.parser
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader  # noqa: E402

NOTEBOOK = os.path.join(ref_loader.REF_ROOT, "notebooks", "scratch", "LaTokenizer.ipynb")


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=None, separators=(",", ":"))
        f.write("\n")
    print("wrote", name)


def cps_of(text):
    return np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.uint32)


class _Table(html.parser.HTMLParser):
    def __init__(self):
        super().__init__()
        self.rows, self.cur, self.cell = [], None, None

    def handle_starttag(self, tag, attrs):
        if tag == "tr":
            self.cur = []
        elif tag in ("td", "th"):
            self.cell = ""

    def handle_endtag(self, tag):
        if tag == "tr" and self.cur is not None:
            self.rows.append(self.cur)
            self.cur = None
        elif tag in ("td", "th") and self.cell is not None and self.cur is not None:
            self.cur.append(self.cell)
            self.cell = None

    def handle_data(self, data):
        if self.cell is not None:
            self.cell += data


def golden_notebook():
    nb = json.load(open(NOTEBOOK))
    cell = nb["cells"][0]
    out = [o for o in cell["outputs"] if "data" in o and "text/html" in o["data"]][0]
    p = _Table()
    p.feed("".join(out["data"]["text/html"]))
    header = p.rows[0]
    assert header[1:3] == ["Chars", "Splits"], header[:4]
    names = header[3:]
    body = [r for r in p.rows[1:] if len(r) == len(header)]
    chars = [r[1] for r in body]
    text = "".join(c if c != "" else " " for c in chars)
    splits = [int(r[2]) for r in body]
    matrix = [[int(x) for x in r[3:]] for r in body]
    assert len(matrix) == 40 and all(len(r) == 25 for r in matrix)
    dump("g1_notebook.json", {
        "source": "reference notebooks/scratch/LaTokenizer.ipynb, cell 0, stored text/html output",
        "text": text, "feature_names": names, "splits": splits, "matrix": matrix})
    return text


def sample_strings():
    rng = random.Random(20261003)
    alpha = list("abcXYZ  \t.,:/@#$^!1 9éあ日🤓́Ⅷ")
    fixed = [
        "This is a #test! Testing, Testing, 1 2 3",
        "can’t wait to get my glasses back 🤓",
        "IKR!! IM LIKE \"\"WHERE'S MY DADDY AT? 👀) https://t.co/jM3qLZijMc",
        "$#@^:a./",
        "camelCaseXMLParser", "foo@bar.com, .@user hi", "日本語のテキスト、です。", "éà x", "x\t\ny", "①②Ⅷ 五 ½",
        "http://a@b X,y z", " ", "a", "@", "#a", "a@b", ".@a", "a:// b", "x://y", "  lead and trail  ",
        "no-whitespace-at-all-http://x.y/z", "end with start #tag", "a@b@c@d x,y p,q r,s t,u",
        "\ud800 lone surrogate \udfff", "\U0010ffff max", "tab\tsep\u00a0nbsp\u2028ls\u3000ideo",
        "Ⓐⓑ ⅷⅧ ǅ ʰ ͅ", "١٢٣ ٤٥", "ÀÉÎõü ΑΒΓαβγ АБВабв",
    ]
    rnd = []
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257):
        rnd.append("".join(rng.choice(alpha) for _ in range(n)))
        rnd.append("".join(rng.choice("abc DEF,.#@:/") for _ in range(n)))
    return fixed + rnd


def golden_strings(dt):
    items = []
    for text in sample_strings():
        m = dt._gen_parse_matrix(text)
        splits = dt.gen_split_mask(m)
        items.append({
            "cps": cps_of(text).tolist(),
            "matrix_sha256": hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest(),
            "splits": splits.astype(int).tolist(),
            "offsets": np.nonzero(splits)[0].astype(int).tolist(),
            "tokens": [cps_of(t).tolist() for t in dt.tokenize(text)],
        })
    dump("ref_strings.json", {"source": "real reference (oracle/_ref + /root/reference/latok/core/*.py)", "items": items})


def golden_native(ext):
    rng = random.Random(7)
    bm = []
    probed = [([0, 0, 1, 0, 0, 0, 0], [0, 1, 0, 0, 1, 0, 0]), ([0, 0, 1, 1, 0, 0, 0, 0], [0, 1, 0, 0, 1, 0, 1, 0]),
              ([0, 0, 0, 0, 0, 1, 0], [0, 1, 0, 0, 1, 0, 0]), ([0, 0, 1, 0], [0, 0, 0, 0]), ([0, 0, 0], [0, 1, 0]),
              ([1], [1]), ([1], [0]), ([0], [0])]
    for a1, a2 in probed:
        bm.append({"a1": a1, "a2": a2, "mask": ext._gen_block_mask(np.array(a1, np.int8), np.array(a2, np.int8)).astype(int).tolist()})
    for n in (2, 3, 5, 17, 64, 65, 200):
        for p1, p2 in ((.2, .3), (.05, .3), (.5, .05), (.3, .0)):
            a1 = [int(rng.random() < p1) for _ in range(n)]
            a2 = [int(rng.random() < p2) for _ in range(n)]
            bm.append({"a1": a1, "a2": a2, "mask": ext._gen_block_mask(np.array(a1, np.int8), np.array(a2, np.int8)).astype(int).tolist()})
    cb = []
    for _ in range(40):
        r, c = rng.randint(1, 8), rng.randint(1, 40)
        m = [[rng.randint(0, 3) for _ in range(c)] for _ in range(r)]
        if rng.random() < 0.6:
            idx = [[rng.randrange(r)] + [rng.choice([-1] + list(range(r))) for _ in range(rng.randint(0, 3))] for _ in range(rng.randint(1, 5))]
            w = max(len(x) for x in idx)
            idx = [x + [-1] * (w - len(x)) for x in idx]
        else:
            idx = [rng.choice([-1] + list(range(r))) for _ in range(rng.randint(1, 6))]
        out = ext._combine_matrix_rows(np.array(m, np.int8), np.array(idx, np.int8))
        cb.append({"m": m, "idx": idx, "out": out.astype(int).tolist()})
    dump("native_vectors.json", {"source": "reference latok.c compiled into oracle/_ref", "block_mask": bm, "combine": cb})


def host_lib():
    sys.path.insert(0, ROOT)
    from latok_amd import _lib
    return _lib, _lib.load()


def corpus(lib_mod, lib, seed, model, n_str, lo, hi, sid0=0):
    row = np.zeros(n_str + 1, np.int64)
    lib_mod.check(lib.latok_corpus_offsets(seed, sid0, n_str, lo, hi, row.ctypes.data))
    cps = np.zeros(int(row[-1]), np.uint32)
    lib_mod.check(lib.latok_corpus_fill_host(seed, model, sid0, n_str, row.ctypes.data, cps.ctypes.data))
    return cps, row


def text_of(cps):
    return cps.astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")


def golden_c1(dt):
    lib_mod, lib = host_lib()
    # config 1: one fixed ~1 KB ASCII paragraph: the head of corpus C2's generator with every rule trigger appended
    tail = " See http://example.com/a?b=1, mail bob@host.org, #tag @user .@user camelCaseWord XMLHttp 42!"
    cps, _ = corpus(lib_mod, lib, 0x1A70C0DE, 0, 1, 1024 - len(tail), 1024 - len(tail))
    text = text_of(cps) + tail
    assert len(text) == 1024 and all(ord(c) < 128 for c in text)
    splits = dt.gen_split_mask(dt._gen_parse_matrix(text))
    dump("c1_paragraph.json", {
        "source": "real reference", "text": text, "n_chars": len(text),
        "splits": splits.astype(int).tolist(), "offsets": np.nonzero(splits)[0].astype(int).tolist(),
        "tokens": list(dt.tokenize(text))})


def golden_corpus(dt):
    lib_mod, lib = host_lib()
    out = {"source": "real reference on host-generated corpora (latok_amd/csrc/corpus_gen.h)", "corpora": {}}
    for name, seed, model, lo, hi in (("C2_ascii", 0x1A70C0DE, 0, 64, 192), ("C3_unicode", 0x1A70C0DF, 1, 128, 384)):
        n_str = 10000
        cps, row = corpus(lib_mod, lib, seed, model, n_str, lo, hi)
        h_off = hashlib.sha256()
        n_bound = 0
        for s in range(n_str):
            text = text_of(cps[row[s]:row[s + 1]])
            nz = np.nonzero(dt.gen_split_mask(dt._gen_parse_matrix(text)))[0].astype("<i8")
            h_off.update(nz.tobytes())
            n_bound += len(nz)
        out["corpora"][name] = {
            "seed": seed, "model": model, "n_str": n_str, "len_lo": lo, "len_hi": hi, "total_chars": int(row[-1]),
            "sha256_cps_u32le": hashlib.sha256(cps.astype("<u4").tobytes()).hexdigest(),
            "sha256_row_off_i64le": hashlib.sha256(row.astype("<i8").tobytes()).hexdigest(),
            "n_boundaries": n_bound,
            "sha256_offsets_i64le": h_off.hexdigest(),
            "first_string": cps[row[0]:row[1]].tolist(),
        }
    dump("corpus_samples.json", out)


def golden_rules(dt):
    """Custom rule tables through the real reference: gen_split_mask reads the module globals C_SPLIT / C_MASK / C_SYM
    (default_tokenizer.py:108-134), so installing other combo matrices there IS the reference's extension mechanism."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import RULE_SETS, random_rule_tables
    rng = random.Random(424242)
    sets = [(name, tabs) for name, tabs in RULE_SETS.items()] + [(f"random{i}", random_rule_tables(rng)) for i in range(6)]
    texts = [t for t in sample_strings() if 0 < len(t) <= 140]
    saved = (dt.C_SPLIT, dt.C_MASK, dt.C_SYM)
    out = []
    try:
        for name, (c_split, c_mask, c_sym) in sets:
            dt.C_SPLIT, dt.C_MASK, dt.C_SYM = (np.asarray(c_split, np.int8), np.asarray(c_mask, np.int8),
                                               np.asarray(c_sym, np.int8))
            items = []
            for text in texts:
                splits = dt.gen_split_mask(dt._gen_parse_matrix(text))
                items.append({"cps": cps_of(text).tolist(), "splits": splits.astype(int).tolist()})
            out.append({"name": name, "c_split": np.asarray(c_split).tolist(), "c_mask": np.asarray(c_mask).tolist(),
                        "c_sym": np.asarray(c_sym).tolist(), "items": items})
    finally:
        dt.C_SPLIT, dt.C_MASK, dt.C_SYM = saved
    dump("rules_strings.json", {"source": "real reference gen_split_mask with other combo matrices installed", "sets": out})


def main():
    dt = ref_loader.load_ref_python()
    ext = ref_loader.load_ref_ext()
    text = golden_notebook()
    assert text == "This is a #test! Testing, Testing, 1 2 3", repr(text)
    golden_strings(dt)
    golden_native(ext)
    golden_c1(dt)
    golden_corpus(dt)
    golden_rules(dt)


if __name__ == "__main__":
    main()
