"""The single-launch path of host batches of at most one tile (4096 chars, <= 512 strings): tokenize(text) /
featurize(text) of the drop-in surface (reference default_tokenizer.py:137-191) and small lists.  One wave computes the
boundaries, the per-string counts, the records and -- for featurize -- the 25 sums of every token (lane = token, popcounts
of the feature planes), stores a completion word into pinned memory, and the host polls that word.  Everything is compared
with the oracle; the sizes sit on the edges of the path (4095 / 4096 / 4097 chars, 512 / 513 strings)."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import ALPHABETS, ROOT, pack, random_strings

pytestmark = pytest.mark.gpu


def _want(oracle, t):
    """tokens of one string as (text, raw_start, raw_end, strip_start, strip_end, sums)"""
    if not t:
        return []
    m = oracle.gen_parse_matrix(t).astype(np.uint8)
    nz = oracle.split_offsets(t).tolist() + [len(t)]
    out = []
    for a, b in zip(nz[:-1], nz[1:]):
        s = t[a:b]
        if s.strip():
            a2 = a + len(s) - len(s.lstrip())
            out.append((s.strip(), a, b, a2, a2 + len(s.strip()), m[a:b].sum(axis=0, dtype=np.uint64).astype(np.uint8).astype(np.int8)))
    return out


def _small_batches():
    rng = random.Random(40964096)
    yield ["This is a #test! Testing, Testing, 1 2 3 -- see http://example.com/x or mail bob@host.org, camelCaseWord."]
    yield ["a"], [" "], ["  \t "], ["#"], ["@a"], ["x" * 64], ["x" * 63 + " "], [" " * 64 + "y"]
    for n in (4095, 4096):                                   # the largest batches the path takes
        yield ["".join(rng.choice(ALPHABETS["mixed"]) for _ in range(n))]
        yield ["http://" + "a" * (n - 7)]                    # one masked token over all 64 words
        yield ["w" * n]
        yield [" " * n]
    yield ["é" * 300 + "@" + "日" * 900 + " end", "", "tail #tag", ""]
    yield random_strings(rng, 512, 0, 8, ALPHABETS["starts"])            # 512 strings: the most the path takes
    yield random_strings(rng, 64, 0, 64, ALPHABETS["mixed"])
    yield random_strings(rng, 65, 0, 60, ALPHABETS["words"])             # counts beyond the first 64 strings
    yield [""] * 5 + ["a b"] + [""] * 70 + ["c"]
    for _ in range(40):
        n_str = rng.randint(1, 80)
        yield random_strings(rng, n_str, 0, 4096 // n_str, ALPHABETS[rng.choice(["mixed", "words", "rare_space_at", "bmp"])])
    # tokens that leave their 64-char word, in every alignment
    for lead in (0, 1, 31, 62, 63, 64, 65):
        yield ["x" * lead + " http://" + "".join(rng.choice("abcXYZ9_/.:é日") for _ in range(n)) + " e #t"
               for n in (50, 64, 130, 300)]


def _flat():
    out = []
    for b in _small_batches():
        for texts in (b if isinstance(b, tuple) else [b]):
            texts = list(texts)
            while sum(map(len, texts)) > 4096:      # (alphabets with multi-char entries overshoot)
                texts.pop()
            out.append(texts)
    return out


def _check_batch(batch, oracle, texts, dtype):
    assert sum(map(len, texts)) <= 4096 and len(texts) <= 512
    cps, row = pack(texts)
    want = [_want(oracle, t) for t in texts]
    counts, offs = batch.split_offsets_csr(cps, row, dtype=dtype)
    w_off = [oracle.split_offsets(t) if t else np.zeros(0, np.int64) for t in texts]
    assert counts.dtype == offs.dtype == dtype
    assert counts.tolist() == [len(x) for x in w_off] and offs.tolist() == [int(v) for x in w_off for v in x]
    tcounts, spans = batch.token_spans_csr(cps, row, dtype=dtype)
    assert tcounts.tolist() == [len(w) for w in want]
    assert spans.reshape(-1, 2).tolist() == [[w[3], w[4]] for ws in want for w in ws]
    fcounts, spans4, feats = batch.token_features_csr(cps, row, dtype=dtype)
    assert fcounts.tolist() == tcounts.tolist() and spans4.dtype == dtype and feats.dtype == np.int8
    assert spans4.reshape(-1, 4).tolist() == [[w[1], w[2], w[3], w[4]] for ws in want for w in ws]
    flat = [w[5] for ws in want for w in ws]
    assert np.array_equal(feats, np.stack(flat) if flat else np.zeros((0, 25), np.int8))


def test_small_batches_equal_the_oracle(gpu, oracle):
    from latok_amd import batch
    for i, texts in enumerate(_flat()):
        _check_batch(batch, oracle, texts, np.int32 if i & 1 else np.int64)


def test_small_batches_with_runtime_rule_tables(gpu, oracle):
    """the same through the rule interpreter (the built-in tables installed as run-time tables give the built-in results)"""
    from latok_amd import batch
    from latok_amd.core import default_tokenizer as dt
    batch.set_rules(dt.C_SPLIT, dt.C_MASK, dt.C_SYM)
    try:
        assert batch.rules_active()
        for i, texts in enumerate(_flat()[::3]):
            _check_batch(batch, oracle, texts, np.int64 if i & 1 else np.int32)
    finally:
        batch.reset_rules()


def test_drop_in_calls_one_string_at_a_time(gpu, oracle):
    from latok_amd.core import default_tokenizer as dt
    rng = random.Random(77)
    texts = [t for b in _flat() for t in b if t][:200] + random_strings(rng, 200, 1, 300, ALPHABETS["mixed"])
    for t in texts:
        want = _want(oracle, t)
        assert list(dt.tokenize(t)) == (oracle.tokenize(t))
        got = list(dt.featurize(t))
        assert [(x.text, x.start_idx, x.end_idx) for x in got] == [(w[0], w[1], w[2]) for w in want]
        assert all(np.array_equal(x.features, w[5]) for x, w in zip(got, want))
    # the vectors of one call are not views of a buffer the next call overwrites
    first = list(dt.featurize(texts[0]))
    keep = [x.features.copy() for x in first]
    list(dt.featurize("something else entirely, #with @other http://tokens.example/x"))
    assert all(np.array_equal(x.features, k) for x, k in zip(first, keep))


def test_edges_of_the_path(gpu, oracle):
    """one char / one string more than the path takes: the large pipeline gives the same answers"""
    from latok_amd import batch
    rng = random.Random(5)
    for texts in (["".join(rng.choice(ALPHABETS["mixed"]) for _ in range(4097))],
                  random_strings(rng, 513, 0, 7, ALPHABETS["starts"]),
                  random_strings(rng, 3, 1300, 1366, ALPHABETS["words"])):
        cps, row = pack(texts)
        counts, offs = batch.split_offsets_csr(cps, row)
        w_off = [oracle.split_offsets(t) if t else np.zeros(0, np.int64) for t in texts]
        assert counts.tolist() == [len(x) for x in w_off] and offs.tolist() == [int(v) for x in w_off for v in x]
        want = [_want(oracle, t) for t in texts]
        _, spans4, feats = batch.token_features_csr(cps, row)
        assert spans4.reshape(-1, 4).tolist() == [[w[1], w[2], w[3], w[4]] for ws in want for w in ws]
        assert np.array_equal(feats, np.stack([w[5] for ws in want for w in ws]))


def test_without_polling_the_call_waits_for_the_stream(gpu):
    """LATOK_SMALL_POLL=0: the same results with hipStreamSynchronize instead of the completion word"""
    code = ("from latok_amd.core import default_tokenizer as dt\n"
            "t = 'This is a #test! see http://a.b/c or mail me@x.org, camelCase 1 2 3'\n"
            "print(list(dt.tokenize(t)))\n"
            "print([(x.text, x.start_idx, x.end_idx, x.features.tolist()) for x in dt.featurize(t)])\n")
    outs = []
    for poll in ("1", "0"):
        env = dict(os.environ, LATOK_SMALL_POLL=poll, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] and "'#test'" in outs[0]


def test_three_native_functions_on_one_string(gpu, oracle):
    """_gen_parse_matrix / _combine_matrix_rows / _gen_block_mask (reference latok.c:373-378) called per string, as the
    reference's gen_split_mask does: up to 4096 chars each is one launch on pinned memory with a polled completion word;
    sizes on both sides of that edge and of the 256-char rounds of the matrix kernel."""
    from latok_amd import latok as lt
    from latok_amd.core import default_tokenizer as dt
    rng = random.Random(2468)
    for n in (1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 513, 1000, 4095, 4096, 4097, 5000):
        t = "".join(rng.choice(ALPHABETS["mixed"] + ["http://a", "b@c", " #t "]) for _ in range(n))[:n]
        m = lt._gen_parse_matrix(t)
        want = oracle.gen_parse_matrix(t)
        assert m.dtype == np.int8 and m.shape == (n, 25) and np.array_equal(m, want), n
        mt = m.T
        for tbl in (dt.C_SPLIT, dt.C_MASK, dt.C_SYM, np.array([0, 3, 5], np.int8)):
            assert np.array_equal(lt._combine_matrix_rows(mt, tbl), oracle.combine_matrix_rows(want.T, tbl)), n
        a1 = np.array([rng.random() < 0.05 for _ in range(n)], np.int8)
        a2 = np.array([rng.random() < 0.2 for _ in range(n)], np.int8)
        for x1, x2 in ((a1, a2), (a1, np.zeros(n, np.int8)), (np.zeros(n, np.int8), a2), (np.ones(n, np.int8), a2)):
            assert np.array_equal(lt._gen_block_mask(x1, x2), oracle.gen_block_mask(x1, x2)), n
        assert np.array_equal(dt.gen_split_mask(m), oracle.gen_split_mask(want)), n


def _multi_tile_batches():
    """host batches of 2 ... 26 tiles (k_one_segment takes up to 24): blocks that start in one tile and close tiles later, so
    that the resolve stage inside the single launch has to patch and to recompute tiles"""
    rng = random.Random(2424)
    for n_tiles in (2, 3, 5, 12, 23, 24, 25, 26):
        total = n_tiles * 4096 - rng.randint(0, 4095)
        for alpha in ("rare_space_at", "nospace_at", "mixed", "words"):
            texts, left = [], total
            while left > 0:
                n = min(left, rng.choice([1, 7, 100, 3000, 9000, 20000]))
                texts.append("".join(rng.choice(ALPHABETS[alpha]) for _ in range(n))[:n])
                left -= len(texts[-1])
            yield texts


def test_one_launch_pipeline_of_small_multi_tile_batches(gpu, oracle):
    from latok_amd import batch
    for i, texts in enumerate(_multi_tile_batches()):
        cps, row = pack(texts)
        vals, bits = oracle.split_batch(cps, row)
        assert np.array_equal(batch.split_mask_batch(cps, row), bits), i
        want = [np.nonzero(vals[row[s]:row[s + 1]])[0] for s in range(len(texts))]
        dtype = np.int32 if i & 1 else np.int64
        counts, offs = batch.split_offsets_csr(cps, row, dtype=dtype)
        assert counts.tolist() == [len(w) for w in want] and np.array_equal(offs, np.concatenate(want)), i
        if i % 4 == 0:
            toks = [_want(oracle, t) for t in texts]
            tcounts, spans = batch.token_spans_csr(cps, row, dtype=dtype)
            assert spans.reshape(-1, 2).tolist() == [[w[3], w[4]] for ws in toks for w in ws], i
            _, spans4, feats = batch.token_features_csr(cps, row, dtype=dtype)
            assert spans4.reshape(-1, 4).tolist() == [[w[1], w[2], w[3], w[4]] for ws in toks for w in ws], i
            assert np.array_equal(feats, np.stack([w[5] for ws in toks for w in ws])), i


def test_one_launch_pipeline_equals_the_three_launch_form(gpu):
    """LATOK_ONE_SEGMENT=0 / LATOK_SMALL_POLL=0: the same offsets from the three-launch pipeline and from the stream wait"""
    code = ("import random, hashlib, numpy as np\n"
            "from latok_amd import batch\n"
            "rng = random.Random(7)\n"
            "h = hashlib.sha256()\n"
            "for n in (5000, 20000, 60000, 98000, 99000, 200000):\n"
            "    texts = [''.join(rng.choice('abcdefgh@ /:.#') for _ in range(rng.choice([3, 50, 4000]))) for _ in range(n // 700 + 1)]\n"
            "    cps, row = batch.pack(texts)\n"
            "    for a in batch.split_offsets_csr(cps, row) + batch.token_spans_csr(cps, row, dtype=np.int32) + (batch.split_mask_batch(cps, row),):\n"
            "        h.update(np.ascontiguousarray(a).tobytes())\n"
            "print(h.hexdigest())\n")
    outs = []
    for env_add in ({}, {"LATOK_ONE_SEGMENT": "0"}, {"LATOK_SMALL_POLL": "0"}):
        env = dict(os.environ, PYTHONPATH=ROOT, **env_add)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.strip())
    assert len(outs[0]) == 64 and outs[0] == outs[1] == outs[2]


def test_c_example_one_string_per_call(gpu, oracle, tmp_path):
    """examples/tokenize_one.c: a plain C caller tokenizes one string per call (UTF-32 in, int32 stripped spans out) and
    prints the reference's tokens; the timing it reports goes to stderr."""
    exe = str(tmp_path / "tokenize_one")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "tokenize_one.c"),
                           "-L" + os.path.join(ROOT, "latok_amd"), "-llatok_hip", "-Wl,-rpath," + os.path.join(ROOT, "latok_amd"),
                           "-o", exe])
    texts = ["This is a #test! Testing, Testing, 1 2 3", "see http://a.b/c or mail me@x.org", "camelCase 日本語 🤓 é", "x" * 5000 + " tail"]
    r = subprocess.run([exe, "50"] + texts, capture_output=True, timeout=120, check=True)
    out = r.stdout.decode("utf-8").splitlines()
    assert out == [f"{i}:" + "".join(f" [{t}]" for t in oracle.tokenize(s)) for i, s in enumerate(texts)]
    assert r.stderr.decode().count("us per call") == len(texts)
    # no arguments: the built-in sentence
    out = subprocess.run([exe, "3"], capture_output=True, timeout=120, check=True).stdout.decode("utf-8")
    assert out.startswith("0: [This] [is] [a] [#test] [!]")


def test_small_narrow_unit_batches_take_the_same_path(gpu, oracle):
    """PEP 393 kind 1 / 2 code units and pure-ASCII UTF-8 in byte space, small host batches (one str per call is what a C
    extension hands over): widened on the host into the pinned small-batch path; the results
    are those of the UTF-32 entry points and of the oracle.  Non-ASCII UTF-8 keeps the byte-space kernel."""
    from latok_amd import batch
    rng = random.Random(393)
    cases = [["This is a #test! see http://a.b/c or mail me@x.org, camelCase 1 2 3"], ["a"], ["", "x y", ""],
             random_strings(rng, 30, 0, 100, ALPHABETS["latin1"]), random_strings(rng, 30, 0, 100, ALPHABETS["bmp"]),
             random_strings(rng, 200, 0, 60, ALPHABETS["latin1"]), random_strings(rng, 3, 1500, 4000, ALPHABETS["bmp"]),
             ["".join(rng.choice("abc XYZ.,:/@#1") for _ in range(n)) for n in (4095, 4096, 4097, 20000)]]
    for i, texts in enumerate(cases):
        cps, row = pack(texts)
        units, krow = batch.pack_kind(texts)
        dtype = np.int32 if i & 1 else np.int64
        want_o = batch.split_offsets_csr(cps, row, dtype=dtype)
        w_off = [oracle.split_offsets(t) if t else np.zeros(0, np.int64) for t in texts]
        assert want_o[1].tolist() == [int(v) for x in w_off for v in x]
        for got, want in ((batch.split_offsets_kind_csr(units, krow, dtype=dtype), want_o),
                          (batch.token_spans_kind_csr(units, krow, dtype=dtype), batch.token_spans_csr(cps, row, dtype=dtype)),
                          (batch.token_features_kind_csr(units, krow, dtype=dtype), batch.token_features_csr(cps, row, dtype=dtype))):
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), (i, units.dtype)
        if all(ord(c) < 128 for t in texts for c in t):
            utf8, boff = batch.pack_utf8([t.encode() for t in texts])
            assert all(np.array_equal(a, b) for a, b in zip(batch.split_offsets_utf8_bytes_csr(utf8, boff, dtype=dtype), want_o))
            assert all(np.array_equal(a, b) for a, b in zip(batch.token_spans_utf8_bytes_csr(utf8, boff, dtype=dtype),
                                                            batch.token_spans_csr(cps, row, dtype=dtype)))


def test_small_utf8_batches_are_decoded_by_the_host(gpu, oracle):
    """UTF-8 host batches up to the pinned path's size: the host decodes (the device decoder's rule) and the call takes the
    UTF-32 small-batch path; byte-space results are mapped back through the byte position of every char.  Against the
    oracle, for both forms; malformed strings are left to the device paths and give what they give inside a large batch."""
    from latok_amd import batch
    rng = random.Random(8)
    cases = [["This is a #test! é日🤓 see http://a.b/c or mail me@x.org"], ["é"], ["", "日本語 x", ""], ["a" * 4096], ["é" * 3000],
             random_strings(rng, 40, 0, 90, ALPHABETS["mixed"]), random_strings(rng, 300, 0, 20, ALPHABETS["bmp"][:18] + ["é", "日"]),
             random_strings(rng, 4, 3000, 9000, ALPHABETS["mixed"])]
    for i, texts in enumerate(cases):
        texts = [t.encode("utf-8", "surrogatepass").decode("utf-8", "surrogatepass") for t in texts]
        blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
        utf8, boff = batch.pack_utf8(blobs)
        dtype = np.int32 if i & 1 else np.int64
        w_off = [oracle.split_offsets(t) if t else np.zeros(0, np.int64) for t in texts]
        toks = [_want(oracle, t) for t in texts]
        bpos = [np.cumsum([0] + [len(c.encode("utf-8", "surrogatepass")) for c in t]) for t in texts]   # byte position of every char
        counts, offs = batch.split_offsets_utf8_csr(utf8, boff, dtype=dtype)
        assert counts.tolist() == [len(x) for x in w_off] and offs.tolist() == [int(v) for x in w_off for v in x], i
        counts, offs = batch.split_offsets_utf8_bytes_csr(utf8, boff, dtype=dtype)
        assert offs.dtype == dtype and offs.tolist() == [int(bp[v]) for x, bp in zip(w_off, bpos) for v in x], i
        _, spans = batch.token_spans_utf8_csr(utf8, boff, dtype=dtype)
        assert spans.reshape(-1, 2).tolist() == [[w[3], w[4]] for ws in toks for w in ws], i
        _, spans = batch.token_spans_utf8_bytes_csr(utf8, boff, dtype=dtype)
        assert spans.reshape(-1, 2).tolist() == [[int(bp[w[3]]), int(bp[w[4]])] for ws, bp in zip(toks, bpos) for w in ws], i
        assert batch.tokenize_utf8_batch(blobs) == [[w[0].encode("utf-8", "surrogatepass") for w in ws] for ws in toks], i
    # malformed: stray continuation bytes, a truncated sequence inside and at the end of a string
    bad = [b"ab \x80\x80 cd", b"x \xe6\x97 y", b"tail \xf0\x9f", b"\xbfstart", b"ok #tag"]
    filler = [b"filler words and a #tag http://x.y/z " * 40] * 300          # > 256 K bytes: the device paths
    for fn, width in ((batch.split_offsets_utf8_bytes_csr, 1), (batch.token_spans_utf8_bytes_csr, 2),
                      (batch.split_offsets_utf8_csr, 1), (batch.token_spans_utf8_csr, 2)):
        small = fn(*batch.pack_utf8(bad))
        big = fn(*batch.pack_utf8(bad + filler))
        n = int(small[0].sum())
        assert n > 0 and np.array_equal(small[0], big[0][:len(bad)]), fn.__name__
        assert np.array_equal(np.ravel(small[1]), np.ravel(big[1])[:n * width]), fn.__name__
