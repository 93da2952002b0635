"""Host-side logic and the C-ABI surface, without a GPU: the library loads and exports every symbol the header
declares, the Python mirror has the reference's names/values, compute fails loudly when no device exists, and the
product never reaches into oracle/."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, has_gpu


def header_symbols():
    text = open(os.path.join(ROOT, "include", "latok_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(latok_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from latok_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    assert sorted(_lib.SIGNATURES) == syms, "latok_amd/_lib.py and include/latok_hip.h disagree"
    for name in syms:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.latok_version()
    assert lib.latok_device_count() >= 0


def test_no_cpu_fallback_without_device():
    if has_gpu():
        pytest.skip("a GPU is present")
    from latok_amd import _lib, batch
    from latok_amd.core import default_tokenizer as dt
    lib = _lib.load()
    assert lib.latok_init(0) == _lib.ERR_HIP and b"no HIP device" in lib.latok_last_error()
    out = np.zeros(1, np.uint64)
    cps = np.array([97, 98], np.uint32)
    row = np.array([0, 2], np.int64)
    assert lib.latok_split_mask_batch(cps.ctypes.data, row.ctypes.data, 1, 2, out.ctypes.data, 0, None) == _lib.ERR_NOT_INIT
    with pytest.raises(RuntimeError):
        batch.split_mask_batch(cps, row)
    with pytest.raises(RuntimeError):
        list(dt.tokenize("no gpu here"))


def test_offsets_constants_match_reference():
    """values of reference latok/core/offsets.py:3-49"""
    from latok_amd.core import offsets as oft
    masks = ["ALPHA", "DECIMAL", "DIGIT", "LOWER", "LINEBREAK", "SPACE", "TITLE", "UPPER", "XID_START", "XID_CONTINUE",
             "PRINTABLE", "NUMERIC", "CASE_IGNORABLE", "CASED", "EXTENDED_CASE", "SPECIALS", "CHAR_AT", "CHAR_COLON",
             "CHAR_SLASH", "CHAR_PERIOD"]
    for i, n in enumerate(masks):
        assert getattr(oft, n + "_MASK") == 1 << i
    assert oft.CHAR_PERIOD_MASK == 0x080000 and oft.SPECIALS_MASK == 0x8000 and oft.PRINTABLE_MASK == 0x400
    g = json.load(open(os.path.join(GOLDEN, "g1_notebook.json")))
    from latok_amd.core.latok_utils import FEATURE_NAMES, NUM_FEATURES
    assert FEATURE_NAMES == g["feature_names"] and NUM_FEATURES == oft.FEATURE_COUNT == 25
    cols = dict(ALPHA=0, ALPHA_NUM=1, NUM=2, LOWER=3, UPPER=4, SPACE=5, SYMBOL=6, TWITTER=7, CHAR_AT=8, CHAR_COLON=9,
                CHAR_SLASH=10, CHAR_PERIOD=11, PREV_ALPHA=12, NEXT_ALPHA=13, PREV_ALPHA_NUM=14, NEXT_ALPHA_NUM=15,
                PREV_LOWER=16, NEXT_LOWER=17, PREV_SPACE=18, NEXT_SPACE=19, PREV_SYMBOL=20, NEXT_AT=21, NEXT_SLASH=22,
                AFTER_NEXT_ALPHA=23, AFTER_NEXT_SLASH=24)
    for n, v in cols.items():
        assert getattr(oft, n + "_IDX") == v


def test_rule_tables_match_reference():
    """C_SPLIT / C_MASK / C_SYM values of reference default_tokenizer.py:108-110."""
    from latok_amd.core import default_tokenizer as dt
    from latok_amd.core.latok_utils import build_combo_matrix
    assert dt.C_SPLIT.dtype == np.int8
    assert dt.C_SPLIT.tolist() == [[5, -1], [6, -1], [20, -1], [4, 17], [4, 16]]
    assert dt.C_MASK.tolist() == [[7, 18, 13, -1], [11, 18, 21, 23], [8, 14, 15, -1], [9, 22, 24, 12]]
    assert dt.C_SYM.tolist() == [[6, 19]]
    assert build_combo_matrix([[1], [2, 3, 4]]).tolist() == [[1, -1, -1], [2, 3, 4]]


def test_pack_and_token_materialisation(oracle):
    """host logic of latok_amd.batch: CSR packing and the reference's slice/strip loop, fed with oracle offsets."""
    from latok_amd import batch
    texts = ["This is a #test! Testing, Testing, 1 2 3", "", " ", "a", "  lead and trail  ", "日本語 テキスト🤓", "x\t\ny"]
    cps, row = batch.pack(texts)
    assert row.tolist() == np.cumsum([0] + [len(t) for t in texts]).tolist()
    assert cps.dtype == np.uint32 and cps.size == row[-1]
    for t in texts:
        if t:
            assert batch.spans_from_offsets(t, oracle.split_offsets(t)) == oracle.tokenize(t)
    assert batch.spans_from_offsets("", np.zeros(0, np.int64)) == []
    with pytest.raises(ValueError):
        batch._csr(np.zeros(3, np.uint32), np.array([0, 5], np.int64))


def test_pack_kind_picks_the_narrowest_pep393_kind():
    """batch.pack_kind: the code units CPython itself would store for the joined text (kind 1 / 2 / 4), same chars as pack()."""
    from latok_amd import batch
    cases = [(["abc", "", "d\xe9\xff"], np.uint8), (["abc", "\u65e5\u672c", "\ud800x"], np.uint16),
             (["abc", "\u65e5", "\U0001f913"], np.uint32), ([], np.uint8), (["", ""], np.uint8)]
    for texts, dtype in cases:
        units, row = batch.pack_kind(texts)
        cps, row32 = batch.pack(texts)
        assert units.dtype == dtype and np.array_equal(row, row32)
        assert np.array_equal(units.astype(np.uint32), cps)
    with pytest.raises(ValueError):
        batch._csr_kind(np.zeros(3, np.int32), np.array([0, 3], np.int64))
    with pytest.raises(ValueError):
        batch._csr_kind(np.zeros(3, np.uint16), np.array([0, 5], np.int64))
    assert batch._narrow_pays(["x" * 300000]) and batch._narrow_pays([""] * 17000) and not batch._narrow_pays(["abc"])


def test_compat_argument_errors_do_not_need_a_gpu():
    from latok_amd import latok as ext
    with pytest.raises(ValueError, match="must specify string"):
        ext._gen_parse_matrix()
    with pytest.raises(ValueError, match="two aligning 1d"):
        ext._gen_block_mask(np.zeros(3))
    with pytest.raises(ValueError, match="1d numpy array args"):
        ext._gen_block_mask(np.zeros((2, 2)), np.zeros(4))
    with pytest.raises(ValueError, match="matching length"):
        ext._gen_block_mask(np.zeros(3), np.zeros(4))
    with pytest.raises(ValueError, match="2d m and idxs"):
        ext._combine_matrix_rows(np.zeros((2, 2)))
    with pytest.raises(ValueError, match="2d numpy array args"):
        ext._combine_matrix_rows(np.zeros(4), np.zeros(2))


def test_product_never_touches_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "latok_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc")) or f == "Makefile":
                src = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"latok_oracle|fused_model|oracle/|ref_loader|import oracle", src):
                    # comments that merely mention the oracle directory are fine; imports / includes / links are not
                    for line in src.splitlines():
                        s = line.strip()
                        if re.search(r"latok_oracle|fused_model|ref_loader", s) and not s.startswith(("//", "#", "*", "/*")):
                            bad.append((f, s))
    assert not bad, bad
    # and the shared object has no dependency on oracle libraries
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "latok_amd", "liblatok_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "fused_model" not in out


def test_install_as_latok_aliases_reference_import_names():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import latok_amd; latok_amd.install_as_latok(); "
            "from latok.core.default_tokenizer import tokenize, featurize, gen_split_mask, C_SPLIT; "
            "from latok.latok import _gen_parse_matrix, _gen_block_mask, _combine_matrix_rows; "
            "from latok.core.latok_utils import gen_parse_matrix, LaToken, FEATURE_NAMES; import latok.core.offsets as oft; "
            "import latok_amd.core.default_tokenizer as d; assert tokenize is d.tokenize and oft.FEATURE_COUNT == 25; print('ok')") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]


def test_latok_import_name_is_served_without_any_setup_call():
    """The reference's own import lines (default_tokenizer.py:33-36, scripts/timing/time_tokenizer.py:20-21) in a FRESH
    interpreter, with nothing but the repo on sys.path and no install_as_latok(): the top-level ``latok`` package serves
    them with the latok_amd objects themselves."""
    import subprocess
    import sys
    code = ("from latok.core.default_tokenizer import tokenize, featurize, gen_split_mask, C_SPLIT, C_MASK, C_SYM\n"
            "from latok.latok import _gen_parse_matrix, _gen_block_mask, _combine_matrix_rows\n"
            "from latok.core.latok_utils import gen_parse_matrix, gen_block_mask, build_combo_matrix, LaToken, FEATURE_NAMES\n"
            "import latok.core.offsets as oft\n"
            "import latok, latok_amd, latok_amd.latok, latok_amd.core.default_tokenizer as d\n"
            "assert tokenize is d.tokenize and _gen_parse_matrix is latok_amd.latok._gen_parse_matrix\n"
            "assert latok.latok is latok_amd.latok and latok.core.default_tokenizer is d and oft.FEATURE_COUNT == 25\n"
            "assert latok.__version__ == latok_amd.__version__\n"
            "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]


def test_latok_shim_refuses_to_shadow_a_foreign_latok(tmp_path):
    import subprocess
    import sys
    other = tmp_path / "elsewhere" / "latok"
    other.mkdir(parents=True)
    (other / "__init__.py").write_text("WHO = 'someone else'\n")
    code = "import sys; sys.path.insert(0, %r); sys.path.append(%r); import latok" % (ROOT, str(tmp_path / "elsewhere"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert out.returncode != 0 and "would shadow another 'latok' package" in out.stderr
    env = dict(os.environ, LATOK_AMD_ALLOW_SHADOW="1")
    out = subprocess.run([sys.executable, "-c", code + "; print(latok.__version__)"], capture_output=True, text=True, timeout=120,
                         cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stderr[-800:]


def test_shard_bounds_balance_and_cover():
    """latok_amd.shard: contiguous string ranges balanced by chars; shards concatenate back to the batch."""
    import random
    from latok_amd import shard
    rng = random.Random(3)
    for n_str, world in [(1, 1), (1, 8), (7, 8), (1000, 8), (1000, 3), (50, 2), (0, 4)]:
        lens = np.array([rng.choice([0, 1, 5, 128, 4000, 100000]) for _ in range(n_str)], np.int64)
        row = np.zeros(n_str + 1, np.int64)
        np.cumsum(lens, out=row[1:])
        cps = np.arange(int(row[-1]), dtype=np.uint32)
        b = shard.shard_bounds(row, world)
        assert b[0] == 0 and b[-1] == n_str and len(b) == world + 1 and np.all(np.diff(b) >= 0)
        parts = [shard.take_shard(cps, row, r, world) for r in range(world)]
        assert np.array_equal(np.concatenate([p[0] for p in parts]) if parts else cps, cps)
        for (c, ro, s0), r in zip(parts, range(world)):
            assert ro[0] == 0 and ro[-1] == c.size and s0 == b[r]
        if n_str >= 8 * world and row[-1] > 0:
            sizes = np.array([p[0].size for p in parts])
            assert sizes.max() - sizes.min() <= 2 * lens.max()      # balanced up to one string per cut


def test_header_is_plain_c_and_example_links(tmp_path):
    """include/latok_hip.h is the C ABI: it has to compile as C99 (no C++ or torch types), and a C program that uses it
    links against the library without anything else."""
    import subprocess
    from conftest import ROOT
    src = os.path.join(ROOT, "examples", "tokenize_utf8.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-fsyntax-only", src])
    exe = str(tmp_path / "tokenize_utf8")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), src, "-L" + os.path.join(ROOT, "latok_amd"),
                           "-llatok_hip", "-Wl,-rpath," + os.path.join(ROOT, "latok_amd"), "-o", exe])
    assert os.path.exists(exe)


def test_host_decoder_of_small_utf8_batches():
    """api.cpp: host_decode_small (small UTF-8 host batches are decoded by the host into the UTF-32 small-batch path): equal
    to Python's decoder on well-formed strings, including surrogatepass forms and 4-byte chars; malformed batches are
    refused (they stay with the device paths).  No device needed."""
    import ctypes as C
    import random
    from latok_amd import _lib
    lib = _lib.load()
    fn = lib.latok_debug_host_decode_utf8
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]

    def run(blobs):
        buf = np.frombuffer(b"".join(blobs) + b"\0", np.uint8)
        off = np.zeros(len(blobs) + 1, np.int64)
        np.cumsum([len(b) for b in blobs], out=off[1:])
        n = int(off[-1])
        cps, row, pos, k = np.zeros(n + 1, np.uint32), np.zeros(len(blobs) + 1, np.int64), np.zeros(n + 2, np.int64), C.c_int64(0)
        rc = fn(buf.ctypes.data, off.ctypes.data, len(blobs), cps.ctypes.data, row.ctypes.data, pos.ctypes.data, C.byref(k))
        return rc, cps[:k.value], row, pos[:k.value + 1]

    rng = random.Random(11)
    alphabet = list("ab \t#@:/.é\xff日あ🤓\U0010ffff́𐏿\x00\x7f\x80߿ࠀ￿")
    for _ in range(200):
        texts = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 40))) for _ in range(rng.randint(1, 8))]
        blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
        if sum(map(len, blobs)) == 0:
            continue
        rc, cps, row, pos = run(blobs)
        assert rc == 1
        assert cps.tolist() == [ord(c) for t in texts for c in t]
        assert row.tolist() == np.cumsum([0] + [len(t) for t in texts]).tolist()
        want_pos, b = [], 0
        for t in texts:
            for c in t:
                want_pos.append(b)
                b += len(c.encode("utf-8", "surrogatepass"))
        assert pos.tolist() == want_pos + [b]
    for bad in ([b"ab\x80"], [b"\xbf"], [b"x\xe6\x97", b"\xa5y"], [b"\xf0\x9f\xa4"], [b"\xc3"], [b"ok", b"\xe6\x97 z"], [b"\xc3\x28"]):
        assert run(bad)[0] == 0, bad


def test_byte_space_class_table_is_the_generated_table_cut_at_six_bits():
    """api.cpp: build_byte_tables -- the byte-space kernel classifies through its own two-stage table (stage 1 by cp >> 6 as
    uint16 block offsets, stage 2 by the last UTF-8 byte's payload; split_code.h LK_B6_*).  Every code point must get the code the
    generated 7-bit table (unicode_tables.inc, pinned by tests/golden/unicode_classes.json through the oracle tests) gives it,
    for split codes and rule codes; ASCII must be the first 128 stage-2 bytes; the last stage-1 entry (cp >= 0x110000, and what
    a decode slot without a lead points at) must be a block of zeros.  No device needed."""
    import ctypes as C
    import re
    from latok_amd import _lib
    fn = _lib.load().latok_debug_byte_tables
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_int64]
    src = open(os.path.join(ROOT, "latok_amd", "csrc", "unicode_tables.inc")).read()

    def arr(name):
        body = re.search(name + r"\[\d+\] = \{(.*?)\};", src, re.S).group(1)
        return np.array([int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", body)], np.int64)

    s1, s2 = arr("kStage1"), arr("kStage2")
    cps = np.arange(0x110000, dtype=np.int64)
    cls = s2[(s1[cps >> 7] << 7) | (cps & 127)]
    n1 = (0x110000 >> 6) + 1
    for rule, name in ((0, "kClassCode"), (1, "kClassRuleCode")):
        blob = np.zeros(1 << 17, np.uint8)
        off = fn(rule, blob.ctypes.data, blob.size)
        assert off == (2 * n1 + 1023) // 1024 * 1024, _lib.load().latok_last_error()
        t1 = blob[:2 * n1].view("<u2").astype(np.int64)
        t2 = blob[off:]
        assert (t1 % 64 == 0).all() and t1.max() + 64 <= 400 * 64
        want = arr(name)[cls]
        assert np.array_equal(t2[t1[cps >> 6] | (cps & 63)], want)
        assert np.array_equal(t2[:128], want[:128])                     # ASCII without stage 1
        assert not t2[t1[-1]:t1[-1] + 64].any()                         # cp >= 0x110000 / "no lead here"
        assert fn(rule, blob.ctypes.data, 1000) < 0


def _router():
    import ctypes as C
    from latok_amd import _lib
    fn = _lib.load().latok_debug_flow_route
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]

    def submit(*ranges, slots=2):
        """ranges: (base, bytes, 'w' | 'r'); returns (slot, drained)"""
        lo = np.array([r[0] for r in ranges], np.uint64)
        nb = np.array([r[1] for r in ranges], np.uint64)
        wr = np.array([r[2] == "w" for r in ranges], np.int32)
        d = C.c_int(0)
        s = fn(slots, lo.ctypes.data, nb.ctypes.data, wr.ctypes.data, len(ranges), C.byref(d))
        assert s >= 0
        return s, d.value

    def reset():
        fn(2, None, None, None, -1, None)

    reset()
    return submit, reset


def test_flow_routing_orders_every_conflicting_pair():
    """flow_hazards.h, the routing of latok_flow_* batches to slots (= streams), without a device.  The reference's contract
    is that every call is independent (default_tokenizer.py:137-160): two batches in flight that touch the same memory must
    share a stream, whichever batches went in between."""
    submit, reset = _router()
    X, Y, Z, IN = 0x10000, 0x20000, 0x30000, 0x900000
    # the round-3 hole: A->X (slot 0), B->Y (1), C->Z (0), D->X would have gone to slot 1 with nothing ordering it behind A
    assert submit((X, 4096, "w"), (IN, 1 << 20, "r")) == (0, 0)
    assert submit((Y, 4096, "w"), (IN, 1 << 20, "r")) == (1, 0)
    assert submit((Z, 4096, "w"), (IN, 1 << 20, "r")) == (0, 0)
    assert submit((X, 4096, "w"), (IN, 1 << 20, "r")) == (0, 0)          # behind A on A's stream
    # the turn has moved on by one for every batch: the next free batch takes the next slot in turn
    assert submit((0x40000, 4096, "w"))[0] == 0
    assert submit((0x50000, 4096, "w"))[0] == 1
    # partial overlap (the tail of Y) is a conflict; touching ranges are not
    reset()
    assert submit((Y, 4096, "w")) == (0, 0)
    assert submit((Z, 4096, "w")) == (1, 0)
    assert submit((0x60000, 64, "w")) == (0, 0)
    assert submit((Y + 4000, 4096, "w")) == (0, 0)    # turn says slot 1; the overlap with Y sends it to slot 0
    assert submit((Z + 4096, 64, "w"))[1] == 0        # starts where Z ends: free
    # a batch in the way of batches on BOTH slots: the flow is drained first
    reset()
    assert submit((X, 4096, "w")) == (0, 0)
    assert submit((Y, 4096, "w")) == (1, 0)
    s, drained = submit((X + 8, 8, "w"), (Y + 8, 8, "w"))
    assert drained == 1
    assert submit((X, 4096, "w"))[1] == 0 and submit((Y, 4096, "w"))[1] == 0
    # reads: two readers of one buffer never conflict; a writer of a buffer that is being read does (and the other way round)
    reset()
    assert submit((X, 64, "w"), (IN, 4096, "r")) == (0, 0)
    assert submit((Y, 64, "w"), (IN, 4096, "r")) == (1, 0)
    assert submit((Z, 64, "w"), (Y, 64, "r")) == (1, 0)                 # reads what the batch on slot 1 writes (turn: 0)
    reset()
    assert submit((X, 64, "w"), (IN, 4096, "r")) == (0, 0)
    assert submit((Y, 64, "w")) == (1, 0)
    assert submit((Z, 64, "w")) == (0, 0)
    assert submit((IN + 100, 8, "w")) == (0, 0)                         # overwrites input of the first batch (turn: 1)
    # secondary outputs count too (counts / result words of a compaction batch): same counts buffer, different records
    reset()
    assert submit((X, 4096, "w"), (0x70000, 800, "w"), (0x80000, 16, "w")) == (0, 0)
    assert submit((Y, 4096, "w"), (0x71000, 800, "w"), (0x80100, 16, "w")) == (1, 0)
    assert submit((Z, 4096, "w"), (0x72000, 800, "w"), (0x80100, 16, "w")) == (1, 0)   # result words of the batch on slot 1
    # alternating two buffers for ever keeps the lists short and the slots in turn
    reset()
    for i in range(1000):
        assert submit((X if i % 2 == 0 else Y, 4096, "w"), (IN, 1 << 20, "r")) == (i % 2, 0)
    # three buffers in rotation over two slots: every reuse lands behind its previous writer
    reset()
    where = {}
    for i in range(300):
        buf = (X, Y, Z)[i % 3]
        s, drained = submit((buf, 4096, "w"))
        if buf in where and not drained:
            assert s == where[buf]
        if drained:
            where = {}
        where[buf] = s
    reset()
