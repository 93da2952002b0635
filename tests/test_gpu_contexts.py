"""Per-device contexts of the C ABI (include/latok_hip.h "contexts") and the in-process sharding driver
(latok_amd.multi) on the GPU: contexts share nothing, so calls on two of them overlap -- also two on ONE device, which is
what a 1-GPU box can show -- and every result is still the oracle's."""
import ctypes as C
import random
import threading

import numpy as np
import pytest

from conftest import ALPHABETS, pack, random_strings

pytestmark = pytest.mark.gpu


def _oracle_offsets(oracle, cps, row):
    vals, bits = oracle.split_batch(cps, row)
    per = [np.nonzero(vals[row[s]:row[s + 1]])[0] for s in range(len(row) - 1)]
    return np.array([len(p) for p in per], np.int64), np.concatenate(per), bits


def test_two_contexts_on_one_device_run_concurrently_and_exactly(gpu, oracle):
    from latok_amd import _lib, batch
    rng = random.Random(99)
    # different batches per thread, sized so that the workspaces of the two contexts would collide if they were shared
    work = []
    for k in range(2):
        texts = random_strings(rng, 3000, 0, 200, ALPHABETS["mixed"]) + random_strings(rng, 3, 20000, 60000, ALPHABETS["rare_space_at"])
        rng.shuffle(texts)
        cps, row = pack(texts)
        work.append((cps, row, _oracle_offsets(oracle, cps, row)))
    errors = []
    gate = threading.Barrier(2, timeout=60)

    def run(k):
        try:
            with _lib.Context(0) as ctx:
                assert ctx.device == 0 and gpu.latok_ctx_device(ctx.handle) == 0
                cps, row, (wc, wo, wb) = work[k]
                gate.wait()
                for _ in range(40):
                    assert np.array_equal(batch.split_mask_batch(cps, row), wb)
                    c, o = batch.split_offsets_csr(cps, row)
                    assert np.array_equal(c, wc) and np.array_equal(o, wo)
                    cs, sp = batch.token_spans_csr(cps, row)
                    assert int(cs.sum()) == len(sp) <= len(o)
            ctx.destroy()
        except BaseException as exc:   # noqa: BLE001 - reported to the main thread
            errors.append(exc)
            try:
                gate.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not errors, errors[0]
    # the default context is untouched and still works from this thread
    cps, row, (wc, wo, wb) = work[0]
    assert gpu.latok_ctx_get_current() is None
    assert np.array_equal(batch.split_mask_batch(cps, row), wb)


def test_device_pool_equals_single_context(gpu, oracle):
    from latok_amd import batch, multi
    rng = random.Random(5)
    texts = random_strings(rng, 20000, 0, 120, ALPHABETS["mixed"]) + ["", "日本語のテキスト、です。 🤓 ok"] + \
        random_strings(rng, 2, 100000, 200000, ALPHABETS["words"])
    cps, row = pack(texts)
    wc, wo, wb = _oracle_offsets(oracle, cps, row)
    with multi.DevicePool([0, 0, 0]) as pool:          # three contexts on the one GPU of this box
        c, o = multi.split_offsets_csr(cps, row, pool)
        assert np.array_equal(c, wc) and np.array_equal(o, wo)
        assert np.array_equal(multi.split_mask_batch(cps, row, pool), wb)
        assert multi.tokenize_batch(texts, pool) == batch.tokenize_batch(texts)
        cs, sp = multi.token_spans_csr(cps, row, pool)
        cs1, sp1 = batch.token_spans_csr(cps, row)
        assert np.array_equal(cs, cs1) and np.array_equal(sp, sp1)
        f = multi.token_features_csr(cps, row, pool)
        f1 = batch.token_features_csr(cps, row)
        assert all(np.array_equal(a, b) for a, b in zip(f, f1))
        small = texts[:50]
        assert [[t.text for t in toks] for toks in multi.featurize_batch(small, pool)] == \
               [[t.text for t in toks] for toks in batch.featurize_batch(small)]
    # the keyword form of the batch API: a list of device ids makes a temporary pool
    assert batch.tokenize_batch(texts[:500], devices=[0, 0]) == batch.tokenize_batch(texts[:500])
    offs = batch.split_offsets_batch(texts[:500], devices=[0, 0])
    assert all(np.array_equal(a, b) for a, b in zip(offs, batch.split_offsets_batch(texts[:500])))


def test_rule_tables_belong_to_their_context(gpu, oracle):
    from conftest import RULE_SETS, oracle_rule_bits
    from latok_amd import _lib, batch, multi
    rng = random.Random(31)
    texts = random_strings(rng, 400, 1, 90, ALPHABETS["mixed"])
    cps, row = pack(texts)
    tables = RULE_SETS["sym_everywhere"]
    want_custom = oracle_rule_bits(oracle, texts, tables)
    _, want_default = oracle.split_batch(cps, row, want_values=False)
    assert not np.array_equal(want_custom, want_default)
    with _lib.Context(0) as ctx:
        batch.set_rules(*tables)
        assert batch.rules_active()
        assert np.array_equal(batch.split_mask_batch(cps, row), want_custom)
    ctx.destroy()
    assert not batch.rules_active()                                    # the default context never saw them
    assert np.array_equal(batch.split_mask_batch(cps, row), want_default)
    with multi.DevicePool([0, 0]) as pool:
        pool.set_rules(*tables)
        assert np.array_equal(multi.split_mask_batch(cps, row, pool), want_custom)
        pool.reset_rules()
        assert np.array_equal(multi.split_mask_batch(cps, row, pool), want_default)


def test_calls_from_a_thread_that_did_not_init(gpu, oracle):
    """HIP's current device is per thread; a worker thread that never called latok_init must still
    allocate and launch on the context's device (DeviceGuard in every entry point)."""
    from latok_amd import batch
    cps, row = pack(["worker thread: a@b.c http://x.y/z #tag camelCase"] * 2000)
    _, want = oracle.split_batch(cps, row, want_values=False)
    got = []
    t = threading.Thread(target=lambda: got.append(batch.split_mask_batch(cps, row)))
    t.start()
    t.join(120)
    assert got and np.array_equal(got[0], want)
    p = []
    t = threading.Thread(target=lambda: p.append(gpu.latok_dev_alloc(1 << 20)))
    t.start()
    t.join(60)
    assert p and p[0]
    assert gpu.latok_dev_free(p[0]) == 0


def test_context_lifecycle_errors(gpu):
    from latok_amd import _lib
    h = C.c_void_p()
    assert gpu.latok_ctx_create(10_000, C.byref(h)) == _lib.ERR_INVALID and not h
    assert gpu.latok_ctx_create(0, None) == _lib.ERR_INVALID
    assert gpu.latok_ctx_create(0, C.byref(h)) == 0 and h
    assert gpu.latok_ctx_set_current(h) == 0 and gpu.latok_ctx_get_current() == h.value
    assert gpu.latok_ctx_destroy(h) == 0                  # destroying the current context falls back to the default one
    assert gpu.latok_ctx_get_current() is None
    assert gpu.latok_ctx_destroy(None) == 0


def test_rebinding_the_module_level_tables_changes_tokenize(gpu, oracle):
    """the reference's extension point is rebinding default_tokenizer.C_SPLIT / C_MASK / C_SYM
    (default_tokenizer.py:108-110,123-129); tokenize() / featurize() must follow it like gen_split_mask does."""
    from conftest import RULE_SETS
    from latok_amd import batch
    from latok_amd.core import default_tokenizer as dt
    text = "see http://a.b/c, mail me@x.org! #ok camelCase 1 2"
    base = list(dt.tokenize(text))
    assert base == oracle.tokenize(text)
    saved = (dt.C_SPLIT, dt.C_MASK, dt.C_SYM)
    try:
        dt.C_SPLIT, dt.C_MASK, dt.C_SYM = RULE_SETS["no_mask"]
        want = np.nonzero(oracle.split_values_rules(text, *RULE_SETS["no_mask"]))[0]
        assert np.array_equal(dt._boundaries(text), want)
        assert np.array_equal(np.nonzero(dt.gen_split_mask(dt._gen_parse_matrix(text)))[0], want)
        assert list(dt.tokenize(text)) != base and batch.rules_active()
        assert [t.text for t in dt.featurize(text)] == list(dt.tokenize(text))
    finally:
        dt.C_SPLIT, dt.C_MASK, dt.C_SYM = saved
    assert list(dt.tokenize(text)) == base and not batch.rules_active()


def test_contexts_come_and_go(gpu, oracle):
    """contexts (and pools) are created and destroyed many times: every one gets working tables / workspaces of its own,
    nothing is left behind that breaks the next one or the default context"""
    from latok_amd import _lib, batch, multi
    texts = ["a http://b.c/d e@f.gh #i camelCase 1 2", "", "日本語 テキスト"] * 50
    cps, row = pack(texts)
    _, want = oracle.split_batch(cps, row, want_values=False)
    free0 = None
    for i in range(40):
        with _lib.Context(0) as ctx:
            assert np.array_equal(batch.split_mask_batch(cps, row), want)
            c, o = batch.split_offsets_csr(cps, row, dtype=np.int32)
            assert int(c.sum()) == len(o)
        ctx.destroy()
        if i % 10 == 9:
            with multi.DevicePool([0, 0]) as pool:
                assert np.array_equal(multi.split_mask_batch(cps, row, pool), want)
    assert np.array_equal(batch.split_mask_batch(cps, row), want)


def test_one_context_shared_by_threads_is_serialised(gpu, oracle):
    """two threads on the SAME (default) context: the calls are serialised by its lock, every result is exact"""
    from latok_amd import batch
    rng = random.Random(12)
    work = []
    for _ in range(2):
        cps, row = pack(random_strings(rng, 2000, 0, 150, ALPHABETS["mixed"]) + random_strings(rng, 2, 9000, 30000, ALPHABETS["words"]))
        work.append((cps, row, _oracle_offsets(oracle, cps, row)))
    errors = []

    def run(k):
        try:
            cps, row, (wc, wo, wb) = work[k]
            for _ in range(25):
                assert np.array_equal(batch.split_mask_batch(cps, row), wb)
                c, o = batch.split_offsets_csr(cps, row, dtype=np.int32)
                assert np.array_equal(c, wc) and np.array_equal(o, wo)
        except BaseException as exc:   # noqa: BLE001
            errors.append(exc)

    ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not errors, errors[0]


def test_default_context_can_be_shut_down_and_come_back(gpu, oracle):
    from latok_amd import _lib, batch
    cps, row = pack(["shut down, then back: a@b.c http://x.y/z"] * 300)
    _, want = oracle.split_batch(cps, row, want_values=False)
    assert np.array_equal(batch.split_mask_batch(cps, row), want)
    _lib.shutdown()
    out = np.zeros(1, np.uint64)
    assert gpu.latok_split_mask_batch(cps.ctypes.data, row.ctypes.data, 1, -1, out.ctypes.data, 0, None) == _lib.ERR_NOT_INIT
    assert np.array_equal(batch.split_mask_batch(cps, row), want)          # ensure_init brings it back
    c, o = batch.split_offsets_csr(cps, row, dtype=np.int32)
    assert int(c.sum()) == len(o) > 0


def test_small_calls_from_three_threads_poll_their_own_completion_words(gpu, oracle):
    """One string / a few tiles per call from three threads at once (two contexts of their own + the default one): every
    call polls the completion word of ITS context while the kernels of the others run on the same GPU; tokenize(text),
    featurize(text) and small batches of every path stay exact."""
    from latok_amd import _lib, batch
    from latok_amd.core import default_tokenizer as dt
    rng = random.Random(31337)
    texts = random_strings(rng, 150, 1, 300, ALPHABETS["mixed"]) + random_strings(rng, 6, 5000, 30000, ALPHABETS["words"])
    want_tok = [oracle.tokenize(t) for t in texts]
    want_off = [oracle.split_offsets(t) for t in texts]
    small = [random_strings(rng, rng.randint(2, 60), 0, 400, ALPHABETS["mixed"]) for _ in range(12)]
    small_want = []
    for b in small:
        cps, row = pack(b)
        small_want.append(_oracle_offsets(oracle, cps, row))
    errors = []

    def run(own_context):
        try:
            ctx = _lib.Context(0) if own_context else None
            if ctx:
                ctx.make_current()
            for rep in range(3):
                for t, wt, wo in zip(texts, want_tok, want_off):
                    assert list(dt.tokenize(t)) == wt
                    assert np.array_equal(batch.split_offsets_one(t), wo)
                    if len(t) <= 300 and rep == 0:
                        assert [x.text for x in dt.featurize(t)] == wt
                for b, (wc, wo, wb) in zip(small, small_want):
                    cps, row = pack(b)
                    c, o = batch.split_offsets_csr(cps, row, dtype=np.int32)
                    assert np.array_equal(c, wc) and np.array_equal(o, wo)
                    assert np.array_equal(batch.split_mask_batch(cps, row), wb)
            if ctx:
                _lib.load().latok_ctx_set_current(None)
                ctx.destroy()
        except BaseException as exc:   # noqa: BLE001 - reported to the main thread
            errors.append(exc)

    ths = [threading.Thread(target=run, args=(k < 2,)) for k in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(600)
    assert not errors, errors[0]


def test_resident_shards_on_three_contexts_equal_the_oracle(gpu, oracle):
    """DevicePool.put_csr / generate: the shards are uploaded (or generated) ONCE into each context's HBM and every later
    call uses the LATOK_DEVICE_PTRS forms; results against the oracle for all four input kinds, repeated calls on the same
    resident batch, device-only results read back by hand, and the caller's HIP device untouched."""
    from latok_amd import _lib, batch, multi
    rng = random.Random(123)
    texts = random_strings(rng, 4000, 0, 200, ALPHABETS["mixed"]) + random_strings(rng, 3, 20000, 50000, ALPHABETS["rare_space_at"]) + ["", "x"]
    rng.shuffle(texts)
    cps, row = pack(texts)
    wc, wo, wb = _oracle_offsets(oracle, cps, row)
    with multi.DevicePool([0, 0, 0]) as pool:
        with pool.put_csr(cps, row) as rb:
            assert len([sh for sh in rb.shards if sh is not None]) == 3
            for _ in range(3):                                    # the resident batch is reusable
                assert np.array_equal(pool.split_mask(rb), wb)
            for dtype in (np.int32, np.int64):
                counts, offs = pool.split_offsets(rb, dtype=dtype)
                assert np.array_equal(counts, wc) and np.array_equal(offs, wo)
            counts, spans = pool.token_spans(rb)
            k = 0
            for t, c in zip(texts, counts.tolist()):
                assert [t[a:b] for a, b in spans[k:k + c].tolist()] == (oracle.tokenize(t) if t else [])
                k += c
            # device-only: read shard 1's records back on its own worker
            shards = pool.split_offsets(rb, dtype=np.int32, to_host=False)
            sh = shards[1]

            def fetch():
                out = np.empty(sh.n_items[1], np.int32)
                _lib.check(pool.lib.latok_memcpy_d2h(out.ctypes.data, sh.bufs[("d_items", 1)], out.nbytes))
                return out
            jobs = [None] * len(pool)
            jobs[sh.worker] = fetch
            got = pool.run(jobs)[sh.worker]
            lo = int(wc[:sh.s0].sum())
            assert np.array_equal(got, wo[lo:lo + sh.n_items[1]])
        # the narrow kinds and UTF-8 bytes
        lat = [t for t in random_strings(rng, 3000, 0, 150, ALPHABETS["latin1"])]
        units = np.frombuffer("".join(lat).encode("latin-1"), np.uint8)
        lrow = np.zeros(len(lat) + 1, np.int64)
        np.cumsum([len(t) for t in lat], out=lrow[1:])
        lc, lo_, lb = _oracle_offsets(oracle, *pack(lat))
        with pool.put_csr(units, lrow, kind="latin1") as rb:
            assert np.array_equal(pool.split_mask(rb), lb)
            c, o = pool.split_offsets(rb, dtype=np.int64)
            assert np.array_equal(c, lc) and np.array_equal(o, lo_)
        blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
        u8, boff = batch.pack_utf8(blobs)
        with pool.put_csr(u8, boff, kind="utf8") as rb:
            assert np.array_equal(pool.split_mask(rb), batch.split_mask_utf8_bytes_csr(u8, boff))
            c, o = pool.split_offsets(rb, dtype=np.int64)
            c1, o1 = batch.split_offsets_utf8_bytes_csr(u8, boff)
            assert np.array_equal(c, c1) and np.array_equal(o, o1)
        # generated on the devices: equals the host generator's corpus
        n = 30000
        grow = np.zeros(n + 1, np.int64)
        gpu.latok_corpus_offsets(0x1A70C0DF, 77, n, 10, 300, grow.ctypes.data)
        gcps = np.zeros(int(grow[-1]), np.uint32)
        gpu.latok_corpus_fill_host(0x1A70C0DF, _lib.CORPUS_UNICODE, 77, n, grow.ctypes.data, gcps.ctypes.data)
        with pool.generate(0x1A70C0DF, _lib.CORPUS_UNICODE, n, 10, 300, sid0=77) as rb:
            assert rb.total == int(grow[-1])
            assert np.array_equal(pool.split_mask(rb), oracle.split_batch(gcps, grow, want_values=False)[1])


def test_pool_over_every_device_of_the_box(gpu, oracle):
    """contexts on devices >= 1 (pinned buffers, events and streams created under a non-zero current device, a
    worker thread per device).  Runs wherever the box has more than one GPU; the caller's current device stays what it was."""
    from latok_amd import multi
    n_dev = gpu.latok_device_count()
    if n_dev < 2:
        pytest.skip("one GPU on this box")
    rng = random.Random(5)
    texts = random_strings(rng, 6000, 0, 300, ALPHABETS["mixed"])
    cps, row = pack(texts)
    wc, wo, wb = _oracle_offsets(oracle, cps, row)
    with multi.DevicePool(list(range(n_dev))) as pool:
        c, o = multi.split_offsets_csr(cps, row, pool)
        assert np.array_equal(c, wc) and np.array_equal(o, wo)
        with pool.put_csr(cps, row) as rb:
            assert np.array_equal(pool.split_mask(rb), wb)
            c, o = pool.split_offsets(rb, dtype=np.int64)
            assert np.array_equal(c, wc) and np.array_equal(o, wo)
    assert gpu.latok_ctx_device(None) == 0      # the default context is still on device 0
