"""The batch flow of the C ABI (include/latok_hip.h "batch flow": latok_flow_split_mask / latok_flow_wait): many
device-resident batches through one context with two in flight.  Every result must be what latok_split_mask_batch and the
oracle give for the same batch -- whatever the sizes of the neighbours in the flow, also when a workspace slot has to grow,
when two batches in flight name the same output buffer, under run-time rule tables, and from two contexts at once."""
import random
import threading

import numpy as np
import pytest

from conftest import ALPHABETS, RULE_SETS, oracle_rule_bits, pack, random_strings

pytestmark = pytest.mark.gpu


class _Resident:
    """one batch in device memory + a poisoned output bitmask"""

    def __init__(self, lib, cps, row):
        from latok_amd import _lib
        self.lib, self.n_str, self.total = lib, len(row) - 1, int(row[-1])
        self.words = (self.total + 63) // 64
        self.d_cps = lib.latok_dev_alloc(max(cps.nbytes, 16))
        self.d_row = lib.latok_dev_alloc(row.nbytes)
        self.d_mask = lib.latok_dev_alloc(max(self.words * 8, 16))
        assert self.d_cps and self.d_row and self.d_mask
        _lib.check(lib.latok_memcpy_h2d(self.d_cps, cps.ctypes.data, cps.nbytes))
        _lib.check(lib.latok_memcpy_h2d(self.d_row, row.ctypes.data, row.nbytes))
        _lib.check(lib.latok_memset_dev(self.d_mask, 0xA5, max(self.words * 8, 16)))

    def mask(self, d_mask=None):
        from latok_amd import _lib
        out = np.empty(self.words, np.uint64)
        _lib.check(self.lib.latok_memcpy_d2h(out.ctypes.data, d_mask or self.d_mask, out.nbytes))
        return out

    def free(self):
        for p in (self.d_cps, self.d_row, self.d_mask):
            self.lib.latok_dev_free(p)


def _batches(rng):
    """sizes from one char to several segments, incl. documents whose pending-start queue crosses tiles and segments"""
    out = [["x"], ["", "a b", ""], random_strings(rng, 50, 0, 40, ALPHABETS["mixed"]),
           random_strings(rng, 3000, 0, 200, ALPHABETS["mixed"]),
           random_strings(rng, 4, 30000, 90000, ALPHABETS["rare_space_at"]) + random_strings(rng, 200, 0, 100, ALPHABETS["starts"]),
           random_strings(rng, 2, 200000, 400000, ALPHABETS["nospace_at"]) + ["@a b"],
           random_strings(rng, 20000, 0, 300, ALPHABETS["words"]),
           random_strings(rng, 700, 0, 64, ALPHABETS["mixed"])]
    rng.shuffle(out)
    return out


def test_flow_results_equal_the_serial_call_and_the_oracle(gpu, oracle):
    from latok_amd import _lib, batch
    rng = random.Random(2024)
    work = []
    for texts in _batches(rng) + _batches(rng):
        cps, row = pack(texts)
        work.append((_Resident(gpu, cps, row), oracle.split_batch(cps, row, want_values=False)[1], cps, row))
    try:
        for rb, _, _, _ in work:                       # sixteen batches of very different sizes, back to back
            batch.flow_split_mask(rb.d_cps, rb.d_row, rb.n_str, rb.total, rb.d_mask)
        batch.flow_wait()
        for rb, want, cps, row in work:
            got = rb.mask()
            assert np.array_equal(got, want)
            assert np.array_equal(batch.split_mask_batch(cps, row), want)
        # again, now with total_chars left to the library (-1) and latok_sync as the barrier
        for rb, _, _, _ in work:
            _lib.check(gpu.latok_memset_dev(rb.d_mask, 0x5A, max(rb.words * 8, 16)))
        for rb, _, _, _ in work:
            batch.flow_split_mask(rb.d_cps, rb.d_row, rb.n_str, -1, rb.d_mask)
        _lib.check(gpu.latok_sync())
        for rb, want, _, _ in work:
            assert np.array_equal(rb.mask(), want)
    finally:
        for rb, _, _, _ in work:
            rb.free()


def test_two_batches_in_flight_may_name_the_same_output_buffer(gpu, oracle):
    """the second one is ordered behind the first (same slot stream): the buffer ends up holding the LAST batch's mask"""
    from latok_amd import batch
    rng = random.Random(7)
    a = pack(random_strings(rng, 6000, 0, 300, ALPHABETS["mixed"]))
    b = pack(random_strings(rng, 5, 40000, 60000, ALPHABETS["rare_space_at"]) + random_strings(rng, 4000, 0, 200, ALPHABETS["mixed"]))
    c = pack(random_strings(rng, 3000, 0, 100, ALPHABETS["starts"]))
    ra, rb_, rc = _Resident(gpu, *a), _Resident(gpu, *b), _Resident(gpu, *c)
    try:
        words = max(ra.words, rb_.words, rc.words)
        shared = gpu.latok_dev_alloc(words * 8)
        for _ in range(3):
            batch.flow_split_mask(ra.d_cps, ra.d_row, ra.n_str, ra.total, shared)
            batch.flow_split_mask(rc.d_cps, rc.d_row, rc.n_str, rc.total, rc.d_mask)     # an unrelated batch in between
            batch.flow_split_mask(rb_.d_cps, rb_.d_row, rb_.n_str, rb_.total, shared)
            batch.flow_wait()
            assert np.array_equal(rb_.mask(shared), oracle.split_batch(*b, want_values=False)[1])
            assert np.array_equal(rc.mask(), oracle.split_batch(*c, want_values=False)[1])
        gpu.latok_dev_free(shared)
    finally:
        for r in (ra, rb_, rc):
            r.free()


def _long_batch(rng, oracle, chars):
    """a batch of >= `chars` chars made of whole repeats of one oracle-checked batch whose length is a multiple of 64 (so
    the mask of the long batch is the repeated mask: strings are independent, default_tokenizer.py:137)"""
    texts = random_strings(rng, 12000, 0, 300, ALPHABETS["mixed"])
    texts.append("p" * ((-sum(len(t) for t in texts)) % 64))
    cps, row = pack(texts)
    want = oracle.split_batch(cps, row, want_values=False)[1]
    reps = -(-chars // len(cps))
    big_row = np.concatenate([[0]] + [row[1:] + i * len(cps) for i in range(reps)]).astype(np.int64)
    return np.tile(cps, reps), big_row, np.tile(want, reps)


def test_output_buffer_reused_behind_other_batches_is_ordered_behind_its_first_writer(gpu, oracle):
    """A (long) -> X, B -> Y, C -> Z, D -> X: between A and D two other batches have passed through BOTH slots, and D must still
    be ordered behind A (flow_hazards.h keeps every in-flight range of a slot, not its last output).  Second leg: D writes only
    the TAIL of X's range (partial overlap; the words A writes last), where an unordered D would be overwritten by A."""
    from latok_amd import _lib, batch
    rng = random.Random(404)
    a_cps, a_row, a_want = _long_batch(rng, oracle, 400_000_000)       # 1.6 GB of code points: ~0.3 ms of tile kernel
    small = [pack(random_strings(rng, 300, 0, 120, ALPHABETS["mixed"])) for _ in range(3)]
    ra = _Resident(gpu, a_cps, a_row)
    rs = [_Resident(gpu, *p) for p in small]
    wants = [oracle.split_batch(*p, want_values=False)[1] for p in small]
    rb_, rc, rd = rs
    try:
        assert a_cps.nbytes >= 200 << 20
        for leg in range(4):
            tail = leg & 1
            # D's place in X: the start of the buffer, or (16-byte aligned) flush with the end of A's words
            d_word0 = ((ra.words - rd.words) & ~1) if tail else 0
            d_ptr = ra.d_mask + 8 * d_word0
            _lib.check(gpu.latok_memset_dev(ra.d_mask, 0xA5, ra.words * 8))
            batch.flow_split_mask(ra.d_cps, ra.d_row, ra.n_str, ra.total, ra.d_mask)     # A -> X
            batch.flow_split_mask(rb_.d_cps, rb_.d_row, rb_.n_str, rb_.total, rb_.d_mask)   # B -> Y
            batch.flow_split_mask(rc.d_cps, rc.d_row, rc.n_str, rc.total, rc.d_mask)     # C -> Z
            batch.flow_split_mask(rd.d_cps, rd.d_row, rd.n_str, rd.total, d_ptr)         # D -> (part of) X
            batch.flow_wait()
            got = ra.mask()
            assert np.array_equal(got[d_word0:d_word0 + rd.words], wants[2]), "D's mask was overwritten"
            assert np.array_equal(got[:d_word0], a_want[:d_word0])
            assert np.array_equal(got[d_word0 + rd.words:], a_want[d_word0 + rd.words:])
            assert np.array_equal(rb_.mask(), wants[0]) and np.array_equal(rc.mask(), wants[1])
    finally:
        for r in [ra] + rs:
            r.free()


def test_compaction_outputs_reused_across_the_flow_are_ordered(gpu, oracle):
    """the same for the compaction calls: records, COUNTS and RESULT words all count as outputs (a batch that reuses only the counts
    or result buffer of a batch in flight is ordered behind it too)"""
    from latok_amd import _lib, batch
    rng = random.Random(405)
    a_cps, a_row, _ = _long_batch(rng, oracle, 120_000_000)
    d_texts = random_strings(rng, 400, 0, 150, ALPHABETS["mixed"])
    d_cps, d_row = pack(d_texts)
    d_vals = oracle.split_batch(d_cps, d_row)[0]
    d_offs = np.concatenate([np.nonzero(d_vals[d_row[i]:d_row[i + 1]])[0] for i in range(len(d_texts))]).astype(np.int32)
    d_counts = np.array([np.count_nonzero(d_vals[d_row[i]:d_row[i + 1]]) for i in range(len(d_texts))], np.int32)
    ra, rd = _Resident(gpu, a_cps, a_row), _Resident(gpu, d_cps, d_row)
    cap_a = ra.total
    bufs = {k: gpu.latok_dev_alloc(n) for k, n in (("items", cap_a * 4 + 64), ("counts", ra.n_str * 4 + 64), ("res", 64), ("items2", rd.total * 4 + 64),
                                                    ("counts2", rd.n_str * 4 + 64), ("res2", 64), ("items3", rd.total * 4 + 64))}
    assert all(bufs.values())
    try:
        for leg in range(3):
            # leg 0: D reuses A's records buffer; leg 1: only A's counts buffer; leg 2: only A's result words
            batch.flow_split_offsets(ra.d_cps, 4, ra.d_row, ra.n_str, ra.total, bufs["counts"], bufs["items"], cap_a, bufs["res"], dtype=np.int32)
            for _ in range(2):   # two unrelated batches: one through each slot
                batch.flow_split_offsets(rd.d_cps, 4, rd.d_row, rd.n_str, rd.total, bufs["counts2"], bufs["items2"], rd.total, bufs["res2"], dtype=np.int32)
            items = bufs["items"] if leg == 0 else bufs["items3"]
            counts = bufs["counts"] if leg == 1 else bufs["counts2"]
            res = bufs["res"] if leg == 2 else bufs["res2"]
            batch.flow_split_offsets(rd.d_cps, 4, rd.d_row, rd.n_str, rd.total, counts, items, rd.total, res, dtype=np.int32)
            batch.flow_wait()
            got_o, got_c, got_r = np.empty(d_offs.size, np.int32), np.empty(rd.n_str, np.int32), np.empty(2, np.int64)
            _lib.check(gpu.latok_memcpy_d2h(got_o.ctypes.data, items, got_o.nbytes))
            _lib.check(gpu.latok_memcpy_d2h(got_c.ctypes.data, counts, got_c.nbytes))
            _lib.check(gpu.latok_memcpy_d2h(got_r.ctypes.data, res, 16))
            assert got_r.tolist() == [d_offs.size, 0], (leg, got_r)
            assert np.array_equal(got_c, d_counts) and np.array_equal(got_o, d_offs), leg
    finally:
        for p_ in bufs.values():
            gpu.latok_dev_free(p_)
        ra.free()
        rd.free()


@pytest.mark.parametrize("name", ["sym_everywhere", "all_starts"])
def test_flow_under_run_time_rule_tables(gpu, oracle, name):
    from latok_amd import batch
    rng = random.Random(31)
    tables = RULE_SETS[name]
    sets = [random_strings(rng, 1500, 0, 200, ALPHABETS["mixed"]) + random_strings(rng, 2, 9000, 20000, ALPHABETS["rare_space_at"])
            for _ in range(3)]
    res = [_Resident(gpu, *pack(t)) for t in sets]
    batch.set_rules(*tables)
    try:
        for r in res:
            batch.flow_split_mask(r.d_cps, r.d_row, r.n_str, r.total, r.d_mask)
        batch.flow_wait()
        for r, t in zip(res, sets):
            assert np.array_equal(r.mask(), oracle_rule_bits(oracle, t, tables))
    finally:
        batch.reset_rules()
        for r in res:
            r.free()


def test_flows_of_two_contexts_run_side_by_side(gpu, oracle):
    from latok_amd import _lib, batch
    rng = random.Random(5)
    work = [[pack(random_strings(rng, 4000 + 500 * k, 0, 250, ALPHABETS["mixed"])) for k in range(4)] for _ in range(2)]
    want = [[oracle.split_batch(c, r, want_values=False)[1] for c, r in w] for w in work]
    errors = []

    def run(k):
        try:
            with _lib.Context(0) as ctx:
                res = [_Resident(gpu, c, r) for c, r in work[k]]
                for _ in range(5):
                    for r in res:
                        batch.flow_split_mask(r.d_cps, r.d_row, r.n_str, r.total, r.d_mask)
                    batch.flow_wait()
                    for r, w in zip(res, want[k]):
                        assert np.array_equal(r.mask(), w)
                for r in res:
                    r.free()
            ctx.destroy()
        except BaseException as exc:   # noqa: BLE001 - reported to the main thread
            errors.append(exc)

    ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not errors, errors[0]


def test_flow_of_mixed_input_forms(gpu, oracle):
    """UTF-32, Latin-1 / UCS-2 units and UTF-8 in byte space follow each other in one flow; each equals its serial call"""
    from latok_amd import _lib, batch
    rng = random.Random(77)
    lat = random_strings(rng, 5000, 0, 200, ALPHABETS["latin1"])
    bmp = random_strings(rng, 4000, 0, 200, ALPHABETS["bmp"]) + random_strings(rng, 2, 30000, 50000, ALPHABETS["rare_space_at"])
    mixed = random_strings(rng, 5000, 0, 150, ALPHABETS["mixed"])

    def dev(a):
        p = gpu.latok_dev_alloc(max(a.nbytes, 16) + 64)
        _lib.check(gpu.latok_memcpy_h2d(p, a.ctypes.data, a.nbytes))
        return p
    jobs = []   # (submit, want, words, d_mask)
    u1, r1 = batch.pack_kind(lat)
    assert u1.dtype == np.uint8
    u2, r2 = batch.pack_kind(bmp)
    assert u2.dtype == np.uint16
    for units, row, kind in ((u1, r1, 1), (u2, r2, 2)):
        want = batch.split_mask_kind_csr(units, row)
        d_u, d_r, d_m = dev(units), dev(row), gpu.latok_dev_alloc(want.nbytes + 16)
        jobs.append((lambda d_u=d_u, d_r=d_r, d_m=d_m, k=kind, row=row: batch.flow_split_mask_kind(d_u, k, d_r, len(row) - 1, int(row[-1]), d_m),
                     want, d_m))
    u8, boff = batch.pack_utf8([t.encode("utf-8", "surrogatepass") for t in mixed])
    want = batch.split_mask_utf8_bytes_csr(u8, boff)
    d_u, d_r, d_m = dev(u8), dev(boff), gpu.latok_dev_alloc(want.nbytes + 16)
    jobs.append((lambda d_u=d_u, d_r=d_r, d_m=d_m: batch.flow_split_mask_utf8_bytes(d_u, d_r, len(boff) - 1, int(boff[-1]), d_m), want, d_m))
    cps, row = pack(mixed)
    want32 = oracle.split_batch(cps, row, want_values=False)[1]
    d_c, d_r4, d_m4 = dev(cps), dev(row), gpu.latok_dev_alloc(want32.nbytes + 16)
    n32, t32 = len(row) - 1, int(row[-1])
    jobs.append((lambda: batch.flow_split_mask_kind(d_c, 4, d_r4, n32, t32, d_m4), want32, d_m4))
    for _ in range(3):
        for _, w_, m_ in jobs:
            _lib.check(gpu.latok_memset_dev(m_, 0x3C, w_.nbytes))
        for submit, _, _ in jobs + jobs[::-1]:
            submit()
        batch.flow_wait()
        for _, w_, m_ in jobs:
            got = np.empty_like(w_)
            _lib.check(gpu.latok_memcpy_d2h(got.ctypes.data, m_, got.nbytes))
            assert np.array_equal(got, w_)
    with pytest.raises(ValueError):
        batch.flow_split_mask_kind(d_c, 3, d_r4, 1, 1, d_m4)      # no such PEP 393 kind


class _Out:
    """device buffers of one compaction result: counts, records, the two result words"""

    def __init__(self, lib, n_str, cap, width, dt):
        self.lib, self.n_str, self.cap, self.width, self.dt = lib, n_str, cap, width, np.dtype(dt)
        self.d_counts = lib.latok_dev_alloc(max(n_str, 1) * self.dt.itemsize + 16)
        self.d_items = lib.latok_dev_alloc(max(cap, 1) * width * self.dt.itemsize + 16)
        self.d_res = lib.latok_dev_alloc(16)
        assert self.d_counts and self.d_items and self.d_res

    def fetch(self):
        from latok_amd import _lib
        res = np.empty(2, np.int64)
        _lib.check(self.lib.latok_memcpy_d2h(res.ctypes.data, self.d_res, 16))
        counts = np.empty(self.n_str, self.dt)
        if self.n_str:
            _lib.check(self.lib.latok_memcpy_d2h(counts.ctypes.data, self.d_counts, counts.nbytes))
        n = int(res[0])
        items = np.empty((min(n, self.cap), self.width) if self.width > 1 else min(n, self.cap), self.dt)
        if items.size and n <= self.cap:
            _lib.check(self.lib.latok_memcpy_d2h(items.ctypes.data, self.d_items, items.nbytes))
        return res, counts, items

    def free(self):
        for p in (self.d_counts, self.d_items, self.d_res):
            self.lib.latok_dev_free(p)


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_flow_offsets_and_spans_equal_the_blocking_calls(gpu, oracle, dtype):
    """latok_flow_split_offsets / latok_flow_token_spans on batches of very different sizes back to back, offsets and spans
    interleaved in one flow: counts, records and totals are those of the blocking calls (which the other GPU tests pin to
    the oracle), and the offsets are checked against the oracle directly"""
    from latok_amd import batch
    rng = random.Random(808)
    work = []
    for texts in _batches(rng):
        cps, row = pack(texts)
        rb = _Resident(gpu, cps, row)
        wc, wo = batch.split_offsets_csr(cps, row, dtype=dtype)
        sc, sp = batch.token_spans_csr(cps, row, dtype=dtype)
        vals = oracle.split_batch(cps, row)[0]
        assert np.array_equal(np.concatenate([np.nonzero(vals[row[i]:row[i + 1]])[0] for i in range(len(row) - 1)] or [np.zeros(0, np.int64)]), wo)
        work.append((rb, (wc, wo), (sc, sp), _Out(gpu, rb.n_str, len(wo), 1, dtype), _Out(gpu, rb.n_str, len(sp), 2, dtype)))
    try:
        for _ in range(2):
            for rb, _, _, oo, os_ in work:
                batch.flow_split_offsets(rb.d_cps, 4, rb.d_row, rb.n_str, rb.total, oo.d_counts, oo.d_items, oo.cap, oo.d_res, dtype=dtype)
                batch.flow_token_spans(rb.d_cps, 4, rb.d_row, rb.n_str, -1, os_.d_counts, os_.d_items, os_.cap, os_.d_res, dtype=dtype)
            batch.flow_wait()
            for rb, (wc, wo), (sc, sp), oo, os_ in work:
                res, counts, items = oo.fetch()
                assert res[0] == len(wo) and res[1] == 0 and np.array_equal(counts, wc) and np.array_equal(items, wo)
                res, counts, items = os_.fetch()
                assert res[0] == len(sp) and res[1] == 0 and np.array_equal(counts, sc) and np.array_equal(items, sp)
    finally:
        for rb, _, _, oo, os_ in work:
            rb.free()
            oo.free()
            os_.free()


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_flow_featurize_equals_the_blocking_call(gpu, oracle, dtype):
    from latok_amd import _lib, batch
    rng = random.Random(4242)
    work = []
    for texts in _batches(rng)[:6]:
        cps, row = pack(texts)
        rb = _Resident(gpu, cps, row)
        c, sp4, ft = batch.token_features_csr(cps, row, dtype=dtype)
        o = _Out(gpu, rb.n_str, len(sp4), 4, dtype)
        d_ft = gpu.latok_dev_alloc(max(len(sp4), 1) * 25 + 16)
        work.append((rb, c, sp4, ft, o, d_ft))
    # a Latin-1 batch (widened on the device, in the slot's own buffer)
    lat = random_strings(rng, 3000, 0, 200, ALPHABETS["latin1"])
    units, lrow = batch.pack_kind(lat)
    lc, lsp4, lft = batch.token_features_kind_csr(units, lrow, dtype=dtype)
    d_lu, d_lr = gpu.latok_dev_alloc(units.nbytes + 64), gpu.latok_dev_alloc(lrow.nbytes)
    _lib.check(gpu.latok_memcpy_h2d(d_lu, units.ctypes.data, units.nbytes))
    _lib.check(gpu.latok_memcpy_h2d(d_lr, lrow.ctypes.data, lrow.nbytes))
    lo = _Out(gpu, len(lrow) - 1, len(lsp4), 4, dtype)
    d_lft = gpu.latok_dev_alloc(max(len(lsp4), 1) * 25 + 16)
    try:
        for _ in range(2):
            for rb, c, sp4, ft, o, d_ft in work:
                batch.flow_token_features(rb.d_cps, 4, rb.d_row, rb.n_str, rb.total, o.d_counts, o.d_items, d_ft, o.cap, o.d_res, dtype=dtype)
            batch.flow_token_features(d_lu, 1, d_lr, len(lrow) - 1, int(lrow[-1]), lo.d_counts, lo.d_items, d_lft, lo.cap, lo.d_res, dtype=dtype)
            batch.flow_wait()
            for rb, c, sp4, ft, o, d_ft in work + [(None, lc, lsp4, lft, lo, d_lft)]:
                res, counts, items = o.fetch()
                assert res[0] == len(sp4) and res[1] == 0 and np.array_equal(counts, c) and np.array_equal(items, sp4)
                got = np.empty((len(sp4), 25), np.int8)
                if got.size:
                    _lib.check(gpu.latok_memcpy_d2h(got.ctypes.data, d_ft, got.nbytes))
                assert np.array_equal(got, ft)
    finally:
        for rb, _, _, _, o, d_ft in work:
            rb.free()
            o.free()
            gpu.latok_dev_free(d_ft)
        lo.free()
        for p in (d_lu, d_lr, d_lft):
            gpu.latok_dev_free(p)


def test_flow_compaction_capacity_protocol_and_other_input_forms(gpu, oracle):
    from latok_amd import _lib, batch
    rng = random.Random(99)
    texts = random_strings(rng, 6000, 0, 200, ALPHABETS["latin1"])
    units, row = batch.pack_kind(texts)
    assert units.dtype == np.uint8
    wc, wo = batch.split_offsets_kind_csr(units, row, dtype=np.int32)
    d_u = gpu.latok_dev_alloc(units.nbytes + 64)
    d_r = gpu.latok_dev_alloc(row.nbytes)
    _lib.check(gpu.latok_memcpy_h2d(d_u, units.ctypes.data, units.nbytes))
    _lib.check(gpu.latok_memcpy_h2d(d_r, row.ctypes.data, row.nbytes))
    n, total = len(row) - 1, int(row[-1])
    small, full = _Out(gpu, n, len(wo) // 2, 1, np.int32), _Out(gpu, n, len(wo), 1, np.int32)
    # UTF-8 in byte space, spans
    blobs = [t.encode("utf-8") for t in random_strings(rng, 4000, 0, 150, ALPHABETS["mixed"])]
    u8, boff = batch.pack_utf8(blobs)
    bc, bs = batch.token_spans_utf8_bytes_csr(u8, boff)
    d_u8 = gpu.latok_dev_alloc(u8.nbytes + 64)
    d_bo = gpu.latok_dev_alloc(boff.nbytes)
    _lib.check(gpu.latok_memcpy_h2d(d_u8, u8.ctypes.data, u8.nbytes))
    _lib.check(gpu.latok_memcpy_h2d(d_bo, boff.ctypes.data, boff.nbytes))
    bo = _Out(gpu, len(boff) - 1, len(bs), 2, np.int64)
    empty = _Out(gpu, 3, 4, 1, np.int64)
    d_er = gpu.latok_dev_alloc(32)
    _lib.check(gpu.latok_memset_dev(d_er, 0, 32))                         # three empty strings: row offsets 0 0 0 0
    try:
        _lib.check(gpu.latok_memset_dev(small.d_items, 0x7F, max(small.cap, 1) * 4))
        batch.flow_split_offsets(d_u, 1, d_r, n, total, small.d_counts, small.d_items, small.cap, small.d_res, dtype=np.int32)
        batch.flow_token_spans(d_u8, 0, d_bo, len(boff) - 1, int(boff[-1]), bo.d_counts, bo.d_items, bo.cap, bo.d_res)
        batch.flow_split_offsets(d_u, 1, d_r, n, total, full.d_counts, full.d_items, full.cap, full.d_res, dtype=np.int32)
        batch.flow_split_offsets(d_u, 1, d_er, 3, 0, empty.d_counts, empty.d_items, empty.cap, empty.d_res)
        batch.flow_wait()
        res, counts, _ = small.fetch()
        assert res[0] == len(wo) > small.cap and res[1] == 0 and np.array_equal(counts, wc)      # too small: total reported, counts valid
        untouched = np.empty(small.cap, np.int32)
        _lib.check(gpu.latok_memcpy_d2h(untouched.ctypes.data, small.d_items, untouched.nbytes))
        assert (untouched == 0x7F7F7F7F).all()                                                    # ... and no record written
        res, counts, items = full.fetch()
        assert res[0] == len(wo) and np.array_equal(counts, wc) and np.array_equal(items, wo)
        res, counts, items = bo.fetch()
        assert res[0] == len(bs) and res[1] == 0 and np.array_equal(counts, bc) and np.array_equal(items, bs)
        res, counts, _ = empty.fetch()
        assert res[0] == 0 and res[1] == 0 and not counts.any()
    finally:
        for o in (small, full, bo, empty):
            o.free()
        for p in (d_u, d_r, d_u8, d_bo, d_er):
            gpu.latok_dev_free(p)


def test_device_pool_pushes_several_resident_batches_through_the_flow(gpu, oracle):
    """DevicePool.split_offsets_many / token_spans_many: every worker submits its shard of every batch to its context's flow
    and waits once; equal to one blocking call per batch, for every input kind, with buffers that have to grow"""
    from latok_amd import batch, multi
    rng = random.Random(31337)
    with multi.DevicePool([0, 0, 0]) as pool:
        rbs, want_o, want_s = [], [], []
        for k in range(5):
            texts = random_strings(rng, 2000 + 700 * k, 0, 150, ALPHABETS["starts" if k == 2 else "mixed"]) + ["", "x"]
            cps, row = pack(texts)
            rbs.append(pool.put_csr(cps, row))
            want_o.append(batch.split_offsets_csr(cps, row, dtype=np.int32))
            want_s.append(batch.token_spans_csr(cps, row, dtype=np.int32))
        lat = random_strings(rng, 3000, 0, 100, ALPHABETS["latin1"])
        units, lrow = batch.pack_kind(lat)
        rbs.append(pool.put_csr(units, lrow, kind="latin1"))
        want_o.append(batch.split_offsets_kind_csr(units, lrow, dtype=np.int32))
        want_s.append(batch.token_spans_kind_csr(units, lrow, dtype=np.int32))
        u8, boff = batch.pack_utf8([t.encode("utf-8", "surrogatepass") for t in random_strings(rng, 2500, 0, 120, ALPHABETS["mixed"])])
        rbs.append(pool.put_csr(u8, boff, kind="utf8"))
        want_o.append(batch.split_offsets_utf8_bytes_csr(u8, boff, dtype=np.int32))
        want_s.append(batch.token_spans_utf8_bytes_csr(u8, boff, dtype=np.int32))
        for got, rb in zip(pool.split_mask_many(rbs), rbs):
            assert np.array_equal(got, pool.split_mask(rb))
        for _ in range(2):
            for (c, o), (wc, wo) in zip(pool.split_offsets_many(rbs), want_o):
                assert np.array_equal(c, wc) and np.array_equal(o, wo)
            for (c, sp), (wc, ws) in zip(pool.token_spans_many(rbs, dtype=np.int32), want_s):
                assert np.array_equal(c, wc) and np.array_equal(sp, ws)
        c64 = pool.split_offsets_many(rbs[:2], dtype=np.int64)
        assert c64[0][1].dtype == np.int64 and np.array_equal(c64[1][1], want_o[1][1])
        for rb in rbs:
            rb.free()


def test_blocking_calls_between_flow_submissions(gpu, oracle):
    """the blocking entry points use the context's own stream and workspace: they may be called while a flow is in flight"""
    from latok_amd import batch
    rng = random.Random(1)
    sets = [pack(random_strings(rng, 3000 + 1000 * k, 0, 200, ALPHABETS["mixed"])) for k in range(4)]
    want = [oracle.split_batch(c, r, want_values=False)[1] for c, r in sets]
    res = [_Resident(gpu, c, r) for c, r in sets]
    try:
        for _ in range(4):
            for k, r in enumerate(res):
                batch.flow_split_mask(r.d_cps, r.d_row, r.n_str, r.total, r.d_mask)
                assert np.array_equal(batch.split_mask_batch(*sets[(k + 1) % 4]), want[(k + 1) % 4])     # host-pointer call, synchronous
                c, o = batch.split_offsets_csr(*sets[k], dtype=np.int32)
                assert int(c.sum()) == len(o)
            batch.flow_wait()
            for r, w in zip(res, want):
                assert np.array_equal(r.mask(), w)
    finally:
        for r in res:
            r.free()


def test_c_example_flow_batches(gpu, oracle, tmp_path):
    """examples/flow_batches.c: a plain C caller pushes five resident UTF-8 batches through the flow (token spans in byte
    space, int32 records, result words read after one latok_flow_wait) and prints the reference's tokens."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "flow_batches")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "flow_batches.c"),
                           "-L" + os.path.join(ROOT, "latok_amd"), "-llatok_hip", "-Wl,-rpath," + os.path.join(ROOT, "latok_amd"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, timeout=120, check=True).stdout.decode("utf-8").splitlines()
    batches = [["This is a #test! Testing, Testing, 1 2 3"], ["see http://a.b/c or mail me@x.org", "camelCase 日本語 🤓"], ["", "x", "  "],
               ["foo@bar.com, .@user hi", "$#@^:a./", "camelCaseXMLParser"], ["one more batch: the flow takes any number"]]
    want = [f"{k}.{i}:" + "".join(f" [{t}]" for t in (oracle.tokenize(s) if s else [])) for k, b in enumerate(batches) for i, s in enumerate(b)]
    assert out == want


def test_flow_refuses_what_the_serial_call_refuses(gpu):
    from latok_amd import batch
    with pytest.raises(ValueError):
        batch.flow_split_mask(0x1000, 0x2000, 3, 10, None)        # NULL output
    with pytest.raises(ValueError):
        batch.flow_split_mask(0x1004, 0x2000, 3, 10, 0x3000)      # misaligned code points
    batch.flow_split_mask(None, None, 0, 0, None)                  # an empty batch is nothing to do
    batch.flow_wait()
