"""latok_amd.multi on CPU: the fan-out / concatenation logic of the in-process multi-GPU driver with FAKE contexts (no
GPU, no HIP): the per-shard work is done by the oracle, so what is tested is exactly what the driver adds -- shard cuts,
one persistent thread per context, result order, bit-exact merge of the shards' bitmasks, error propagation."""
import random
import threading

import numpy as np
import pytest

from conftest import ALPHABETS, pack, random_strings


class FakeCtx:
    """stands in for _lib.Context: records which thread made it current"""
    made = []

    def __init__(self, device):
        self.device = device
        self.thread = None
        self.destroyed = False
        FakeCtx.made.append(self)

    def make_current(self):
        self.thread = threading.get_ident()

    def destroy(self):
        self.destroyed = True


@pytest.fixture
def fake_batch(monkeypatch, oracle):
    """latok_amd.batch's entry points re-implemented on the oracle (CPU), each recording the thread it ran on"""
    from latok_amd import batch
    ran_on = []

    def offsets_csr(cps, row):
        ran_on.append(threading.get_ident())
        cps, row = np.ascontiguousarray(cps, np.uint32), np.ascontiguousarray(row, np.int64)
        vals, _ = oracle.split_batch(cps, row, want_bits=False)
        per = [np.nonzero(vals[row[s]:row[s + 1]])[0] for s in range(len(row) - 1)]
        return np.array([len(p) for p in per], np.int64), (np.concatenate(per) if per else np.zeros(0, np.int64))

    def mask(cps, row):
        ran_on.append(threading.get_ident())
        return oracle.split_batch(np.ascontiguousarray(cps, np.uint32), np.ascontiguousarray(row, np.int64), want_values=False)[1]

    def tokenize(texts):
        ran_on.append(threading.get_ident())
        return [oracle.tokenize(t) if t else [] for t in texts]

    monkeypatch.setattr(batch, "split_offsets_csr", offsets_csr)
    monkeypatch.setattr(batch, "split_mask_batch", mask)
    monkeypatch.setattr(batch, "tokenize_batch", tokenize)
    return ran_on


@pytest.mark.parametrize("n_workers", [1, 2, 3, 8])
def test_sharded_results_equal_the_whole_batch(fake_batch, n_workers):
    from latok_amd import batch, multi
    rng = random.Random(7 * n_workers)
    FakeCtx.made.clear()
    with multi.DevicePool(list(range(n_workers)), ctx_factory=FakeCtx) as pool:
        ctxs = list(FakeCtx.made)
        assert [c.device for c in ctxs] == list(range(n_workers))
        assert len({c.thread for c in ctxs}) == n_workers and threading.get_ident() not in {c.thread for c in ctxs}
        for texts in (random_strings(rng, 300, 0, 80, ALPHABETS["mixed"]), random_strings(rng, 5, 0, 3000, ALPHABETS["words"]),
                      ["", "", "x"], ["only one string, with a http://u.rl/ in it"], [""] * 3,
                      random_strings(rng, 40, 60, 70, ALPHABETS["starts"])):
            cps, row = pack(texts)
            fake_batch.clear()
            c1, o1 = batch.split_offsets_csr(cps, row)
            whole_thread = set(fake_batch)
            fake_batch.clear()
            c2, o2 = multi.split_offsets_csr(cps, row, pool)
            assert np.array_equal(c1, c2) and np.array_equal(o1, o2)
            assert set(fake_batch) <= {c.thread for c in ctxs} and not (set(fake_batch) & whole_thread)
            # the shards' bitmasks start at arbitrary bit positions of the batch: merged bit-exactly
            assert np.array_equal(batch.split_mask_batch(cps, row), multi.split_mask_batch(cps, row, pool))
            assert batch.tokenize_batch(texts) == multi.tokenize_batch(texts, pool)
    assert all(c.destroyed for c in ctxs)


def test_shards_are_balanced_by_chars_and_run_concurrently(fake_batch):
    from latok_amd import multi, shard
    texts = ["a" * 10] * 1000 + ["b" * 10000]          # one heavy string at the end
    cps, row = pack(texts)
    b = shard.shard_bounds(row, 2)
    assert b.tolist() == [0, 1000, 1001]               # half of the chars each, not half of the strings
    gate = threading.Barrier(2, timeout=20)

    def meet(u, r):                                    # both shards must be inside their call at the same time
        gate.wait()
        return len(r) - 1

    with multi.DevicePool([0, 1], ctx_factory=FakeCtx) as pool:
        bounds, res = multi.map_shards(meet, cps, row, pool)
        assert res == [1000, 1]


def test_errors_propagate_after_all_shards_finished(fake_batch):
    from latok_amd import multi
    cps, row = pack(["abc def"] * 100)
    finished = []

    def job(u, r):
        if threading.current_thread().name.endswith("dev1"):
            raise ValueError("bad shard")
        finished.append(1)
        return 0

    with multi.DevicePool([0, 1, 2], ctx_factory=FakeCtx) as pool:
        with pytest.raises(ValueError, match="bad shard"):
            multi.map_shards(job, cps, row, pool)
        assert len(finished) == 2
        # the pool is still usable afterwards
        assert sum(multi.map_shards(lambda u, r: len(r) - 1, cps, row, pool)[1]) == 100


def test_pool_creation_failure_is_reported():
    from latok_amd import multi

    class Broken(FakeCtx):
        def __init__(self, device):
            if device == 1:
                raise RuntimeError("no such device")
            super().__init__(device)

    with pytest.raises(RuntimeError, match="no such device"):
        multi.DevicePool([0, 1], ctx_factory=Broken)
    with pytest.raises(ValueError):
        multi.DevicePool([])


# ---- device-resident shards (DevicePool.put_csr / generate / split_mask / split_offsets / token_spans) ------------------
class FakeDeviceLib:
    """The slice of the C ABI the resident forms use, over FAKE device memory (a dict of numpy buffers per pointer; every
    buffer remembers the thread = context that allocated it), compute done by the oracle.  Checks what the pool must get
    right: every pointer is used only on the worker that owns it, nothing leaks, the capacity protocol is honoured."""

    def __init__(self, oracle):
        import ctypes
        from latok_amd import _lib
        self.ct, self.real, self.oracle = ctypes, _lib.load(), oracle
        self.lock = threading.Lock()
        self.mem, self.owner, self.next = {}, {}, 0x10000
        self.calls = []

    def _buf(self, p, dtype=np.uint8, count=-1):
        if p not in self.mem:      # an address inside an allocation (base + offset)
            base = max(b for b in self.mem if b <= p)
            assert p < base + self.mem[base].size and self.owner[base] == threading.get_ident()
            return self.mem[base][p - base:].view(dtype)[:count] if count >= 0 else self.mem[base][p - base:].view(dtype)
        assert self.owner[p] == threading.get_ident(), "device pointer used on a context that does not own it"
        return self.mem[p].view(dtype)[:count] if count >= 0 else self.mem[p].view(dtype)

    def latok_dev_alloc(self, nbytes):
        with self.lock:
            self.next += 0x100000
            self.mem[self.next] = np.zeros((nbytes + 7) // 8 * 8, np.uint8)
            self.owner[self.next] = threading.get_ident()
            return self.next

    def latok_dev_free(self, p):
        with self.lock:
            assert self.owner.pop(p) == threading.get_ident()
            del self.mem[p]
        return 0

    def latok_memcpy_h2d(self, d, s, n):
        self._buf(d)[:n] = np.frombuffer(self.ct.string_at(s, n), np.uint8)
        return 0

    def latok_memcpy_d2h(self, d, s, n):
        self.ct.memmove(d, self._buf(s)[:n].ctypes.data, n)
        return 0

    def latok_sync(self):
        return 0

    def latok_corpus_offsets(self, *a):
        return self.real.latok_corpus_offsets(*a)

    def latok_corpus_fill_device(self, seed, model, sid0, n_str, d_row, d_cps, stream):
        row = self._buf(d_row, np.int64, n_str + 1).copy()
        cps = np.zeros(int(row[-1]), np.uint32)
        rc = self.real.latok_corpus_fill_host(seed, model, sid0, n_str, row.ctypes.data, cps.ctypes.data)
        self._buf(d_cps, np.uint32)[:cps.size] = cps
        return rc

    def _csr(self, d_cps, d_row, n_str, total):
        return self._buf(d_cps, np.uint32, total).copy(), self._buf(d_row, np.int64, n_str + 1).copy()

    def latok_split_mask_batch(self, d_cps, d_row, n_str, total, d_bits, flags, stream):
        from latok_amd import _lib
        assert flags & _lib.DEVICE_PTRS
        cps, row = self._csr(d_cps, d_row, n_str, total)
        bits = self.oracle.split_batch(cps, row, want_values=False)[1]
        self._buf(d_bits, np.uint64)[:bits.size] = bits
        self.calls.append(("mask", threading.get_ident(), n_str))
        return 0

    def _records(self, d_cps, d_row, n_str, total, d_counts, d_items, cap, n_out, flags, spans):
        from latok_amd import _lib
        assert flags & _lib.DEVICE_PTRS
        cps, row = self._csr(d_cps, d_row, n_str, total)
        dt = np.int32 if flags & _lib.OUT_INT32 else np.int64
        vals = self.oracle.split_batch(cps, row, want_bits=False)[0]
        per = []
        for s in range(n_str):
            nz = np.nonzero(vals[row[s]:row[s + 1]])[0]
            if not spans:
                per.append(nz.astype(dt))
                continue
            text = cps[row[s]:row[s + 1]].astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")
            cuts = nz.tolist() + [len(text)]
            rec = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                tok = text[a:b]
                if tok.strip():
                    lead = len(tok) - len(tok.lstrip())
                    rec.append((a + lead, a + lead + len(tok.strip())))
            per.append(np.array(rec, dt).reshape(-1, 2))
        n = sum(len(p) for p in per)
        n_out._obj.value = n
        self._buf(d_counts, dt)[:n_str] = [len(p) for p in per]
        self.calls.append(("spans" if spans else "offsets", threading.get_ident(), n_str, cap, n))
        if n > cap:                 # as api.cpp does: the total is reported, nothing is written, LATOK_ERR_INVALID
            return -1
        if n:
            flat = np.concatenate(per).ravel()
            self._buf(d_items, dt)[:flat.size] = flat
        return 0

    def latok_split_offsets_batch(self, *a):
        return self._records(*a[:9], spans=False)

    def latok_token_spans_batch(self, *a):
        return self._records(*a[:9], spans=True)

    # the batch flow (latok_flow_*): submissions are queued per thread (= context) and only run at latok_flow_wait, so a caller
    # that read a result word before the wait would see nothing -- the late capacity protocol is what is being tested
    def _flow(self, spans, d_units, kind, d_row, n_str, total, d_counts, d_items, cap, d_result, flags):
        from latok_amd import _lib
        assert kind == 4, "the fake flow knows UTF-32 only"
        self.__dict__.setdefault("flow_q", {}).setdefault(threading.get_ident(), []).append(
            (spans, d_units, d_row, n_str, total, d_counts, d_items, cap, d_result, flags | _lib.DEVICE_PTRS))
        return 0

    def latok_flow_split_mask(self, d_cps, d_row, n_str, total, d_bits):
        from latok_amd import _lib
        self.__dict__.setdefault("flow_masks", {}).setdefault(threading.get_ident(), []).append((d_cps, d_row, n_str, total, d_bits))
        return 0

    def latok_flow_split_offsets(self, *a):
        return self._flow(False, *a)

    def latok_flow_token_spans(self, *a):
        return self._flow(True, *a)

    def latok_flow_wait(self):
        import ctypes
        from latok_amd import _lib
        for a in self.__dict__.get("flow_masks", {}).pop(threading.get_ident(), []):
            self.latok_split_mask_batch(*a, _lib.DEVICE_PTRS, None)
        for spans, d_units, d_row, n_str, total, d_counts, d_items, cap, d_result, flags in self.__dict__.get("flow_q", {}).pop(threading.get_ident(), []):
            n = ctypes.c_int64(0)
            self._records(d_units, d_row, n_str, total, d_counts, d_items, cap, ctypes.byref(n), flags, spans=spans)
            res = self._buf(d_result, np.int64)    # (the caller may pass an address inside a larger allocation)
            res[:2] = (n.value, 0)
        return 0


@pytest.mark.parametrize("n_workers", [1, 3, 8])
def test_resident_shards_bookkeeping_and_results(oracle, n_workers):
    """put_csr -> split_mask / split_offsets / token_spans on device-resident shards == the oracle on the whole batch; the
    device-only form returns per-shard buffers; free() releases everything on the right workers."""
    from latok_amd import multi
    rng = random.Random(11 * n_workers)
    fake = FakeDeviceLib(oracle)
    with multi.DevicePool(list(range(n_workers)), ctx_factory=FakeCtx, lib=fake) as pool:
        for texts in (random_strings(rng, 200, 0, 80, ALPHABETS["mixed"]), ["", "", "x"], ["one string only, http://u.rl/ #tag"],
                      random_strings(rng, 5, 0, 3000, ALPHABETS["words"]), ["spaces   everywhere   "] * 40):
            cps, row = pack(texts)
            with pool.put_csr(cps, row) as rb:
                assert rb.n_str == len(texts) and rb.total == int(row[-1])
                live = [sh for sh in rb.shards if sh is not None]
                assert sum(sh.n_str for sh in live) == len(texts) and len({sh.worker for sh in live}) == len(live)
                vals, bits = oracle.split_batch(cps, row)
                assert np.array_equal(pool.split_mask(rb), bits)
                exp = [np.nonzero(vals[row[s]:row[s + 1]])[0] for s in range(len(texts))]
                for dtype in (np.int32, np.int64):
                    counts, offs = pool.split_offsets(rb, dtype=dtype)
                    assert counts.dtype == dtype and counts.tolist() == [len(e) for e in exp]
                    assert offs.tolist() == [int(v) for e in exp for v in e]
                counts, spans = pool.token_spans(rb)
                want = [[t for t in oracle.tokenize(x)] if x else [] for x in texts]
                got, k = [], 0
                for x, c in zip(texts, counts.tolist()):
                    got.append([x[a:b] for a, b in spans[k:k + c].tolist()])
                    k += c
                assert got == want
                # device-only form: nothing comes back but the shards with their buffers and item totals
                shards = pool.split_offsets(rb, to_host=False)
                assert sum(sh.n_items[1] for sh in shards if sh.n_str) == sum(len(e) for e in exp)
                assert all(("d_items", 1) in sh.bufs for sh in shards if sh.n_str)
                shards = pool.split_mask(rb, to_host=False)
                assert all(sh.d_bits for sh in shards if sh.total)
            assert not fake.mem, "free() must release every device buffer"
        # every shard ran on its own worker thread
        threads = {c[1] for c in fake.calls}
        assert threading.get_ident() not in threads and len(threads) <= n_workers


def test_resident_capacity_protocol_grows_the_record_buffer(oracle):
    from latok_amd import multi
    texts = ["a b c d e f g h i j k l m n o p"] * 300          # every other char is a boundary: more items than total / 3
    cps, row = pack(texts)
    fake = FakeDeviceLib(oracle)
    with multi.DevicePool([0], ctx_factory=FakeCtx, lib=fake) as pool, pool.put_csr(cps, row) as rb:
        counts, offs = pool.split_offsets(rb, dtype=np.int32)
        calls = [c for c in fake.calls if c[0] == "offsets"]
        assert len(calls) == 2 and calls[0][4] > calls[0][3] and calls[1][3] >= calls[1][4]     # too small, then grown
        assert counts.tolist() == [len(np.nonzero(oracle.split_values(t))[0]) for t in texts]
        assert offs.size == int(counts.sum())
        fake.calls.clear()
        pool.split_offsets(rb, dtype=np.int32)
        assert len([c for c in fake.calls if c[0] == "offsets"]) == 1                           # the grown buffer is kept


@pytest.mark.parametrize("n_workers", [1, 3])
def test_several_resident_batches_through_the_flow(oracle, n_workers):
    """DevicePool.split_offsets_many / token_spans_many: every worker submits its shard of EVERY batch to its context's flow,
    waits once, and re-submits the shards whose records did not fit (the total is only known after the wait)"""
    from latok_amd import multi
    rng = random.Random(77 + n_workers)
    fake = FakeDeviceLib(oracle)
    sets = [random_strings(rng, 150, 0, 60, ALPHABETS["mixed"]), ["a b c d e f g h i j k l m n o p"] * 200, ["", "x", ""],
            random_strings(rng, 4, 0, 2000, ALPHABETS["words"])]
    with multi.DevicePool(list(range(n_workers)), ctx_factory=FakeCtx, lib=fake) as pool:
        rbs = [pool.put_csr(*pack(t)) for t in sets]
        for dtype in (np.int32, np.int64):
            fake.calls.clear()
            res = pool.split_offsets_many(rbs, dtype=dtype)
            for texts, (counts, offs) in zip(sets, res):
                exp = [np.nonzero(oracle.split_values(t))[0] if t else np.zeros(0, np.int64) for t in texts]
                assert counts.dtype == dtype and counts.tolist() == [len(e) for e in exp]
                assert offs.tolist() == [int(v) for e in exp for v in e]
        grown = [c for c in fake.calls if c[0] == "offsets" and c[4] > c[3]]
        assert not grown, "the second pass (int64) reuses the buffers the first one grew"
        for texts, bits in zip(sets, pool.split_mask_many(rbs)):
            assert np.array_equal(bits, oracle.split_batch(*pack(texts), want_values=False)[1])
        res = pool.token_spans_many(rbs)
        for texts, (counts, spans) in zip(sets, res):
            k = 0
            for x, c in zip(texts, counts.tolist()):
                assert [x[a:b] for a, b in spans[k:k + c].tolist()] == (oracle.tokenize(x) if x else [])
                k += c
        for rb in rbs:
            rb.free()
    assert not fake.mem


def test_resident_generate_owns_disjoint_string_ids(oracle):
    from latok_amd import _lib, multi
    fake = FakeDeviceLib(oracle)
    lib = _lib.load()
    n = 1000
    row = np.zeros(n + 1, np.int64)
    lib.latok_corpus_offsets(0x1A70C0DE, 5000, n, 64, 192, row.ctypes.data)
    cps = np.zeros(int(row[-1]), np.uint32)
    lib.latok_corpus_fill_host(0x1A70C0DE, 0, 5000, n, row.ctypes.data, cps.ctypes.data)
    with multi.DevicePool([0, 1, 2], ctx_factory=FakeCtx, lib=fake) as pool, pool.generate(0x1A70C0DE, 0, n, 64, 192, sid0=5000) as rb:
        assert rb.total == int(row[-1]) and [sh.s0 for sh in rb.shards] == [0, 333, 666]
        assert np.array_equal(pool.split_mask(rb), oracle.split_batch(cps, row, want_values=False)[1])
