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
