"""In-process sharding of one batch over several GPUs (SURVEY 8e, 8b "one host thread + stream(s) per GPU").

Every string is tokenized independently (reference tokenize() takes one str, default_tokenizer.py:137; no cross-string
state in latok.c), so a CSR batch is cut into contiguous string-id ranges balanced by char count
(``shard.shard_bounds``), each range runs on its own library context from its own host thread -- ctypes releases the
GIL while a call is inside the library -- and the per-string results concatenate in range order to exactly what one
context would have produced for the whole batch.  No collective, no device-to-device traffic.

    pool = DevicePool([0, 1, 2, 3, 4, 5, 6, 7])          # one context per GPU of the node
    tokens = tokenize_batch(texts, pool)                  # == batch.tokenize_batch(texts)
    counts, offsets = split_offsets_csr(cps, row_off, pool)

A device may be listed more than once (two contexts on one GPU overlap one shard's copies with the other's kernels).
One process per GPU (bench.py, torchrun) remains the other way to use a node; this is the one for a caller that holds
the whole batch in one process.
"""
import queue
import threading

import numpy as np

from . import _lib, batch, shard


class _Worker(threading.Thread):
    """A host thread bound to one context for its whole life (the context is the thread's current one)."""

    def __init__(self, device, ctx_factory):
        super().__init__(daemon=True, name=f"latok-dev{device}")
        self.device = device
        self._factory = ctx_factory
        self._jobs = queue.Queue()
        self._ready = threading.Event()
        self.error = None
        self.start()
        self._ready.wait()
        if self.error is not None:
            raise self.error

    def run(self):
        try:
            ctx = self._factory(self.device)
            ctx.make_current()
        except BaseException as exc:   # reported to the creating thread
            self.error = exc
            self._ready.set()
            return
        self._ready.set()
        while True:
            job = self._jobs.get()
            if job is None:
                break
            fn, box, done = job
            try:
                box.append((True, fn()))
            except BaseException as exc:
                box.append((False, exc))
            done.set()
        ctx.destroy()

    def submit(self, fn):
        box, done = [], threading.Event()
        self._jobs.put((fn, box, done))
        return box, done

    def close(self):
        self._jobs.put(None)
        self.join()


class DevicePool:
    """One context + one worker thread per entry of ``devices``."""

    def __init__(self, devices, ctx_factory=_lib.Context):
        devices = list(devices)
        if not devices:
            raise ValueError("DevicePool needs at least one device")
        self.devices = devices
        self._workers = []
        try:
            for d in devices:
                self._workers.append(_Worker(d, ctx_factory))
        except BaseException:
            self.close()
            raise

    def __len__(self):
        return len(self._workers)

    def run(self, jobs):
        """jobs[i] runs on worker i (None = nothing for that worker); returns the results in the same order.  The first
        failure is re-raised after every job has finished (no worker is left mid-call)."""
        if len(jobs) != len(self._workers):
            raise ValueError("one job per worker")
        pending = [w.submit(j) if j is not None else None for w, j in zip(self._workers, jobs)]
        out, err = [], None
        for p in pending:
            if p is None:
                out.append(None)
                continue
            box, done = p
            done.wait()
            ok, val = box[0]
            if ok:
                out.append(val)
            else:
                out.append(None)
                err = err or val
        if err is not None:
            raise err
        return out

    def broadcast(self, fn):
        """run fn() once on every worker (e.g. per-context state like run-time rule tables)"""
        return self.run([fn] * len(self._workers))

    def set_rules(self, c_split, c_mask, c_sym):
        self.broadcast(lambda: batch.set_rules(c_split, c_mask, c_sym))

    def reset_rules(self):
        self.broadcast(batch.reset_rules)

    def close(self):
        for w in self._workers:
            w.close()
        self._workers = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def _as_pool(devices):
    """(pool, owned): a DevicePool as given, or a temporary one over a list of device ids"""
    if isinstance(devices, DevicePool):
        return devices, False
    return DevicePool(devices), True


def map_shards(fn, units, row_off, pool):
    """Cut the CSR batch (units = code points / code units / bytes, row_off in units) into len(pool) contiguous string
    ranges balanced by unit count, run fn(units_r, row_off_r) for range r on worker r, return (bounds, results).  Empty
    ranges (more workers than strings) are skipped: their result is None."""
    row_off = np.ascontiguousarray(row_off, dtype=np.int64)
    units = np.asarray(units)
    bounds = shard.shard_bounds(row_off, len(pool))
    jobs = []
    for r in range(len(pool)):
        s0, s1 = int(bounds[r]), int(bounds[r + 1])
        if s1 <= s0:
            jobs.append(None)
            continue
        lo, hi = int(row_off[s0]), int(row_off[s1])
        u_r = units[lo:hi]
        row_r = row_off[s0:s1 + 1] - lo
        jobs.append(lambda u_r=u_r, row_r=row_r: fn(u_r, row_r))
    return bounds, pool.run(jobs)


def _concat_counts_items(results, n_str, width):
    parts = [r for r in results if r is not None]
    if not parts:
        return np.zeros(n_str, np.int64), np.zeros((0, width) if width > 1 else 0, np.int64)
    counts = np.concatenate([p[0] for p in parts])
    items = np.concatenate([p[1] for p in parts])
    return counts, items


def split_offsets_csr(cps, row_off, devices):
    """batch.split_offsets_csr over several devices: (counts int64[n], offsets int64[sum(counts)])."""
    pool, owned = _as_pool(devices)
    try:
        _, res = map_shards(batch.split_offsets_csr, np.ascontiguousarray(cps, dtype=np.uint32), row_off, pool)
        return _concat_counts_items(res, len(row_off) - 1, 1)
    finally:
        if owned:
            pool.close()


def token_spans_csr(cps, row_off, devices):
    """batch.token_spans_csr over several devices: (counts, spans int64[n_tokens, 2])."""
    pool, owned = _as_pool(devices)
    try:
        _, res = map_shards(batch.token_spans_csr, np.ascontiguousarray(cps, dtype=np.uint32), row_off, pool)
        return _concat_counts_items(res, len(row_off) - 1, 2)
    finally:
        if owned:
            pool.close()


def token_features_csr(cps, row_off, devices):
    """batch.token_features_csr over several devices: (counts, spans int64[n_tokens, 4], features int8[n_tokens, 25])."""
    pool, owned = _as_pool(devices)
    try:
        _, res = map_shards(batch.token_features_csr, np.ascontiguousarray(cps, dtype=np.uint32), row_off, pool)
        parts = [r for r in res if r is not None]
        if not parts:
            return np.zeros(len(row_off) - 1, np.int64), np.zeros((0, 4), np.int64), np.zeros((0, _lib.FEATURE_COUNT), np.int8)
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))
    finally:
        if owned:
            pool.close()


def split_mask_batch(cps, row_off, devices):
    """batch.split_mask_batch over several devices.  The shards' bitmasks start at bit 0 of their own buffers, the
    batch's bits are packed over the whole buffer, so they are merged bit-exactly on the host (shift + or)."""
    pool, owned = _as_pool(devices)
    try:
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        bounds, res = map_shards(batch.split_mask_batch, np.ascontiguousarray(cps, dtype=np.uint32), row_off, pool)
        total = int(row_off[-1]) if row_off.size > 1 else 0
        out = np.zeros((total + 63) // 64 + 1, np.uint64)
        for r, bits in enumerate(res):
            if bits is None or bits.size == 0:
                continue
            lo = int(row_off[int(bounds[r])])
            w, sh = lo >> 6, np.uint64(lo & 63)
            out[w:w + bits.size] |= bits << sh
            if sh:
                out[w + 1:w + 1 + bits.size] |= bits >> (np.uint64(64) - sh)
        return out[:(total + 63) // 64]
    finally:
        if owned:
            pool.close()


def _texts_sharded(fn_texts, texts, devices):
    """shard a list of str by char count; fn_texts(sub_list) runs on each worker; results (lists) are concatenated"""
    pool, owned = _as_pool(devices)
    try:
        lens = np.fromiter((len(t) for t in texts), dtype=np.int64, count=len(texts))
        row_off = np.zeros(len(texts) + 1, np.int64)
        np.cumsum(lens, out=row_off[1:])
        bounds = shard.shard_bounds(row_off, len(pool))
        jobs = []
        for r in range(len(pool)):
            s0, s1 = int(bounds[r]), int(bounds[r + 1])
            jobs.append((lambda sub=texts[s0:s1]: fn_texts(sub)) if s1 > s0 else None)
        out = []
        for part in pool.run(jobs):
            if part is not None:
                out.extend(part)
        return out
    finally:
        if owned:
            pool.close()


def tokenize_batch(texts, devices):
    """list[str] -> list[list[str]] like batch.tokenize_batch, the strings sharded over ``devices`` (a DevicePool or a
    list of device ids)."""
    return _texts_sharded(batch.tokenize_batch, list(texts), devices) if len(texts) else []


def split_offsets_batch(texts, devices):
    return _texts_sharded(batch.split_offsets_batch, list(texts), devices) if len(texts) else []


def featurize_batch(texts, devices):
    return _texts_sharded(batch.featurize_batch, list(texts), devices) if len(texts) else []
