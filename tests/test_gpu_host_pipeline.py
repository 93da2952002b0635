"""Large host-pointer batches take a chunked pipeline (api.cpp: compact_host_pipelined): chunks of whole strings move up
the bus, through the kernels and down again on three streams with double buffers.  The results must be exactly what the
one-shot device-pointer call gives for the whole batch -- for every record kind, width and input form, with pageable and
with pinned host arrays, with a string far longer than a chunk, and when the caller's buffer is too small."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus(lib, n_str, lo, hi, seed=0x1A70C0DE):
    from latok_amd import _lib
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n_str, lo, hi, row.ctypes.data))
    cps = np.zeros(int(row[-1]), np.uint32)
    _lib.check(lib.latok_corpus_fill_host(seed, 0, 0, n_str, row.ctypes.data, cps.ctypes.data))
    return cps, row


def _device_reference(lib, fn, units, kind_args, row, width, dt, feats=False):
    """the same entry point with device pointers (one shot, no pipeline)"""
    from latok_amd import _lib
    n_str, total = len(row) - 1, int(row[-1])
    flags = _lib.DEVICE_PTRS | (_lib.OUT_INT32 if dt == np.int32 else 0)
    d_units, d_row = lib.latok_dev_alloc(units.nbytes + 64), lib.latok_dev_alloc(row.nbytes)
    _lib.check(lib.latok_memcpy_h2d(d_units, units.ctypes.data, units.nbytes))
    _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
    isz = np.dtype(dt).itemsize
    d_counts = lib.latok_dev_alloc(n_str * isz + 16)
    cap = total // 2 + 4096
    d_items = lib.latok_dev_alloc(cap * width * isz)
    d_feat = lib.latok_dev_alloc(cap * 25) if feats else None
    n = C.c_int64(0)
    args = [d_units] + kind_args + [d_row, n_str, total, d_counts, d_items] + ([d_feat] if feats else []) + [cap, C.byref(n), flags, None]
    _lib.check(fn(*args))
    counts = np.empty(n_str, dt)
    items = np.empty((n.value, width) if width > 1 else n.value, dt)
    _lib.check(lib.latok_memcpy_d2h(counts.ctypes.data, d_counts, counts.nbytes))
    _lib.check(lib.latok_memcpy_d2h(items.ctypes.data, d_items, items.nbytes))
    out = [counts, items]
    if feats:
        f = np.empty((n.value, 25), np.int8)
        _lib.check(lib.latok_memcpy_d2h(f.ctypes.data, d_feat, f.nbytes))
        out.append(f)
    for p in (d_units, d_row, d_counts, d_items, d_feat):
        if p:
            lib.latok_dev_free(p)
    return out


def test_pipelined_host_batches_equal_one_shot_device_calls(gpu, oracle):
    from latok_amd import batch
    # 300 K strings ~ 38 M chars: 5 chunks; one 20 M-char string in the middle (a chunk of its own, > 2 x the chunk size)
    cps, row = _corpus(gpu, 300_000, 64, 192)
    big = np.tile(np.frombuffer("lorem ipsum http://x.y/z dolor, Sit@amet.com #tag ".encode("utf-32-le"), "<u4"), 400_000).astype(np.uint32)
    mid = 150_000
    cps = np.concatenate([cps[:row[mid]], big, cps[row[mid]:]])
    row = np.concatenate([row[:mid + 1], row[mid:] + big.size])
    assert int(row[-1]) > 3 * (8 << 20)
    u8 = cps.astype(np.uint8)
    for dt in (np.int64, np.int32):
        ref = _device_reference(gpu, gpu.latok_split_offsets_batch, cps, [], row, 1, dt)
        got = batch.split_offsets_csr(cps, row, dtype=dt)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))
        ref = _device_reference(gpu, gpu.latok_token_spans_kind_batch, u8, [1], row, 2, dt)
        got = batch.token_spans_kind_csr(u8, row, dtype=dt)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))
        ref = _device_reference(gpu, gpu.latok_split_offsets_utf8_bytes_batch, u8, [], row, 1, dt)
        got = batch.split_offsets_utf8_bytes_csr(u8, row, dtype=dt)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    ref = _device_reference(gpu, gpu.latok_token_features_batch, cps, [], row, 4, np.int32, feats=True)
    got = batch.token_features_csr(cps, row, dtype=np.int32)
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    # oracle parity on a slice that straddles the first chunk boundary region and on the tail
    counts, offs = batch.split_offsets_csr(cps, row, dtype=np.int32)
    starts = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])
    for s0, s1 in ((60_000, 70_000), (len(row) - 3001, len(row) - 1)):
        sub_row = row[s0:s1 + 1] - row[s0]
        vals, _ = oracle.split_batch(cps[row[s0]:row[s1]], sub_row, want_bits=False)
        want = np.concatenate([np.nonzero(vals[sub_row[i]:sub_row[i + 1]])[0] for i in range(s1 - s0)])
        assert np.array_equal(offs[starts[s0]:starts[s1]], want)


def test_pipelined_with_pinned_arrays_and_small_capacity(gpu):
    from latok_amd import _lib, batch
    cps, row = _corpus(gpu, 200_000, 64, 192)
    total = int(row[-1])
    want = batch.split_offsets_csr(cps, row, dtype=np.int32)
    p_cps = batch.pinned_empty(cps.size, np.uint32)
    p_cps[:] = cps
    p_row = batch.pinned_empty(row.size, np.int64)
    p_row[:] = row
    got = batch._compact(gpu.latok_split_offsets_batch, [p_cps.ctypes.data, p_row.ctypes.data], len(row) - 1, total, 1, np.int32, pinned=True)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    # too small a buffer: the needed size comes back, the counts are complete, nothing is written past the capacity
    n = len(want[1])
    counts = np.zeros(len(row) - 1, np.int32)
    items = np.full(n, -5, np.int32)
    n_out = C.c_int64(0)
    rc = gpu.latok_split_offsets_batch(cps.ctypes.data, row.ctypes.data, len(row) - 1, total, counts.ctypes.data, items.ctypes.data,
                                       n // 2, C.byref(n_out), _lib.OUT_INT32, None)
    assert rc == _lib.ERR_INVALID and n_out.value == n and np.array_equal(counts, want[0])
    assert (items[n // 2:] == -5).all()
    del got, p_cps, p_row


def test_hundreds_of_tiny_chunks_against_the_oracle():
    """the same pipeline with 4 K-char chunks (test hook LATOK_PIPE_CHUNK_CHARS): every double-buffer hand-over, the
    rebasing of the row offsets, chunks that are one long string, chunks of empty strings -- a few hundred chunks per
    batch, every record kind, checked against the oracle.  Runs in a process of its own (the hook is read once)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import random, sys, numpy as np
sys.path[:0] = [%r, %r + "/oracle", %r + "/tests"]
from conftest import ALPHABETS, pack, random_strings
from latok_amd import batch
import latok_oracle as orc
rng = random.Random(2024)
for rep in range(6):
    texts = random_strings(rng, rng.randint(2000, 6000), 0, 300, ALPHABETS[rng.choice(["mixed", "words", "starts"])])
    texts[rng.randrange(len(texts))] = "".join(rng.choice(ALPHABETS["rare_space_at"]) for _ in range(30000))   # >> one chunk
    for k in range(0, len(texts), 700):
        texts[k:k + 40] = [""] * 40                                                                           # runs of empties
    cps, row = pack(texts)
    assert int(row[-1]) > 40 * 4096
    vals, _ = orc.split_batch(cps, row, want_bits=False)
    want = [np.nonzero(vals[row[s]:row[s + 1]])[0] for s in range(len(texts))]
    for dt in (np.int64, np.int32):
        c, o = batch.split_offsets_csr(cps, row, dtype=dt)
        assert np.array_equal(c, [len(w) for w in want]) and np.array_equal(o, np.concatenate(want)), (rep, dt)
    toks = batch.tokenize_batch(texts)
    assert toks == [orc.tokenize(t) if t else [] for t in texts], rep
    u8 = cps.astype(np.uint8) if int(cps.max()) < 256 else None
    fc, fs, ff = batch.token_features_csr(cps, row, dtype=np.int32)
    k = 0
    for t, n in zip(texts, fc.tolist()):
        if n and len(t) < 400:
            m = orc.gen_parse_matrix(t).astype(np.uint8)
            for a, b, _, _ in fs[k:k + n].tolist():
                assert np.array_equal(ff[k], m[a:b].sum(axis=0, dtype=np.uint64).astype(np.uint8).astype(np.int8)), (rep, t[:30])
                k += 1
        else:
            k += n
print("ok")
''' % (ROOT, ROOT, ROOT)
    env = dict(os.environ, LATOK_PIPE_CHUNK_CHARS="4096")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-3000:])
