#!/usr/bin/env python3
"""Dev tool: device-resident timing of the compaction entry points (offsets / token spans / token features)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib
lib = _lib.ensure_init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
model = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lo, hi = (64, 192) if model == 0 else (128, 384)
if len(sys.argv) > 4:
    lo, hi = int(sys.argv[3]), int(sys.argv[4])
row = np.zeros(n + 1, np.int64)
lib.latok_corpus_offsets(0x1A70C0DE + model, 0, n, lo, hi, row.ctypes.data)
total = int(row[-1])
d_row = lib.latok_dev_alloc(row.nbytes); d_cps = lib.latok_dev_alloc(total * 4)
lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes)
lib.latok_corpus_fill_device(0x1A70C0DE + model, model, 0, n, d_row, d_cps, None)
cap = total // 2 + 1024
d_counts = lib.latok_dev_alloc(n * 8); d_items = lib.latok_dev_alloc(cap * 32); d_feat = lib.latok_dev_alloc(cap * 25)
nout = C.c_int64(0)
def timeit(name, fn, reps=5):
    fn(); lib.latok_sync()
    t = time.perf_counter()
    for _ in range(reps): fn()
    lib.latok_sync()
    dt = (time.perf_counter() - t) / reps
    print(f"{name:28s} {dt * 1e3:8.3f} ms  -> {total / dt / 1e9:7.1f} G chars/s, items={nout.value}")
D = _lib.DEVICE_PTRS
timeit("split_offsets (device ptrs)", lambda: _lib.check(lib.latok_split_offsets_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None)))
timeit("token_spans (device ptrs)", lambda: _lib.check(lib.latok_token_spans_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None)))
timeit("token_features (device ptrs)", lambda: _lib.check(lib.latok_token_features_batch(d_cps, d_row, n, total, d_counts, d_items, d_feat, cap, C.byref(nout), D, None)))
