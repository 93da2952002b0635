#!/usr/bin/env python3
"""Dev tool: PCIe-inclusive rate of the host-pointer API (H2D of the batch + pipeline + D2H of the bitmask)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib, batch
lib = _lib.ensure_init()
n = 1_000_000
row = np.zeros(n + 1, np.int64)
lib.latok_corpus_offsets(0x1A70C0DE, 0, n, 64, 192, row.ctypes.data)
cps = np.zeros(int(row[-1]), np.uint32)
lib.latok_corpus_fill_host(0x1A70C0DE, 0, 0, n, row.ctypes.data, cps.ctypes.data)
batch.split_mask_batch(cps, row)
t = time.perf_counter()
for _ in range(5):
    batch.split_mask_batch(cps, row)
dt = (time.perf_counter() - t) / 5
print(f"host-pointer split_mask_batch: {dt * 1e3:.2f} ms per 1M-string batch = {cps.size / dt / 1e9:.2f} GB/s UTF-8 (PCIe-inclusive, pageable host memory)")
t = time.perf_counter()
counts, offs = batch.split_offsets_csr(cps, row)
dt = time.perf_counter() - t
print(f"host-pointer split_offsets_csr: {dt * 1e3:.2f} ms ({offs.size} boundaries)")
# same batch handed over as UTF-8 (ASCII corpus: 1 byte per char over PCIe instead of 4)
u8 = cps.astype(np.uint8)
t = time.perf_counter()
for _ in range(3):
    c2, o2 = batch.split_offsets_utf8_csr(u8, row)
dt8 = (time.perf_counter() - t) / 3
t = time.perf_counter()
for _ in range(3):
    c1, o1 = batch.split_offsets_csr(cps, row)
dt32 = (time.perf_counter() - t) / 3
assert np.array_equal(o1, o2)
print(f"host-pointer offsets: UTF-32 input {dt32 * 1e3:.1f} ms, UTF-8 input {dt8 * 1e3:.1f} ms per 1M-string batch "
      f"({cps.size / dt32 / 1e9:.2f} vs {cps.size / dt8 / 1e9:.2f} GB/s UTF-8, output copy of {o1.size} int64 offsets included)")
t = time.perf_counter()
for _ in range(5):
    mb, mrow = batch.split_mask_utf8_csr(u8, row)
dtm = (time.perf_counter() - t) / 5
print(f"host-pointer split_mask from UTF-8: {dtm * 1e3:.2f} ms per batch = {cps.size / dtm / 1e9:.2f} GB/s UTF-8 (PCIe-inclusive)")
