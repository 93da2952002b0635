#!/usr/bin/env python3
"""Dev tool: does the placement of the code-point buffer (offset inside one big allocation) change the tile kernel's time?"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib
lib = _lib.ensure_init()
n_str = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
row = np.zeros(n_str + 1, np.int64)
lib.latok_corpus_offsets(0x1A70C0DE, 0, n_str, 64, 192, row.ctypes.data)
total = int(row[-1])
slack = 64 << 20
big = lib.latok_dev_alloc(total * 4 + slack)
d_row = lib.latok_dev_alloc(row.nbytes)
d_bits = lib.latok_dev_alloc(((total + 63) // 64) * 8 + (1 << 20))
lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes)
print(f"base {big:#x} rows {d_row:#x} bits {d_bits:#x}")
for off in [0, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 65536, 3 << 20, 8 << 20, (16 << 20) + 16384, 33 << 20]:
    p = big + off
    _lib.check(lib.latok_corpus_fill_device(0x1A70C0DE, 0, 0, n_str, d_row, p, None))
    best = 1e9
    for _ in range(3):
        ms_tot, ms_tiles = C.c_float(0), C.c_float(0)
        _lib.check(lib.latok_bench_split_mask(p, d_row, n_str, total, d_bits, 2, 20, C.byref(ms_tot), C.byref(ms_tiles), None))
        best = min(best, ms_tiles.value / 20)
    print(f"offset {off:>10d}  tiles kernel {best * 1e3:7.1f} us   pipeline {ms_tot.value / 20 * 1e3:7.1f} us")
for boff in [0, 4096, 65536, 1 << 19]:
    ms_tot, ms_tiles = C.c_float(0), C.c_float(0)
    _lib.check(lib.latok_bench_split_mask(big, d_row, n_str, total, d_bits + boff, 2, 20, C.byref(ms_tot), C.byref(ms_tiles), None))
    print(f"bits offset {boff:>8d}  tiles kernel {ms_tiles.value / 20 * 1e3:7.1f} us")
