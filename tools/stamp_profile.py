#!/usr/bin/env python3
"""Dev tool: phase shares of the tile function from the stamped diagnostic build (make -C latok_amd/csrc diag)."""
import ctypes as C, os, sys
import numpy as np
ROOT = __file__.rsplit("/tools/", 1)[0]
os.environ["LATOK_HIP_LIB"] = os.path.join(ROOT, "latok_amd", "liblatok_hip_diag.so")
sys.path.insert(0, ROOT)
from latok_amd import _lib
lib = _lib.ensure_init()
n_str = 1_000_000
row = np.zeros(n_str + 1, np.int64)
lib.latok_corpus_offsets(0x1A70C0DE, 0, n_str, 64, 192, row.ctypes.data)
total = int(row[-1])
d_row = lib.latok_dev_alloc(row.nbytes); d_cps = lib.latok_dev_alloc(total * 4); d_bits = lib.latok_dev_alloc(((total + 63) // 64) * 8)
lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes)
lib.latok_corpus_fill_device(0x1A70C0DE, 0, 0, n_str, d_row, d_cps, None)
raw = C.CDLL(os.environ["LATOK_HIP_LIB"])
out = (C.c_ulonglong * 16)()
ms = C.c_float(0)
_lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 2, 0, None, None, None))
raw.latok_diag_stamps(out, 1)
_lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 0, 10, C.byref(ms), None, None))
raw.latok_diag_stamps(out, 1)
names = ["tiles(non-first)", "issue loads -> before classify", "classify (waits for HBM)", "B words (+row_off)", "LDS reads + bitslice + rules",
         "forward + wave scan", "summary", "backward", "output store"]
tiles = out[0] or 1
tot = sum(out[i] for i in range(1, 9))
print(f"stamped build: {ms.value / 10 * 1e3:.1f} us/pass, {tiles} stamped tiles")
for i in range(1, 9):
    print(f"  {names[i]:34s} {out[i] / tiles:9.0f} clk  {100.0 * out[i] / tot:5.1f} %")
print(f"  total per tile {tot / tiles:.0f} clk (s_memtime ticks)")
