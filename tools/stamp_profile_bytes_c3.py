#!/usr/bin/env python3
"""Dev tool: phase shares of the tile function in BYTE mode on the C3 (mixed-Unicode) corpus from the stamped diagnostic build."""
import ctypes as C, os, sys
import numpy as np
ROOT = __file__.rsplit("/tools/", 1)[0]
os.environ["LATOK_HIP_LIB"] = os.path.join(ROOT, "latok_amd", "liblatok_hip_diag.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from latok_amd import _lib
from path_bench import utf8_of
lib = _lib.ensure_init()
n_str = 300_000
row = np.zeros(n_str + 1, np.int64)
lib.latok_corpus_offsets(0x1A70C0DF, 0, n_str, 128, 384, row.ctypes.data)
total = int(row[-1])
cps = np.zeros(total, np.uint32)
lib.latok_corpus_fill_host(0x1A70C0DF, 1, 0, n_str, row.ctypes.data, cps.ctypes.data)
u8, boff = utf8_of(cps, row)
n8 = int(u8.size)
d_row = lib.latok_dev_alloc(boff.nbytes); d_u8 = lib.latok_dev_alloc(n8 + 64); d_bits = lib.latok_dev_alloc(((n8 + 63) // 64) * 8)
lib.latok_memcpy_h2d(d_row, boff.ctypes.data, boff.nbytes)
lib.latok_memcpy_h2d(d_u8, u8.ctypes.data, n8)
raw = C.CDLL(os.environ["LATOK_HIP_LIB"])
out = (C.c_ulonglong * 16)()
D = _lib.DEVICE_PTRS
for _ in range(2):
    _lib.check(lib.latok_split_mask_utf8_bytes_batch(d_u8, d_row, n_str, n8, d_bits, D, None))
lib.latok_sync()
raw.latok_diag_stamps(out, 1)
for _ in range(10):
    _lib.check(lib.latok_split_mask_utf8_bytes_batch(d_u8, d_row, n_str, n8, d_bits, D, None))
lib.latok_sync()
raw.latok_diag_stamps(out, 1)
names = ["tiles", "-", "phase 1 (loads + classify)", "B words (+row_off)", "LDS reads + bitslice + rules",
         "forward + wave scan", "summary", "backward", "output store", "(share of 4) ds reads", "(share of 4) slicing", "(of phase 1) until the bytes arrive", "(of phase 1) owner before the tile",
         "(of phase 1) the four rows"]
tiles = out[0] or 1
tot = sum(out[i] for i in range(1, 9))
print(f"byte mode on C3 text, stamped build: {tiles} stamped tiles, {n8 / total:.2f} bytes per char")
for i in range(1, 14):
    print(f"  {names[i]:34s} {out[i] / tiles:9.0f} clk  {100.0 * out[i] / tot:5.1f} %")
print(f"  total per tile {tot / tiles:.0f} clk (s_memtime ticks)")
