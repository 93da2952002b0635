#!/usr/bin/env python3
"""Dev tool: where a wave's time goes inside the tile function, per input mode, from the stamped diagnostic build
(make -C latok_amd/csrc diag).  usage: tools/stamp_profile_modes.py [utf32|latin1|bytes] [diag .so]"""
import ctypes as C, os, sys
import numpy as np
ROOT = __file__.rsplit("/tools/", 1)[0]
mode = sys.argv[1] if len(sys.argv) > 1 else "latin1"
os.environ["LATOK_HIP_LIB"] = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "latok_amd", "liblatok_hip_diag.so")
sys.path.insert(0, ROOT)
from latok_amd import _lib
lib = _lib.ensure_init()
n_str = 1_000_000
row = np.zeros(n_str + 1, np.int64)
lib.latok_corpus_offsets(0x1A70C0DE, 0, n_str, 64, 192, row.ctypes.data)
total = int(row[-1])
cps = np.zeros(total, np.uint32)
lib.latok_corpus_fill_host(0x1A70C0DE, 0, 0, n_str, row.ctypes.data, cps.ctypes.data)
data = cps if mode == "utf32" else cps.astype(np.uint8)
d_row = lib.latok_dev_alloc(row.nbytes); d_in = lib.latok_dev_alloc(data.nbytes + 64); d_bits = lib.latok_dev_alloc(((total + 63) // 64) * 8)
lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes)
lib.latok_memcpy_h2d(d_in, data.ctypes.data, data.nbytes)
raw = C.CDLL(os.environ["LATOK_HIP_LIB"])
out = (C.c_ulonglong * 16)()
D = _lib.DEVICE_PTRS
def call():
    if mode == "utf32":
        return lib.latok_split_mask_batch(d_in, d_row, n_str, total, d_bits, D, None)
    if mode == "latin1":
        return lib.latok_split_mask_kind_batch(d_in, 1, d_row, n_str, total, d_bits, D, None)
    return lib.latok_split_mask_utf8_bytes_batch(d_in, d_row, n_str, total, d_bits, D, None)
for _ in range(2):
    _lib.check(call())
lib.latok_sync()
raw.latok_diag_stamps(out, 1)
import time
t = time.perf_counter()
for _ in range(10):
    _lib.check(call())
lib.latok_sync()
dt = (time.perf_counter() - t) / 10
raw.latok_diag_stamps(out, 1)
names = {1: "issue the tile's loads (utf32)", 2: "phase 1 (loads + wait + classify / stage)", 3: "B words (+row_off)",
         9: "phase 2: LDS reads", 10: "phase 2: LUT slice", 4: "phase 2: (slice +) rules", 5: "forward + wave scan",
         6: "summary", 7: "backward", 8: "output store / values"}
tiles = out[0] or 1
tot = sum(out[i] for i in range(1, 16))
print(f"{mode}, stamped build {os.path.basename(os.environ['LATOK_HIP_LIB'])}: {tiles} stamped tiles, {dt * 1e6:.1f} us per call (stamped: slower than the product)")
for i in (1, 2, 3, 9, 10, 4, 5, 6, 7, 8):
    if out[i]:
        print(f"  {names[i]:44s} {out[i] / tiles:9.0f} clk  {100.0 * out[i] / tot:5.1f} %")
print(f"  total per tile {tot / tiles:.0f} clk (s_memtime ticks)")
