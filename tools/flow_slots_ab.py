#!/usr/bin/env python3
"""A/B of the number of flow slots (LATOK_FLOW_SLOTS) with as many output bitmasks as slots, submissions from Python.
  LATOK_FLOW_SLOTS=3 python3 tools/flow_slots_ab.py [n_str] [model] [iters] [rounds]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib  # noqa: E402


def main():
    n_str = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    model = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    slots = int(os.environ.get("LATOK_FLOW_SLOTS", "2"))
    lo, hi = (64, 192) if model == 0 else (128, 384)
    lib = _lib.ensure_init()
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(0x1A70C0DE + model, 0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    n_words = (total + 63) // 64
    d_row = lib.latok_dev_alloc(row.nbytes)
    d_cps = lib.latok_dev_alloc(total * 4)
    masks = [lib.latok_dev_alloc(n_words * 8) for _ in range(slots)]
    _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
    _lib.check(lib.latok_corpus_fill_device(0x1A70C0DE + model, model, 0, n_str, d_row, d_cps, None))
    _lib.check(lib.latok_sync())
    n8 = C.c_int64(0)
    _lib.check(lib.latok_utf8_bytes(d_cps, total, C.byref(n8), _lib.DEVICE_PTRS))
    for r in range(rounds + 1):
        t = time.perf_counter()
        for i in range(iters):
            _lib.check(lib.latok_flow_split_mask(d_cps, d_row, n_str, total, masks[i % slots]))
        t_sub = time.perf_counter() - t
        _lib.check(lib.latok_flow_wait())
        dt = (time.perf_counter() - t) / iters
        if r:
            print(f"slots {slots}: {dt * 1e6:.2f} us/batch = {n8.value / dt / 1e9:.1f} GB/s (host submit {t_sub / iters * 1e6:.1f} us/batch)", flush=True)


if __name__ == "__main__":
    main()
