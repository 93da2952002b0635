#!/usr/bin/env python3
"""First-light timing of the fused path on a device-generated corpus (not the contract bench; see bench.py)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib  # noqa: E402


def main():
    n_str = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    model = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lo, hi = (64, 192) if model == 0 else (128, 384)
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lib = _lib.ensure_init()
    row = np.zeros(n_str + 1, np.int64)
    t = time.time()
    _lib.check(lib.latok_corpus_offsets(0x1A70C0DE + model, 0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    print(f"n_str={n_str} total_chars={total} offsets_host_s={time.time() - t:.2f}", flush=True)
    d_row = lib.latok_dev_alloc(row.nbytes)
    d_cps = lib.latok_dev_alloc(total * 4)
    d_bits = lib.latok_dev_alloc(((total + 63) // 64) * 8)
    _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
    t = time.time()
    _lib.check(lib.latok_corpus_fill_device(0x1A70C0DE + model, model, 0, n_str, d_row, d_cps, None))
    _lib.check(lib.latok_sync())
    print(f"device fill s={time.time() - t:.3f}", flush=True)
    n8 = C.c_int64(0)
    _lib.check(lib.latok_utf8_bytes(d_cps, total, C.byref(n8), _lib.DEVICE_PTRS))
    if len(sys.argv) > 4 and sys.argv[4] == "rules":   # the built-in tables, but interpreted at run time (kModeRules)
        from latok_amd import batch
        from latok_amd.core import default_tokenizer as dt
        batch.set_rules(dt.C_SPLIT, dt.C_MASK, dt.C_SYM)
        print("runtime rule tables installed", flush=True)
    ms_total, ms_tiles, nfix = C.c_float(0), C.c_float(0), C.c_int64(0)
    _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 3, iters, C.byref(ms_total),
                                          C.byref(ms_tiles), C.byref(nfix)))
    per = ms_total.value / iters
    per_tiles = ms_tiles.value / iters
    alg = 4 * total + 8 * (n_str + 1)
    print(f"pipeline {per * 1e3:.1f} us/pass  tiles-kernel {per_tiles * 1e3:.1f} us  fix_tiles={nfix.value} "
          f"of {(total + 4095) // 4096}")
    print(f"utf8 GB/s={n8.value / per / 1e6:.1f}  alg_read TB/s pipeline={alg / per / 1e9:.3f} "
          f"tiles-kernel={alg / per_tiles / 1e9:.3f}  frac_of_8TB/s={alg / per_tiles / 1e9 / 8.0:.3f}")


if __name__ == "__main__":
    main()
