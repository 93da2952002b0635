#!/usr/bin/env python3
"""A/B of the batch flow (latok_flow_split_mask: stages of consecutive batches overlapped on three streams) against the
serial pipeline on one resident batch; also checks that both leave the same bitmask.
  python3 tools/flow_ab.py [n_str] [model 0|1] [iters] [rounds]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib  # noqa: E402


def main():
    n_str = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    model = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    lo, hi = (64, 192) if model == 0 else (128, 384)
    lib = _lib.ensure_init()
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(0x1A70C0DE + model, 0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    n_words = (total + 63) // 64
    d_row = lib.latok_dev_alloc(row.nbytes)
    d_cps = lib.latok_dev_alloc(total * 4)
    d_a = lib.latok_dev_alloc(n_words * 8)
    d_b = lib.latok_dev_alloc(n_words * 8)
    d_c = lib.latok_dev_alloc(n_words * 8)
    _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
    _lib.check(lib.latok_corpus_fill_device(0x1A70C0DE + model, model, 0, n_str, d_row, d_cps, None))
    _lib.check(lib.latok_sync())
    n8 = C.c_int64(0)
    _lib.check(lib.latok_utf8_bytes(d_cps, total, C.byref(n8), _lib.DEVICE_PTRS))
    # parity: serial result in c, flow results in a / b
    _lib.check(lib.latok_split_mask_batch(d_cps, d_row, n_str, total, d_c, _lib.DEVICE_PTRS, None))
    _lib.check(lib.latok_sync())
    for buf in (d_a, d_b):
        _lib.check(lib.latok_memset_dev(buf, 0xA5, n_words * 8))
    for i in range(5):
        _lib.check(lib.latok_flow_split_mask(d_cps, d_row, n_str, total, d_a if i % 2 == 0 else d_b))
    _lib.check(lib.latok_flow_wait())
    ref = np.empty(n_words, np.uint64)
    got = np.empty(n_words, np.uint64)
    _lib.check(lib.latok_memcpy_d2h(ref.ctypes.data, d_c, ref.nbytes))
    for name, buf in (("a", d_a), ("b", d_b)):
        _lib.check(lib.latok_memcpy_d2h(got.ctypes.data, buf, got.nbytes))
        same = bool((ref == got).all())
        print(f"flow mask {name} == serial mask: {same}", flush=True)
        if not same:
            sys.exit(1)
    ms, t0, t1 = C.c_float(0), C.c_int64(0), C.c_int64(0)
    for r in range(rounds):
        _lib.check(lib.latok_bench_split_mask_gated(d_cps, d_row, n_str, total, d_c, iters, None, C.byref(ms), C.byref(t0),
                                                    C.byref(t1)))
        s_ev, s_wall = ms.value / iters, (t1.value - t0.value) / 1e6 / iters
        _lib.check(lib.latok_bench_split_mask_flow_gated(d_cps, d_row, n_str, total, d_a, d_b, iters, None, C.byref(ms),
                                                         C.byref(t0), C.byref(t1)))
        f_ev, f_wall = ms.value / iters, (t1.value - t0.value) / 1e6 / iters
        print(f"round {r}: serial {s_ev * 1e3:.2f} us/step (wall {s_wall * 1e3:.2f}) = {n8.value / s_wall / 1e6:.1f} GB/s | "
              f"flow {f_ev * 1e3:.2f} us/step (wall {f_wall * 1e3:.2f}) = {n8.value / f_wall / 1e6:.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
