#!/usr/bin/env python3
"""Dev / measurement tool: latency of the host-pointer C ABI calls against the batch size (strings of ~105 chars)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib, batch
lib = _lib.ensure_init()
text = "This is a #test! Testing, Testing, 1 2 3 -- see http://example.com/x or mail bob@host.org, camelCaseWord."
print(f"{'strings':>8s} {'chars':>9s} {'offsets us':>11s} {'spans us':>10s} {'features us':>12s} {'ns/char (offsets)':>18s}")
for n_str in (1, 4, 16, 39, 40, 64, 156, 157, 256, 512, 1024, 4096, 16384, 65536):
    cps, row = batch.pack([text] * n_str)
    total = int(row[-1])
    counts = np.zeros(n_str, np.int32); items = np.empty(4 * total, np.int32); feats = np.empty((total, 25), np.int8); n_out = C.c_int64(0)
    def call(fn, with_feats=False):
        args = [cps.ctypes.data, row.ctypes.data, n_str, total, counts.ctypes.data, items.ctypes.data]
        if with_feats: args.append(feats.ctypes.data)
        args += [total, C.byref(n_out), _lib.OUT_INT32, None]
        for _ in range(3):      # (the runtime's pageable-copy staging grows on the second call at a new size: ~8 ms once)
            _lib.check(fn(*args))
        reps = max(5, min(2000, int(2e5 / max(total, 100))))
        t = time.perf_counter()
        for _ in range(reps): fn(*args)
        return (time.perf_counter() - t) / reps * 1e6
    o = call(lib.latok_split_offsets_batch); s = call(lib.latok_token_spans_batch); f = call(lib.latok_token_features_batch, True)
    print(f"{n_str:8d} {total:9d} {o:11.1f} {s:10.1f} {f:12.1f} {o * 1e3 / total:18.2f}")
