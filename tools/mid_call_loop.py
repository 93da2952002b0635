#!/usr/bin/env python3
"""Dev tool: N offsets calls on a host batch of <n_str> strings of 105 chars (to look at the launches under rocprofv3)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib, batch
lib = _lib.ensure_init()
text = "This is a #test! Testing, Testing, 1 2 3 -- see http://example.com/x or mail bob@host.org, camelCaseWord."
n_str = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
cps, row = batch.pack([text] * n_str)
total = int(row[-1])
counts = np.zeros(n_str, np.int32); items = np.empty(total, np.int32); n_out = C.c_int64(0)
args = [cps.ctypes.data, row.ctypes.data, n_str, total, counts.ctypes.data, items.ctypes.data, total, C.byref(n_out), _lib.OUT_INT32, None]
lib.latok_split_offsets_batch(*args)
t = time.perf_counter()
for _ in range(n):
    lib.latok_split_offsets_batch(*args)
print(f"{n_str} strings, {total} chars: {(time.perf_counter() - t) / n * 1e6:.1f} us per call")
