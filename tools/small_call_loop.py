#!/usr/bin/env python3
"""Dev tool: N single-string calls of the C ABI in a loop (to look at k_small_batch under rocprofv3)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib, batch
lib = _lib.ensure_init()
text = "This is a #test! Testing, Testing, 1 2 3 -- see http://example.com/x or mail bob@host.org, camelCaseWord."
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cps, row = batch.pack([text])
counts = np.zeros(1, np.int64); offs = np.empty(len(text), np.int64); n_out = C.c_int64(0)
args = (cps.ctypes.data, row.ctypes.data, 1, len(text), counts.ctypes.data, offs.ctypes.data, offs.size, C.byref(n_out), 0, None)
fn = lib.latok_split_offsets_batch
if len(sys.argv) > 2 and sys.argv[2] == "features":
    spans4 = np.empty(4 * len(text), np.int64); feats = np.empty((len(text), 25), np.int8)
    args = (cps.ctypes.data, row.ctypes.data, 1, len(text), counts.ctypes.data, spans4.ctypes.data, feats.ctypes.data, len(text),
            C.byref(n_out), 0, None)
    fn = lib.latok_token_features_batch
fn(*args)
t = time.perf_counter()
for _ in range(n):
    fn(*args)
print(f"{(time.perf_counter() - t) / n * 1e6:.1f} us per call, {n_out.value} items")
