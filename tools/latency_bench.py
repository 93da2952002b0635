#!/usr/bin/env python3
"""Dev tool: per-call latency of the single-string entry points (drop-in surface)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd.core import default_tokenizer as dt
from latok_amd import batch
text = "This is a #test! Testing, Testing, 1 2 3 -- see http://example.com/x or mail bob@host.org, camelCaseWord."
list(dt.tokenize(text))
def timeit(name, fn, n=300):
    fn()
    t = time.perf_counter()
    for _ in range(n): fn()
    dt_ = (time.perf_counter() - t) / n
    print(f"{name:40s} {dt_ * 1e6:9.1f} us/call")
timeit("tokenize(text) [fused, batch of one]", lambda: list(dt.tokenize(text)))
timeit("featurize(text)", lambda: list(dt.featurize(text)))
timeit("_gen_parse_matrix(text)", lambda: dt._gen_parse_matrix(text))
m = dt._gen_parse_matrix(text)
timeit("gen_split_mask(m) [compat kernels]", lambda: dt.gen_split_mask(m), 100)
texts = [text] * 1000
timeit("tokenize_batch(1000 strings)", lambda: batch.tokenize_batch(texts), 20)
timeit("token_spans_batch(1000 strings): spans, no per-token str", lambda: batch.token_spans_batch(texts), 20)
# the C call alone (arrays prebuilt): what a C / Cython caller of the ABI pays per string
import ctypes as C
from latok_amd import _lib
lib = _lib.ensure_init()
cps, row = batch.pack([text])
counts = np.zeros(1, np.int64); offs = np.empty(len(text), np.int64); n_out = C.c_int64(0)
args = (cps.ctypes.data, row.ctypes.data, 1, len(text), counts.ctypes.data, offs.ctypes.data, offs.size, C.byref(n_out), 0, None)
timeit("latok_split_offsets_batch, 1 string (C ABI)", lambda: lib.latok_split_offsets_batch(*args), 2000)
timeit("latok_token_spans_batch, 1 string (C ABI)", lambda: lib.latok_token_spans_batch(*args), 2000)
c32 = np.zeros(1, np.int32); o32 = np.empty(2 * len(text), np.int32)
a32 = (cps.ctypes.data, row.ctypes.data, 1, len(text), c32.ctypes.data, o32.ctypes.data, len(text), C.byref(n_out), _lib.OUT_INT32, None)
timeit("latok_token_spans_batch, int32 (C ABI)", lambda: lib.latok_token_spans_batch(*a32), 2000)
timeit("batch.pack([text])", lambda: batch.pack([text]), 2000)
timeit("dt._sync_rules()", lambda: dt._sync_rules(), 2000)
f32 = np.empty((len(text), 25), np.int8); s32 = np.empty(4 * len(text), np.int32)
af = (cps.ctypes.data, row.ctypes.data, 1, len(text), c32.ctypes.data, s32.ctypes.data, f32.ctypes.data, len(text), C.byref(n_out), _lib.OUT_INT32, None)
timeit("latok_token_features_batch, int32 (C ABI)", lambda: lib.latok_token_features_batch(*af), 2000)
u8 = np.frombuffer(text.encode("latin-1"), np.uint8)
ak = (u8.ctypes.data, 1, row.ctypes.data, 1, len(text), c32.ctypes.data, o32.ctypes.data, len(text), C.byref(n_out), _lib.OUT_INT32, None)
timeit("latok_split_offsets_kind_batch, kind 1 (C ABI)", lambda: lib.latok_split_offsets_kind_batch(*ak), 2000)
ab = (u8.ctypes.data, row.ctypes.data, 1, len(text), c32.ctypes.data, o32.ctypes.data, len(text), C.byref(n_out), _lib.OUT_INT32, None)
timeit("latok_split_offsets_utf8_bytes_batch, ASCII (C ABI)", lambda: lib.latok_split_offsets_utf8_bytes_batch(*ab), 2000)
