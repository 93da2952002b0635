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
