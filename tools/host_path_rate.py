#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer API on C2 (1 M ASCII strings, 128 M chars): the batch starts and ends in HOST
memory -- upload, kernels, download of the per-string counts and the records -- through the C ABI's host-pointer calls
(large batches: the chunked three-stream pipeline of api.cpp).  One line per input form x record width x host memory
kind; GB/s = UTF-8 bytes of the corpus / wall time of the call.  Never the `value` of bench.py (that one starts with the
data resident in HBM); reported in DESIGN.md."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from latok_amd import _lib, batch  # noqa: E402

lib = _lib.ensure_init()
n = 1_000_000
row = np.zeros(n + 1, np.int64)
lib.latok_corpus_offsets(0x1A70C0DE, 0, n, 64, 192, row.ctypes.data)
cps = np.zeros(int(row[-1]), np.uint32)
lib.latok_corpus_fill_host(0x1A70C0DE, 0, 0, n, row.ctypes.data, cps.ctypes.data)
total = int(row[-1])
u8 = cps.astype(np.uint8)
utf8_bytes = total  # ASCII corpus


def pinned(a):
    p = batch.pinned_empty(a.shape, a.dtype)
    p[...] = a
    return p


def timed(label, fn, lead, width, dtype, pin, reps=5):
    """the C call alone: output arrays (sized for the result, touched once) are allocated outside the timed region"""
    isz = np.dtype(dtype).itemsize
    alloc = batch.pinned_empty if pin else np.empty
    cap = total // 4
    counts, items = alloc(n, dtype), alloc((cap, width) if width > 1 else cap, dtype)
    counts[...] = 0
    items[...] = 0
    flags = _lib.OUT_INT32 if np.dtype(dtype) == np.dtype(np.int32) else 0
    n_out = C.c_int64(0)
    best = None
    for _ in range(reps + 1):
        t = time.perf_counter()
        _lib.check(fn(*lead, n, total, counts.ctypes.data, items.ctypes.data, cap, C.byref(n_out), flags, None))
        dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best   # (first pass warms the library's buffers up)
    n_items = n_out.value
    print(json.dumps({"path": label, "records": np.dtype(dtype).name, "host_memory": "pinned" if pin else "pageable",
                      "ms_per_batch": best * 1e3, "utf8_GBps": utf8_bytes / best / 1e9, "items": n_items,
                      "bytes_up": int(sum(np.asarray(x).nbytes for x in lead_arrays[label])),
                      "bytes_down": n * isz + n_items * width * isz}), flush=True)


lead_arrays = {}
for pin in (False, True):
    a_cps, a_u8, a_row = (pinned(cps), pinned(u8), pinned(row)) if pin else (cps, u8, row)
    for dtype in (np.int64, np.int32):
        for label, fn, lead, width in (
                ("offsets, UTF-32 in", lib.latok_split_offsets_batch, [a_cps, a_row], 1),
                ("offsets, Latin-1 units in", lib.latok_split_offsets_kind_batch, [a_u8, 1, a_row], 1),
                ("offsets, UTF-8 in (byte space)", lib.latok_split_offsets_utf8_bytes_batch, [a_u8, a_row], 1),
                ("token spans, UTF-8 in (byte space)", lib.latok_token_spans_utf8_bytes_batch, [a_u8, a_row], 2)):
            lead_arrays[label] = [x for x in lead if isinstance(x, np.ndarray)]
            args = [x.ctypes.data if isinstance(x, np.ndarray) else x for x in lead]
            timed(label, fn, args, width, dtype, pin)
# the bitmask only (the north-star output): 1 bit per char down
for pin in (False, True):
    a_u8, a_row = (pinned(u8), pinned(row)) if pin else (u8, row)
    bits = batch.pinned_empty((total + 63) // 64, np.uint64) if pin else np.empty((total + 63) // 64, np.uint64)
    best = None
    for _ in range(6):
        t = time.perf_counter()
        _lib.check(lib.latok_split_mask_utf8_bytes_batch(a_u8.ctypes.data, a_row.ctypes.data, n, total, bits.ctypes.data, 0, None))
        dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    print(json.dumps({"path": "mask, UTF-8 in (byte space)", "host_memory": "pinned" if pin else "pageable",
                      "ms_per_batch": best * 1e3, "utf8_GBps": utf8_bytes / best / 1e9}), flush=True)

# device-resident shards of a DevicePool (latok_amd.multi): the batch is uploaded once per context, every call after that
# uses device pointers -- the rate each context reaches is the device rate, not the bus's.  Contexts here: 1, 2 and 3 on
# device 0 (what a one-GPU box can show; on a node list its devices), records stay in HBM (to_host=False).
from latok_amd import multi  # noqa: E402

for devices in ([0], [0, 0], [0, 0, 0]):
    with multi.DevicePool(devices) as pool:
        for kind, units in (("utf32", cps), ("latin1", u8), ("utf8", u8)):
            with pool.put_csr(units, row, kind=kind) as rb:
                for label, call in (("mask", lambda: pool.split_mask(rb, to_host=False)),
                                    ("offsets int32", lambda: pool.split_offsets(rb, dtype=np.int32, to_host=False))):
                    call()
                    best = None
                    for _ in range(5):
                        t = time.perf_counter()
                        call()
                        dt = time.perf_counter() - t
                        best = dt if best is None or dt < best else best
                    print(json.dumps({"path": f"resident pool, {label}, {kind} in", "contexts": len(devices), "devices": devices,
                                      "ms_per_batch": best * 1e3, "utf8_GBps": utf8_bytes / best / 1e9,
                                      "note": "shards resident in HBM, device-pointer calls, results stay on the device"}), flush=True)
