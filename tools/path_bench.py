#!/usr/bin/env python3
"""Measurement of the SURVEY 8(f) rows on device-resident data: one JSON line per path.

  mask            latok_split_mask_batch            (the north-star path; here for reference, bench.py is the contract)
  offsets         latok_split_offsets_batch         (8f-1: np.nonzero per string)
  spans           latok_token_spans_batch           (8f-1: slice / strip / drop-empty)
  features        latok_token_features_batch        (8f-2: featurize)
  utf8_mask       latok_split_mask_utf8_batch       (8f-3: UTF-8 in, code-point row offsets + mask out)
  utf8_offsets    latok_split_offsets_utf8_batch
  utf8_spans      latok_token_spans_utf8_batch
  bytes_mask / bytes_offsets / bytes_spans   latok_*_utf8_bytes_batch (8f-3 fused: the tile kernel reads the bytes)
  rules_mask      latok_split_mask_batch after latok_set_rules(built-in tables)   (8f-4)
  offsets32 / spans32 / features32 / kind_offsets32 / kind_spans32   the same entry points with LATOK_OUT_INT32 records
  mask_flow / bytes_mask_flow / kind_mask_flow   the same mask paths through the batch flow (latok_flow_split_mask*: two
                  batches in flight, the small launches of one in the shadow of the other's tile kernel)
  kind_mask / kind_offsets / kind_spans      latok_*_kind_batch on PEP 393 code units: kind 1 (uint8) when every char
                  of the corpus is <= U+00FF (C2), else kind 2 (uint16) with the corpus' astral chars folded into the BMP
                  (cp & 0xFFFF; timing only -- C3 as CPython would store it without its emoji)

Every line carries: ms per call (wall clock around `--iters` blocking calls, inputs and outputs in HBM; the
compaction calls contain one 8-byte blocking read of the item total), the UTF-8 GB/s of the corpus through that path,
the path's ALGORITHMIC bytes (inputs that must be read + outputs that must be written, stated per line) and their rate
as a fraction of the 8 TB/s HBM peak.  `--cpu N` adds the reference's own C functions + its Python glue (oracle/_ref,
test infrastructure, timed here as the baseline only) on the first N strings for offsets and tokens.

usage: tools/path_bench.py [--workload C2|C3] [--strings N] [--iters K] [--cpu N]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latok_amd import _lib  # noqa: E402

HBM_PEAK = 8000.0
WORKLOADS = {"C2": (_lib.CORPUS_ASCII, 0x1A70C0DE, 64, 192), "C3": (_lib.CORPUS_UNICODE, 0x1A70C0DF, 128, 384)}


def utf8_of(cps, row):
    """packed UTF-32 -> (utf8 bytes uint8[], byte offsets int64[n+1]) on the host (setup, not timed)."""
    lens = 1 + (cps >= 0x80).astype(np.int64) + (cps >= 0x800) + (cps >= 0x10000)
    pref = np.zeros(cps.size + 1, np.int64)
    np.cumsum(lens, out=pref[1:])
    boff = pref[row]
    if int(pref[-1]) == cps.size:
        u8 = cps.astype(np.uint8)
    else:
        u8 = np.frombuffer(cps.astype("<u4").tobytes().decode("utf-32-le", "surrogatepass").encode("utf-8", "surrogatepass"),
                           np.uint8)
    assert u8.size == int(pref[-1])
    return np.ascontiguousarray(u8), np.ascontiguousarray(boff)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--strings", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu", type=int, default=0, help="strings of CPU baseline (0 = skip)")
    ap.add_argument("--paths", default="mask,mask_flow,bytes_mask_flow,kind_mask_flow,offsets32_flow,spans32_flow,features32_flow,offsets,offsets32,spans,spans32,features,features32,utf8_mask,utf8_offsets,utf8_spans,"
                                       "bytes_mask,bytes_offsets,bytes_spans,rules_mask,kind_mask,kind_offsets,kind_offsets32,"
                                       "kind_spans,kind_spans32")
    args = ap.parse_args()
    lib = _lib.ensure_init()
    model, seed, lo, hi = WORKLOADS[args.workload]
    n = args.strings
    row = np.zeros(n + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n, lo, hi, row.ctypes.data))
    total = int(row[-1])
    words = (total + 63) // 64
    d_row = lib.latok_dev_alloc(row.nbytes)
    d_cps = lib.latok_dev_alloc(total * 4)
    _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
    _lib.check(lib.latok_corpus_fill_device(seed, model, 0, n, d_row, d_cps, None))
    cps = np.empty(total, np.uint32)
    _lib.check(lib.latok_memcpy_d2h(cps.ctypes.data, d_cps, total * 4))
    u8, boff = utf8_of(cps, row)
    n8 = int(u8.size)
    d_u8 = lib.latok_dev_alloc(n8 + 64)
    d_boff = lib.latok_dev_alloc(boff.nbytes)
    _lib.check(lib.latok_memcpy_h2d(d_u8, u8.ctypes.data, n8))
    _lib.check(lib.latok_memcpy_h2d(d_boff, boff.ctypes.data, boff.nbytes))
    cap = total // 2 + 4096
    d_bits = lib.latok_dev_alloc(words * 8 + 8)
    d_counts = lib.latok_dev_alloc(n * 8)
    d_items = lib.latok_dev_alloc(cap * 32)
    d_feat = lib.latok_dev_alloc(cap * 25)
    d_cprow = lib.latok_dev_alloc((n + 1) * 8)
    for p in (d_row, d_cps, d_u8, d_boff, d_bits, d_counts, d_items, d_feat, d_cprow):
        if not p:
            raise RuntimeError(_lib.last_error())
    nout, tcp = C.c_int64(0), C.c_int64(0)
    D = _lib.DEVICE_PTRS
    csr = 8 * (n + 1)

    def run(name, fn, alg, note):
        _lib.check(fn())
        _lib.check(lib.latok_sync())
        t = time.perf_counter()
        for _ in range(args.iters):
            _lib.check(fn())
        _lib.check(lib.latok_sync())
        dt = (time.perf_counter() - t) / args.iters
        a = alg()
        print(json.dumps({"path": name, "workload": args.workload, "strings": n, "chars": total, "utf8_bytes": n8,
                          "items": nout.value, "ms_per_call": dt * 1e3, "utf8_GBps": n8 / dt / 1e9,
                          "alg_bytes": a, "alg_GBps": a / dt / 1e9, "frac_of_hbm_peak": a / dt / 1e9 / HBM_PEAK,
                          "alg_bytes_are": note}), flush=True)

    paths = args.paths.split(",")
    flow_buf = {}

    def flow_pair(n_words):   # two output bitmasks: consecutive submissions of a flow alternate between them
        if n_words not in flow_buf:
            flow_buf[n_words] = (lib.latok_dev_alloc(n_words * 8 + 8), lib.latok_dev_alloc(n_words * 8 + 8))
        return flow_buf[n_words]

    def run_flow(name, submit, alg, note):
        """the same measurement through the batch flow (include/latok_hip.h): `--iters` submissions, then one latok_flow_wait"""
        a_, b_ = None, None
        for i in range(4):
            _lib.check(submit(i))
        _lib.check(lib.latok_flow_wait())
        k = max(args.iters, 100)   # long enough that filling and draining the two-batch pipeline is noise
        t = time.perf_counter()
        for i in range(k):
            _lib.check(submit(i))
        _lib.check(lib.latok_flow_wait())
        dt = (time.perf_counter() - t) / k
        a = alg()
        print(json.dumps({"path": name, "workload": args.workload, "strings": n, "chars": total, "utf8_bytes": n8,
                          "items": 0, "ms_per_call": dt * 1e3, "utf8_GBps": n8 / dt / 1e9,
                          "alg_bytes": a, "alg_GBps": a / dt / 1e9, "frac_of_hbm_peak": a / dt / 1e9 / HBM_PEAK,
                          "alg_bytes_are": note + "; batch flow: two batches in flight, per-batch time of " + str(k) + " submissions + one wait"}),
              flush=True)

    if "mask_flow" in paths:
        fa, fb = flow_pair(words)
        run_flow("mask_flow", lambda i: lib.latok_flow_split_mask(d_cps, d_row, n, total, fb if i & 1 else fa),
                 lambda: 4 * total + csr + words * 8, "4 B/char + 8 B/string read, 1 bit/char written")
    if "bytes_mask_flow" in paths:
        bw_ = (n8 + 63) // 64
        fa, fb = flow_pair(bw_)
        run_flow("bytes_mask_flow", lambda i: lib.latok_flow_split_mask_utf8_bytes(d_u8, d_boff, n, n8, fb if i & 1 else fa),
                 lambda: n8 + csr + bw_ * 8, "UTF-8 bytes + 8 B/string read; 1 bit/byte written (byte space, fused ingest)")
    d_res = lib.latok_dev_alloc(64)
    for name, spans, width in (("offsets32_flow", False, 4), ("spans32_flow", True, 8)):
        if name in paths:
            if "items2" not in flow_buf:
                flow_buf["items2"] = (lib.latok_dev_alloc(cap * 32), lib.latok_dev_alloc(n * 8))
            d_items2, d_counts2 = flow_buf["items2"]
            fn = lib.latok_flow_token_spans if spans else lib.latok_flow_split_offsets
            run("offsets32" if not spans else "spans32", (lambda s_=spans: (lib.latok_token_spans_batch if s_ else lib.latok_split_offsets_batch)(
                d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D | _lib.OUT_INT32, None)), lambda: 0, "(item count for the flow line)")
            items_n = nout.value
            run_flow(name, lambda i, fn=fn: fn(d_cps, 4, d_row, n, total, d_counts2 if i & 1 else d_counts, d_items2 if i & 1 else d_items, cap,
                                               C.c_void_p(d_res + 16 * (i & 1)), _lib.OUT_INT32),
                     lambda: 4 * total + csr + 4 * n + width * items_n,
                     f"4 B/char + 8 B/string read; 4 B/string counts + {width} B/item written (LATOK_OUT_INT32)")
    if "features32_flow" in paths:
        if "feat2" not in flow_buf:
            flow_buf["feat2"] = (lib.latok_dev_alloc(cap * 32), lib.latok_dev_alloc(n * 8), lib.latok_dev_alloc(cap * 25))
        d_items3, d_counts3, d_feat3 = flow_buf["feat2"]
        run("features32", lambda: lib.latok_token_features_batch(d_cps, d_row, n, total, d_counts, d_items, d_feat, cap, C.byref(nout), D | _lib.OUT_INT32, None),
            lambda: 0, "(item count for the flow line)")
        items_f = nout.value
        run_flow("features32_flow", lambda i: lib.latok_flow_token_features(d_cps, 4, d_row, n, total, d_counts3 if i & 1 else d_counts,
                                                                            d_items3 if i & 1 else d_items, d_feat3 if i & 1 else d_feat, cap,
                                                                            C.c_void_p(d_res + 16 * (i & 1)), _lib.OUT_INT32),
                 lambda: 4 * total + csr + 4 * n + (16 + 25) * items_f,
                 "SURVEY 8f-2 accounting: 4 B/char + 8 B/string read (the input ONCE); 4 B/string + 41 B/token written (LATOK_OUT_INT32)")
    if "mask" in paths:
        run("mask", lambda: lib.latok_split_mask_batch(d_cps, d_row, n, total, d_bits, D, None),
            lambda: 4 * total + csr + words * 8, "4 B/char + 8 B/string read, 1 bit/char written")
    if "offsets" in paths:
        run("offsets", lambda: lib.latok_split_offsets_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None),
            lambda: 4 * total + csr + 8 * n + 8 * nout.value, "4 B/char + 8 B/string read; 8 B/string counts + 8 B/boundary written")
    if "spans" in paths:
        run("spans", lambda: lib.latok_token_spans_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None),
            lambda: 4 * total + csr + 8 * n + 16 * nout.value, "4 B/char + 8 B/string read; 8 B/string counts + 16 B/token written")
    D32 = D | _lib.OUT_INT32
    if "offsets32" in paths:
        run("offsets32", lambda: lib.latok_split_offsets_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D32, None),
            lambda: 4 * total + csr + 4 * n + 4 * nout.value, "4 B/char + 8 B/string read; 4 B/string counts + 4 B/boundary written (LATOK_OUT_INT32)")
    if "spans32" in paths:
        run("spans32", lambda: lib.latok_token_spans_batch(d_cps, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D32, None),
            lambda: 4 * total + csr + 4 * n + 8 * nout.value, "4 B/char + 8 B/string read; 4 B/string counts + 8 B/token written (LATOK_OUT_INT32)")
    if "features32" in paths:
        run("features32", lambda: lib.latok_token_features_batch(d_cps, d_row, n, total, d_counts, d_items, d_feat, cap, C.byref(nout), D32, None),
            lambda: 4 * total + csr + 4 * n + (16 + 25) * nout.value,
            "SURVEY 8f-2 accounting: 4 B/char + 8 B/string read (the input ONCE); 4 B/string + 41 B/token written (LATOK_OUT_INT32)")
    if "features" in paths:
        run("features", lambda: lib.latok_token_features_batch(d_cps, d_row, n, total, d_counts, d_items, d_feat, cap, C.byref(nout), D, None),
            lambda: 4 * total + csr + 8 * n + (32 + 25) * nout.value,
            "SURVEY 8f-2 accounting: 4 B/char + 8 B/string read (the input ONCE); 8 B/string + 57 B/token written")
    if "utf8_mask" in paths:
        nout.value = 0
        run("utf8_mask", lambda: lib.latok_split_mask_utf8_batch(d_u8, d_boff, n, n8, d_bits, words + 1, d_cprow, C.byref(tcp), D, None),
            lambda: n8 + csr + words * 8 + csr, "UTF-8 bytes + 8 B/string read; 1 bit/char + 8 B/string cp offsets written")
        assert tcp.value == total
    if "utf8_offsets" in paths:
        run("utf8_offsets", lambda: lib.latok_split_offsets_utf8_batch(d_u8, d_boff, n, n8, d_counts, d_items, cap, C.byref(nout), D, None),
            lambda: n8 + csr + 8 * n + 8 * nout.value, "UTF-8 bytes + 8 B/string read; 8 B/string + 8 B/boundary written")
    if "utf8_spans" in paths:
        run("utf8_spans", lambda: lib.latok_token_spans_utf8_batch(d_u8, d_boff, n, n8, d_counts, d_items, cap, C.byref(nout), D, None),
            lambda: n8 + csr + 8 * n + 16 * nout.value, "UTF-8 bytes + 8 B/string read; 8 B/string + 16 B/token written")
    bwords = (n8 + 63) // 64
    if "bytes_mask" in paths:
        d_bbits = lib.latok_dev_alloc(bwords * 8 + 8)
        nout.value = 0
        run("bytes_mask", lambda: lib.latok_split_mask_utf8_bytes_batch(d_u8, d_boff, n, n8, d_bbits, D, None),
            lambda: n8 + csr + bwords * 8, "UTF-8 bytes + 8 B/string read; 1 bit/byte written (byte space, fused ingest)")
    if "bytes_offsets" in paths:
        run("bytes_offsets", lambda: lib.latok_split_offsets_utf8_bytes_batch(d_u8, d_boff, n, n8, d_counts, d_items, cap, C.byref(nout), D, None),
            lambda: n8 + csr + 8 * n + 8 * nout.value, "UTF-8 bytes + 8 B/string read; 8 B/string + 8 B/boundary written (byte offsets)")
    if "bytes_spans" in paths:
        run("bytes_spans", lambda: lib.latok_token_spans_utf8_bytes_batch(d_u8, d_boff, n, n8, d_counts, d_items, cap, C.byref(nout), D, None),
            lambda: n8 + csr + 8 * n + 16 * nout.value, "UTF-8 bytes + 8 B/string read; 8 B/string + 16 B/token written (byte ranges)")
    if "rules_mask" in paths:
        from latok_amd import batch
        from latok_amd.core import default_tokenizer as dt
        batch.set_rules(dt.C_SPLIT, dt.C_MASK, dt.C_SYM)
        nout.value = 0
        try:
            run("rules_mask", lambda: lib.latok_split_mask_batch(d_cps, d_row, n, total, d_bits, D, None),
                lambda: 4 * total + csr + words * 8, "as mask; tables interpreted at run time")
        finally:
            batch.reset_rules()

    if any(p.startswith("kind_") for p in paths):
        kind = 1 if int(cps.max()) < 256 else 2
        units = cps.astype(np.uint8) if kind == 1 else (cps & 0xFFFF).astype(np.uint16)
        d_units = lib.latok_dev_alloc(units.nbytes + 64)
        _lib.check(lib.latok_memcpy_h2d(d_units, units.ctypes.data, units.nbytes))
        note = f"{kind} B/char + 8 B/string read"
        if "kind_mask_flow" in paths:
            fa, fb = flow_pair(words)
            run_flow("kind_mask_flow", lambda i: lib.latok_flow_split_mask_kind(d_units, kind, d_row, n, total, fb if i & 1 else fa),
                     lambda: kind * total + csr + words * 8, note + "; 1 bit/char written")
        if "kind_mask" in paths:
            nout.value = 0
            run("kind_mask", lambda: lib.latok_split_mask_kind_batch(d_units, kind, d_row, n, total, d_bits, D, None),
                lambda: kind * total + csr + words * 8, note + "; 1 bit/char written")
        if "kind_offsets" in paths:
            run("kind_offsets", lambda: lib.latok_split_offsets_kind_batch(d_units, kind, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None),
                lambda: kind * total + csr + 8 * n + 8 * nout.value, note + "; 8 B/string counts + 8 B/boundary written")
        if "kind_offsets32" in paths:
            run("kind_offsets32", lambda: lib.latok_split_offsets_kind_batch(d_units, kind, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D32, None),
                lambda: kind * total + csr + 4 * n + 4 * nout.value, note + "; 4 B/string counts + 4 B/boundary written (LATOK_OUT_INT32)")
        if "kind_spans32" in paths:
            run("kind_spans32", lambda: lib.latok_token_spans_kind_batch(d_units, kind, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D32, None),
                lambda: kind * total + csr + 4 * n + 8 * nout.value, note + "; 4 B/string counts + 8 B/token written (LATOK_OUT_INT32)")
        if "kind_spans" in paths:
            run("kind_spans", lambda: lib.latok_token_spans_kind_batch(d_units, kind, d_row, n, total, d_counts, d_items, cap, C.byref(nout), D, None),
                lambda: kind * total + csr + 8 * n + 16 * nout.value, note + "; 8 B/string counts + 16 B/token written")

    if args.cpu > 0:   # baseline only: the reference's own C (oracle/_ref) under its restated glue, one string at a time
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import latok_oracle as orc
        m = min(args.cpu, n)
        text = cps[:row[m]].astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")
        strs = [text[row[i]:row[i + 1]] for i in range(m)]
        b8 = int(boff[m])
        try:
            glue = orc.RefGlue()
            kind = "reference"
            offsets = lambda s: np.nonzero(glue.split_values(s))[0]  # noqa: E731
        except Exception:
            kind = "port"
            offsets = orc.split_offsets
        t = time.perf_counter()
        for s in strs:
            offsets(s)
        dt_off = time.perf_counter() - t
        t = time.perf_counter()
        for s in strs:
            nz = offsets(s)
            a, b, toks = int(nz[0]), 0, []
            for b in nz[1:]:
                b = int(b)
                w = s[a:b].strip()
                if w:
                    toks.append(w)
                a = b
            w = s[b:].strip()
            if w:
                toks.append(w)
        dt_tok = time.perf_counter() - t
        for name, dt_ in (("offsets", dt_off), ("tokens", dt_tok)):
            print(json.dumps({"cpu_baseline": name, "kind": kind, "cores": 1, "strings": m, "utf8_bytes": b8,
                              "seconds": dt_, "utf8_GBps": b8 / dt_ / 1e9}), flush=True)


if __name__ == "__main__":
    main()
