#!/usr/bin/env python3
"""Full-size runs of BASELINE configs[3] / configs[4] on ONE GPU with the size-independent property checks.

    python tools/full_size_check.py C4            # 100 M strings x 128 chars avg: 51.2 GB of UTF-32 resident
    python tools/full_size_check.py C5            # 10 K documents x 1 M chars: 40 GB resident
    python tools/full_size_check.py C4 --strings 12500000     # one 8-GPU shard of C4

Everything stays on the device (C ABI with LATOK_DEVICE_PTRS); only the bitmask, the per-string counts and SAMPLES of the
code points / offsets travel to the host.  Checks (the ones tests/test_gpu_parity.py::test_full_size_properties makes
at 1 M strings, at the size the config names):
  1. every string start is a boundary, no bit is set beyond the last char
  2. determinism: a second pass gives the same bitmask
  3. compaction: sum(counts) == popcount(mask) == n_offsets; sampled strings' offsets == nonzero of their mask bits
  4. shard consistency (= translation invariance): sampled string ranges, run as batches of their own, give the bit
     range the big batch holds for them -- this is what makes the multi-GPU sharding of the same batch exact
  5. oracle parity on the same samples (oracle/ is the checker here, test infrastructure)
Prints one JSON line; tests/test_gpu_full_size.py runs it inside the GPU test suite.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

SHAPES = {  # name: (model, seed, lo, hi, strings, sample ranges (count, strings per range))
    "C4": (0, 0x1A70C0DE, 64, 192, 100_000_000, (6, 20_000)),
    "C5": (0, 0x1A70C0E0, 1_000_000, 1_000_000, 10_000, (4, 2)),
}


def _bit_at(bits, pos):
    return (bits[pos >> 6] >> (pos & 63).astype(np.uint64)) & np.uint64(1)


def _bit_range(bits, lo, hi):
    """bits [lo, hi) of a little-endian uint64 bitmask as a bool array"""
    w0, w1 = lo >> 6, (hi + 63) >> 6
    flat = np.unpackbits(bits[w0:w1].view(np.uint8), bitorder="little")
    return flat[lo - (w0 << 6):hi - (w0 << 6)].astype(bool)


def run(workload: str, n_str: int = 0, verbose: bool = True) -> dict:
    from latok_amd import _lib, batch
    import latok_oracle as orc
    lib = _lib.ensure_init()
    model, seed, lo, hi, n_default, (n_samples, per_sample) = SHAPES[workload]
    n_str = n_str or n_default
    t_all = time.perf_counter()

    def say(msg):
        if verbose:
            print(f"[{time.perf_counter() - t_all:7.1f}s] {msg}", file=sys.stderr, flush=True)

    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    words = (total + 63) // 64
    say(f"{workload}: {n_str} strings, {total} chars = {total * 4 / 1e9:.1f} GB of UTF-32")
    dev = {}

    def alloc(name, nbytes):
        p = lib.latok_dev_alloc(max(int(nbytes), 16))
        if not p:
            raise MemoryError(_lib.last_error())
        dev[name] = p
        return p

    try:
        d_row, d_cps = alloc("row", row.nbytes), alloc("cps", total * 4)
        d_bits, d_bits2 = alloc("bits", words * 8), alloc("bits2", words * 8)
        _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
        _lib.check(lib.latok_corpus_fill_device(seed, model, 0, n_str, d_row, d_cps, None))
        _lib.check(lib.latok_sync())
        say("corpus resident")
        # ---- the mask, twice ---------------------------------------------------------------------------------------
        ms = C.c_float(0)
        _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 1, 3, C.byref(ms), None, None))
        _lib.check(lib.latok_split_mask_batch(d_cps, d_row, n_str, total, d_bits2, _lib.DEVICE_PTRS, None))
        _lib.check(lib.latok_sync())
        bits = np.empty(words, np.uint64)
        bits2 = np.empty(words, np.uint64)
        _lib.check(lib.latok_memcpy_d2h(bits.ctypes.data, d_bits, bits.nbytes))
        _lib.check(lib.latok_memcpy_d2h(bits2.ctypes.data, d_bits2, bits2.nbytes))
        say(f"mask: {ms.value / 3:.3f} ms per pass")
        assert np.array_equal(bits, bits2), "the second pass gave a different bitmask"
        del bits2
        # (1)
        assert (_bit_at(bits, row[:-1][row[:-1] < total]) == 1).all(), "a string start is not a boundary"
        if total & 63:
            assert int(bits[-1]) >> (total & 63) == 0, "bits beyond the last char"
        n_bound = int(np.bitwise_count(bits).sum(dtype=np.int64))
        say(f"{n_bound} boundaries")
        # (3) compaction on the device
        d_counts = alloc("counts", n_str * 8)
        d_offs = alloc("offs", n_bound * 8)
        n_off = C.c_int64(0)
        t = time.perf_counter()
        _lib.check(lib.latok_split_offsets_batch(d_cps, d_row, n_str, total, d_counts, d_offs, n_bound, C.byref(n_off),
                                                 _lib.DEVICE_PTRS, None))
        _lib.check(lib.latok_sync())
        t_offsets = time.perf_counter() - t
        counts = np.empty(n_str, np.int64)
        _lib.check(lib.latok_memcpy_d2h(counts.ctypes.data, d_counts, counts.nbytes))
        assert n_off.value == n_bound == int(counts.sum()), (n_off.value, n_bound, int(counts.sum()))
        starts = np.zeros(n_str + 1, np.int64)
        np.cumsum(counts, out=starts[1:])
        say(f"offsets: {t_offsets * 1e3:.1f} ms (blocking call), {n_off.value} offsets")
        # token spans: only the count (every kept token starts at a boundary)
        lib.latok_dev_free(dev.pop("offs"))
        lib.latok_dev_free(dev.pop("bits2"))
        n_tok = C.c_int64(0)
        d_spans = alloc("spans", n_bound * 16)
        _lib.check(lib.latok_token_spans_batch(d_cps, d_row, n_str, total, d_counts, d_spans, n_bound, C.byref(n_tok),
                                               _lib.DEVICE_PTRS, None))
        _lib.check(lib.latok_sync())
        tcounts = np.empty(n_str, np.int64)
        _lib.check(lib.latok_memcpy_d2h(tcounts.ctypes.data, d_counts, tcounts.nbytes))
        assert 0 < n_tok.value <= n_bound and int(tcounts.sum()) == n_tok.value and (tcounts <= counts).all()
        tstarts = np.zeros(n_str + 1, np.int64)
        np.cumsum(tcounts, out=tstarts[1:])
        say(f"{n_tok.value} tokens")
        # re-run offsets for the sample comparison (the spans call reused the counts buffer)
        lib.latok_dev_free(dev.pop("spans"))
        d_offs = alloc("offs", n_bound * 8)
        _lib.check(lib.latok_split_offsets_batch(d_cps, d_row, n_str, total, d_counts, d_offs, n_bound, C.byref(n_off),
                                                 _lib.DEVICE_PTRS, None))
        # (4) + (5) samples: head, tail and evenly spaced ranges
        rng = np.random.default_rng(12345)
        firsts = [0, n_str - per_sample] + [int(x) for x in rng.integers(0, n_str - per_sample, max(0, n_samples - 2))]
        sampled_chars = 0
        for s0 in firsts:
            s1 = s0 + per_sample
            c0, c1 = int(row[s0]), int(row[s1])
            sub = np.empty(c1 - c0, np.uint32)
            _lib.check(lib.latok_memcpy_d2h(sub.ctypes.data, d_cps + c0 * 4, sub.nbytes))
            sub_row = row[s0:s1 + 1] - c0
            want = _bit_range(bits, c0, c1)
            own = batch.split_mask_batch(sub, sub_row)                       # the range as a batch of its own
            assert np.array_equal(_bit_range(own, 0, c1 - c0), want), f"shard of strings [{s0}, {s1}) differs from the big batch"
            ov, _ = orc.split_batch(sub, sub_row, want_bits=False)           # oracle
            assert np.array_equal(ov != 0, want), f"oracle parity fails in strings [{s0}, {s1})"
            k0, k1 = int(starts[s0]), int(starts[s1])
            offs = np.empty(k1 - k0, np.int64)
            _lib.check(lib.latok_memcpy_d2h(offs.ctypes.data, d_offs + k0 * 8, offs.nbytes))
            glob = offs + np.repeat(row[s0:s1], counts[s0:s1])
            assert np.array_equal(glob - c0, np.nonzero(want)[0]), f"offsets of strings [{s0}, {s1}) differ from the mask"
            sampled_chars += c1 - c0
        say(f"{len(firsts)} sampled ranges ({sampled_chars} chars): shard-consistent, oracle-exact, offsets == mask")
        res = {"workload": workload, "strings": n_str, "chars": total, "utf32_GB": total * 4 / 1e9,
               "mask_ms_per_pass": ms.value / 3, "boundaries": n_bound, "tokens": n_tok.value,
               "offsets_call_ms": t_offsets * 1e3, "sampled_ranges": len(firsts), "sampled_chars": sampled_chars,
               "checks": ["string starts are boundaries", "deterministic", "sum(counts) == popcount(mask) == n_offsets",
                          "sampled shards == big batch", "sampled oracle parity", "sampled offsets == mask"],
               "seconds": time.perf_counter() - t_all, "ok": True}
        return res
    finally:
        for p in dev.values():
            lib.latok_dev_free(p)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(SHAPES))
    ap.add_argument("--strings", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(run(a.workload, a.strings)), flush=True)
