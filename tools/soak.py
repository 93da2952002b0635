#!/usr/bin/env python3
"""Randomised differential soak of the HIP path against the oracle (test infrastructure; not part of pytest).

Worker processes build random batches from adversarial alphabets in several shape regimes (many tiny strings, tweets,
multi-tile documents, no-whitespace documents with starts, dense starts) and compute the oracle's split values / bitmask on
the CPU; the main process runs the same batch through the C ABI (values, bitmask, offsets, token spans, the bitmask
under run-time rule tables, featurize sums, the UTF-8 entry points in byte space and in code-point units, and the PEP 393 kind-1 / kind-2 entry points) and compares bit for bit.  Stops after --seconds.

usage: tools/soak.py [--seconds 120] [--workers 12] [--seed 1]
"""
import argparse
import multiprocessing as mp
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

REGIMES = [
    # (alphabet, n_strings range, length range)
    ("mixed", (1, 3000), (0, 12)), ("mixed", (1, 400), (0, 200)), ("words", (1, 300), (0, 400)),
    ("starts", (1, 200), (0, 600)), ("mixed", (1, 4), (3000, 30000)), ("nospace_at", (1, 4), (4000, 40000)),
    ("rare_space_at", (1, 4), (4000, 40000)), ("starts", (500, 4000), (1, 3)), ("words", (1, 6), (8000, 20000)),
    ("latin1", (1, 400), (0, 200)), ("bmp", (1, 400), (0, 200)), ("latin1", (1, 4), (3000, 30000)), ("bmp", (1, 4), (3000, 30000)),
    ("bmp", (500, 3000), (1, 3)),
    # at most one tile: the single-launch path of small host batches (k_small_batch: offsets, spans, featurize sums)
    ("mixed", (1, 40), (0, 100)), ("words", (1, 8), (0, 500)), ("bmp", (1, 30), (0, 130)), ("starts", (1, 500), (0, 8)),
    ("rare_space_at", (1, 2), (1000, 2040)),
    # beyond the small-batch size in UTF-8 bytes: code-point results from byte space (no UTF-32 copy), the chunk-parallel paths
    ("mixed", (2500, 6000), (0, 200)), ("bmp", (1500, 4000), (0, 300)), ("words", (3, 8), (60000, 120000)),
]


def make_batch(seed):
    import latok_oracle as orc
    from conftest import ALPHABETS, DEFAULT_RULES, RULE_SETS, pack, random_rule_tables, random_strings
    rng = random.Random(seed)
    kind, (n_lo, n_hi), (l_lo, l_hi) = rng.choice(REGIMES)
    texts = random_strings(rng, rng.randint(n_lo, n_hi), l_lo, l_hi, ALPHABETS[kind])
    if rng.random() < 0.3:   # shift every tile boundary by an odd prefix
        texts.insert(0, "x" * rng.randint(1, 5000))
    cps, row = pack(texts)
    vals, bits = orc.split_batch(cps, row)
    # space flags for the span check
    uniq, inv = np.unique(cps, return_inverse=True)
    space = np.array([(orc.base_word(int(c)) >> 5) & 1 for c in uniq], bool)[inv] if cps.size else np.zeros(0, bool)
    rules = None
    rule_bits = None
    if rng.random() < 0.25 and cps.size < 60000:
        name = rng.choice(sorted(RULE_SETS) + ["random"])
        rules = random_rule_tables(rng) if name == "random" else RULE_SETS[name]
        flags = np.zeros(cps.size, bool)
        k = 0
        for t in texts:
            if len(t):
                flags[k:k + len(t)] = orc.split_values_rules(t, *rules) != 0
            k += len(t)
        rule_bits = np.packbits(np.concatenate([flags, np.zeros((-cps.size) % 64, bool)]), bitorder="little").view(np.uint64)
    # UTF-8 view of the batch: bytes, byte offsets, byte position of every code point (multi-byte material is added to
    # some batches by replacing a few ASCII letters)
    u8 = boff = bpos = None
    if rng.random() < 0.5 and cps.size < 200000:
        lens = 1 + (cps >= 0x80).astype(np.int64) + (cps >= 0x800) + (cps >= 0x10000)
        pref = np.zeros(cps.size + 1, np.int64)
        np.cumsum(lens, out=pref[1:])
        u8 = np.frombuffer("".join(texts).encode("utf-8", "surrogatepass"), np.uint8)
        assert u8.size == int(pref[-1])
        boff = np.ascontiguousarray(pref[row])
        bpos = pref[:-1]
    # featurize: per-token column sums from the oracle's n x 25 matrix (small batches only)
    feats = None
    if cps.size < 6000 and rng.random() < 0.5:
        rows_, spans4_ = [], []
        for s_, t in enumerate(texts):
            if not t:
                continue
            m = orc.gen_parse_matrix(t).astype(np.uint8)
            a = int(row[s_])
            v = vals[a:a + len(t)]
            sp = space[a:a + len(t)]
            nz = np.nonzero(v)[0].tolist() + [len(t)]
            for p_, e_ in zip(nz[:-1], nz[1:]):
                if (~sp[p_:e_]).any():
                    rows_.append(m[p_:e_].sum(axis=0, dtype=np.uint64).astype(np.uint8).astype(np.int8))
                    ns_ = np.nonzero(~sp[p_:e_])[0]
                    spans4_.append((p_, e_, p_ + int(ns_[0]), p_ + int(ns_[-1]) + 1))
        feats = (np.array(rows_, np.int8).reshape(-1, 25), np.array(spans4_, np.int64).reshape(-1, 4))
    return seed, kind, cps, row, vals, bits, space, rules, rule_bits, u8, boff, bpos, feats


def spans_from(vals, space, row):
    """reference tokenize() slicing on arrays: spans between consecutive boundaries, stripped, empty dropped"""
    out, counts = [], []
    for s in range(len(row) - 1):
        a, b = int(row[s]), int(row[s + 1])
        nz = np.nonzero(vals[a:b])[0].tolist() + [b - a]
        c = 0
        for p, e in zip(nz[:-1], nz[1:]):
            seg = ~space[a + p:a + e]
            if seg.any():
                idx = np.nonzero(seg)[0]
                out.append((p + int(idx[0]), p + int(idx[-1]) + 1))
                c += 1
        counts.append(c)
    return np.array(counts, np.int64), np.array(out, np.int64).reshape(-1, 2)


class FlowLeg:
    """Every group of batches also goes through the batch flow (latok_flow_*: two batches in flight per context): each batch is
    uploaded, its mask / offsets / spans are submitted back to back with the other batches' and checked after ONE wait."""

    def __init__(self, lib, group=4):
        self.lib, self.group, self.items, self.n_checked = lib, group, [], 0

    def add(self, cps, row, bits, counts, offs, spans, dt, tag):
        if cps.size == 0:
            return
        self.items.append((cps, row, bits, counts, offs, spans, dt, tag))
        if len(self.items) >= self.group:
            self.run()

    def run(self):
        from latok_amd import _lib, batch
        lib, live, jobs = self.lib, [], []

        def dev(nbytes, src=None):
            p = lib.latok_dev_alloc(int(nbytes) + 64)
            assert p, _lib.last_error()
            live.append(p)
            if src is not None and src.nbytes:
                _lib.check(lib.latok_memcpy_h2d(p, src.ctypes.data, src.nbytes))
            return p

        def get(p, shape, dt_):
            a = np.empty(shape, dt_)
            if a.nbytes:
                _lib.check(lib.latok_memcpy_d2h(a.ctypes.data, p, a.nbytes))
            return a
        for cps, row, bits, counts, offs, spans, dt, tag in self.items:
            n, total, isz = len(row) - 1, int(row[-1]), np.dtype(dt).itemsize
            d_c, d_r, d_m = dev(cps.nbytes, cps), dev(row.nbytes, row), dev(bits.nbytes)
            d_oc, d_oo, d_or = dev(n * isz), dev(max(len(offs), 1) * isz), dev(16)
            job = dict(tag=tag, n=n, dt=dt, bits=bits, counts=counts, offs=offs, spans=spans, d_m=d_m, d_oc=d_oc, d_oo=d_oo, d_or=d_or)
            batch.flow_split_offsets(d_c, 4, d_r, n, total, d_oc, d_oo, len(offs), d_or, dtype=dt)
            batch.flow_split_mask(d_c, d_r, n, total, d_m)
            if spans is not None:
                job["d_sc"], job["d_ss"], job["d_sr"] = dev(n * isz), dev(max(len(spans[1]), 1) * 2 * isz), dev(16)
                batch.flow_token_spans(d_c, 4, d_r, n, -1, job["d_sc"], job["d_ss"], len(spans[1]), job["d_sr"], dtype=dt)
            jobs.append(job)
        batch.flow_wait()
        for j in jobs:
            assert np.array_equal(get(j["d_m"], j["bits"].shape, np.uint64), j["bits"]), "flow bitmask differs: " + j["tag"]
            res = get(j["d_or"], 2, np.int64)
            assert res[0] == len(j["offs"]) and res[1] == 0, "flow offsets total differs: " + j["tag"]
            assert np.array_equal(get(j["d_oc"], j["n"], j["dt"]), j["counts"]), "flow offset counts differ: " + j["tag"]
            assert np.array_equal(get(j["d_oo"], len(j["offs"]), j["dt"]), j["offs"]), "flow offsets differ: " + j["tag"]
            if j["spans"] is not None:
                wc, ws = j["spans"]
                res = get(j["d_sr"], 2, np.int64)
                assert res[0] == len(ws) and res[1] == 0, "flow span total differs: " + j["tag"]
                assert np.array_equal(get(j["d_sc"], j["n"], j["dt"]), wc), "flow span counts differ: " + j["tag"]
                assert np.array_equal(get(j["d_ss"], (len(ws), 2), j["dt"]), ws), "flow spans differ: " + j["tag"]
        for p in live:
            lib.latok_dev_free(p)
        self.n_checked += len(jobs)
        self.items = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--workers", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from latok_amd import _lib, batch
    _lib.ensure_init()
    second = _lib.Context(0)            # every other batch runs on a second context of the same device
    flow = FlowLeg(_lib.load())
    t_end = time.time() + args.seconds
    n_batches = n_chars = n_rules = n_spans = n_u8 = n_feat = n_kind = 0
    seed = args.seed * 1_000_003
    last = time.time()
    with mp.Pool(args.workers) as pool:
        pending = [pool.apply_async(make_batch, (seed + i,)) for i in range(args.workers * 2)]
        seed += len(pending)
        while pending:
            sd, kind, cps, row, vals, bits, space, rules, rule_bits, u8, boff, bpos, feats = pending.pop(0).get()
            if time.time() < t_end:
                pending.append(pool.apply_async(make_batch, (seed,)))
                seed += 1
            dt = np.int32 if (sd >> 1) & 1 else np.int64      # record width of the compaction calls (LATOK_OUT_INT32)
            tag = f"seed {sd} kind {kind} n_str {len(row) - 1} chars {cps.size} records {np.dtype(dt).name} ctx {sd & 1}"
            if sd & 1:
                second.make_current()
            else:
                _lib.load().latok_ctx_set_current(None)
            gv = batch.split_values_batch(cps, row)
            assert np.array_equal(gv, vals), "values differ: " + tag
            gb = batch.split_mask_batch(cps, row)
            assert np.array_equal(gb, bits), "bitmask differs: " + tag
            counts, offs = batch.split_offsets_csr(cps, row, dtype=dt)
            assert counts.dtype == offs.dtype == dt
            exp = [np.nonzero(vals[row[s]:row[s + 1]])[0] for s in range(len(row) - 1)]
            assert np.array_equal(counts, [len(e) for e in exp]), "offset counts differ: " + tag
            assert np.array_equal(offs, np.concatenate(exp) if exp else np.zeros(0, np.int64)), "offsets differ: " + tag
            wc = ws = None
            if cps.size < 20000:
                wc, ws = spans_from(vals, space, row)
                gc, gs = batch.token_spans_csr(cps, row, dtype=dt)
                assert np.array_equal(gc, wc) and np.array_equal(gs, ws), "token spans differ: " + tag
                n_spans += 1
            # (the flow leg runs on whatever context is current when its group is full: both contexts get their share)
            flow.add(cps, row, bits, counts, offs, (wc, ws) if ws is not None else None, dt, tag)
            if rules is not None:
                batch.set_rules(*rules)
                try:
                    rb = batch.split_mask_batch(cps, row)
                finally:
                    batch.reset_rules()
                assert np.array_equal(rb, rule_bits), "rule-table bitmask differs: " + tag
                n_rules += 1
            if feats is not None and cps.size > 0:
                fc, fs, ff = batch.token_features_csr(cps, row, dtype=dt)
                assert ff.shape == feats[0].shape and np.array_equal(ff, feats[0]), "featurize sums differ: " + tag
                assert np.array_equal(fs, feats[1]), "featurize span records differ: " + tag
                n_feat += 1
            if cps.size > 0 and int(cps.max()) < 65536:
                # PEP 393 kinds: the same text as 1- or 2-byte code units
                units = cps.astype(np.uint8 if int(cps.max()) < 256 else np.uint16)
                assert np.array_equal(batch.split_mask_kind_csr(units, row), bits), "kind bitmask differs: " + tag
                kc, ko = batch.split_offsets_kind_csr(units, row, dtype=dt)
                assert np.array_equal(kc, counts) and np.array_equal(ko, offs), "kind offsets differ: " + tag
                if cps.size < 20000:
                    kc, ks = batch.token_spans_kind_csr(units, row, dtype=dt)
                    assert np.array_equal(kc, wc) and np.array_equal(ks, ws), "kind token spans differ: " + tag
                if rules is not None:
                    batch.set_rules(*rules)
                    try:
                        rb = batch.split_mask_kind_csr(units, row)
                    finally:
                        batch.reset_rules()
                    assert np.array_equal(rb, rule_bits), "kind rule-table bitmask differs: " + tag
                if feats is not None:
                    fc, fs, ff = batch.token_features_kind_csr(units, row, dtype=dt)
                    assert np.array_equal(ff, feats[0]) and np.array_equal(fs, feats[1]), "kind featurize differs: " + tag
                n_kind += 1
            if u8 is not None and cps.size > 0:
                # byte space: boundaries at the lead byte of every boundary char; staged path: code-point units
                flags = np.zeros(u8.size, bool)
                flags[bpos[vals != 0]] = True
                bb = batch.split_mask_utf8_bytes_csr(u8, boff)
                got = np.unpackbits(bb.view(np.uint8), bitorder="little")[:u8.size].astype(bool)
                assert np.array_equal(got, flags), "byte-space bitmask differs: " + tag
                bc, bo = batch.split_offsets_utf8_bytes_csr(u8, boff, dtype=dt)
                exp_b = [bpos[row[s]:row[s + 1]][vals[row[s]:row[s + 1]] != 0] - boff[s] for s in range(len(row) - 1)]
                assert np.array_equal(bc, [len(e) for e in exp_b]), "byte offset counts differ: " + tag
                assert np.array_equal(bo, np.concatenate(exp_b)), "byte offsets differ: " + tag
                cb, crow = batch.split_mask_utf8_csr(u8, boff)
                assert np.array_equal(crow, row) and np.array_equal(cb, bits), "code-point UTF-8 path differs: " + tag
                # offsets and token spans in code-point units (large batches: the compaction runs on masks packed from byte space)
                cc, co = batch.split_offsets_utf8_csr(u8, boff, dtype=dt)
                assert np.array_equal(cc, counts) and np.array_equal(co, offs), "code-point UTF-8 offsets differ: " + tag
                tc, ts = batch.token_spans_utf8_csr(u8, boff, dtype=dt)
                if ws is not None:
                    assert np.array_equal(tc, wc) and np.array_equal(ts, ws), "code-point UTF-8 token spans differ: " + tag
                else:
                    gc2, gs2 = batch.token_spans_csr(cps, row, dtype=dt)      # (the UTF-32 path: checked against the oracle on the smaller batches)
                    assert np.array_equal(tc, gc2) and np.array_equal(ts, gs2), "code-point UTF-8 token spans differ from the UTF-32 path: " + tag
                if rules is not None:
                    # run-time rule tables in byte space (k_tiles_main<kModeBytesRules>) and through the code-point UTF-8 path
                    rflags = np.zeros(u8.size, bool)
                    rflags[bpos[np.unpackbits(rule_bits.view(np.uint8), bitorder="little")[:cps.size].astype(bool)]] = True
                    batch.set_rules(*rules)
                    try:
                        rbb = batch.split_mask_utf8_bytes_csr(u8, boff)
                        rcb, _ = batch.split_mask_utf8_csr(u8, boff)
                    finally:
                        batch.reset_rules()
                    got = np.unpackbits(rbb.view(np.uint8), bitorder="little")[:u8.size].astype(bool)
                    assert np.array_equal(got, rflags), "byte-space rule-table bitmask differs: " + tag
                    assert np.array_equal(rcb, rule_bits), "code-point UTF-8 rule-table bitmask differs: " + tag
                n_u8 += 1
            n_batches += 1
            n_chars += cps.size
            if time.time() - last > 30:
                last = time.time()
                print(f"[soak] {n_batches} batches, {n_chars} chars, {n_rules} with rule tables, {n_spans} span checks, {n_u8} UTF-8, {n_feat} featurize, {n_kind} PEP 393 kinds ... ok",
                      flush=True)
    if flow.items:
        flow.run()
    _lib.load().latok_ctx_set_current(None)
    second.destroy()
    print(f"soak passed (two contexts alternating, int32 / int64 records alternating): {n_batches} batches, {n_chars} chars, {flow.n_checked} also through the batch flow (mask + offsets + spans), {n_rules} with rule tables, {n_spans} span checks, "
          f"{n_u8} UTF-8 (byte space + code-point) checks, {n_feat} featurize checks, {n_kind} PEP 393 kind checks, {args.seconds:.0f} s")


if __name__ == "__main__":
    main()
