#!/usr/bin/env python3
"""Measure tokenization speed over a file of strings -- the counterpart of the reference's
``scripts/timing/time_tokenizer.py`` (same positional argument and flags), driven through the fused batch path.

usage:
  tools/time_tokenizer.py <file> [--split | --matrix | --features] [--mincount N] [--outfile OUT]
                                 [--format csv-json|lines|jsonl] [--batch STRINGS] [--field KEY]

Input formats
  csv-json (default, what the reference reads, time_tokenizer.py:35-37): CSV rows whose SECOND column is a
           JSON-encoded string; the text is ``json.loads(row[1]).strip()``.
  lines    one string per line (stripped).
  jsonl    one JSON value per line: a string, or an object whose ``--field`` (default "text") is the string.
``.gz`` files are read through gzip like the reference does (time_tokenizer.py:30-33).

Modes (reference wrappers, time_tokenizer.py:43-62)
  default     tokens of every string            (LaTokenizeWrapper)  -> batch.tokenize_batch
  --split     boundary offsets only             (LaSplitWrapper)     -> batch.split_offsets_csr
  --features  LaToken feature vectors           (LaFeatureWrapper)   -> batch.featurize_batch
  --matrix    the n x 25 matrix of every string (LaMatrixWrapper)    -> latok._gen_parse_matrix, string by string
``--outfile`` writes the tokens of each string tab-separated on one line (time_tokenizer.py:105-108) and, like the
reference, overrides --split / --matrix.

Strings are tokenized ``--batch`` at a time (default 65 536); progress goes to stderr once ``--mincount`` strings
have been processed.  There is no CPU fallback: this needs a HIP device.
"""
import argparse
import csv
import gzip
import json
import os
import sys
import time
from datetime import datetime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_texts(path, fmt="csv-json", field="text"):
    """Yield the strings of ``path`` (see module docstring for the formats)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8", newline="" if fmt == "csv-json" else None) as f:
        if fmt == "csv-json":
            for row in csv.reader(f):
                yield json.loads(row[1]).strip()
        elif fmt == "lines":
            for line in f:
                yield line.strip()
        elif fmt == "jsonl":
            for line in f:
                line = line.strip()
                if not line:
                    continue
                v = json.loads(line)
                yield (v if isinstance(v, str) else v[field]).strip()
        else:
            raise ValueError(f"unknown format {fmt!r}")


def batches(it, size):
    buf = []
    for x in it:
        buf.append(x)
        if len(buf) >= size:
            yield buf
            buf = []
    if buf:
        yield buf


def make_worker(mode):
    """mode -> function(list[str]) -> list of per-string token lists (or None when nothing is to be written)."""
    from latok_amd import batch
    if mode == "tokens":
        return batch.tokenize_batch
    if mode == "split":
        def split(texts):
            cps, row = batch.pack(texts)
            batch.split_offsets_csr(cps, row)
            return None
        return split
    if mode == "features":
        def feats(texts):
            return [[t.text for t in toks] for toks in batch.featurize_batch(texts)]
        return feats
    if mode == "matrix":
        from latok_amd.latok import _gen_parse_matrix

        def matrix(texts):
            for t in texts:
                _gen_parse_matrix(t)
            return None
        return matrix
    raise ValueError(mode)


def run(infile, mode="tokens", fmt="csv-json", field="text", batch_size=65536, mincount=100000, outfile=None,
        log=sys.stderr):
    worker = make_worker(mode)
    worker(["This is a test line, just to get things warmed up..."])
    out = open(outfile, "w", encoding="utf-8") if outfile else None
    n_lines = n_chars = n_bytes = 0
    t_work = 0.0
    t0 = time.perf_counter()
    print(f"{datetime.now()}: Beginning tokenization...", file=log)
    try:
        for texts in batches(read_texts(infile, fmt, field), batch_size):
            t = time.perf_counter()
            res = worker(texts)
            t_work += time.perf_counter() - t
            if out is not None:
                for toks in res:
                    print("\t".join(toks), file=out)
            n_lines += len(texts)
            n_chars += sum(len(x) for x in texts)
            n_bytes += sum(len(x.encode("utf-8", "surrogatepass")) for x in texts)
            if n_lines > mincount - 1:
                dt = time.perf_counter() - t0
                print(f"{datetime.now()}: processed {n_lines} lines @ {n_lines / dt:.4f}/sec in {dt:.3f}s", file=log)
    finally:
        if out is not None:
            out.close()
    wall = time.perf_counter() - t0
    print(f"{datetime.now()}: ...tokenized {n_lines} lines", file=log)
    summary = {"lines": n_lines, "chars": n_chars, "utf8_bytes": n_bytes, "mode": mode, "batch": batch_size,
               "wall_s": wall, "tokenizer_s": t_work,
               "lines_per_s_wall": n_lines / wall if wall > 0 else None,
               "lines_per_s_tokenizer": n_lines / t_work if t_work > 0 else None,
               "utf8_MBps_tokenizer": n_bytes / t_work / 1e6 if t_work > 0 else None}
    print(json.dumps(summary), file=log)
    return summary


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("infile", help="path to file with texts (see --format)")
    ap.add_argument("--split", action="store_true", help="only generate the split offsets, no token strings")
    ap.add_argument("--matrix", action="store_true", help="only generate the parse matrix of every string")
    ap.add_argument("--features", action="store_true", help="featurize the tokens")
    ap.add_argument("--mincount", type=int, default=100000, help="count at which to begin displaying progress")
    ap.add_argument("--outfile", help="file to which tokenizations are written, one string per line")
    ap.add_argument("--format", default="csv-json", choices=["csv-json", "lines", "jsonl"])
    ap.add_argument("--field", default="text", help="jsonl: key of the string inside each object")
    ap.add_argument("--batch", type=int, default=65536, help="strings per kernel batch")
    args = ap.parse_args(argv)
    print(f"{datetime.now()}: {args}", file=sys.stderr)
    mode = "tokens"
    if args.outfile:
        mode = "features" if args.features else "tokens"   # reference: --outfile turns --split / --matrix off
    elif args.split:
        mode = "split"
    elif args.matrix:
        mode = "matrix"
    elif args.features:
        mode = "features"
    run(args.infile, mode, args.format, args.field, args.batch, args.mincount, args.outfile)


if __name__ == "__main__":
    main()
