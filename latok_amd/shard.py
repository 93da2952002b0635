"""Sharding a CSR batch across GPUs: every string is tokenized independently (reference tokenize() takes one
str, default_tokenizer.py:137; no cross-string state in latok.c), so a batch splits into contiguous string-id ranges,
one per rank, balanced by cumulative char count, and the data path needs no collective.  One process per GPU; each
rank calls the batch API on its own slice.  Results of the shards concatenate to the result of the whole batch.
"""
import numpy as np


def shard_bounds(row_off, world: int):
    """String-id cut points [b_0 = 0, b_1, ..., b_world = n_str]: rank r owns strings [b_r, b_{r+1}).  Cuts are placed
    where the cumulative char count crosses r/world of the total (a string is never split)."""
    row_off = np.asarray(row_off, dtype=np.int64)
    if row_off.ndim != 1 or row_off.size < 1:
        raise ValueError("row_off must be a 1-D array of n_str + 1 offsets")
    if world < 1:
        raise ValueError("world must be >= 1")
    n_str = row_off.size - 1
    total = int(row_off[-1])
    targets = (np.arange(1, world, dtype=np.int64) * total) // world
    cuts = np.searchsorted(row_off, targets, side="left").astype(np.int64)
    cuts = np.minimum(cuts, n_str)
    return np.concatenate([[0], cuts, [n_str]]).astype(np.int64)


def take_shard(cps, row_off, rank: int, world: int):
    """(cps_r, row_off_r, first_string_id) of rank `rank`: views into the batch, row offsets rebased to 0."""
    b = shard_bounds(row_off, world)
    s0, s1 = int(b[rank]), int(b[rank + 1])
    row_off = np.asarray(row_off, dtype=np.int64)
    lo, hi = int(row_off[s0]), int(row_off[s1])
    return np.asarray(cps)[lo:hi], row_off[s0:s1 + 1] - lo, s0
