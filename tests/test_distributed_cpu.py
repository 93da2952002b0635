"""N > 1 path of bench.py on CPU: world_size-2 gloo, the same helpers bench.py uses for sharding and for its only
cross-rank traffic (max of wall time, sums of byte counts).  The data path itself has no collective."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import bench
    import latok_oracle as orc
    from latok_amd import _lib
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        n_per = 500
        sid0, n = bench.shard_string_ids(n_per, rank)
        lib = _lib.load()
        model, seed, lo, hi, _, _ = bench.WORKLOADS["C2"]
        row = np.zeros(n + 1, np.int64)
        lib.latok_corpus_offsets(seed, sid0, n, lo, hi, row.ctypes.data)
        cps = np.zeros(int(row[-1]), np.uint32)
        lib.latok_corpus_fill_host(seed, model, sid0, n, row.ctypes.data, cps.ctypes.data)
        vals, _ = orc.split_batch(cps, row, want_bits=False)
        total_chars = bench.reduce_sum_int(dist, int(row[-1]))
        total_bound = bench.reduce_sum_int(dist, int(np.count_nonzero(vals)))
        slowest = bench.reduce_max_seconds(dist, 0.25 * (rank + 1))
        # per-rank times as bench.py reports them (every rank sees every rank's HIP-event time, in rank order)
        assert bench.gather_per_rank(dist, 1.5 + rank, rank, world) == [1.5 + r for r in range(world)]
        # strong-scaling workloads (C4 / C5): the batch's string ids are cut into contiguous ranges that tile it exactly
        cuts = [bench.split_string_ids(1001, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and sum(n_ for _, n_ in cuts) == 1001
        assert all(cuts[r][0] + cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
        # product sharding helper: rank r tokenizes its char-balanced slice of one common batch; the shards' boundary
        # counts add up to the whole batch's (no exchange on the data path, only this reduction for the check)
        from latok_amd import shard
        m_row = np.zeros(801, np.int64)
        lib.latok_corpus_offsets(seed, 0, 800, lo, hi, m_row.ctypes.data)
        m_cps = np.zeros(int(m_row[-1]), np.uint32)
        lib.latok_corpus_fill_host(seed, model, 0, 800, m_row.ctypes.data, m_cps.ctypes.data)
        c_r, row_r, _ = shard.take_shard(m_cps, m_row, rank, world)
        v_r, _ = orc.split_batch(np.ascontiguousarray(c_r), np.ascontiguousarray(row_r), want_bits=False)
        shard_bound = bench.reduce_sum_int(dist, int(np.count_nonzero(v_r)))
        whole_bound = int(np.count_nonzero(orc.split_batch(m_cps, m_row, want_bits=False)[0]))
        assert shard_bound == whole_bound, (shard_bound, whole_bound)
        dist.barrier()
        q.put((rank, sid0, int(row[-1]), total_chars, total_bound, slowest, cps[:8].tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_sharding_and_reductions(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, s0, c0, tot0, b0, slow0, head0), (r1, s1, c1, tot1, b1, slow1, head1) = res
    assert (s0, s1) == (0, 500)                       # disjoint contiguous shards
    assert tot0 == tot1 == c0 + c1                    # sum over ranks
    assert b0 == b1 > 0
    assert slow0 == slow1 == 0.5                      # max over ranks
    assert head0 != head1                             # different strings on different ranks
    # the union of the two shards is the single-process corpus of 1000 strings
    sys.path.insert(0, ROOT)
    import bench
    from latok_amd import _lib
    lib = _lib.load()
    model, seed, lo, hi, _, _ = bench.WORKLOADS["C2"]
    row = np.zeros(1001, np.int64)
    lib.latok_corpus_offsets(seed, 0, 1000, lo, hi, row.ctypes.data)
    assert int(row[-1]) == tot0 and int(row[500]) == c0


def _launcher_worker(rank, world, port, q, take_turns):
    """what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` does in each rank, with the device half
    of the C ABI faked (tests/test_bench_launcher.py): bench.main's launcher branch end to end over gloo"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import io
    import bench
    from test_bench_launcher import FakeApi
    api = FakeApi(n_dev=1)     # a launcher that narrows HIP_VISIBLE_DEVICES to one GPU per rank: LOCAL_RANK 1 -> device 0
    out = io.StringIO()
    argv = ["--gpus", str(world), "--steps", "4", "--warmup", "1", "--strings", "1500", "--no-cpu-baseline", "--sustain-s", "0", "--settle-s", "0"]
    rc = bench.main(argv + (["--take-turns"] if take_turns else []), api=api, out=out)
    q.put((rank, rc, out.getvalue(), [c.device for c in api.contexts], list(api.lib.fills), list(api.lib.timed)))


@pytest.mark.timeout(240)
@pytest.mark.parametrize("take_turns", [False, True])
def test_two_rank_gloo_bench_main_under_a_launcher(take_turns):
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_launcher_worker, args=(r, 2, port, q, take_turns)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=200) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, rc0, out0, dev0, fills0, timed0), (_, rc1, out1, dev1, fills1, timed1) = res
    assert rc0 == rc1 == 0 and out1.strip() == ""          # ONE line, from rank 0
    line = json.loads(out0)
    assert dev0 == dev1 == [0]                              # LOCAL_RANK mod visible device count
    assert sorted(set(fills0)) == [(0, 0, 1500)] and sorted(set(fills1)) == [(0, 1500, 1500)]   # (twice each: the flow reads two copies)
    assert line["n_gpus"] == 2 and line["config"]["strings_total"] == 3000 and len(line["ms_per_rank"]) == 2
    assert "launcher" in line["config"]["launch"]
    (a0, b0), (a1, b1) = timed0[0][2:], timed1[0][2:]       # CLOCK_MONOTONIC is one clock for every process of the host
    if take_turns:
        assert b0 <= a1 or b1 <= a0
        assert line["value"] is None and line["value_projected"] > 0
    else:
        assert line["value"] == pytest.approx(line["config"]["utf8_bytes_total"] * 4 / (max(b0 - a0, b1 - a1) / 1e9) / 1e9, rel=1e-6)
