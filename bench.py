#!/usr/bin/env python3
"""bench.py -- headline benchmark: input UTF-8 GB/s tokenized by the fused feature+split-mask path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                        # one rank per GPU

A "step" is one pass of the whole pipeline (fused tiles kernel -> resolve/repair kernel) over one batch
of synthetic strings that is already resident in HBM.  Workload at every N = BASELINE.json configs[1] per GPU
("1 M synthetic ASCII strings, avg 128 chars"); with N ranks each rank owns an independent shard of 1 M strings of the
same corpus (string ids rank*1M ...), i.e. configs[3]'s sharding with per-GPU work held fixed -> "scaling": "weak".
No collective on the data path: strings are independent.  torch is imported only for N > 1, and only for
the barrier and the max-over-ranks of the wall time.

Prints ONE JSON line on rank 0.  Besides the contract keys it carries
  roofline     -- dominant kernel (k_tiles_main): algorithmic HBM bytes per launch / its HIP-event time, vs 8 TB/s
  cpu_baseline -- the reference's own C functions (oracle/_ref, built from the reference's latok.c) under a restated
                  NumPy glue, 1 thread, timed on this host on the same corpus (N = 1, rank 0 only); "port" numbers of
                  oracle/latok_oracle.c ride along in cpu_baseline_port.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from latok_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)

WORKLOADS = {
    # name: (model, seed, len_lo, len_hi, default strings per GPU, description)
    "C2": (_lib.CORPUS_ASCII, 0x1A70C0DE, 64, 192, 1_000_000, "1M synthetic ASCII strings, avg 128 chars (BASELINE configs[1])"),
    "C3": (_lib.CORPUS_UNICODE, 0x1A70C0DF, 128, 384, 1_000_000, "1M mixed-Unicode strings, avg 256 chars (BASELINE configs[2])"),
    "C5": (_lib.CORPUS_ASCII, 0x1A70C0E0, 1_000_000, 1_000_000, 1_000, "long documents x 1M chars (BASELINE configs[4], reduced count)"),
}


def shard_string_ids(n_per_gpu: int, rank: int):
    """Rank r owns string ids [r*n, (r+1)*n) of the corpus: contiguous, disjoint, no exchange needed."""
    return rank * n_per_gpu, n_per_gpu


def reduce_max_seconds(dist, seconds: float, device=None) -> float:
    """max over ranks of a wall time (the only cross-rank traffic of the benchmark)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum_int(dist, value: int, device=None) -> int:
    import torch
    t = torch.tensor([value], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def cpu_baselines(workload: str, n_strings: int):
    """Time the CPU paths on this host, 1 thread, on the head of the same corpus.  Test infrastructure (oracle/) is
    used here ONLY as the thing being timed for the baseline line, never by the product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import latok_oracle as orc
    model, seed, lo, hi, _, _ = WORKLOADS[workload]
    lib = _lib.load()
    row = np.zeros(n_strings + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n_strings, lo, hi, row.ctypes.data))
    cps = np.zeros(int(row[-1]), np.uint32)
    _lib.check(lib.latok_corpus_fill_host(seed, model, 0, n_strings, row.ctypes.data, cps.ctypes.data))
    n8 = C.c_int64(0)
    _lib.check(lib.latok_utf8_bytes(cps.ctypes.data, cps.size, C.byref(n8), 0))
    out = {}
    orc.lib()
    t = time.perf_counter()
    orc.split_batch(cps, row, want_values=False, want_bits=True)
    dt = time.perf_counter() - t
    out["cpu_baseline_port"] = {
        "value": n8.value / dt / 1e9, "unit": "GB/s (input UTF-8)", "cores": 1, "kind": "port",
        "sample": f"{n_strings} strings / {cps.size} chars of the {workload} corpus, oracle/latok_oracle.c "
                  f"(reference-shaped: n x 25 matrix, 3 combines, sequential block mask), {dt:.1f} s"}
    try:
        glue = orc.RefGlue()
        n_ref = min(n_strings, 400_000)
        text = cps[:row[n_ref]].astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")
        strs = [text[row[i]:row[i + 1]] for i in range(n_ref)]
        b8 = C.c_int64(0)
        _lib.check(lib.latok_utf8_bytes(cps.ctypes.data, int(row[n_ref]), C.byref(b8), 0))
        t = time.perf_counter()
        for s in strs:
            np.nonzero(glue.split_values(s))
        dt = time.perf_counter() - t
        out["cpu_baseline"] = {
            "value": b8.value / dt / 1e9, "unit": "GB/s (input UTF-8)", "cores": 1, "kind": "reference",
            "sample": f"first {n_ref} strings / {int(row[n_ref])} chars of the {workload} corpus through the reference's own "
                      f"compiled C functions (oracle/_ref) one string at a time, as the reference runs, {dt:.1f} s"}
    except Exception as exc:  # oracle/_ref not built: report the port as the baseline
        out["cpu_baseline"] = dict(out["cpu_baseline_port"])
        out["cpu_baseline"]["note"] = f"oracle/_ref unavailable ({exc})"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--strings", type=int, default=0, help="strings per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-strings", type=int, default=1_000_000)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real runs; gloo lets several ranks rehearse on one GPU")
    ap.add_argument("--device", type=int, default=-1, help="HIP device for this rank (default: LOCAL_RANK)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU", file=sys.stderr)
        sys.exit(2)

    device = args.device if args.device >= 0 else local_rank
    dist = tdev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(device)
            tdev = torch.device("cuda", device)
            dist.init_process_group(backend="nccl", device_id=tdev)
        else:
            dist.init_process_group(backend="gloo")

    lib = _lib.ensure_init(device)
    model, seed, lo, hi, n_default, desc = WORKLOADS[args.workload]
    n_str = args.strings or n_default
    sid0, n_str = shard_string_ids(n_str, rank)

    # ---- build this rank's shard directly in HBM (offsets on host: 8 B/string; code points on device) -------------
    row = np.zeros(n_str + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, sid0, n_str, lo, hi, row.ctypes.data))
    total = int(row[-1])
    d_row = lib.latok_dev_alloc(row.nbytes)
    d_cps = lib.latok_dev_alloc(total * 4)
    d_bits = lib.latok_dev_alloc(((total + 63) // 64) * 8)
    if not (d_row and d_cps and d_bits):
        raise RuntimeError(_lib.last_error())
    _lib.check(lib.latok_memcpy_h2d(d_row, row.ctypes.data, row.nbytes))
    _lib.check(lib.latok_corpus_fill_device(seed, model, sid0, n_str, d_row, d_cps, None))
    utf8 = C.c_int64(0)
    _lib.check(lib.latok_utf8_bytes(d_cps, total, C.byref(utf8), _lib.DEVICE_PTRS))
    _lib.check(lib.latok_reserve(total, n_str))

    def sync_all():
        _lib.check(lib.latok_sync())
        if dist is not None:
            if tdev is not None:
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    # ---- W untimed warm-up steps, then exactly K timed steps ------------------------------------------------------
    if args.warmup > 0:
        _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, args.warmup, 0, None, None, None))
    sync_all()
    ms_events = C.c_float(0)
    t0 = time.perf_counter()
    _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 0, args.steps, C.byref(ms_events), None, None))
    _lib.check(lib.latok_sync())
    if tdev is not None:
        import torch
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        wall = reduce_max_seconds(dist, wall, tdev)
        utf8_all = reduce_sum_int(dist, utf8.value, tdev)
        chars_all = reduce_sum_int(dist, total, tdev)
        strs_all = reduce_sum_int(dist, n_str, tdev)
    else:
        utf8_all, chars_all, strs_all = utf8.value, total, n_str

    # ---- dominant kernel alone: `steps` back-to-back launches of k_tiles_main between one pair of HIP events on the launch
    #      stream (per-launch event pairs charge each interval with ~6 us of marker dispatch), outside the timed region ---
    ms_tiles, n_fix = C.c_float(0), C.c_int64(0)
    _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 0, args.steps, None, C.byref(ms_tiles),
                                          C.byref(n_fix)))
    # ---- streaming-read ceiling of this box on the same buffer, also outside the timed region ----------
    ms_read = C.c_float(0)
    read_bytes = (total * 4 // 16384) * 16384
    if rank == 0 and read_bytes > 0:
        _lib.check(lib.latok_bench_stream_read(d_cps, read_bytes, 3, 20, C.byref(ms_read)))
    for p in (d_row, d_cps, d_bits):
        lib.latok_dev_free(p)

    if rank == 0:
        alg_read = 4 * total + 8 * (n_str + 1)          # 4 B/code point + 8 B/string row offset
        t_kernel = ms_tiles.value / args.steps / 1e3     # s per launch
        achieved = alg_read / t_kernel / 1e9
        measured_read = (read_bytes / (ms_read.value / 20 / 1e3) / 1e9) if ms_read.value > 0 else None
        traffic = None
        try:  # HBM bytes per launch from the committed PMC pass of the same workload, if any (else null)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
                pmc = json.load(f)
            if pmc.get("workload") == args.workload and pmc.get("total_chars") == total:
                traffic = pmc.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        line = {
            "metric": "input UTF-8 GB/s tokenized (fused feature+split-mask path)",
            "value": utf8_all * args.steps / wall / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "dtype_note": "u32 code points in, 64-bit bit-sliced boolean words, u64 bitmask out (integer / bit ops)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "strings_per_gpu": n_str, "strings_total": strs_all,
                       "chars_total": chars_all, "utf8_bytes_total": utf8_all, "sharding": f"{world} x independent string shards"},
            "ms_per_step_hip_events_rank0": ms_events.value / args.steps,
            "fix_tiles_rank0": n_fix.value, "tiles_rank0": (total + _lib.TILE_CHARS - 1) // _lib.TILE_CHARS,
            "roofline": {"bound": "hbm", "kernel": "k_tiles_main", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": alg_read, "kernel_ms": t_kernel * 1e3,
                         "kernel_timing": f"{args.steps} back-to-back launches between one HIP event pair on the launch stream",
                         "pipeline_frac": alg_read / (ms_events.value / args.steps / 1e3) / 1e9 / HBM_PEAK_GBS,
                         "measured_stream_read": measured_read,
                         "frac_of_measured_read": (achieved / measured_read) if measured_read else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            line.update(cpu_baselines(args.workload, min(args.cpu_strings, n_str)))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
