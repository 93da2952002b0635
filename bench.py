#!/usr/bin/env python3
"""bench.py -- headline benchmark: input UTF-8 GB/s tokenized by the fused feature+split-mask path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                        # one rank per GPU

A "step" is one pass of the whole pipeline (tile index -> fused tiles kernel -> resolve/repair) over one batch of
synthetic strings that is already resident in HBM.  Default workload at every N = BASELINE.json configs[1] per GPU
("1 M synthetic ASCII strings, avg 128 chars"): with N ranks each rank owns an independent shard of 1 M strings of the
same corpus (string ids rank*1M ...) -> "scaling": "weak".  The other workloads:
    C3   configs[2]  1 M mixed-Unicode strings per GPU (weak)
    C4   configs[3]  100 M strings as C2, the WHOLE batch split over the N ranks (strong; 51 GB resident at N = 1)
    C5   configs[4]  10 K documents x 1 M chars, split over the N ranks (strong; 40 GB resident at N = 1)
No collective on the data path: strings are independent.  torch is imported only for N > 1, and only for
the barriers and the reductions of the report (max of the per-rank times, sums of byte counts).

Timing: W untimed warm-up steps, then EXACTLY K steps between a barrier + device synchronisation on both sides.  Every
rank times its K steps twice over the same region: with one pair of HIP events on the launch stream (GPU time) and with
the host clock (wall).  `value` = bytes of all ranks x K / max over ranks of the HIP-event time; the max of the wall
times rides along as `ms_per_step_wall` (with 8 Python processes around a 2 ms region the slowest host's jitter, not a
GPU, would otherwise decide the number).  `ms_per_rank` lists every rank's event time so that a straggler is visible.

Prints ONE JSON line on rank 0.  Besides the contract keys it carries
  roofline     -- dominant kernel (k_tiles_main): algorithmic HBM bytes per launch / its HIP-event time, vs 8 TB/s
  sustained    -- >= 1 s of back-to-back steps outside the timed region (clock / thermal drift shows here)
  cpu_baseline -- the reference's own C functions (oracle/_ref, built from the reference's latok.c) under a restated
                  NumPy glue, 1 thread, timed on this host on the same corpus (N = 1, rank 0 only);
                  cpu_baseline_port = oracle/latok_oracle.c (1 thread); cpu_baseline_fused_allcores = the fused CPU
                  model (oracle/fused_model.cpp) over all host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from latok_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)

WORKLOADS = {
    # name: (model, seed, len_lo, len_hi, default strings (per GPU if weak, in total if strong), description)
    "C2": (_lib.CORPUS_ASCII, 0x1A70C0DE, 64, 192, 1_000_000, "1M synthetic ASCII strings, avg 128 chars (BASELINE configs[1])"),
    "C3": (_lib.CORPUS_UNICODE, 0x1A70C0DF, 128, 384, 1_000_000, "1M mixed-Unicode strings, avg 256 chars (BASELINE configs[2])"),
    "C4": (_lib.CORPUS_ASCII, 0x1A70C0DE, 64, 192, 100_000_000, "100M strings avg 128 chars, the batch split over the ranks (BASELINE configs[3])"),
    "C5": (_lib.CORPUS_ASCII, 0x1A70C0E0, 1_000_000, 1_000_000, 10_000, "10K documents x 1M chars, split over the ranks (BASELINE configs[4])"),
}
SCALING = {"C2": "weak", "C3": "weak", "C4": "strong", "C5": "strong"}
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r02_pmc_summary.json")


def shard_string_ids(n_per_gpu: int, rank: int):
    """Weak scaling: rank r owns string ids [r*n, (r+1)*n) of the corpus: contiguous, disjoint, no exchange needed."""
    return rank * n_per_gpu, n_per_gpu


def split_string_ids(n_total: int, rank: int, world: int):
    """Strong scaling: the batch of n_total strings is cut into `world` contiguous id ranges (sizes differ by <= 1)."""
    lo = rank * n_total // world
    hi = (rank + 1) * n_total // world
    return lo, hi - lo


def reduce_max_seconds(dist, seconds: float, device=None) -> float:
    """max over ranks of a time (the only cross-rank traffic of the benchmark besides the sums below)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum_int(dist, value: int, device=None) -> int:
    import torch
    t = torch.tensor([value], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_per_rank(dist, value: float, rank: int, world: int, device=None):
    """every rank's value, in rank order, on every rank (a sum of one-hot vectors: works on gloo and on RCCL)"""
    import torch
    t = torch.zeros(world, dtype=torch.float64, device=device if device is not None else "cpu")
    t[rank] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def host_corpus(workload: str, n_strings: int):
    model, seed, lo, hi, _, _ = WORKLOADS[workload]
    lib = _lib.load()
    row = np.zeros(n_strings + 1, np.int64)
    _lib.check(lib.latok_corpus_offsets(seed, 0, n_strings, lo, hi, row.ctypes.data))
    cps = np.zeros(int(row[-1]), np.uint32)
    _lib.check(lib.latok_corpus_fill_host(seed, model, 0, n_strings, row.ctypes.data, cps.ctypes.data))
    return cps, row


def cpu_baselines(workload: str, n_strings: int):
    """Time the CPU paths on this host on the head of the same corpus.  Test infrastructure (oracle/) is used here ONLY
    as the thing being timed for the baseline line, never by the product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import latok_oracle as orc
    lib = _lib.load()
    cps, row = host_corpus(workload, n_strings)
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
    try:
        out["cpu_baseline_fused_allcores"] = cpu_fused_allcores(workload, cps, row, n8.value)
    except Exception as exc:
        out["cpu_baseline_fused_allcores"] = {"value": None, "note": f"oracle/libfused_model.so unavailable ({exc})"}
    return out


def cpu_fused_allcores(workload, cps, row, utf8_bytes, repeats=3):
    """the fused CPU restatement (oracle/fused_model.cpp = the GPU pipeline's lane math run by a host loop
    over 64 lanes) over contiguous string shards, one host thread per core (ctypes releases the GIL)."""
    model = C.CDLL(os.path.join(ROOT, "oracle", "libfused_model.so"))
    model.fused_split_batch.restype = C.c_int
    model.fused_split_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n = row.size - 1

    def shards(parts):
        out = []
        for i in range(parts):
            a, b = i * n // parts, (i + 1) * n // parts
            r = np.ascontiguousarray(row[a:b + 1] - row[a])
            c = cps[row[a]:row[b]]
            out.append((c, r, b - a, np.zeros((c.size + 63) // 64, np.uint64)))
        return out

    def run(parts):
        sh = shards(parts)
        best = None
        for _ in range(repeats):
            ths = [threading.Thread(target=lambda s=s: model.fused_split_batch(s[0].ctypes.data, s[1].ctypes.data, s[2], None,
                                                                               s[3].ctypes.data, None)) for s in sh]
            t = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t
            best = dt if best is None or dt < best else best
        return best

    t1, tn = run(1), run(cores)
    return {"value": utf8_bytes / tn / 1e9, "unit": "GB/s (input UTF-8)", "cores": cores, "kind": "port",
            "one_core_value": utf8_bytes / t1 / 1e9,
            "sample": f"{n} strings / {cps.size} chars of the {workload} corpus, oracle/fused_model.cpp (bit-sliced 64-char "
                      f"words, no n x 25 matrix) on {cores} threads over contiguous string shards, best of {repeats}: "
                      f"{tn:.2f} s ({t1:.2f} s on 1 thread)"}


def pmc_traffic(workload: str, total_chars: int):
    """HBM bytes per k_tiles_main launch from the committed PMC passes of the same workload (profiles/, collected with
    rocprofv3 --pmc as MI355X_MICROARCH.md prescribes: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections); a
    counter pass cannot run inside this process, so the figure is null for a workload / size that has none."""
    try:
        with open(PMC_SUMMARY) as f:
            pmc = json.load(f)
        for rec in pmc.get("runs", []):
            if (rec.get("label") == "bench" and rec.get("workload") == workload and rec.get("kernel") == "k_tiles_main"
                    and rec.get("total_chars") == total_chars):
                return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--strings", type=int, default=0, help="strings per GPU (weak workloads) / in total (strong); default: the workload's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-strings", type=int, default=1_000_000)
    ap.add_argument("--sustain-s", type=float, default=1.0, help="length of the sustained run after the timed region (0 = off)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real runs; gloo lets several ranks rehearse on one GPU")
    ap.add_argument("--device", type=int, default=-1, help="HIP device for this rank (default: LOCAL_RANK)")
    ap.add_argument("--take-turns", action="store_true",
                    help="rehearsal on ONE shared GPU: the ranks run their timed regions one after the other, so each rank's "
                         "time is what a GPU of its own would give; the line is marked as a rehearsal")
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
    scaling = SCALING[args.workload]
    if scaling == "weak":
        sid0, n_str = shard_string_ids(args.strings or n_default, rank)
    else:
        sid0, n_str = split_string_ids(args.strings or n_default, rank, world)

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
    del row

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
    wall = 0.0
    for turn in range(world if args.take_turns else 1):
        if not args.take_turns or turn == rank:
            t0 = time.perf_counter()
            # K pipeline passes between one pair of HIP events on the launch stream; returns after the second event
            _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 0, args.steps, C.byref(ms_events), None, None))
            _lib.check(lib.latok_sync())
            if tdev is not None:
                import torch
                torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
    ev_s = ms_events.value / 1e3
    if dist is not None:
        per_rank_ms = [x * 1e3 / args.steps for x in gather_per_rank(dist, ev_s, rank, world, tdev)]
        ev_max = reduce_max_seconds(dist, ev_s, tdev)
        wall_max = reduce_max_seconds(dist, wall, tdev)
        utf8_all = reduce_sum_int(dist, utf8.value, tdev)
        chars_all = reduce_sum_int(dist, total, tdev)
        strs_all = reduce_sum_int(dist, n_str, tdev)
    else:
        per_rank_ms = [ev_s * 1e3 / args.steps]
        ev_max, wall_max, utf8_all, chars_all, strs_all = ev_s, wall, utf8.value, total, n_str

    # ---- everything below is outside the timed region ------------------------------------------------------------------
    # dominant kernel alone: `steps` back-to-back launches of k_tiles_main between one pair of HIP events on the launch
    # stream (per-launch event pairs charge each interval with ~6 us of marker dispatch)
    ms_tiles, n_fix = C.c_float(0), C.c_int64(0)
    sustained = None
    ms_read = C.c_float(0)
    read_bytes = min((total * 4 // 16384) * 16384, 1 << 31)
    if rank == 0 or not args.take_turns:
        _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 0, args.steps, None, C.byref(ms_tiles),
                                              C.byref(n_fix)))
    if rank == 0:
        # sustained: >= sustain_s of back-to-back pipeline passes, in chunks of <= 2000 passes per event pair
        if args.sustain_s > 0 and ev_s > 0:
            per = ev_s / args.steps
            want = max(args.steps, int(args.sustain_s / per) + 1)
            done, t_ms, chunks = 0, 0.0, []
            while done < want:
                k = min(2000, want - done)
                ms = C.c_float(0)
                _lib.check(lib.latok_bench_split_mask(d_cps, d_row, n_str, total, d_bits, 0, k, C.byref(ms), None, None))
                chunks.append(ms.value / k)
                t_ms += ms.value
                done += k
            sustained = {"steps": done, "seconds": t_ms / 1e3, "ms_per_step": t_ms / done,
                         "value": utf8.value * done / (t_ms / 1e3) / 1e9, "unit": "GB/s (rank 0)",
                         "ms_per_step_first_chunk": chunks[0], "ms_per_step_last_chunk": chunks[-1]}
        # streaming-read ceiling of this box on the same buffer
        if read_bytes > 0:
            _lib.check(lib.latok_bench_stream_read(d_cps, read_bytes, 3, 20, C.byref(ms_read)))
    for p in (d_row, d_cps, d_bits):
        lib.latok_dev_free(p)

    if rank == 0:
        alg_read = 4 * total + 8 * (n_str + 1)          # 4 B/code point + 8 B/string row offset
        t_kernel = ms_tiles.value / args.steps / 1e3     # s per launch
        achieved = alg_read / t_kernel / 1e9
        measured_read = (read_bytes / (ms_read.value / 20 / 1e3) / 1e9) if ms_read.value > 0 else None
        traffic = pmc_traffic(args.workload, total)
        line = {
            "metric": "input UTF-8 GB/s tokenized (fused feature+split-mask path)",
            "value": utf8_all * args.steps / ev_max / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ev_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u64", "dtype_note": "u32 code points in, 64-bit bit-sliced boolean words, u64 bitmask out (integer / bit ops)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "strings_per_gpu": n_str, "strings_total": strs_all,
                       "chars_total": chars_all, "utf8_bytes_total": utf8_all,
                       "sharding": f"{world} x independent contiguous string-id shards, no collective on the data path"},
            "timing": "max over ranks of the HIP-event time of the K steps (one event pair on the launch stream per rank), "
                      "region bracketed by barrier + device sync on both sides",
            "ms_per_step_wall": wall_max / args.steps * 1e3,
            "value_wall": utf8_all * args.steps / wall_max / 1e9,
            "ms_per_rank": per_rank_ms,
            "sustained": sustained,
            "fix_tiles_rank0": n_fix.value, "tiles_rank0": (total + _lib.TILE_CHARS - 1) // _lib.TILE_CHARS,
            "roofline": {"bound": "hbm", "kernel": "k_tiles_main", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": alg_read, "kernel_ms": t_kernel * 1e3,
                         "kernel_timing": f"{args.steps} back-to-back launches between one HIP event pair on the launch stream (rank 0)",
                         "pipeline_frac": alg_read / (per_rank_ms[0] / 1e3) / 1e9 / HBM_PEAK_GBS,
                         "measured_stream_read": measured_read,
                         "frac_of_measured_read": (achieved / measured_read) if measured_read else None},
        }
        if args.take_turns:
            line["rehearsal"] = ("ranks took turns on ONE shared GPU: per-rank times are single-GPU times and `value` is what "
                                 f"{world} such GPUs would give -- a rehearsal of the N > 1 code path, not a measurement of {world} GPUs")
        if world == 1 and not args.no_cpu_baseline:
            line.update(cpu_baselines("C2" if args.workload in ("C4",) else args.workload,
                                      min(args.cpu_strings, n_str, 1_000_000 if args.workload != "C5" else 64)))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
