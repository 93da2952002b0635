#!/usr/bin/env python3
"""bench.py -- headline benchmark: input UTF-8 GB/s tokenized by the fused feature+split-mask path on MI355X.

    python bench.py --gpus N --steps K --warmup W                        # N GPUs of one node from ONE process, no launcher
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                        # one process per GPU (gloo carries the barriers)

Without a launcher (WORLD_SIZE unset) bench.py starts the N ranks itself.  N > 1: one child PROCESS per GPU, spawned before
this process has touched a GPU; each child owns one library context on its device, the children meet at a gate in shared
memory (latok_gate_create_shared), the parent collects their records and prints the line.  Every rank then has a HIP
runtime of its own -- exactly the N = 1 run, N times -- so nothing in one rank's launch path can wait for another's
(3 launches per 0.1 ms step and GPU: N host threads of ONE process would share the runtime's locks; `--launch threads` runs
that form: N contexts (latok_ctx_create(device)) + N host threads, the product's in-process way of using a node,
include/latok_hip.h "contexts").  Every string is independent, so there is no collective on the data path,
RCCL is not linked and torch is not imported.  `--devices 0,0 [--take-turns]` rehearses N ranks on fewer GPUs.  Under a
launcher every process is one rank and torch.distributed (gloo) is used for the barriers and the gather of the records only.

A "step" is one pass of the whole pipeline (tile index -> fused tiles kernel -> resolve/repair) over one batch of
synthetic strings that is already resident in HBM.  Default workload at every N = BASELINE.json configs[1] per GPU
("1 M synthetic ASCII strings, avg 128 chars"): with N ranks each rank owns an independent shard of 1 M strings of the
same corpus (string ids rank*1M ...) -> "scaling": "weak".  The other workloads:
    C3   configs[2]  1 M mixed-Unicode strings per GPU (weak)
    C4   configs[3]  100 M strings as C2, the WHOLE batch split over the N ranks (strong; 51 GB resident at N = 1)
    C5   configs[4]  10 K documents x 1 M chars, split over the N ranks (strong; 40 GB resident at N = 1)

Timing: W untimed warm-up steps (then untimed passes until the GPU has been under load for `--settle-s`, 50 ms: the GPU
idles while the shard is built and reaches its steady clocks only after tens of ms -- without it the first 20-step region
after 5 warm-up steps runs 2 % (C2) to 14 % (C3) below the fourth; reported as `warmup_settle`), then EXACTLY K steps.  The timed region of a rank is ONE library call
(latok_bench_split_mask_flow_gated / latok_bench_split_mask_gated): rendezvous of all ranks -> host monotonic clock -> HIP
event -> K passes -> HIP event -> stream(s) synchronise -> host clock -> rendezvous.  By default the K passes go through the
product's batch flow (include/latok_hip.h "batch flow", `--in-flight 2`): consecutive batches alternate between two streams,
two workspaces and two output bitmasks, so the string-index and resolve launches of one batch run in the shadow of the other
batch's tile kernel; every pass still runs all three kernels and every result is complete when the clock stops.  The same K
steps one batch at a time (`--in-flight 1`, what earlier rounds reported as `value`) are timed right after and ride along as
`serial`.  `value` = UTF-8 bytes of all ranks x K / whole-job WALL time (in
process: last rank out - first rank in, one clock; under a launcher: max over ranks of the rank's wall time).  The
HIP-event time of the same K steps rides along (`ms_per_step_events`, `value_events`, `ms_per_rank`); round 2 reported
that one as `value` (0.7-2 % higher), round 1 and this round the wall time, as the contract reads.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline     -- dominant kernel (k_tiles_main): algorithmic HBM bytes per launch / its HIP-event time, vs 8 TB/s; measured
                  on every rank (frac_per_rank), `frac` is the slowest rank's
  sustained    -- >= 1 s of back-to-back steps outside the timed region (clock / thermal drift shows here)
  cpu_baseline -- at every N: the reference's own C functions (oracle/_ref, built from the reference's latok.c) under a
                  restated NumPy glue, 1 thread, timed on this host on the same corpus;
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

GATE_TIMEOUT_S = 900.0
CHILD_TIMEOUT_S = 1500.0   # --launch procs: a rank that has not finished by then is killed (its own process handle)
AUTO_FLOW_MAX_CHARS = 4_000_000_000   # --in-flight 0: batches below ~1 M tiles go through the batch flow
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)

WORKLOADS = {
    # name: (model, seed, len_lo, len_hi, default strings (per GPU if weak, in total if strong), description)
    "C2": (_lib.CORPUS_ASCII, 0x1A70C0DE, 64, 192, 1_000_000, "1M synthetic ASCII strings, avg 128 chars (BASELINE configs[1])"),
    "C3": (_lib.CORPUS_UNICODE, 0x1A70C0DF, 128, 384, 1_000_000, "1M mixed-Unicode strings, avg 256 chars (BASELINE configs[2])"),
    "C4": (_lib.CORPUS_ASCII, 0x1A70C0DE, 64, 192, 100_000_000, "100M strings avg 128 chars, the batch split over the ranks (BASELINE configs[3])"),
    "C5": (_lib.CORPUS_ASCII, 0x1A70C0E0, 1_000_000, 1_000_000, 10_000, "10K documents x 1M chars, split over the ranks (BASELINE configs[4])"),
}
SCALING = {"C2": "weak", "C3": "weak", "C4": "strong", "C5": "strong"}
PMC_SUMMARIES = [os.path.join(ROOT, "profiles", n) for n in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json")]


# ---- host-side facts about a GPU, read from sysfs without touching the HIP runtime ------------------------------------------
def kfd_gpu_nodes(sysfs="/sys"):
    """The GPUs in KFD topology order (= HIP device order when no *_VISIBLE_DEVICES filter is set): for each the DRM render
    minor, from which the PCI device directory (NUMA node, local CPUs, clocks, power) is reached."""
    import glob
    nodes = []
    for d in sorted(glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*")), key=lambda x: int(os.path.basename(x))):
        props = {}
        try:
            with open(os.path.join(d, "properties")) as f:
                for ln in f:
                    k, _, v = ln.strip().partition(" ")
                    props[k] = v
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0 and "drm_render_minor" in props:
            nodes.append({"kfd_node": int(os.path.basename(d)), "render_minor": int(props["drm_render_minor"])})
    return nodes


def gpu_device_dir(device, sysfs="/sys"):
    nodes = kfd_gpu_nodes(sysfs)
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if vis:      # a launcher narrowed the devices: index into its list (numeric entries only)
        try:
            order = [int(x) for x in vis.split(",") if x.strip() != ""]
            nodes = [nodes[i] for i in order if 0 <= i < len(nodes)]
        except ValueError:
            return None
    if device < 0 or device >= len(nodes):
        return None
    path = os.path.join(sysfs, "class/drm", "renderD%d" % nodes[device]["render_minor"], "device")
    return path if os.path.isdir(path) else None


def parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def format_cpulist(cpus):
    out, run = [], []
    for c in sorted(cpus) + [None]:
        if run and (c is None or c != run[-1] + 1):
            out.append(str(run[0]) if len(run) == 1 else "%d-%d" % (run[0], run[-1]))
            run = []
        if c is not None:
            run.append(c)
    return ",".join(out)


def gpu_numa(device, sysfs="/sys"):
    """{numa_node, cpus} of the PCI device behind HIP device `device` (None when sysfs does not say)."""
    d = gpu_device_dir(device, sysfs)
    if not d:
        return None
    try:
        with open(os.path.join(d, "numa_node")) as f:
            node = int(f.read().strip())
        with open(os.path.join(d, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        return None
    return {"numa_node": node, "cpus": cpus}


def pin_to_gpu_node(device, sysfs="/sys", setaffinity=None, getaffinity=None):
    """Pin the calling process to the CPUs of the NUMA node its GPU hangs on (a rank issues 3 launches per 0.09 ms step: on a
    two-socket node a launch thread on the far socket pays the inter-socket hop on every doorbell and every completion poll).
    Called before the first HIP call, so that the runtime's own threads inherit the mask.  Returns what it did, for the line."""
    getaffinity = getaffinity or (lambda: os.sched_getaffinity(0))
    setaffinity = setaffinity or (lambda cpus: os.sched_setaffinity(0, cpus))
    info = {"device": device, "numa_node": None, "pinned": False}
    try:
        allowed = set(getaffinity())
    except (AttributeError, OSError):
        return info
    info["cpus_allowed"] = len(allowed)
    nm = gpu_numa(device, sysfs)
    if nm is None:
        return info
    info["numa_node"] = nm["numa_node"]
    want = allowed & nm["cpus"]
    if nm["numa_node"] < 0 or not want or want == allowed:      # one node, or nothing to narrow
        return info
    try:
        setaffinity(want)
        info["pinned"] = True
        info["cpus_allowed"] = len(want)
        info["cpus"] = format_cpulist(want)
    except OSError:
        pass
    return info


def gpu_sensors(device, sysfs="/sys"):
    """sclk / mclk (MHz, the level marked active in pp_dpm_*), socket power (W) and busy % of a GPU from sysfs; {} when unreadable.
    Sampled before and after the timed region and during the sustained run: box-to-box and run-to-run differences of the
    dominant kernel (88-110 us within one profile in round 3) then have a clock next to them."""
    d = gpu_device_dir(device, sysfs)
    out = {}
    if not d:
        return out

    def active_mhz(name):
        try:
            with open(os.path.join(d, name)) as f:
                lines = f.read().strip().splitlines()
        except OSError:
            return None
        cur = [ln for ln in lines if ln.rstrip().endswith("*")] or lines[-1:]
        for tok in cur[0].replace("*", " ").split():
            t = tok.lower()
            if t.endswith("mhz"):
                try:
                    return int(float(t[:-3]))
                except ValueError:
                    pass
        return None

    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
        v = active_mhz(name)
        if v is not None:
            out[key] = v
    import glob
    for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("temp1_input", "temp_c", 1e-3),
                                 ("freq1_input", "sclk_hwmon_mhz", 1e-6)):
            if key in out:
                continue
            try:
                with open(os.path.join(hw, name)) as f:
                    out[key] = round(int(f.read().strip()) * scale, 1)
            except (OSError, ValueError):
                pass
    try:
        with open(os.path.join(d, "gpu_busy_percent")) as f:
            out["busy_pct"] = int(f.read().strip())
    except (OSError, ValueError):
        pass
    return out


class SensorWatch:
    """polls gpu_sensors from a side thread while a measurement runs (the sustained run: >= 1 s)"""

    def __init__(self, device, period_s=0.1):
        self.device, self.period, self.samples, self._stop, self._t = device, period_s, [], threading.Event(), None

    def __enter__(self):
        def loop():
            while not self._stop.is_set():
                s = gpu_sensors(self.device)
                if s:
                    self.samples.append(s)
                self._stop.wait(self.period)
        self._t = threading.Thread(target=loop, name="bench-sensors", daemon=True)
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join()
        return False

    def summary(self):
        if not self.samples:
            return None
        out = {"samples": len(self.samples)}
        for k in ("sclk_mhz", "mclk_mhz", "power_w", "temp_c", "sclk_hwmon_mhz"):
            v = [s[k] for s in self.samples if k in s]
            if v:
                out[k] = {"min": min(v), "max": max(v), "mean": round(sum(v) / len(v), 1)}
        return out


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


def pmc_traffic(workload: str, total_chars: int, in_flight: int, distinct: bool):
    """(HBM bytes per k_tiles_main launch, where the figure comes from) out of the committed PMC passes of the same workload, size
    AND launch scheme (profiles/, collected with rocprofv3 --pmc as MI355X_MICROARCH.md prescribes: separate FETCH_SIZE / WRITE_SIZE
    passes, gfx950 corrections); a counter pass cannot run inside this process, so the figure is null for a workload / size that
    has none.  The pass of the line's own command (batches in flight, one or two input copies) is preferred; any other is named
    as what it is."""
    best, best_score = None, (-1, -1)
    for path in PMC_SUMMARIES:
        try:
            with open(path) as f:
                pmc = json.load(f)
        except Exception:
            continue
        for rec in pmc.get("runs", []):
            if not (rec.get("label") == "bench" and rec.get("workload") == workload and rec.get("kernel") == "k_tiles_main"
                    and rec.get("total_chars") == total_chars):
                continue
            same_flow = int(rec.get("in_flight", 1)) == in_flight
            same_inputs = in_flight < 2 or bool(rec.get("distinct_inputs", 0)) == bool(distinct)
            # (a flow pass holds two instantiations of the kernel: its own and the one-batch warm-up passes'; the one launched more wins)
            score = (2 * same_flow + (same_flow and same_inputs), int(rec.get("launches") or 0))
            if score > best_score:
                best_score = score
                best = (rec.get("hbm_bytes_per_launch"),
                        {"file": os.path.relpath(path, ROOT), "command": rec.get("command"), "in_flight": int(rec.get("in_flight", 1)),
                         "distinct_inputs": bool(rec.get("distinct_inputs", 0)) if "distinct_inputs" in rec else None,
                         "same_command_as_this_line": bool(same_flow and same_inputs), "launches_averaged": rec.get("launches"),
                         "note": "a counter pass runs the dispatches one after the other: bytes per launch, not the overlap, are what it shows"})
    return best if best else (None, None)


class RealApi:
    """What the job runner needs from the product: the C ABI and a context per device.  tests/test_bench_launcher.py
    drives the same runner with a fake of this (no GPU)."""

    def __init__(self):
        self.lib = _lib.load()

    def device_count(self):
        return int(self.lib.latok_device_count())

    def context(self, device):
        return _lib.Context(device)

    def check(self, rc):
        _lib.check(rc)

    def last_error(self):
        return _lib.last_error()


class Shard:
    """One rank's work: an independent contiguous string-id range of the corpus, resident in ITS device's HBM.  Every
    method must run on the host thread whose current context is the rank's (in-process mode: the rank's own thread)."""

    def __init__(self, api, args, rank, world, device=0):
        self.api, self.lib, self.args, self.rank, self.world, self.device = api, api.lib, args, rank, world, device
        model, seed, lo, hi, n_default, _ = WORKLOADS[args.workload]
        if SCALING[args.workload] == "weak":
            self.sid0, self.n_str = shard_string_ids(args.strings or n_default, rank)
        else:
            self.sid0, self.n_str = split_string_ids(args.strings or n_default, rank, world)
        self.total = self.utf8 = 0
        self.d_row = self.d_cps = self.d_bits = self.d_bits2 = self.d_row2 = self.d_cps2 = None
        self.flow = args.in_flight >= 2     # (0 = decided in build() by the size of the rank's batch)
        self.distinct = False

    def build(self):
        """the shard directly in HBM: offsets on the host (8 B/string), code points generated on the device"""
        lib, chk = self.lib, self.api.check
        model, seed, lo, hi, _, _ = WORKLOADS[self.args.workload]
        row = np.zeros(self.n_str + 1, np.int64)
        chk(lib.latok_corpus_offsets(seed, self.sid0, self.n_str, lo, hi, row.ctypes.data))
        self.total = int(row[-1])
        if self.args.in_flight == 0:
            # one huge batch per GPU (C4: 51 GB, C5: 40 GB at N = 1): start-up and end of a 7-9 ms tile kernel are noise and two
            # such kernels in flight only contend (C5 -4 %, C4 +-0 %; 10 M strings = 0.3 M tiles still +3-8 %): one batch at a time
            self.flow = self.total < AUTO_FLOW_MAX_CHARS
        self.d_row = lib.latok_dev_alloc(row.nbytes)
        self.d_cps = lib.latok_dev_alloc(self.total * 4)
        self.d_bits = lib.latok_dev_alloc(((self.total + 63) // 64) * 8)
        # two batches in flight write two bitmasks (consecutive steps alternate between them)
        self.d_bits2 = lib.latok_dev_alloc(((self.total + 63) // 64) * 8) if self.flow else None
        if not (self.d_row and self.d_cps and self.d_bits and (self.d_bits2 or not self.flow)):
            raise RuntimeError(self.api.last_error())
        chk(lib.latok_memcpy_h2d(self.d_row, row.ctypes.data, row.nbytes))
        chk(lib.latok_corpus_fill_device(seed, model, self.sid0, self.n_str, self.d_row, self.d_cps, None))
        u = C.c_int64(0)
        chk(lib.latok_utf8_bytes(self.d_cps, self.total, C.byref(u), _lib.DEVICE_PTRS))
        self.utf8 = int(u.value)
        chk(lib.latok_reserve(self.total, self.n_str))
        if self.flow and not self.args.shared_input:
            # the two batches in flight read two COPIES of the shard at different addresses (as two different batches would):
            # nothing one batch fetched can be an L2 / MALL hit for the other
            self.d_row2 = lib.latok_dev_alloc(row.nbytes)
            self.d_cps2 = lib.latok_dev_alloc(self.total * 4)
            if not (self.d_row2 and self.d_cps2):
                raise RuntimeError(self.api.last_error())
            chk(lib.latok_memcpy_h2d(self.d_row2, row.ctypes.data, row.nbytes))
            chk(lib.latok_corpus_fill_device(seed, model, self.sid0, self.n_str, self.d_row2, self.d_cps2, None))
            chk(lib.latok_bench_set_second_input(self.d_cps2, self.d_row2))
            self.distinct = True

    def warmup(self, n):
        if n > 0:
            self.api.check(self.lib.latok_bench_split_mask(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits, n, 0,
                                                           None, None, None))
        self.api.check(self.lib.latok_sync())

    def warmup_flow(self, n):
        if self.flow and n > 0:
            ms, t0, t1 = C.c_float(0), C.c_int64(0), C.c_int64(0)
            self.api.check(self.lib.latok_bench_split_mask_flow_gated(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits,
                                                                      self.d_bits2, n, None, C.byref(ms), C.byref(t0), C.byref(t1)))

    def settle(self, seconds):
        """untimed passes (the form the timed region uses) until `seconds` of GPU time have gone by: the clocks of a GPU that
        sat idle while the shard was built take tens of milliseconds of load to reach their steady state (the first 20-step
        region after W = 5 warm-up steps runs 2-8 % slower than the fourth: profiles/r03_flow_k.txt).  Returns the step count."""
        done, t_ms = 0, 0.0
        while seconds > 0 and t_ms < seconds * 1e3 and done < 100000:
            k = 50
            ms, t0, t1 = C.c_float(0), C.c_int64(0), C.c_int64(0)
            if self.flow:
                self.api.check(self.lib.latok_bench_split_mask_flow_gated(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits,
                                                                          self.d_bits2, k, None, C.byref(ms), C.byref(t0), C.byref(t1)))
            else:
                self.api.check(self.lib.latok_bench_split_mask_gated(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits, k, None,
                                                                     C.byref(ms), C.byref(t0), C.byref(t1)))
            dt = (t1.value - t0.value) / 1e6
            if dt <= 0:      # (a fake device layer in the tests: no time passes)
                break
            t_ms += dt
            done += k
        return done

    def timed(self, steps, gate, flow=None):
        """EXACTLY `steps` pipeline passes: gate -> host clock -> HIP event -> passes -> HIP event -> stream sync -> host
        clock -> gate, all inside one library call (no interpreter between the clocks).  flow: the passes go through the
        batch flow (latok_flow_split_mask: two batches in flight, alternating between two output bitmasks), every pass
        still launches all three kernels and completes inside the region."""
        flow = self.flow if flow is None else flow
        ms, t0, t1 = C.c_float(0), C.c_int64(0), C.c_int64(0)
        if flow:
            self.api.check(self.lib.latok_bench_split_mask_flow_gated(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits,
                                                                      self.d_bits2, steps, gate, C.byref(ms), C.byref(t0), C.byref(t1)))
            return {"ms_events": float(ms.value), "t0_ns": int(t0.value), "t1_ns": int(t1.value), "graph": False}
        self.api.check(self.lib.latok_bench_split_mask_gated(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits, steps, gate,
                                                             C.byref(ms), C.byref(t0), C.byref(t1)))
        used_graph = getattr(self.lib, "latok_debug_bench_used_graph", None)
        return {"ms_events": float(ms.value), "t0_ns": int(t0.value), "t1_ns": int(t1.value),
                "graph": bool(used_graph()) if used_graph is not None else False}

    def kernel_only(self, steps):
        """the dominant kernel alone: back-to-back launches of k_tiles_main between one HIP event pair -- at least `steps`, and as
        many as ~20 ms of them (<= 200): 20 launches right behind another phase gave 94-101 us on one box within a minute, the
        rocprofv3 average over 10 K launches of the same box 93.7"""
        ms, n_fix = C.c_float(0), C.c_int64(0)
        self.api.check(self.lib.latok_bench_split_mask(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits, 0, max(2, min(steps, 5)), None,
                                                       C.byref(ms), C.byref(n_fix)))
        per = float(ms.value) / max(2, min(steps, 5))
        n = max(steps, min(200, int(20.0 / per) + 1)) if per > 0 else steps
        self.api.check(self.lib.latok_bench_split_mask(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits, 0, n, None,
                                                       C.byref(ms), C.byref(n_fix)))
        self.kernel_launches = n
        return float(ms.value) / n, int(n_fix.value)

    def kernel_in_flow(self, steps):
        """the dominant kernel alone in the flow's launch scheme (two streams, 7/8 of the CUs per launch, launches overlap):
        wall time per launch over `steps` launches"""
        fn = getattr(self.lib, "latok_bench_tiles_flow", None)
        if not self.flow or fn is None:
            return None
        ms = C.c_float(0)
        self.api.check(fn(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits, self.d_bits2, max(steps, 400), C.byref(ms)))
        return float(ms.value) / max(steps, 400)

    def sustained(self, seconds, ms_per_step):
        """>= `seconds` of back-to-back pipeline passes, in chunks of <= 1000 passes (one library call each)"""
        want = max(self.args.steps, int(seconds / (ms_per_step / 1e3)) + 1)
        done, t_ms, chunks = 0, 0.0, []
        while done < want:
            k = min(1000, want - done)
            ms = C.c_float(0)
            if self.flow:
                t0, t1 = C.c_int64(0), C.c_int64(0)
                self.api.check(self.lib.latok_bench_split_mask_flow_gated(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits,
                                                                          self.d_bits2, k, None, C.byref(ms), C.byref(t0), C.byref(t1)))
                ms = C.c_float((t1.value - t0.value) / 1e6)   # wall: the event pair sits on two streams
            else:
                self.api.check(self.lib.latok_bench_split_mask(self.d_cps, self.d_row, self.n_str, self.total, self.d_bits, 0, k,
                                                               C.byref(ms), None, None))
            chunks.append(ms.value / k)
            t_ms += ms.value
            done += k
        return {"steps": done, "seconds": t_ms / 1e3, "ms_per_step": t_ms / done, "in_flight": 2 if self.flow else 1,
                "value": self.utf8 * done / (t_ms / 1e3) / 1e9, "unit": "GB/s (rank 0)",
                "ms_per_step_first_chunk": chunks[0], "ms_per_step_last_chunk": chunks[-1],
                "ms_per_step_min_chunk": min(chunks), "ms_per_step_max_chunk": max(chunks), "chunks": len(chunks)}

    def stream_read(self):
        """streaming-read ceiling of this GPU on the same buffer"""
        nbytes = min((self.total * 4 // 16384) * 16384, 1 << 31)
        if nbytes <= 0:
            return None
        ms = C.c_float(0)
        self.api.check(self.lib.latok_bench_stream_read(self.d_cps, nbytes, 3, 20, C.byref(ms)))
        return nbytes / (ms.value / 20 / 1e3) / 1e9 if ms.value > 0 else None

    def free(self):
        if self.distinct:
            self.lib.latok_bench_set_second_input(None, None)
            self.distinct = False
        for p in (self.d_row, self.d_cps, self.d_bits, self.d_bits2, self.d_row2, self.d_cps2):
            if p:
                self.lib.latok_dev_free(p)
        self.d_row = self.d_cps = self.d_bits = self.d_bits2 = self.d_row2 = self.d_cps2 = None

    def alg_read(self):
        return 4 * self.total + 8 * (self.n_str + 1)   # 4 B/code point + 8 B/string row offset


def measure_shard(sh, args, gate, phase):
    """The measurement protocol of one rank of the in-process job.  `phase()` is a no-op context when every rank has a GPU
    of its own (the gate inside latok_bench_split_mask_gated then starts the timed regions together) and one global lock
    under --take-turns, where ranks share a GPU and must not overlap."""
    lib, chk = sh.lib, sh.api.check
    sh.build()
    sh.warmup(args.warmup)
    sh.warmup_flow(args.warmup)
    dev = getattr(sh, "device", 0)
    sens_before = gpu_sensors(dev)                       # (before the settle passes: nothing may sit between them and the timed region --
    n_settle = sh.settle(args.settle_s)                  # a few ms of sysfs reads there would let the clocks drop again)
    chk(lib.latok_gate_wait(gate, GATE_TIMEOUT_S))      # every shard is resident and warm before anyone's clock starts
    with phase():
        rec = sh.timed(args.steps, None if args.take_turns else gate)
    rec["sensors"] = {"before_settle_passes": sens_before, "after_timed_region": gpu_sensors(dev)}
    serial = None
    if sh.flow:   # the same K steps one batch at a time (what `value` was before the batch flow), beside the headline
        chk(lib.latok_gate_wait(gate, GATE_TIMEOUT_S))
        with phase():
            serial = sh.timed(args.steps, None if args.take_turns else gate, flow=False)
    chk(lib.latok_gate_wait(gate, GATE_TIMEOUT_S))
    with phase():
        k_ms, n_fix = sh.kernel_only(args.steps)
    rec["serial"] = serial
    rec["settle_steps"] = n_settle
    with phase():
        rec["kernel_flow_ms"] = sh.kernel_in_flow(args.steps)
    rec.update(rank=sh.rank, n_str=sh.n_str, total=sh.total, utf8=sh.utf8, alg_read=sh.alg_read(), kernel_ms=k_ms, n_fix=n_fix,
               sustained=None, measured_read=None, distinct_inputs=sh.distinct, flow=sh.flow, kernel_launches=getattr(sh, "kernel_launches", None))
    chk(lib.latok_gate_wait(gate, GATE_TIMEOUT_S))
    if sh.rank == 0:   # outside the timed region, the other ranks are done
        if args.sustain_s > 0 and rec["ms_events"] > 0:
            with SensorWatch(dev) as watch:
                rec["sustained"] = sh.sustained(args.sustain_s, rec["ms_events"] / args.steps)
            rec["sensors"]["during_sustained_run"] = watch.summary()
        rec["measured_read"] = sh.stream_read()
    sh.free()
    return rec


class _NoLock:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def run_in_process(api, args, devices):
    """N > 1 (and N = 1) without a launcher: one context + one host thread per entry of `devices`; the threads meet at a
    gate inside the library, so the timed regions start within microseconds of each other whatever the interpreter does.
    No torch, no RCCL: the shards are independent."""
    world = len(devices)
    if world > 1 and not args.take_turns:
        os.environ.setdefault("LATOK_BENCH_GRAPH", "1")   # N launching threads in one process: replay the K passes as one hipGraph
    gate = C.c_void_p()
    api.check(api.lib.latok_gate_create(world, C.byref(gate)))
    turn_lock = threading.Lock()
    phase = (lambda: turn_lock) if args.take_turns else _NoLock
    results, errors = [None] * world, []

    def body(rank):
        ctx = None
        try:
            pin = pin_to_gpu_node(devices[rank]) if args.pin else {"device": devices[rank], "numa_node": None, "pinned": False}
            ctx = api.context(devices[rank])
            ctx.make_current()
            results[rank] = measure_shard(Shard(api, args, rank, world, devices[rank]), args, gate, phase)
            results[rank]["host"] = pin
        except BaseException as exc:
            errors.append((rank, exc))
            api.lib.latok_gate_break(gate)   # the other ranks fail at their next wait instead of sitting out the timeout
        finally:
            if ctx is not None:
                try:
                    api.lib.latok_ctx_set_current(None)
                    ctx.destroy()
                except Exception:
                    pass

    threads = [threading.Thread(target=body, args=(r,), name=f"bench-rank{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    api.lib.latok_gate_destroy(gate)
    if errors:
        errors.sort(key=lambda e: "gate: broken" in str(e[1]))   # the root cause first, not the ranks it took along
        rank, exc = errors[0]
        raise RuntimeError(f"rank {rank} (device {devices[rank]}) failed: {exc}") from exc
    return results


def run_under_launcher(api, args, rank, world, local_rank):
    """One process per GPU (torchrun): this process is ONE rank.  torch.distributed carries the barriers and the gather of
    the per-rank records only -- gloo by default (nothing here needs RCCL: no collective on the data path)."""
    import torch.distributed as dist
    tdev = None   # the reductions run on host tensors
    # The rank's device, if possible without a HIP call (so that the pin below comes before the runtime starts): the GPUs sysfs lists,
    # unless a launcher narrowed the visible devices or sysfs says nothing -- then the runtime is asked (counting devices does not
    # initialise one).  Never guess: a wrong count would put several ranks on one GPU.
    n_sys = len(kfd_gpu_nodes())
    narrowed = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"))
    n_dev = n_sys if (args.device < 0 and args.pin and n_sys > 0 and not narrowed) else max(1, api.device_count())
    device = args.device if args.device >= 0 else local_rank % n_dev   # a launcher may narrow HIP_VISIBLE_DEVICES per rank
    pin = pin_to_gpu_node(device) if args.pin else {"device": device, "numa_node": None, "pinned": False}   # (before the first HIP call below)
    n_hip = api.device_count()
    if args.device < 0 and n_hip > 0 and n_dev != n_hip:     # sysfs and the runtime disagree: the runtime decides
        device = local_rank % n_hip
    dist.init_process_group(backend="gloo")

    class _Barrier:
        def __init__(self, phase):
            pass

        def __enter__(self):
            if args.take_turns:
                for _ in range(rank):
                    dist.barrier()

        def __exit__(self, *exc):
            if args.take_turns:
                for _ in range(world - rank):
                    dist.barrier()
            return False

    ctx = api.context(device)
    ctx.make_current()
    try:
        sh = Shard(api, args, rank, world, device)
        sh.build()
        sh.warmup(args.warmup)
        sh.warmup_flow(args.warmup)
        sens_before = gpu_sensors(device)
        n_settle = sh.settle(args.settle_s)
        dist.barrier()
        with _Barrier("timed"):
            rec = sh.timed(args.steps, None)
        rec["sensors"] = {"before_settle_passes": sens_before, "after_timed_region": gpu_sensors(device)}
        rec["host"] = pin
        serial = None
        if sh.flow:
            dist.barrier()
            with _Barrier("timed-serial"):
                serial = sh.timed(args.steps, None, flow=False)
        dist.barrier()
        with _Barrier("kernel"):
            k_ms, n_fix = sh.kernel_only(args.steps)
        rec["serial"] = serial
        rec["settle_steps"] = n_settle
        with _Barrier("kernel-flow"):
            rec["kernel_flow_ms"] = sh.kernel_in_flow(args.steps)
        rec.update(rank=rank, n_str=sh.n_str, total=sh.total, utf8=sh.utf8, alg_read=sh.alg_read(), kernel_ms=k_ms, n_fix=n_fix,
                   sustained=None, measured_read=None, distinct_inputs=sh.distinct, flow=sh.flow, kernel_launches=getattr(sh, "kernel_launches", None))
        dist.barrier()
        if rank == 0:
            if args.sustain_s > 0 and rec["ms_events"] > 0:
                with SensorWatch(device) as watch:
                    rec["sustained"] = sh.sustained(args.sustain_s, rec["ms_events"] / args.steps)
                rec["sensors"]["during_sustained_run"] = watch.summary()
            rec["measured_read"] = sh.stream_read()
        sh.free()
        gathered = [None] * world
        dist.all_gather_object(gathered, rec)
        # the contract's reductions, spelled out (max of the per-rank times, sum of the bytes): must agree with the records
        wall_max = reduce_max_seconds(dist, (rec["t1_ns"] - rec["t0_ns"]) / 1e9, tdev)
        assert abs(wall_max - max((g["t1_ns"] - g["t0_ns"]) / 1e9 for g in gathered)) < 1e-9
        return gathered, dist
    finally:
        api.lib.latok_ctx_set_current(None)
        ctx.destroy()


def build_line(args, recs, mode, devices, same_start):
    """the ONE JSON line, from the per-rank records (rank order)"""
    world = len(recs)
    _, _, _, _, _, desc = WORKLOADS[args.workload]
    K = args.steps
    utf8_all = sum(r["utf8"] for r in recs)
    walls = [(r["t1_ns"] - r["t0_ns"]) / 1e9 for r in recs]
    # whole-job wall time: first rank in to last rank out when the ranks share one clock and one start (in-process gate),
    # otherwise the slowest rank's own wall time
    if same_start:
        job_s = (max(r["t1_ns"] for r in recs) - min(r["t0_ns"] for r in recs)) / 1e9
    else:
        job_s = max(walls)
    ev_max = max(r["ms_events"] for r in recs) / 1e3
    flow = all(r.get("serial") for r in recs)
    serial = None
    if flow:   # the same K steps, one batch at a time on one stream
        sr = [r["serial"] for r in recs]
        s_job = ((max(x["t1_ns"] for x in sr) - min(x["t0_ns"] for x in sr)) / 1e9 if same_start
                 else max((x["t1_ns"] - x["t0_ns"]) / 1e9 for x in sr))
        serial = {"ms_per_step": s_job / K * 1e3, "value": utf8_all * K / s_job / 1e9,
                  "ms_per_rank": [x["ms_events"] / K for x in sr],
                  "what": "the same K steps with ONE batch in flight (index -> tiles -> resolve back to back on one stream): "
                          "`value` of rounds 1-3 before the batch flow"}
    traffic, traffic_src = pmc_traffic(args.workload, recs[0]["total"], 2 if flow else 1, bool(flow and all(r.get("distinct_inputs") for r in recs)))
    fracs = [r["alg_read"] / (r["kernel_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS for r in recs]
    worst = min(range(world), key=lambda i: fracs[i])
    r0 = recs[0]
    achieved = recs[worst]["alg_read"] / (recs[worst]["kernel_ms"] / 1e3) / 1e9
    value = utf8_all * K / job_s / 1e9
    line = {
        "metric": "input UTF-8 GB/s tokenized (fused feature+split-mask path)",
        "value": value, "unit": "GB/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup,
        "warmup_effective": {"passes_rank0": args.warmup * (2 if flow else 1) + (r0.get("settle_steps") or 0),
                             "what": "every untimed pass before the timed region on rank 0: W one-batch passes" +
                                     (" + W passes through the flow" if flow else "") + " + the settle passes below (`warmup` is the flag's value)"},
        "warmup_settle": {"seconds": args.settle_s, "steps_rank0": r0.get("settle_steps"),
                          "why": "after the W warm-up steps, untimed passes until the GPU has been under load for this long: a GPU that "
                                 "idled while the shard was built reaches its steady clocks only after tens of ms (outside the timed region)"},
        "ms_per_step": job_s / K * 1e3,
        "higher_is_better": True, "scaling": SCALING[args.workload], "vs_baseline": None,
        "dtype": "u64", "dtype_note": "u32 code points in, 64-bit bit-sliced boolean words, u64 bitmask out (integer / bit ops)",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc}", "strings_per_gpu": r0["n_str"], "strings_total": sum(r["n_str"] for r in recs),
                   "chars_total": sum(r["total"] for r in recs), "utf8_bytes_total": utf8_all,
                   "sharding": f"{world} x independent contiguous string-id shards, no collective on the data path",
                   "launch": mode, "devices": devices},
        "in_flight": 2 if flow else 1,
        "distinct_inputs": bool(flow and all(r.get("distinct_inputs") for r in recs)),
        "ranks": [{"rank": r["rank"], **(r.get("host") or {})} for r in recs],
        "sensors_rank0": r0.get("sensors"),
        "in_flight_note": ("batch flow (latok_flow_split_mask): step i+1 is submitted while step i runs, on a second stream with its own "
                           "workspace and its own output bitmask; every step launches all three kernels and is complete when the "
                           "clock stops; with `distinct_inputs` the odd steps read a second copy of the shard at another address, so the two "
                           "batches in flight share no input lines in L2 / MALL; `serial` = the same K steps one at a time") if flow else "one batch at a time",
        "serial": serial,
        "timing": ("host monotonic clock, inputs resident in HBM: every rank's region = [gate ->] clock -> K pipeline passes -> "
                   "stream(s) synchronise -> clock inside one library call; value = bytes of all ranks x K / "
                   + ("(last rank out - first rank in)" if same_start else "max over ranks of the rank's own wall time")
                   + "; the HIP-event time of the same K steps rides along as ms_per_step_events / value_events"),
        "ms_per_step_events": ev_max / K * 1e3,
        "value_events": utf8_all * K / ev_max / 1e9,
        "ms_per_rank": [r["ms_events"] / K for r in recs],
        "ms_per_rank_wall": [w / K * 1e3 for w in walls],
        "start_skew_us": (max(r["t0_ns"] for r in recs) - min(r["t0_ns"] for r in recs)) / 1e3 if same_start else None,
        "timed_region_launches": ("one hipGraph replay of the K passes per rank" if all(r.get("graph") for r in recs)
                                  else "3 kernel launches per pass" + (", passes alternating between two streams" if flow else "")),
        "sustained": r0["sustained"],
        "fix_tiles_rank0": r0["n_fix"], "tiles_rank0": (r0["total"] + _lib.TILE_CHARS - 1) // _lib.TILE_CHARS,
        "roofline": {"bound": "hbm", "kernel": "k_tiles_main", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": fracs[worst], "traffic": traffic, "traffic_source": traffic_src,
                     "alg_bytes_per_launch": recs[worst]["alg_read"], "kernel_ms": recs[worst]["kernel_ms"],
                     "kernel_timing": f"{r0.get('kernel_launches') or K} back-to-back launches between one HIP event pair on the launch stream, per rank; "
                                      "frac / achieved = the SLOWEST rank's",
                     "frac_per_rank": fracs, "kernel_ms_per_rank": [r["kernel_ms"] for r in recs],
                     "in_flow": ({"kernel_ms_per_launch": max(r["kernel_flow_ms"] for r in recs),
                                  "frac": min(r["alg_read"] / (r["kernel_flow_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS for r in recs),
                                  "how": "max(K, 400) launches of k_tiles_main ALONE in the flow's launch scheme (alternating between the "
                                         "two slot streams, each launch planned for 7/8 of the CUs, no other kernel): host wall time / launches; "
                                         "the launches overlap their start-up and ragged end, so this is the kernel's average cost per launch in "
                                         "the product's scheme, not one launch's duration (that is kernel_ms above, which rocprofv3 confirms)"}
                                 if all(r.get("kernel_flow_ms") for r in recs) else None),
                     "pipeline_frac": min(r["alg_read"] / (r["ms_events"] / K / 1e3) / 1e9 / HBM_PEAK_GBS for r in recs),
                     "pipeline_frac_sustained": (r0["alg_read"] / (r0["sustained"]["ms_per_step"] / 1e3) / 1e9 / HBM_PEAK_GBS) if r0.get("sustained") else None,
                     "pipeline_note": ("pipeline_frac = algorithmic bytes / per-step time of the K timed steps (includes filling and draining the "
                                       "two-batch flow); _sustained = the same over the >= 1 s run.  In the flow the tile kernels of consecutive "
                                       "batches overlap (each is planned for 7/8 of the CUs), so a step can take less than one isolated launch "
                                       "of the kernel (`kernel_ms`, all CUs, nothing else on the GPU)") if flow else None,
                     "measured_stream_read": r0["measured_read"],
                     "frac_of_measured_read": (r0["alg_read"] / (r0["kernel_ms"] / 1e3) / 1e9 / r0["measured_read"]) if r0["measured_read"] else None},
    }
    if args.take_turns:
        line["rehearsal"] = ("ranks took turns on shared GPU(s): per-rank times are single-GPU times; value_projected is what "
                             f"{world} such GPUs would give -- a rehearsal of the N > 1 code path, not a measurement of {world} GPUs")
        proj = max(walls)
        line["value_projected"] = utf8_all * K / proj / 1e9
        line["ms_per_step_projected"] = proj / K * 1e3
        line["value"] = None
        line["ms_per_step"] = None
        if serial:
            serial["value"] = serial["ms_per_step"] = None
            serial["ms_per_step_projected"] = max((x["t1_ns"] - x["t0_ns"]) / 1e9 for x in sr) / K * 1e3
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--strings", type=int, default=0, help="strings per GPU (weak workloads) / in total (strong); default: the workload's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-strings", type=int, default=1_000_000)
    ap.add_argument("--sustain-s", type=float, default=1.0, help="length of the sustained run after the timed region (0 = off)")
    ap.add_argument("--devices", default="", help="in-process mode: comma-separated HIP device of each rank (default 0..N-1); "
                                                  "a device may repeat (rehearsal on fewer GPUs)")
    ap.add_argument("--dist-backend", default="gloo", choices=["gloo"],
                    help="under torchrun only: what carries the barriers and the gather of the report.  The data path has no "
                         "collective and RCCL is not linked, so there is nothing for an RCCL backend to do")
    ap.add_argument("--device", type=int, default=-1, help="under torchrun only: HIP device of this rank (default LOCAL_RANK mod device count)")
    ap.add_argument("--launch", default="auto", choices=["auto", "procs", "threads"],
                    help="without a launcher, how the N ranks run: procs = one child PROCESS per GPU, spawned by this script before "
                         "it touches a GPU (every rank has a HIP runtime of its own: nothing in one rank's launch path can wait "
                         "for another's), threads = N contexts + N host threads in this process; auto = procs for N > 1")
    ap.add_argument("--child-rank", type=int, default=-1, help=argparse.SUPPRESS)      # set by the parent of --launch procs
    ap.add_argument("--child-gate", default="", help=argparse.SUPPRESS)
    ap.add_argument("--child-lock", default="", help=argparse.SUPPRESS)
    ap.add_argument("--child-api", default="", help=argparse.SUPPRESS)               # tests: module:factory of a fake device layer
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2],
                    help="batches in flight per GPU in the timed region: 2 = the batch flow (latok_flow_split_mask), 1 = one batch at "
                         "a time; 0 (default) = by the size of the rank's batch (the flow below 4 G chars per GPU: C2, C3, the "
                         "shards of C4 / C5 from 4 / 3 GPUs on), 1 for N > 1 host threads in one process (--launch threads)")
    ap.add_argument("--settle-s", type=float, default=0.05,
                    help="untimed passes after the W warm-up steps until the GPU has been busy this long (clock settling; 0 = off)")
    ap.add_argument("--shared-input", action="store_true",
                    help="batch flow: both batches in flight read the SAME input buffer (rounds 3's form); default: each slot reads "
                         "its own copy of the shard at another address")
    ap.add_argument("--no-pin", dest="pin", action="store_false",
                    help="do not pin a rank to the CPUs of its GPU's NUMA node (default: pinned, before the first HIP call)")
    ap.add_argument("--take-turns", action="store_true",
                    help="rehearsal on shared GPU(s): the ranks run their timed regions one after the other, so each rank's "
                         "time is what a GPU of its own would give; the line is marked as a rehearsal and carries no `value`")
    args = ap.parse_args(argv)
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error("--gpus and --steps must be >= 1, --warmup >= 0")
    if args.in_flight == 0:
        threads_n = args.gpus > 1 and args.launch == "threads" and args.child_rank < 0 and not args.take_turns
        # (0 stays 0 = by batch size, see Shard.build: a rank whose batch is huge runs one batch at a time)
        args.in_flight = 1 if threads_n else 0
    return args


def pick_devices(args, n_dev):
    """in-process mode: the HIP device of each of the N ranks"""
    if args.devices:
        devices = [int(x) for x in args.devices.split(",") if x != ""]
        if len(devices) != args.gpus:
            raise SystemExit(f"bench.py: --devices lists {len(devices)} devices for --gpus {args.gpus}")
    else:
        devices = list(range(args.gpus))
    bad = [d for d in devices if d < 0 or d >= n_dev]
    if bad:
        raise SystemExit(f"bench.py: device(s) {bad} not present ({n_dev} HIP device(s) visible); "
                         f"use --devices (a device may repeat) to rehearse --gpus {args.gpus} on fewer GPUs")
    return devices


class _FileLock:
    """--take-turns across processes: one flock for every GPU phase"""

    def __init__(self, path):
        self.path, self.f = path, None

    def __enter__(self):
        import fcntl
        self.f = open(self.path, "a+")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def child_main(args, api):
    """One rank of --launch procs: this process owns ONE GPU.  Rank 0 creates the shared-memory gate, the others attach; the
    record goes to stdout as one line for the parent."""
    rank, world = args.child_rank, args.gpus
    devices = [int(x) for x in args.devices.split(",")]
    # first of all, before anything can start the HIP runtime (and its helper threads): onto the cores next to this rank's GPU
    pin = pin_to_gpu_node(devices[rank]) if args.pin else {"device": devices[rank], "numa_node": None, "pinned": False}
    lib = api.lib
    gate = C.c_void_p()
    name = args.child_gate.encode()
    if rank == 0:
        api.check(lib.latok_gate_create_shared(name, world, C.byref(gate)))
    else:
        t_end = time.time() + 60.0
        while lib.latok_gate_attach_shared(name, C.byref(gate)) != 0:      # rank 0 has not created it yet
            if time.time() > t_end:
                raise RuntimeError("rank %d: the gate %s never appeared" % (rank, args.child_gate))
            time.sleep(0.01)
    try:
        n_dev = api.device_count()
        if devices[rank] < 0 or devices[rank] >= n_dev:
            raise RuntimeError(f"device {devices[rank]} not present ({n_dev} HIP device(s) visible); use --devices "
                               f"(a device may repeat) to rehearse --gpus {world} on fewer GPUs")
        ctx = api.context(devices[rank])
        ctx.make_current()
        phase = (lambda: _FileLock(args.child_lock)) if args.take_turns else _NoLock
        rec = measure_shard(Shard(api, args, rank, world, devices[rank]), args, gate, phase)
        rec["host"] = pin
        lib.latok_ctx_set_current(None)
        ctx.destroy()
        print("LATOK_BENCH_REC " + json.dumps(rec), flush=True)
        return 0
    except BaseException:
        lib.latok_gate_break(gate)     # the other ranks fail at their next wait instead of sitting out the timeout
        raise
    finally:
        lib.latok_gate_detach_shared(gate)


def run_processes(args, devices, argv):
    """N > 1 without a launcher, one child process per GPU.  The parent has NOT touched a GPU (nor loaded the library) when
    it spawns them -- a process that has initialised the GPU must not start other programs on this pool -- and never does
    afterwards either, except for host-only calls (CPU baseline, unlinking the gate)."""
    import subprocess
    import tempfile
    world = len(devices)
    gate_name = "/latok_bench_%d_%d" % (os.getpid(), int(time.time() * 1e3) & 0xFFFFFF)
    lock = tempfile.NamedTemporaryFile(prefix="latok_bench_lock_", delete=False)
    lock.close()
    base = [a for a in (argv if argv is not None else sys.argv[1:])]
    cmd0 = [sys.executable, os.path.abspath(__file__)] + base + ["--devices", ",".join(map(str, devices)), "--child-gate", gate_name,
                                                                   "--child-lock", lock.name, "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # Every child writes into FILES of its own: with pipes drained one child after the other, a rank that fills its pipe (HIP
    # warnings, AMD_LOG_LEVEL, a long traceback) blocks in write() before it reaches the gate and takes every other rank into
    # the gate's timeout, with a misleading "gate" error on top.
    files = [(tempfile.TemporaryFile(mode="w+", prefix="latok_bench_out_"), tempfile.TemporaryFile(mode="w+", prefix="latok_bench_err_"))
             for _ in range(world)]
    procs = [subprocess.Popen(cmd0 + ["--child-rank", str(r)], stdout=files[r][0], stderr=files[r][1], text=True, env=env)
             for r in range(world)]
    outs, deadline = [], time.time() + CHILD_TIMEOUT_S
    for r, p in enumerate(procs):
        note = ""
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()                       # (this child, by its own handle: never by pattern)
            p.wait()
            note = f"\nbench.py: rank killed after {CHILD_TIMEOUT_S:.0f} s"
        texts = []
        for f in files[r]:
            f.seek(0)
            texts.append(f.read())
            f.close()
        outs.append((texts[0], texts[1] + note))
    try:
        os.unlink(lock.name)
    except OSError:
        pass
    lib = _lib.load()                      # (after the children are gone)
    lib.latok_gate_unlink_shared(gate_name.encode())
    recs, errors = [None] * world, []
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        line = [ln for ln in so.splitlines() if ln.startswith("LATOK_BENCH_REC ")]
        if p.returncode == 0 and line:
            recs[r] = json.loads(line[-1][len("LATOK_BENCH_REC "):])
        else:
            errors.append((r, p.returncode, (se.strip().splitlines() or ["(no output)"])[-1]))
    if errors:
        errors.sort(key=lambda e: "gate: broken" in e[2])       # the root cause first
        r, rc, msg = errors[0]
        raise SystemExit(f"bench.py: rank {r} (device {devices[r]}) failed with exit code {rc}: {msg}")
    return recs


def load_api(spec):
    """tests only: 'module:factory' of a fake device layer for the children of --launch procs"""
    import importlib
    mod, attr = spec.split(":")
    return getattr(importlib.import_module(mod), attr)()


def main(argv=None, api=None, out=None):
    args = parse_args(argv)
    if args.child_rank >= 0:              # one rank of --launch procs
        return child_main(args, api or (load_api(args.child_api) if args.child_api else RealApi()))
    env_world = os.environ.get("WORLD_SIZE", "")
    dist = None
    rank = 0
    if env_world != "" and int(env_world) > 1:
        # started by a launcher, one process per GPU (python -m torch.distributed.run ... bench.py --gpus N)
        api = api or RealApi()
        rank, world, local_rank = int(os.environ.get("RANK", "0")), int(env_world), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            if rank == 0:
                print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
            sys.exit(2)
        recs, dist = run_under_launcher(api, args, rank, world, local_rank)
        mode, devices, same_start = f"one process per GPU (launcher, {args.dist_backend} for barriers only)", None, False
    elif args.gpus > 1 and args.launch in ("auto", "procs") and api is None:
        # no launcher, N > 1: one child process per GPU, spawned before this process touches a GPU
        devices = [int(x) for x in args.devices.split(",") if x != ""] if args.devices else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit(f"bench.py: --devices lists {len(devices)} devices for --gpus {args.gpus}")
        recs = run_processes(args, devices, argv)
        mode = "one process per GPU, spawned by bench.py (no launcher, no torch, no RCCL; shared-memory gate)"
        same_start = not args.take_turns
    else:
        # N contexts + N host threads in THIS process (N = 1; --launch threads)
        api = api or RealApi()
        devices = pick_devices(args, api.device_count())
        recs = run_in_process(api, args, devices)
        mode, same_start = "one process, one context + host thread per GPU (no launcher, no torch, no RCCL)", not args.take_turns
    if rank == 0:
        line = build_line(args, recs, mode, devices, same_start)
        if not args.no_cpu_baseline:   # next to the GPU number at every N: the host's cores, same corpus (rank 0's shard head)
            n_cpu = min(args.cpu_strings, recs[0]["n_str"], 1_000_000 if args.workload != "C5" else 64)
            line.update(cpu_baselines("C2" if args.workload == "C4" else args.workload, n_cpu))
        print(json.dumps(line), file=out or sys.stdout, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
