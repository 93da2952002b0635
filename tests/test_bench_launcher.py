"""bench.py's own launcher (one context + one host thread per GPU, in ONE process, no torch / RCCL) driven on
CPU with a FAKE device layer: the real library supplies everything that needs no GPU (corpus offsets, the gate), the fake
supplies the device calls.  What is tested is what the launcher adds: device selection, one thread per rank, every rank on
its own context, disjoint shards, the gate (timed regions overlap; --take-turns: they do not), the JSON line of every N.
"""
import ctypes as C
import io
import json
import os
import subprocess
import sys
import threading
import time

import pytest

from conftest import ROOT

import bench
from latok_amd import _lib


class FakeLib:
    """the device half of the C ABI, faked; everything else is forwarded to the real library"""

    def __init__(self, n_dev, ms_per_pass=2.0, fail_on_device=None):
        self.real = _lib.load()
        self.n_dev = n_dev
        self.ms_per_pass = ms_per_pass
        self.fail_on_device = fail_on_device
        self.tl = threading.local()          # the fake "current context" of each thread
        self.lock = threading.Lock()
        self.next_ptr = 0x1000
        self.live = {}                       # ptr -> (device, bytes)
        self.fills = []                      # (device, sid0, n_str)
        self.timed = []                      # (device, thread, t0, t1) of the headline region of every rank
        self.timed_serial = []               # ... of the `serial` region behind it (batch flow on)
        self.flow_calls = 0
        self.second_inputs = []              # (device, cps_b, row_b) of every latok_bench_set_second_input

    def __getattr__(self, name):            # gate functions, corpus offsets: the real thing
        return getattr(self.real, name)

    def _dev(self):
        return self.tl.ctx.device

    def latok_device_count(self):
        return self.n_dev

    def latok_ctx_set_current(self, h):
        return 0

    def latok_dev_alloc(self, nbytes):
        with self.lock:
            self.next_ptr += 0x1000
            self.live[self.next_ptr] = (self._dev(), nbytes)
            return self.next_ptr

    def latok_dev_free(self, p):
        with self.lock:
            assert self.live.pop(p)[0] == self._dev(), "freed on another context than allocated"
        return 0

    def latok_memcpy_h2d(self, d, s, n):
        return 0

    def latok_corpus_fill_device(self, seed, model, sid0, n_str, d_row, d_cps, stream):
        if self.fail_on_device is not None and self._dev() == self.fail_on_device:
            return _lib.ERR_HIP
        with self.lock:
            self.fills.append((self._dev(), sid0, n_str))
        return 0

    def latok_utf8_bytes(self, cps, n, out, flags):
        out._obj.value = n       # ASCII corpus: one byte per char
        return 0

    def latok_reserve(self, a, b):
        return 0

    def latok_sync(self):
        return 0

    def latok_bench_split_mask(self, cps, row, n_str, total, bits, warmup, iters, ms_total, ms_tiles, n_fix):
        if ms_total is not None:
            ms_total._obj.value = self.ms_per_pass * iters
        if ms_tiles is not None:
            ms_tiles._obj.value = 0.9 * self.ms_per_pass * iters
        if n_fix is not None:
            n_fix._obj.value = 7
        return 0

    def _region(self, iters, gate, ms, t0, t1, sink):
        rc = self.real.latok_gate_wait(gate, C.c_double(20.0))
        if rc:
            return rc
        a = time.monotonic_ns()
        time.sleep(self.ms_per_pass * iters / 1e3)
        b = time.monotonic_ns()
        rc = self.real.latok_gate_wait(gate, C.c_double(20.0))
        ms._obj.value = (b - a) / 1e6
        t0._obj.value, t1._obj.value = a, b
        with self.lock:
            sink.append((self._dev(), threading.get_ident(), a, b))
        return rc

    def latok_bench_split_mask_gated(self, cps, row, n_str, total, bits, iters, gate, ms, t0, t1):
        # with the batch flow on (the default) this is the `serial` region that follows the headline one
        return self._region(iters, gate, ms, t0, t1, self.timed_serial if self.flow_calls else self.timed)

    def latok_bench_split_mask_flow_gated(self, cps, row, n_str, total, bits_a, bits_b, iters, gate, ms, t0, t1):
        assert bits_a and bits_b and bits_a != bits_b, "the flow alternates between two output bitmasks"
        if iters <= 1:      # the flow's warm-up pass (BASE: --warmup 1, --steps 5)
            ms._obj.value = 0.0
            return 0
        with self.lock:
            self.flow_calls += 1
        return self._region(iters, gate, ms, t0, t1, self.timed)

    def latok_bench_set_second_input(self, cps_b, row_b):
        with self.lock:
            self.second_inputs.append((self._dev(), cps_b, row_b))
        return 0

    def latok_bench_tiles_flow(self, cps, row, n_str, total, bits_a, bits_b, iters, ms):
        ms._obj.value = 0.8 * self.ms_per_pass * iters
        return 0

    def latok_bench_stream_read(self, buf, nbytes, warmup, iters, ms):
        ms._obj.value = 1.0
        return 0


class FakeApi:
    def __init__(self, n_dev, **kw):
        self.lib = FakeLib(n_dev, **kw)
        self.contexts = []

    def device_count(self):
        return self.lib.n_dev

    def context(self, device):
        api = self

        class Ctx:
            def __init__(self):
                self.device = device
                self.thread = None
                self.destroyed = False

            def make_current(self):
                self.thread = threading.get_ident()
                api.lib.tl.ctx = self

            def destroy(self):
                self.destroyed = True
        c = Ctx()
        self.contexts.append(c)
        return c

    def check(self, rc):
        if rc:
            raise RuntimeError(f"rc {rc}: {self.last_error()}")

    def last_error(self):
        return _lib.last_error()


def _run(argv, api):
    out = io.StringIO()
    assert bench.main(argv, api=api, out=out) == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly ONE JSON line"
    return json.loads(lines[0])


BASE = ["--steps", "5", "--warmup", "1", "--strings", "2000", "--no-cpu-baseline", "--sustain-s", "0", "--settle-s", "0"]


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_in_process_launch_runs_every_rank_on_its_own_context(n, monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    api = FakeApi(n_dev=8)
    line = _run(["--gpus", str(n)] + BASE, api)
    assert line["n_gpus"] == n and line["steps"] == 5 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["devices"] == list(range(n))
    # one context per rank, each made current on a thread of its own, all destroyed, nothing leaked
    assert [c.device for c in sorted(api.contexts, key=lambda c: c.device)] == list(range(n))
    assert len({c.thread for c in api.contexts}) == n and all(c.destroyed for c in api.contexts)
    assert not api.lib.live
    # disjoint contiguous shards of the corpus: rank r owns string ids [r * 2000, (r + 1) * 2000) on device r
    # (filled twice: the two batches in flight read two copies of the shard at different addresses, installed and removed again)
    assert sorted(api.lib.fills) == sorted([(r, r * 2000, 2000) for r in range(n)] * 2)
    assert line["distinct_inputs"] is True
    for r in range(n):
        mine = [x for x in api.lib.second_inputs if x[0] == r]
        assert len(mine) == 2 and mine[0][1] and mine[0][2] and mine[0][1] != mine[0][2] and mine[1][1:] == (None, None)
    assert line["config"]["strings_total"] == 2000 * n and line["config"]["strings_per_gpu"] == 2000
    assert line["warmup_effective"]["passes_rank0"] == 2 and [x["rank"] for x in line["ranks"]] == list(range(n))
    assert line["roofline"]["traffic_source"] is None or "file" in line["roofline"]["traffic_source"]
    # the gate: all timed regions overlap (started together), so the job took about one rank's time, not n of them
    t0s, t1s = [t[2] for t in api.lib.timed], [t[3] for t in api.lib.timed]
    assert max(t0s) < min(t1s)
    assert len(line["ms_per_rank"]) == n == len(line["ms_per_rank_wall"]) == len(line["roofline"]["frac_per_rank"])
    job_ms = line["ms_per_step"] * 5
    assert job_ms == pytest.approx((max(t1s) - min(t0s)) / 1e6, rel=1e-6)
    assert job_ms < 1.6 * max(line["ms_per_rank_wall"]) * 5
    assert line["value"] == pytest.approx(line["config"]["utf8_bytes_total"] * 5 / (job_ms / 1e3) / 1e9, rel=1e-6)
    assert line["roofline"]["frac"] == min(line["roofline"]["frac_per_rank"]) and line["fix_tiles_rank0"] == 7
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and line["data"] == "synthetic"
    # the headline region went through the batch flow (two batches in flight), the same K steps one at a time ride along
    assert line["in_flight"] == 2 and api.lib.flow_calls == n and len(api.lib.timed_serial) == n
    assert line["serial"]["value"] > 0 and len(line["serial"]["ms_per_rank"]) == n
    assert line["roofline"]["in_flow"]["frac"] > line["roofline"]["frac"]      # (the fake: 0.8 vs 0.9 of a pass)


def test_in_flight_1_times_one_batch_at_a_time(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    api = FakeApi(n_dev=2)
    line = _run(["--gpus", "2", "--in-flight", "1"] + BASE, api)
    assert line["in_flight"] == 1 and line["serial"] is None and api.lib.flow_calls == 0 and len(api.lib.timed) == 2


def test_devices_may_repeat_and_take_turns_serialises(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    api = FakeApi(n_dev=1, ms_per_pass=4.0)
    line = _run(["--gpus", "3", "--devices", "0,0,0", "--take-turns"] + BASE, api)
    assert line["n_gpus"] == 3 and line["config"]["devices"] == [0, 0, 0]
    spans = sorted((t[2], t[3]) for t in api.lib.timed)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(2)), "timed regions must not overlap under --take-turns"
    assert line["value"] is None and "rehearsal" in line and line["value_projected"] > 0
    assert len(line["ms_per_rank"]) == 3


def test_strong_scaling_workload_cuts_the_whole_batch(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    api = FakeApi(n_dev=4)
    line = _run(["--gpus", "4", "--workload", "C4", "--steps", "2", "--warmup", "0", "--strings", "1001", "--no-cpu-baseline",
                 "--sustain-s", "0"], api)
    fills = sorted(set(api.lib.fills), key=lambda f: f[1])      # (each shard is filled twice: the flow's two input copies)
    assert fills[0][1] == 0 and sum(f[2] for f in fills) == 1001
    assert all(fills[i][1] + fills[i][2] == fills[i + 1][1] for i in range(3))
    assert line["scaling"] == "strong" and line["config"]["strings_total"] == 1001


def test_missing_devices_are_refused_with_advice(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"] + BASE, api=FakeApi(n_dev=1), out=io.StringIO())
    assert "--devices" in str(e.value)
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "2", "--devices", "0"] + BASE, api=FakeApi(n_dev=1), out=io.StringIO())


def test_a_failing_rank_fails_the_job_quickly(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    t = time.time()
    with pytest.raises(RuntimeError, match=r"rank 2 \(device 2\)"):
        bench.main(["--gpus", "4"] + BASE, api=FakeApi(n_dev=4, fail_on_device=2), out=io.StringIO())
    assert time.time() - t < 30, "the other ranks must not sit out the gate timeout"


def test_gate_is_a_reusable_rendezvous():
    lib = _lib.load()
    g = C.c_void_p()
    assert lib.latok_gate_create(4, C.byref(g)) == 0
    order = []

    def party(i):
        for rnd in range(3):
            time.sleep(0.01 * i)
            order.append(("in", rnd))
            assert lib.latok_gate_wait(g, 10.0) == 0
            order.append(("out", rnd))
    th = [threading.Thread(target=party, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for rnd in range(3):   # nobody leaves round r before everybody has entered it
        last_in = max(i for i, e in enumerate(order) if e == ("in", rnd))
        first_out = min(i for i, e in enumerate(order) if e == ("out", rnd))
        assert last_in < first_out
    assert lib.latok_gate_break(g) == 0 and lib.latok_gate_wait(g, 1.0) == _lib.ERR_INVALID
    assert lib.latok_gate_destroy(g) == 0


def test_without_a_gpu_the_driver_spelling_fails_loudly_not_with_a_usage_error():
    """`python3 bench.py --gpus 2` with no launcher and no WORLD_SIZE: round 2 exited 2 with 'launch one rank per GPU'.
    Now it starts the in-process job; in this container that stops at the device check with a clear message."""
    if _lib.load().latok_device_count() > 0:
        pytest.skip("needs a box without GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "HIP device(s) visible" in p.stderr and "launch one rank per GPU" not in p.stderr


def fake_api():
    """factory for the children of `bench.py --launch procs --child-api test_bench_launcher:fake_api`"""
    return FakeApi(n_dev=8, ms_per_pass=3.0)


def _run_procs(extra, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, env.get("PYTHONPATH", "")])
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--child-api", "test_bench_launcher:fake_api"] + BASE + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_process_per_gpu_launch_is_the_default_for_n_gt_1():
    """`python3 bench.py --gpus N` with no launcher: the script spawns one child process per GPU itself (before it touches a
    GPU), the children meet at a gate in shared memory and the parent prints the ONE line.  Device layer faked in the
    children; processes, gate, clocks and the line are real."""
    p = _run_procs(["--gpus", "3"])
    assert p.returncode == 0, p.stderr[-1500:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["config"]["devices"] == [0, 1, 2] and "spawned by bench.py" in line["config"]["launch"]
    assert line["config"]["strings_total"] == 6000 and len(line["ms_per_rank"]) == 3 == len(line["roofline"]["frac_per_rank"])
    # gated start across processes (one monotonic clock per host): the job took about one rank's time
    assert line["start_skew_us"] is not None and line["start_skew_us"] < 50_000
    assert line["ms_per_step"] * 5 < 1.6 * max(line["ms_per_rank_wall"]) * 5
    assert line["value"] == pytest.approx(line["config"]["utf8_bytes_total"] * 5 / (line["ms_per_step"] * 5 / 1e3) / 1e9, rel=1e-6)


def test_process_per_gpu_take_turns_and_failure():
    p = _run_procs(["--gpus", "2", "--devices", "0,0", "--take-turns"])
    assert p.returncode == 0, p.stderr[-1500:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["value"] is None and line["value_projected"] > 0 and line["config"]["devices"] == [0, 0]
    # a rank whose device does not exist: the job fails quickly and says which rank and why
    t = time.time()
    p = _run_procs(["--gpus", "2", "--devices", "0,11"])
    assert p.returncode != 0 and "rank 1 (device 11)" in p.stderr and "not present" in p.stderr
    assert time.time() - t < 60


def _fake_sysfs(root, gpus):
    """gpus: list of (kfd node, render minor, numa node, local cpulist); plus one CPU-only KFD node in front (as on real hosts)"""
    def put(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    put(os.path.join(root, "class/kfd/kfd/topology/nodes/0/properties"), "cpu_cores_count 96\nsimd_count 0\ndrm_render_minor 0\n")
    for node, minor, numa, cpus in gpus:
        put(os.path.join(root, f"class/kfd/kfd/topology/nodes/{node}/properties"), f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\n")
        dev = os.path.join(root, f"class/drm/renderD{minor}/device")
        put(os.path.join(dev, "numa_node"), f"{numa}\n")
        put(os.path.join(dev, "local_cpulist"), cpus + "\n")
        put(os.path.join(dev, "pp_dpm_sclk"), "0: 132Mhz\n1: 2100Mhz *\n2: 2400Mhz\n")
        put(os.path.join(dev, "pp_dpm_mclk"), "0: 900Mhz\n1: 2000Mhz *\n")
        put(os.path.join(dev, "hwmon/hwmon3/power1_average"), "612000000\n")
        put(os.path.join(dev, "gpu_busy_percent"), "97\n")


def test_rank_is_pinned_to_the_cpus_of_its_gpus_numa_node(tmp_path, monkeypatch):
    """bench.pin_to_gpu_node: HIP device -> KFD topology order -> DRM render node -> numa_node / local_cpulist, all from sysfs (no
    HIP call: the pin happens before the runtime starts), on a fake two-socket host with 8 GPUs."""
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    root = str(tmp_path)
    gpus = [(2 + i, 128 + i, 0 if i < 4 else 1, "0-47,96-143" if i < 4 else "48-95,144-191") for i in range(8)]
    _fake_sysfs(root, gpus)
    assert [g["render_minor"] for g in bench.kfd_gpu_nodes(root)] == [128 + i for i in range(8)]
    everything = set(range(192))
    for dev in range(8):
        got = {}
        info = bench.pin_to_gpu_node(dev, sysfs=root, setaffinity=lambda cpus: got.update(cpus=set(cpus)), getaffinity=lambda: everything)
        want = set(range(0, 48)) | set(range(96, 144)) if dev < 4 else set(range(48, 96)) | set(range(144, 192))
        assert info["pinned"] and info["numa_node"] == (0 if dev < 4 else 1) and info["cpus_allowed"] == 96 and got["cpus"] == want
    # a cgroup that already narrowed the CPUs: only the intersection; nothing left of it: no pin
    got = {}
    info = bench.pin_to_gpu_node(5, sysfs=root, setaffinity=lambda cpus: got.update(cpus=set(cpus)), getaffinity=lambda: set(range(40, 56)))
    assert info["pinned"] and got["cpus"] == set(range(48, 56)) and info["cpus"] == "48-55"
    info = bench.pin_to_gpu_node(5, sysfs=root, setaffinity=lambda cpus: got.update(bad=True), getaffinity=lambda: set(range(0, 16)))
    assert not info["pinned"] and "bad" not in got and info["numa_node"] == 1
    # one NUMA node (or numa_node = -1), a device sysfs does not know, a launcher's HIP_VISIBLE_DEVICES
    _fake_sysfs(root + "/one", [(1, 128, -1, "0-15")])
    assert not bench.pin_to_gpu_node(0, sysfs=root + "/one", setaffinity=lambda c: got.update(bad=True), getaffinity=lambda: set(range(16)))["pinned"]
    assert bench.pin_to_gpu_node(9, sysfs=root, setaffinity=lambda c: got.update(bad=True), getaffinity=lambda: everything)["numa_node"] is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6,1")
    assert bench.pin_to_gpu_node(0, sysfs=root, setaffinity=lambda c: None, getaffinity=lambda: everything)["numa_node"] == 1
    assert bench.pin_to_gpu_node(1, sysfs=root, setaffinity=lambda c: None, getaffinity=lambda: everything)["numa_node"] == 0
    assert "bad" not in got
    # the clocks / power sample of the same device
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    s = bench.gpu_sensors(3, sysfs=root)
    assert s == {"sclk_mhz": 2100, "mclk_mhz": 2000, "power_w": 612.0, "busy_pct": 97}
    assert bench.gpu_sensors(0, sysfs=str(tmp_path / "nothing")) == {}


def test_bench_line_names_numa_node_and_sensors_of_every_rank(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = []
    monkeypatch.setattr(bench, "pin_to_gpu_node", lambda device, **kw: seen.append(device) or {"device": device, "numa_node": device // 4, "pinned": True,
                                                                                                    "cpus_allowed": 96})
    monkeypatch.setattr(bench, "gpu_sensors", lambda device, **kw: {"sclk_mhz": 2400 - device, "mclk_mhz": 2000, "power_w": 600.0})
    line = _run(["--gpus", "8"] + BASE, FakeApi(n_dev=8))
    assert sorted(seen) == list(range(8))
    assert [(x["rank"], x["device"], x["numa_node"], x["pinned"]) for x in line["ranks"]] == [(r, r, r // 4, True) for r in range(8)]
    assert line["sensors_rank0"]["before_settle_passes"]["sclk_mhz"] == 2400 and line["sensors_rank0"]["after_timed_region"]["power_w"] == 600.0
    seen.clear()
    line = _run(["--gpus", "2", "--no-pin"] + BASE, FakeApi(n_dev=2))
    assert seen == [] and [x["pinned"] for x in line["ranks"]] == [False, False]
