"""ctypes binding of liblatok_hip.so -- the only place that touches the shared library.

This is the binding a reference maintainer would add in place of ``from latok.latok import ...``
(reference latok/core/default_tokenizer.py:36, latok/core/latok_utils.py:7); see INTEGRATION.md.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LATOK_HIP_LIB", os.path.join(_HERE, "liblatok_hip.so"))

OK, ERR_INVALID, ERR_HIP, ERR_NOT_INIT, ERR_NOMEM = 0, -1, -2, -3, -4
DEVICE_PTRS = 1
OUT_INT32 = 2
FEATURE_COUNT = 25
TILE_CHARS = 4096
CORPUS_ASCII, CORPUS_UNICODE = 0, 1

_lib = None
_lock = threading.Lock()
_inited = False

i64, u64, vp, ci = C.c_int64, C.c_uint64, C.c_void_p, C.c_int

# every symbol declared in include/latok_hip.h: name -> (restype, argtypes)
SIGNATURES = {
    "latok_device_count": (ci, []),
    "latok_init": (ci, [ci]),
    "latok_shutdown": (ci, []),
    "latok_last_error": (C.c_char_p, []),
    "latok_version": (C.c_char_p, []),
    "latok_ctx_create": (ci, [ci, C.POINTER(vp)]),
    "latok_ctx_destroy": (ci, [vp]),
    "latok_ctx_set_current": (ci, [vp]),
    "latok_ctx_get_current": (vp, []),
    "latok_ctx_device": (ci, [vp]),
    "latok_reserve": (ci, [i64, i64]),
    "latok_split_mask_batch": (ci, [vp, vp, i64, i64, vp, ci, vp]),
    "latok_split_values_batch": (ci, [vp, vp, i64, i64, vp, ci, vp]),
    "latok_split_offsets_batch": (ci, [vp, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_token_spans_batch": (ci, [vp, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_utf8_decode_batch": (ci, [vp, vp, i64, i64, vp, i64, vp, C.POINTER(i64), ci, vp]),
    "latok_split_mask_utf8_batch": (ci, [vp, vp, i64, i64, vp, i64, vp, C.POINTER(i64), ci, vp]),
    "latok_split_offsets_utf8_batch": (ci, [vp, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_token_spans_utf8_batch": (ci, [vp, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_split_mask_utf8_bytes_batch": (ci, [vp, vp, i64, i64, vp, ci, vp]),
    "latok_split_offsets_utf8_bytes_batch": (ci, [vp, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_token_spans_utf8_bytes_batch": (ci, [vp, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_token_features_batch": (ci, [vp, vp, i64, i64, vp, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_split_mask_kind_batch": (ci, [vp, ci, vp, i64, i64, vp, ci, vp]),
    "latok_split_offsets_kind_batch": (ci, [vp, ci, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_token_spans_kind_batch": (ci, [vp, ci, vp, i64, i64, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_token_features_kind_batch": (ci, [vp, ci, vp, i64, i64, vp, vp, vp, i64, C.POINTER(i64), ci, vp]),
    "latok_parse_matrix": (ci, [vp, i64, vp, ci, vp]),
    "latok_combine_matrix_rows": (ci, [vp, i64, i64, i64, i64, vp, ci, ci, ci, vp, ci, vp]),
    "latok_block_mask": (ci, [vp, vp, i64, vp, ci, vp]),
    "latok_dev_alloc": (vp, [C.c_size_t]),
    "latok_dev_free": (ci, [vp]),
    "latok_host_alloc": (vp, [C.c_size_t]),
    "latok_host_free": (ci, [vp]),
    "latok_memcpy_h2d": (ci, [vp, vp, C.c_size_t]),
    "latok_memcpy_d2h": (ci, [vp, vp, C.c_size_t]),
    "latok_memset_dev": (ci, [vp, ci, C.c_size_t]),
    "latok_sync": (ci, []),
    "latok_device_props": (ci, [C.POINTER(ci), C.POINTER(i64), C.c_char_p, ci]),
    "latok_corpus_offsets": (ci, [u64, u64, i64, i64, i64, vp]),
    "latok_corpus_fill_host": (ci, [u64, ci, u64, i64, vp, vp]),
    "latok_corpus_fill_device": (ci, [u64, ci, u64, i64, vp, vp, vp]),
    "latok_utf8_bytes": (ci, [vp, i64, C.POINTER(i64), ci]),
    "latok_bench_split_mask": (ci, [vp, vp, i64, i64, vp, ci, ci, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(i64)]),
    "latok_set_rules": (ci, [vp, ci, ci, vp, ci, ci, vp, ci, ci]),
    "latok_reset_rules": (ci, []),
    "latok_rules_active": (ci, []),
    "latok_bench_stream_read": (ci, [vp, i64, ci, ci, C.POINTER(C.c_float)]),
    "latok_gate_create": (ci, [ci, C.POINTER(vp)]),
    "latok_gate_destroy": (ci, [vp]),
    "latok_gate_wait": (ci, [vp, C.c_double]),
    "latok_gate_break": (ci, [vp]),
    "latok_gate_create_shared": (ci, [C.c_char_p, ci, C.POINTER(vp)]),
    "latok_gate_attach_shared": (ci, [C.c_char_p, C.POINTER(vp)]),
    "latok_gate_detach_shared": (ci, [vp]),
    "latok_gate_unlink_shared": (ci, [C.c_char_p]),
    "latok_bench_split_mask_gated": (ci, [vp, vp, i64, i64, vp, ci, vp, C.POINTER(C.c_float), C.POINTER(i64), C.POINTER(i64)]),
    "latok_bench_set_second_input": (ci, [vp, vp]),
    "latok_bench_tiles_flow": (ci, [vp, vp, i64, i64, vp, vp, ci, C.POINTER(C.c_float)]),
    "latok_flow_split_mask": (ci, [vp, vp, i64, i64, vp]),
    "latok_flow_split_mask_kind": (ci, [vp, ci, vp, i64, i64, vp]),
    "latok_flow_split_mask_utf8_bytes": (ci, [vp, vp, i64, i64, vp]),
    "latok_flow_split_offsets": (ci, [vp, ci, vp, i64, i64, vp, vp, i64, vp, ci]),
    "latok_flow_token_spans": (ci, [vp, ci, vp, i64, i64, vp, vp, i64, vp, ci]),
    "latok_flow_token_features": (ci, [vp, ci, vp, i64, i64, vp, vp, vp, i64, vp, ci]),
    "latok_flow_wait": (ci, []),
    "latok_bench_split_mask_flow_gated": (ci, [vp, vp, i64, i64, vp, vp, ci, vp, C.POINTER(C.c_float), C.POINTER(i64),
                                               C.POINTER(i64)]),
}


def load():
    """dlopen the library (no GPU needed for this step) and declare every signature."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `make -C latok_amd/csrc` (or __graft_entry__.build()). "
                    "latok_amd has no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def last_error() -> str:
    return load().latok_last_error().decode("utf-8", "replace")


def check(rc: int):
    """Map a status code to the exception the reference would raise (ValueError for bad arguments,
    reference latok.c:40-50,151-171,292-312) or RuntimeError for everything else."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def default_device() -> int:
    for key in ("LATOK_DEVICE", "LOCAL_RANK"):
        if os.environ.get(key, "") != "":
            return int(os.environ[key])
    return 0


def ensure_init(device=None):
    """Initialise the default context on first use.  A thread that runs inside ``with Context(...)`` needs no default
    context: its calls go to the current one."""
    global _inited
    lib = load()
    if not _inited and not lib.latok_ctx_get_current():
        check(lib.latok_init(default_device() if device is None else int(device)))
        _inited = True
    return lib


class Context:
    """One library context (include/latok_hip.h "contexts"): a device, a stream, a workspace set and a lock of its own.
    ``with ctx:`` makes it the calling THREAD's current context, so every ``latok_amd.batch`` call inside runs on it;
    contexts on different devices (or two on one device) run concurrently from different threads -- ctypes releases the
    GIL for the duration of a call."""

    def __init__(self, device: int):
        lib = load()
        h = vp()
        check(lib.latok_ctx_create(int(device), C.byref(h)))
        self._lib, self.handle, self.device = lib, h, int(device)
        # the saved "previous current context" stack is PER THREAD: the current context is a thread-local of the library,
        # and one Context object may be entered from several threads at once
        self._tls = threading.local()
        self._entered = 0
        self._count_lock = threading.Lock()

    def make_current(self):
        check(self._lib.latok_ctx_set_current(self.handle))

    def __enter__(self):
        if not self.handle:
            raise RuntimeError("context has been destroyed")
        stack = self._tls.__dict__.setdefault("prev", [])
        stack.append(self._lib.latok_ctx_get_current())
        with self._count_lock:
            self._entered += 1
        self.make_current()
        return self

    def __exit__(self, *exc):
        with self._count_lock:
            self._entered -= 1
        check(self._lib.latok_ctx_set_current(self._tls.prev.pop()))
        return False

    def destroy(self):
        """Destroy the library context.  Refused while some thread is still inside ``with ctx:`` (its thread-local
        current context would dangle)."""
        if self.handle:
            with self._count_lock:
                if self._entered > 0:
                    raise RuntimeError(f"context is still entered by {self._entered} thread(s)")
            check(self._lib.latok_ctx_destroy(self.handle))
            self.handle = vp()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def shutdown():
    global _inited
    if _lib is not None and _inited:
        _lib.latok_shutdown()
    _inited = False
