"""TEST INFRASTRUCTURE ONLY -- ctypes/NumPy front-end of oracle/liblatok_oracle.so (the plain-C CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(latok_amd/) never does.

Also offers ``RefGlue``: the reference's *own* compiled C functions (oracle/_ref, built from the reference's latok.c
where it lies) driven by a restatement of the 20-line NumPy glue ``gen_split_mask`` (reference
latok/core/default_tokenizer.py:113-134) -- this is what can run on the GPU box, where /root/reference is absent.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblatok_oracle.so")
_lib = None

FEATURE_COUNT = 25


def build(force: bool = False) -> str:
    """Compile the restatement (gcc, ~1 s).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("latok_oracle.c", "latok_oracle.h", "latok_oracle_tables.inc")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "restate"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i64, vp = C.c_int64, C.c_void_p
        L.oracle_base_word.restype = C.c_uint
        L.oracle_base_word.argtypes = [C.c_uint32]
        L.oracle_gen_parse_matrix.restype = None
        L.oracle_gen_parse_matrix.argtypes = [vp, i64, vp]
        L.oracle_combine_matrix_rows.restype = None
        L.oracle_combine_matrix_rows.argtypes = [vp, i64, i64, i64, vp, C.c_int, C.c_int, C.c_int, vp]
        L.oracle_gen_block_mask.restype = None
        L.oracle_gen_block_mask.argtypes = [vp, vp, i64, vp]
        L.oracle_gen_split_mask.restype = C.c_int
        L.oracle_gen_split_mask.argtypes = [vp, i64, vp]
        L.oracle_split_values.restype = C.c_int
        L.oracle_split_values.argtypes = [vp, i64, vp]
        L.oracle_nonzero.restype = i64
        L.oracle_nonzero.argtypes = [vp, i64, vp]
        L.oracle_split_batch.restype = C.c_int
        L.oracle_split_batch.argtypes = [vp, vp, i64, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def to_cps(text) -> np.ndarray:
    """str -> uint32 code points (what PyUnicode_READ yields per index, reference latok.c:53-55,79)."""
    if isinstance(text, np.ndarray):
        return np.ascontiguousarray(text, dtype=np.uint32)
    return np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.uint32)


def base_word(cp: int) -> int:
    return int(lib().oracle_base_word(C.c_uint32(cp)))


def gen_parse_matrix(text) -> np.ndarray:
    cps = to_cps(text)
    out = np.empty((len(cps), FEATURE_COUNT), np.int8)
    lib().oracle_gen_parse_matrix(_p(cps), len(cps), _p(out))
    return out


def combine_matrix_rows(m: np.ndarray, idxs: np.ndarray) -> np.ndarray:
    if m.ndim != 2 or idxs.ndim > 2:
        raise ValueError("must specify 2d numpy array args")
    m8 = m if m.dtype == np.int8 else m.astype(np.int8)
    idx = np.ascontiguousarray(idxs, dtype=np.int8)
    out = np.empty(m8.shape[1], np.int8)
    irows, icols = (idx.shape if idx.ndim == 2 else (1, idx.shape[0]))
    lib().oracle_combine_matrix_rows(_p(m8), m8.strides[0], m8.strides[1], m8.shape[1], _p(idx), idx.ndim, irows,
                                     icols, _p(out))
    return out


def gen_block_mask(a1: np.ndarray, a2: np.ndarray) -> np.ndarray:
    if a1.ndim != 1 or a2.ndim != 1:
        raise ValueError("must specify 1d numpy array args")
    if a1.size != a2.size:
        raise ValueError("must specify 1d numpy arrays of matching length")
    b1 = np.ascontiguousarray(a1 != 0, dtype=np.int8)
    b2 = np.ascontiguousarray(a2 != 0, dtype=np.int8)
    out = np.empty(a1.size, np.int8)
    lib().oracle_gen_block_mask(_p(b1), _p(b2), a1.size, _p(out))
    return out


def gen_split_mask(m: np.ndarray) -> np.ndarray:
    m = np.ascontiguousarray(m, dtype=np.int8)
    if m.shape[0] == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    out = np.empty(m.shape[0], np.int8)
    lib().oracle_gen_split_mask(_p(m), m.shape[0], _p(out))
    return out


def split_values_rules(text, c_split, c_mask, c_sym) -> np.ndarray:
    """gen_split_mask (default_tokenizer.py:113-134) with caller-supplied combo matrices, composed from the restated
    C functions exactly as the reference composes its own."""
    m = gen_parse_matrix(text)
    if m.shape[0] == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    mt = m.T
    splits = combine_matrix_rows(mt, np.asarray(c_split, np.int8)) * gen_block_mask(
        combine_matrix_rows(mt, np.asarray(c_mask, np.int8)), mt[5])
    splits += combine_matrix_rows(mt, np.asarray(c_sym, np.int8))
    splits[0] = 1
    return splits


def split_values(text) -> np.ndarray:
    cps = to_cps(text)
    if len(cps) == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    out = np.empty(len(cps), np.int8)
    lib().oracle_split_values(_p(cps), len(cps), _p(out))
    return out


def split_offsets(text) -> np.ndarray:
    """np.nonzero(gen_split_mask(_gen_parse_matrix(text)))[0] -- the bit-exact deliverable (int64, ascending)."""
    sp = split_values(text)
    off = np.empty(len(sp), np.int64)
    k = lib().oracle_nonzero(_p(sp), len(sp), _p(off))
    return off[:k].copy()


def tokenize(text: str):
    """reference tokenize(): default_tokenizer.py:137-160."""
    nz = split_offsets(text)
    out = []
    if len(nz) > 0:
        a, b = int(nz[0]), 0
        for b in nz[1:]:
            b = int(b)
            tok = text[a:b].strip()
            if tok:
                out.append(tok)
            a = b
        last = text[b:].strip()
        if last:
            out.append(last)
    else:
        out.append("")
    return out


def split_batch(cps: np.ndarray, row_off: np.ndarray, want_values=True, want_bits=True):
    cps = np.ascontiguousarray(cps, dtype=np.uint32)
    row_off = np.ascontiguousarray(row_off, dtype=np.int64)
    n_str = len(row_off) - 1
    total = int(row_off[-1]) if n_str >= 0 else 0
    vals = np.zeros(total, np.uint8) if want_values else None
    bits = np.zeros((total + 63) // 64, np.uint64) if want_bits else None
    rc = lib().oracle_split_batch(_p(cps), _p(row_off), n_str, _p(vals), _p(bits))
    if rc != 0:
        raise ValueError("bad row offsets")
    return vals, bits


# ------------------------------------------------------------------------------------------------------------------
# The reference's own compiled C functions + a restatement of its NumPy glue (works wherever oracle/_ref exists).
# ------------------------------------------------------------------------------------------------------------------
class RefGlue:
    """latok's own C (oracle/_ref) under a restated gen_split_mask (default_tokenizer.py:113-134)."""

    C_SPLIT = np.array([[5, -1], [6, -1], [20, -1], [4, 17], [4, 16]], np.int8)           # :49-55
    C_MASK = np.array([[7, 18, 13, -1], [11, 18, 21, 23], [8, 14, 15, -1], [9, 22, 24, 12]], np.int8)  # :80-91
    C_SYM = np.array([[6, 19]], np.int8)                                                   # :100-102
    SPACE_IDX = 5

    def __init__(self):
        import ref_loader  # sibling module in oracle/
        self.ext = ref_loader.load_ref_ext()

    def gen_parse_matrix(self, text: str):
        return self.ext._gen_parse_matrix(text)

    def gen_split_mask(self, m, tables=None):
        c_split, c_mask, c_sym = tables if tables is not None else (self.C_SPLIT, self.C_MASK, self.C_SYM)
        mt = m.T
        splits = (self.ext._combine_matrix_rows(mt, c_split) *
                  self.ext._gen_block_mask(self.ext._combine_matrix_rows(mt, c_mask), mt[self.SPACE_IDX]))
        splits += self.ext._combine_matrix_rows(mt, c_sym)
        splits[0] = 1
        return splits

    def split_values(self, text: str):
        return self.gen_split_mask(self.gen_parse_matrix(text))

    def split_offsets(self, text: str):
        return np.nonzero(self.split_values(text))[0]
