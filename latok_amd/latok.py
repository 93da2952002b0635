"""Stand-in for the reference's C extension module ``latok.latok``.

The reference exports three METH_VARARGS functions (latok/core/src/latok/latok.c:373-378); here each is a thin NumPy
front-end over the C ABI of liblatok_hip.so, which runs them as HIP kernels.  Same names, argument meaning, return
types (fresh int8 ndarrays) and ValueError messages for the argument checks the reference makes.
"""
import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def text_to_cps(text) -> np.ndarray:
    """str -> uint32 code points, i.e. what PyUnicode_READ yields per index (reference latok.c:53-55,79)."""
    if not isinstance(text, str):
        raise ValueError("Input string not in 'ready' state")  # latok.c:47-50
    return np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.uint32, copy=False)


def _gen_parse_matrix(*args) -> np.ndarray:
    """reference gen_parse_matrix, latok.c:31-138: str -> int8[n, 25]."""
    if len(args) < 1:
        raise ValueError("must specify string to generate the parse matrix for")  # latok.c:40-43
    cps = np.ascontiguousarray(text_to_cps(args[0]))
    out = np.empty((len(cps), _lib.FEATURE_COUNT), np.int8)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_parse_matrix(_ptr(cps), len(cps), _ptr(out), 0, None))
    return out


def _gen_block_mask(*args) -> np.ndarray:
    """reference gen_block_mask, latok.c:140-258: two aligned 1-D arrays -> int8[n] mask of ones with zeroed blocks."""
    if len(args) < 2:
        raise ValueError("must specify two aligning 1d numpy array args")  # latok.c:151-154
    a1, a2 = np.asarray(args[0]), np.asarray(args[1])
    if a1.ndim != 1 or a2.ndim != 1:
        raise ValueError("must specify 1d numpy array args")  # latok.c:159-162
    if a1.size != a2.size:
        raise ValueError("must specify 1d numpy arrays of matching length")  # latok.c:168-171
    b1 = np.ascontiguousarray(a1 != 0, dtype=np.int8)  # PyArray_Nonzero semantics (latok.c:178,198)
    b2 = np.ascontiguousarray(a2 != 0, dtype=np.int8)
    out = np.empty(a1.size, np.int8)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_block_mask(_ptr(b1), _ptr(b2), a1.size, _ptr(out), 0, None))
    return out


def _combine_matrix_rows(*args) -> np.ndarray:
    """reference combine_matrix_rows, latok.c:275-370: "sum over idx rows of the product over idx columns" of matrix
    rows (2-D idxs, -1 = padding) or plain sum of rows (1-D idxs); uint8 wrap-around; returns int8[m.shape[1]]."""
    if len(args) < 2:
        raise ValueError("must specify 2d m and idxs matrices")  # latok.c:292-295
    m, idxs = np.asarray(args[0]), np.asarray(args[1])
    if m.ndim != 2 or idxs.ndim > 2 or idxs.ndim < 1:
        raise ValueError("must specify 2d numpy array args")  # latok.c:309-312
    m8 = m if m.dtype == np.int8 else m.astype(np.int8)        # convert_to_byte_array, latok.c:260-273
    idx = np.ascontiguousarray(idxs, dtype=np.int8)
    irows, icols = idx.shape if idx.ndim == 2 else (1, idx.shape[0])
    out = np.empty(m8.shape[1], np.int8)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_combine_matrix_rows(_ptr(m8), m8.shape[0], m8.shape[1], m8.strides[0], m8.strides[1],
                                             _ptr(idx), idx.ndim, irows, icols, _ptr(out), 0, None))
    return out
