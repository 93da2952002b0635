"""Whole-batch entry points over the fused HIP kernel (additive to the reference surface).

A batch is CSR: ``cps`` = packed UTF-32 code points of all strings, ``row_off[n+1]`` = start of each string.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(texts):
    """list[str] -> (cps uint32[total], row_off int64[n+1])."""
    lens = np.fromiter((len(t) for t in texts), dtype=np.int64, count=len(texts))
    row_off = np.zeros(len(texts) + 1, np.int64)
    np.cumsum(lens, out=row_off[1:])
    blob = "".join(texts).encode("utf-32-le", "surrogatepass")
    cps = np.frombuffer(blob, dtype="<u4").astype(np.uint32, copy=False)
    return np.ascontiguousarray(cps), row_off


def _csr(cps, row_off):
    cps = np.ascontiguousarray(cps, dtype=np.uint32)
    row_off = np.ascontiguousarray(row_off, dtype=np.int64)
    if row_off.ndim != 1 or row_off.size < 1:
        raise ValueError("row_off must be a 1-D array of n_str + 1 offsets")
    if cps.ndim != 1 or (row_off.size > 1 and cps.size < int(row_off[-1])):
        raise ValueError("cps is shorter than row_off[-1]")
    return cps, row_off



class _Pinned:
    """keeps a pinned allocation alive for as long as the numpy array over it"""

    def __init__(self, lib, nbytes):
        self.lib, self.ptr = lib, lib.latok_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise MemoryError(_lib.last_error())

    def __del__(self):
        try:
            self.lib.latok_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """np.empty in pinned (page-locked) host memory (latok_host_alloc): host-pointer batches held in such arrays cross the
    bus at full speed and asynchronously, which is what lets the chunked pipeline of the large-batch calls overlap the
    upload of one chunk with the download of another.  The memory is freed when the array (and its views) are gone."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    owner = _Pinned(_lib.ensure_init(), n * dt.itemsize)
    buf = (C.c_char * max(n * dt.itemsize, 1)).from_address(owner.ptr)
    buf._latok_owner = owner
    a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    return a


def _out_dtype(dtype):
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.int64), np.dtype(np.int32)):
        raise ValueError("dtype must be int64 or int32")
    return dt, (_lib.OUT_INT32 if dt == np.dtype(np.int32) else 0)


def _compact(fn, lead, n_str, total, width, dtype, feats=False, pinned=False):
    """Shared body of the compaction wrappers: fn(*lead, n_str, total, counts, items[, features], cap, &n, flags, stream).
    Returns (counts[n_str], items[n] or items[n, width][, features int8[n, 25]]); counts / items in `dtype` (int64, or
    int32 = LATOK_OUT_INT32: half the bytes written on the device and moved over the bus).  pinned: the result arrays
    live in pinned host memory and are returned as views (no copy)."""
    dt, flags = _out_dtype(dtype)
    alloc = pinned_empty if pinned else np.empty
    counts = alloc(n_str, dt)
    cap = max(total, 1)                                  # a string has at most len items
    items = alloc((cap, width) if width > 1 else cap, dt)
    feat = alloc((cap, _lib.FEATURE_COUNT), np.int8) if feats else None
    n = C.c_int64(0)
    args = list(lead) + [n_str, total, _ptr(counts), _ptr(items)] + ([_ptr(feat)] if feats else []) + [cap, C.byref(n), flags, None]
    _lib.check(fn(*args))
    keep = (lambda a: a) if pinned else (lambda a: a.copy())
    if feats:
        return counts, keep(items[:n.value]), keep(feat[:n.value])
    return counts, keep(items[:n.value])


def split_mask_batch(cps, row_off) -> np.ndarray:
    """Boundary bitmask uint64[ceil(total/64)]: bit i = packed char i starts a token."""
    cps, row_off = _csr(cps, row_off)
    n_str = row_off.size - 1
    total = int(row_off[-1]) if n_str > 0 else 0
    bits = np.zeros((total + 63) // 64, np.uint64)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_split_mask_batch(_ptr(cps), _ptr(row_off), n_str, total, _ptr(bits), 0, None))
    return bits


def split_values_batch(cps, row_off) -> np.ndarray:
    """The reference's split values (0..5) for every packed char, uint8[total]."""
    cps, row_off = _csr(cps, row_off)
    n_str = row_off.size - 1
    total = int(row_off[-1]) if n_str > 0 else 0
    vals = np.zeros(total, np.uint8)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_split_values_batch(_ptr(cps), _ptr(row_off), n_str, total, _ptr(vals), 0, None))
    return vals


def split_offsets_csr(cps, row_off, dtype=np.int64):
    """(counts[n], offsets[sum(counts)]): per-string boundary offsets, concatenated (dtype int64 or int32)."""
    cps, row_off = _csr(cps, row_off)
    n_str = row_off.size - 1
    total = int(row_off[-1]) if n_str > 0 else 0
    return _compact(_lib.ensure_init().latok_split_offsets_batch, [_ptr(cps), _ptr(row_off)], n_str, total, 1, dtype)


# host batches beyond the library's small-batch path (api.cpp: kSmallChars / kSmallStrings, served from pinned memory
# in UTF-32) are shipped as the narrowest PEP 393 kind: 1 or 2 bytes per char over the bus instead of 4
_SMALL_CHARS, _SMALL_STRINGS = 262144, 16384     # api.cpp: kSmallChars / kSmallStrings (pinned zero-copy path)
_ONE_MAX = 16384                                   # split_offsets_one: the per-thread output array


_ROW1 = None


def split_offsets_one(text: str) -> np.ndarray:
    """np.nonzero(split mask)[0] of ONE non-empty string with as little Python around the C call as possible (the drop-in
    tokenize(text) surface): a string of at most 4096 chars is one single-wave launch in the library, which polls the
    kernel's completion word; what is left here is the UTF-32 encode (the bytes object goes to the C call as it is) and
    a copy of the offsets out of a per-thread output array."""
    global _ROW1
    n = len(text)
    if n > _ONE_MAX:
        return split_offsets_batch([text])[0]
    lib = _lib.ensure_init()
    if _ROW1 is None:
        _ROW1 = threading.local()
    st = getattr(_ROW1, "st", None)
    if st is None:
        row, count, offs, n_out = np.zeros(2, np.int64), np.zeros(1, np.int32), np.empty(_ONE_MAX, np.int32), C.c_int64(0)
        st = _ROW1.st = (row, count, offs, n_out, row.ctypes.data, count.ctypes.data, offs.ctypes.data, C.byref(n_out))
    row, _, offs, n_out, p_row, p_count, p_offs, p_n = st
    row[1] = n
    rc = lib.latok_split_offsets_batch(text.encode("utf-32-le", "surrogatepass"), p_row, 1, n, p_count, p_offs, n, p_n,
                                       _lib.OUT_INT32, None)
    if rc:
        _lib.check(rc)
    return offs[:n_out.value].copy()


def featurize_one(text: str):
    """list(featurize(text)) of the reference for ONE non-empty string of at most 4096 chars: two small launches in the
    library (boundaries + kept tokens, then the sums), per-thread output arrays, one LaToken per kept token."""
    from .core.latok_utils import LaToken
    global _FEAT1
    n = len(text)
    if n > 4096:
        return featurize_batch([text])[0]
    lib = _lib.ensure_init()
    if _FEAT1 is None:
        _FEAT1 = threading.local()
    st = getattr(_FEAT1, "st", None)
    if st is None:
        row, count, spans, n_out = np.zeros(2, np.int64), np.zeros(1, np.int32), np.empty((4096, 4), np.int32), C.c_int64(0)
        feats = np.empty((4096, 25), np.int8)
        st = _FEAT1.st = (row, count, spans, feats, n_out, row.ctypes.data, count.ctypes.data, spans.ctypes.data,
                          feats.ctypes.data, C.byref(n_out))
    row, _, spans, feats, n_out, p_row, p_count, p_spans, p_feats, p_n = st
    row[1] = n
    rc = lib.latok_token_features_batch(text.encode("utf-32-le", "surrogatepass"), p_row, 1, n, p_count, p_spans, p_feats, n,
                                        p_n, _lib.OUT_INT32, None)
    if rc:
        _lib.check(rc)
    k = n_out.value
    rows = feats[:k].copy()            # the tokens' vectors are views of one fresh array, not of the per-thread buffer
    return [LaToken(text[c:d], a, b, rows[j]) for j, (a, b, c, d) in enumerate(spans[:k].tolist())]


_FEAT1 = None


def _record_dtype(row_off):
    """int32 records (LATOK_OUT_INT32: half the device writes and bus traffic) unless a string has 2^31 chars or more"""
    return np.int32 if row_off.size < 2 or int(np.diff(row_off).max()) <= 0x7FFFFFFF else np.int64


def _narrow_pays(texts):
    return len(texts) > _SMALL_STRINGS or sum(map(len, texts)) > _SMALL_CHARS


def split_offsets_batch(texts, devices=None):
    """list[str] -> list of int64 arrays = np.nonzero(split mask)[0] of every string ('' -> empty array).
    devices: a multi.DevicePool or a list of device ids -> the strings are sharded over them (latok_amd.multi)."""
    if len(texts) == 0:
        return []
    if devices is not None:
        from . import multi
        return multi.split_offsets_batch(texts, devices)
    if _narrow_pays(texts):
        counts, offsets = split_offsets_kind_csr(*pack_kind(texts))
    else:
        counts, offsets = split_offsets_csr(*pack(texts))
    return np.split(offsets, np.cumsum(counts)[:-1])


def token_spans_csr(cps, row_off, dtype=np.int64):
    """(counts[n], spans[n_tokens, 2]): [start, end) of every token of every string, already stripped and with
    whitespace-only tokens dropped -- everything reference tokenize() does after np.nonzero, on the device."""
    cps, row_off = _csr(cps, row_off)
    n_str = row_off.size - 1
    total = int(row_off[-1]) if n_str > 0 else 0
    return _compact(_lib.ensure_init().latok_token_spans_batch, [_ptr(cps), _ptr(row_off)], n_str, total, 2, dtype)


def token_features_csr(cps, row_off, dtype=np.int64):
    """(counts[n], spans[n_tokens, 4] = {raw_start, raw_end, strip_start, strip_end}, features int8[n_tokens, 25]):
    reference featurize() for a whole batch, without the n x 25 matrix."""
    cps, row_off = _csr(cps, row_off)
    n_str = row_off.size - 1
    total = int(row_off[-1]) if n_str > 0 else 0
    return _compact(_lib.ensure_init().latok_token_features_batch, [_ptr(cps), _ptr(row_off)], n_str, total, 4, dtype, feats=True)


def featurize_batch(texts, devices=None):
    """list[str] -> list[list[LaToken]], each as list(featurize(text)) of the reference.
    devices: a multi.DevicePool or a list of device ids -> the strings are sharded over them."""
    from .core.latok_utils import LaToken
    if len(texts) == 0:
        return []
    if devices is not None:
        from . import multi
        return multi.featurize_batch(texts, devices)
    if _narrow_pays(texts):
        units, row_off = pack_kind(texts)
        counts, spans, feats = token_features_kind_csr(units, row_off, dtype=_record_dtype(row_off))
    else:
        cps, row_off = pack(texts)
        counts, spans, feats = token_features_csr(cps, row_off, dtype=_record_dtype(row_off))
    out, k = [], 0
    for text, n in zip(texts, counts.tolist()):
        out.append([LaToken(text[c:d], a, b, feats[k + j]) for j, (a, b, c, d) in enumerate(spans[k:k + n].tolist())])
        k += n
    return out


# ---- UTF-8 ingest -------------------------------------------------------------------------------------------------------
def pack_utf8(blobs):
    """list[bytes] (each valid UTF-8) -> (utf8 uint8[total_bytes], byte_off int64[n+1])."""
    lens = np.fromiter((len(b) for b in blobs), dtype=np.int64, count=len(blobs))
    byte_off = np.zeros(len(blobs) + 1, np.int64)
    np.cumsum(lens, out=byte_off[1:])
    return np.frombuffer(b"".join(blobs), dtype=np.uint8), byte_off


def _csr_u8(utf8, byte_off):
    utf8 = np.ascontiguousarray(utf8, dtype=np.uint8)
    byte_off = np.ascontiguousarray(byte_off, dtype=np.int64)
    if byte_off.ndim != 1 or byte_off.size < 1 or (byte_off.size > 1 and utf8.size < int(byte_off[-1])):
        raise ValueError("byte_off must be n_str + 1 offsets into utf8")
    return utf8, byte_off


def utf8_decode_csr(utf8, byte_off):
    """Device decode of a UTF-8 CSR batch -> (cps uint32[total_cps], cp_row_off int64[n+1])."""
    utf8, byte_off = _csr_u8(utf8, byte_off)
    n_str = byte_off.size - 1
    total = int(byte_off[-1]) if n_str > 0 else 0
    cps = np.empty(max(total, 1), np.uint32)
    row = np.zeros(n_str + 1, np.int64)
    n = C.c_int64(0)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_utf8_decode_batch(_ptr(utf8), _ptr(byte_off), n_str, total, _ptr(cps), cps.size, _ptr(row),
                                           C.byref(n), 0, None))
    return cps[:n.value].copy(), row


def split_mask_utf8_csr(utf8, byte_off):
    """(bits uint64[ceil(total_cps/64)], cp_row_off int64[n+1]): boundary bitmask over the decoded code points."""
    utf8, byte_off = _csr_u8(utf8, byte_off)
    n_str = byte_off.size - 1
    total = int(byte_off[-1]) if n_str > 0 else 0
    bits = np.zeros((total + 63) // 64, np.uint64)
    row = np.zeros(n_str + 1, np.int64)
    n = C.c_int64(0)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_split_mask_utf8_batch(_ptr(utf8), _ptr(byte_off), n_str, total, _ptr(bits), bits.size, _ptr(row),
                                               C.byref(n), 0, None))
    return bits[:(n.value + 63) // 64], row


def split_offsets_utf8_csr(utf8, byte_off, dtype=np.int64):
    """(counts, offsets) like split_offsets_csr, input handed over as UTF-8 (1 byte per ASCII char over PCIe).
    Offsets are code-point indices, as the reference reports them for the decoded str."""
    utf8, byte_off = _csr_u8(utf8, byte_off)
    n_str = byte_off.size - 1
    total = int(byte_off[-1]) if n_str > 0 else 0
    return _compact(_lib.ensure_init().latok_split_offsets_utf8_batch, [_ptr(utf8), _ptr(byte_off)], n_str, total, 1, dtype)


def token_spans_utf8_csr(utf8, byte_off, dtype=np.int64):
    """(counts, spans[n_tokens, 2]) like token_spans_csr for a UTF-8 CSR batch; spans are code-point indices."""
    utf8, byte_off = _csr_u8(utf8, byte_off)
    n_str = byte_off.size - 1
    total = int(byte_off[-1]) if n_str > 0 else 0
    return _compact(_lib.ensure_init().latok_token_spans_utf8_batch, [_ptr(utf8), _ptr(byte_off)], n_str, total, 2, dtype)


# byte-space forms: the tile kernel reads the UTF-8 bytes itself; every position is a BYTE position in `utf8`
def split_mask_utf8_bytes_csr(utf8, byte_off) -> np.ndarray:
    """uint64 bitmask over the BYTES of the batch: bit i set = byte i is the lead byte of a boundary char."""
    utf8, byte_off = _csr_u8(utf8, byte_off)
    n_str = byte_off.size - 1
    total = int(byte_off[-1]) if n_str > 0 else 0
    bits = np.zeros((total + 63) // 64, np.uint64)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_split_mask_utf8_bytes_batch(_ptr(utf8), _ptr(byte_off), n_str, total, _ptr(bits), 0, None))
    return bits


def split_offsets_utf8_bytes_csr(utf8, byte_off, dtype=np.int64):
    """(counts, offsets): boundary BYTE offsets relative to each string's first byte."""
    utf8, byte_off = _csr_u8(utf8, byte_off)
    n_str = byte_off.size - 1
    total = int(byte_off[-1]) if n_str > 0 else 0
    return _compact(_lib.ensure_init().latok_split_offsets_utf8_bytes_batch, [_ptr(utf8), _ptr(byte_off)], n_str, total, 1, dtype)


def token_spans_utf8_bytes_csr(utf8, byte_off, dtype=np.int64):
    """(counts, spans[n_tokens, 2]): [start, end) BYTE ranges of the stripped, non-empty tokens of each string."""
    utf8, byte_off = _csr_u8(utf8, byte_off)
    n_str = byte_off.size - 1
    total = int(byte_off[-1]) if n_str > 0 else 0
    return _compact(_lib.ensure_init().latok_token_spans_utf8_bytes_batch, [_ptr(utf8), _ptr(byte_off)], n_str, total, 2, dtype)


def tokenize_utf8_batch(blobs):
    """list[bytes] (UTF-8) -> list[list[bytes]]: the reference's tokens of every string, as UTF-8 slices of the input
    (byte-space path: nothing is transcoded on the host or on the device)."""
    if len(blobs) == 0:
        return []
    utf8, byte_off = pack_utf8(blobs)
    counts, spans = token_spans_utf8_bytes_csr(utf8, byte_off, dtype=_record_dtype(byte_off))
    out, k = [], 0
    for blob, n in zip(blobs, counts.tolist()):
        out.append([blob[a:b] for a, b in spans[k:k + n].tolist()])
        k += n
    return out


# ---- PEP 393 code units: 1 / 2 / 4 bytes per char, the buffer the reference itself reads (latok.c:53-55,79) -----------
def pack_kind(texts):
    """list[str] -> (units, row_off): units uint8 / uint16 / uint32 = the narrowest PEP 393 kind that holds every char of
    the batch (what CPython stores for the joined text), row_off int64[n+1] in chars."""
    lens = np.fromiter((len(t) for t in texts), dtype=np.int64, count=len(texts))
    row_off = np.zeros(len(texts) + 1, np.int64)
    np.cumsum(lens, out=row_off[1:])
    joined = "".join(texts)
    try:
        units = np.frombuffer(joined.encode("latin-1"), dtype=np.uint8)
    except UnicodeEncodeError:
        blob = joined.encode("utf-16-le", "surrogatepass")
        if len(blob) == 2 * len(joined):
            units = np.frombuffer(blob, dtype="<u2").astype(np.uint16, copy=False)
        else:   # astral chars: kind 4
            units = np.frombuffer(joined.encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.uint32, copy=False)
    return np.ascontiguousarray(units), row_off


def _csr_kind(units, row_off):
    units = np.ascontiguousarray(units)
    if units.dtype not in (np.uint8, np.uint16, np.uint32):
        raise ValueError("units must be uint8 (Latin-1), uint16 (UCS-2) or uint32 (UCS-4)")
    row_off = np.ascontiguousarray(row_off, dtype=np.int64)
    if row_off.ndim != 1 or row_off.size < 1:
        raise ValueError("row_off must be a 1-D array of n_str + 1 offsets")
    if units.ndim != 1 or (row_off.size > 1 and units.size < int(row_off[-1])):
        raise ValueError("units is shorter than row_off[-1]")
    return units, row_off, int(units.dtype.itemsize)


def split_mask_kind_csr(units, row_off) -> np.ndarray:
    """split_mask_batch for PEP 393 code units (dtype picks the kind); bit i = packed char i starts a token."""
    units, row_off, kind = _csr_kind(units, row_off)
    n_str = row_off.size - 1
    total = int(row_off[-1]) if n_str > 0 else 0
    bits = np.zeros((total + 63) // 64, np.uint64)
    lib = _lib.ensure_init()
    _lib.check(lib.latok_split_mask_kind_batch(_ptr(units), kind, _ptr(row_off), n_str, total, _ptr(bits), 0, None))
    return bits


def _compact_kind(fn, width, units, row_off, dtype, feats=False):
    units, row_off, kind = _csr_kind(units, row_off)
    n_str = row_off.size - 1
    total = int(row_off[-1]) if n_str > 0 else 0
    return _compact(fn, [_ptr(units), kind, _ptr(row_off)], n_str, total, width, dtype, feats)


def split_offsets_kind_csr(units, row_off, dtype=np.int64):
    """(counts, offsets) like split_offsets_csr for PEP 393 code units."""
    return _compact_kind(_lib.ensure_init().latok_split_offsets_kind_batch, 1, units, row_off, dtype)


def token_spans_kind_csr(units, row_off, dtype=np.int64):
    """(counts, spans[n_tokens, 2]) like token_spans_csr for PEP 393 code units."""
    return _compact_kind(_lib.ensure_init().latok_token_spans_kind_batch, 2, units, row_off, dtype)


def token_features_kind_csr(units, row_off, dtype=np.int64):
    """(counts, spans[n_tokens, 4], features int8[n_tokens, 25]) like token_features_csr for PEP 393 code units."""
    return _compact_kind(_lib.ensure_init().latok_token_features_kind_batch, 4, units, row_off, dtype, feats=True)


def spans_from_offsets(text, nz):
    """Token strings of one text from its boundary offsets, as the reference's loop builds them
    (default_tokenizer.py:149-158): slice between consecutive boundaries, strip, drop empties."""
    toks = []
    if len(nz) > 0:
        bounds = [int(x) for x in nz]
        a, end = bounds[0], 0
        for end in bounds[1:]:
            tok = text[a:end].strip()
            if tok:
                toks.append(tok)
            a = end
        tok = text[end:].strip()
        if tok:
            toks.append(tok)
    return toks


def tokenize_batch(texts, devices=None):
    """list[str] -> list[list[str]], each as list(tokenize(text)) of the reference (default_tokenizer.py:137-160);
    an empty string yields [] instead of the reference's IndexError.
    devices: a multi.DevicePool or a list of device ids -> the strings are sharded over them, one host thread and one
    library context per entry (latok_amd.multi); the result is the same list."""
    if len(texts) == 0:
        return []
    if devices is not None:
        from . import multi
        return multi.tokenize_batch(texts, devices)
    if _narrow_pays(texts):
        units, row_off = pack_kind(texts)
        counts, spans = token_spans_kind_csr(units, row_off, dtype=_record_dtype(row_off))
    else:
        cps, row_off = pack(texts)
        counts, spans = token_spans_csr(cps, row_off, dtype=_record_dtype(row_off))
    out, k = [], 0
    for text, n in zip(texts, counts.tolist()):
        out.append([text[a:b] for a, b in spans[k:k + n].tolist()])
        k += n
    return out


class TokenSpans:
    """The tokens of a batch WITHOUT the per-token Python objects: ``spans[k] = (start, end)`` of token k inside its own
    string (whitespace stripped, empties dropped -- what the reference's loop keeps, default_tokenizer.py:149-160),
    ``counts[i]`` tokens for string i, strings in order.  Building ``list[list[str]]`` costs one slice per token (1.5 ms per
    1000 short strings, five orders of magnitude below the device path); a caller that filters, counts, hashes or looks only
    at some strings slices lazily:

        ts = batch.token_spans_batch(texts)
        ts.counts, ts.spans                     # numpy: int32 / int64 [n_str], [n_tokens, 2]
        ts.tokens(i)                            # list[str] of string i, sliced on demand
        for toks in ts: ...                     # == batch.tokenize_batch(texts), one string at a time
    """

    def __init__(self, texts, counts, spans):
        self.texts, self.counts, self.spans = texts, counts, spans
        self.first = np.zeros(len(texts) + 1, np.int64)       # index of each string's first token
        np.cumsum(counts, out=self.first[1:])

    def __len__(self):
        return len(self.texts)

    def row(self, i):
        """(start, end) pairs of string i: a view into ``spans``"""
        return self.spans[int(self.first[i]):int(self.first[i + 1])]

    def tokens(self, i):
        t = self.texts[i]
        return [t[a:b] for a, b in self.row(i).tolist()]

    def __iter__(self):
        return (self.tokens(i) for i in range(len(self.texts)))


def token_spans_batch(texts):
    """list[str] -> TokenSpans: the same device work as tokenize_batch, none of its per-token Python."""
    texts = list(texts)
    if len(texts) == 0:
        return TokenSpans(texts, np.zeros(0, np.int64), np.zeros((0, 2), np.int64))
    if _narrow_pays(texts):
        units, row_off = pack_kind(texts)
        counts, spans = token_spans_kind_csr(units, row_off, dtype=_record_dtype(row_off))
    else:
        cps, row_off = pack(texts)
        counts, spans = token_spans_csr(cps, row_off, dtype=_record_dtype(row_off))
    return TokenSpans(texts, counts, spans.reshape(-1, 2))


# ---- runtime rule tables (the reference's extension point, default_tokenizer.py:9-30,108-110) ------------------------
# ---- batch flow: many device-resident batches through the current context, two in flight ------------------------------
def flow_split_mask(d_cps, d_row_off, n_str, total_chars, d_mask):
    """include/latok_hip.h "batch flow": enqueue the split mask of one DEVICE-RESIDENT batch (device addresses as ints /
    c_void_p: UTF-32 code points, int64 row offsets, uint64 mask words out) and return at once; up to two batches run
    overlapped (the string-index and resolve launches of one in the shadow of the other's tile kernel).  The inputs must
    be complete in device memory; results are complete after ``flow_wait()``.  The reference's unit of independence is
    the single ``tokenize`` call (default_tokenizer.py:137-160)."""
    lib = _lib.ensure_init()
    _lib.check(lib.latok_flow_split_mask(d_cps, d_row_off, int(n_str), int(total_chars), d_mask))


def flow_split_mask_kind(d_units, kind, d_row_off, n_str, total_chars, d_mask):
    """``flow_split_mask`` for PEP 393 units in device memory (kind 1 = Latin-1 bytes, 2 = UCS-2 uint16, 4 = UTF-32)."""
    lib = _lib.ensure_init()
    _lib.check(lib.latok_flow_split_mask_kind(d_units, int(kind), d_row_off, int(n_str), int(total_chars), d_mask))


def flow_split_mask_utf8_bytes(d_utf8, d_byte_off, n_str, total_bytes, d_mask):
    """``flow_split_mask`` in byte space: UTF-8 bytes + byte offsets in device memory, bit i of the mask = byte i."""
    lib = _lib.ensure_init()
    _lib.check(lib.latok_flow_split_mask_utf8_bytes(d_utf8, d_byte_off, int(n_str), int(total_bytes), d_mask))


def flow_split_offsets(d_units, kind, d_row_off, n_str, total_units, d_counts, d_offsets, cap, d_result, dtype=np.int64):
    """Boundary offsets of one device-resident batch through the flow (``latok_flow_split_offsets``): kind 4 / 1 / 2 =
    UTF-32 / Latin-1 / UCS-2 units, 0 = UTF-8 bytes in byte space.  ``d_result`` = int64[2] the device can write: item
    total and error word, valid after ``flow_wait()``; nothing is written to ``d_offsets`` when the total exceeds ``cap``."""
    lib = _lib.ensure_init()
    _, flag32 = _out_dtype(dtype)
    _lib.check(lib.latok_flow_split_offsets(d_units, int(kind), d_row_off, int(n_str), int(total_units), d_counts, d_offsets, int(cap),
                                            d_result, flag32))


def flow_token_spans(d_units, kind, d_row_off, n_str, total_units, d_counts, d_spans, cap, d_result, dtype=np.int64):
    """Token spans (start, end per kept token) of one device-resident batch through the flow (``latok_flow_token_spans``)."""
    lib = _lib.ensure_init()
    _, flag32 = _out_dtype(dtype)
    _lib.check(lib.latok_flow_token_spans(d_units, int(kind), d_row_off, int(n_str), int(total_units), d_counts, d_spans, int(cap),
                                          d_result, flag32))


def flow_token_features(d_units, kind, d_row_off, n_str, total_chars, d_counts, d_spans4, d_features, cap, d_result, dtype=np.int64):
    """featurize of one device-resident batch through the flow (``latok_flow_token_features``): 4 span values + 25 int8
    feature sums per kept token (reference default_tokenizer.py:163-191)."""
    lib = _lib.ensure_init()
    _, flag32 = _out_dtype(dtype)
    _lib.check(lib.latok_flow_token_features(d_units, int(kind), d_row_off, int(n_str), int(total_chars), d_counts, d_spans4, d_features,
                                             int(cap), d_result, flag32))


def flow_wait():
    """Block until every batch submitted with ``flow_split_mask`` on the current context is complete."""
    _lib.check(_lib.ensure_init().latok_flow_wait())


def _rule_table(name, idx):
    """A combo matrix as build_combo_matrix returns it -> C-contiguous int8 [rows, cols].  A 1-D index vector means
    "sum of those feature rows" to _combine_matrix_rows (latok.c:340-353): one single-column row per entry."""
    a = np.asarray(idx)
    if a.size and not np.issubdtype(a.dtype, np.integer):
        raise ValueError(f"{name}: feature ids must be integers")
    if a.size and not ((a == -1) | ((a >= 0) & (a < _lib.FEATURE_COUNT))).all():
        raise ValueError(f"{name}: feature ids must be -1 (padding) or 0..{_lib.FEATURE_COUNT - 1}")
    a = a.astype(np.int8)
    if a.ndim == 1:
        a = a[a != -1].reshape(-1, 1)
    if a.ndim != 2:
        raise ValueError(f"{name}: must be a 1-D or 2-D index matrix")
    if a.shape[0] and a.shape[1] == 0:
        raise ValueError(f"{name}: rows have no columns")
    return np.ascontiguousarray(a)


def set_rules(c_split, c_mask, c_sym):
    """Install custom C_SPLIT / C_MASK / C_SYM combo matrices: every batch entry point (and ``tokenize`` / ``featurize``
    of latok_amd.core.default_tokenizer) then evaluates them inside the fused kernel, bit-exact with the reference's
    ``gen_split_mask`` recipe (default_tokenizer.py:113-134) run on the same tables.  ``reset_rules()`` restores the
    built-in tables."""
    lib = _lib.ensure_init()
    t = [_rule_table(n, m) for n, m in (("C_SPLIT", c_split), ("C_MASK", c_mask), ("C_SYM", c_sym))]
    args = []
    for a in t:
        args += [_ptr(a) if a.size else None, a.shape[0], a.shape[1] if a.shape[0] else 0]
    _lib.check(lib.latok_set_rules(*args))


def reset_rules():
    _lib.check(_lib.ensure_init().latok_reset_rules())


def rules_active() -> bool:
    return bool(_lib.ensure_init().latok_rules_active())
