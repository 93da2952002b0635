"""Default LaTok tokenizer -- mirror of reference latok/core/default_tokenizer.py (same public names).

``tokenize`` / ``featurize`` get their boundary offsets from the fused HIP kernel (one string = a batch of one);
``gen_split_mask`` stays the generic, user-editable recipe over the three native functions exactly as in the reference
(default_tokenizer.py:113-134), so customised combo matrices keep working through the compat kernels.

Extension point: the reference's ``tokenize`` / ``featurize`` evaluate the module-level ``C_SPLIT`` / ``C_MASK`` /
``C_SYM`` (default_tokenizer.py:108-110,123-129), so a user customises them by rebinding those names.  The same works
here: before every call the three names are compared with the built-in tables and, when they differ, installed in the
fused kernel as run-time rule tables (``batch.set_rules``) -- and removed again when they are restored.
"""
import numpy as np

from . import offsets as oft
from .latok_utils import LaToken, build_combo_matrix, gen_block_mask
from ..latok import _combine_matrix_rows, _gen_parse_matrix
from .. import batch as _batch


def build_split_combo_matrix():
    """Split on whitespace, on a symbol, after a symbol, and at camelCase humps (an upper-case letter next to a
    lower-case one, either side).  Reference default_tokenizer.py:39-55."""
    return build_combo_matrix([
        [oft.SPACE_IDX],
        [oft.SYMBOL_IDX],
        [oft.PREV_SYMBOL_IDX],
        [oft.UPPER_IDX, oft.NEXT_LOWER_IDX],
        [oft.UPPER_IDX, oft.PREV_LOWER_IDX],
    ])


def build_mask_combo_matrix():
    """Starts of spans that must not be split: twitter specials (``#tag``, ``@user``, ``.@user`` after a space),
    e-mail addresses (``@`` between alphanumerics) and URLs (``:`` after a letter, before ``//``).
    Reference default_tokenizer.py:58-91."""
    return build_combo_matrix([
        [oft.TWITTER_IDX, oft.PREV_SPACE_IDX, oft.NEXT_ALPHA_IDX],
        [oft.CHAR_PERIOD_IDX, oft.PREV_SPACE_IDX, oft.NEXT_AT_IDX, oft.AFTER_NEXT_ALPHA_IDX],
        [oft.CHAR_AT_IDX, oft.PREV_ALPHA_NUM_IDX, oft.NEXT_ALPHA_NUM_IDX],
        [oft.CHAR_COLON_IDX, oft.NEXT_SLASH_IDX, oft.AFTER_NEXT_SLASH_IDX, oft.PREV_ALPHA_IDX],
    ])


def build_symbol_combo_matrix():
    """A symbol followed by whitespace (a symbol that ends a token).  Reference default_tokenizer.py:94-102."""
    return build_combo_matrix([
        [oft.SYMBOL_IDX, oft.NEXT_SPACE_IDX],
    ])


# rule tables used by gen_split_mask (reference default_tokenizer.py:108-110)
C_SPLIT = build_split_combo_matrix()
C_MASK = build_mask_combo_matrix()
C_SYM = build_symbol_combo_matrix()


def gen_split_mask(m: np.ndarray):
    """Split-mask vector of a feature matrix: non-zero entries mark the characters at which to split
    (reference default_tokenizer.py:113-134)."""
    mt = m.T  # features as rows
    splits = _combine_matrix_rows(mt, C_SPLIT) * gen_block_mask(_combine_matrix_rows(mt, C_MASK), mt[oft.SPACE_IDX])
    splits += _combine_matrix_rows(mt, C_SYM)
    splits[0] = 1  # start of string is always a boundary; IndexError on an empty matrix, like the reference
    return splits


_BUILTIN = (C_SPLIT.copy(), C_MASK.copy(), C_SYM.copy())
_installed = {}   # context handle -> the tables THIS module installed there (an explicit batch.set_rules is left alone)


def _same_as_builtin() -> bool:
    for a, b in zip((C_SPLIT, C_MASK, C_SYM), _BUILTIN):
        if a is b:
            continue
        a = np.asarray(a)
        if a.shape != b.shape:
            return False
        if a.dtype == b.dtype and a.flags.c_contiguous:
            if a.tobytes() != b.tobytes():     # (a tenth of the time of np.array_equal on arrays this small)
                return False
        elif not np.array_equal(a, b):
            return False
    return True


def _sync_rules():
    """Make the fused kernel evaluate whatever C_SPLIT / C_MASK / C_SYM are bound to right now."""
    if not _installed and _same_as_builtin():
        return                              # the usual case: untouched tables, nothing installed by this module
    from .. import _lib
    key = _lib.load().latok_ctx_get_current()
    cur = (np.asarray(C_SPLIT), np.asarray(C_MASK), np.asarray(C_SYM))
    custom = not all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(cur, _BUILTIN))
    have = _installed.get(key)
    if have is not None and not _batch.rules_active():
        have = None                     # (a new context at the address of a destroyed one, or an explicit reset_rules)
        _installed.pop(key, None)
    if custom:
        if have is None or not all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(cur, have)):
            _batch.set_rules(*cur)
            _installed[key] = tuple(np.array(x, copy=True) for x in cur)
    elif have is not None:
        _batch.reset_rules()
        _installed.pop(key, None)


def _boundaries(text: str) -> np.ndarray:
    """np.nonzero(gen_split_mask(_gen_parse_matrix(text)))[0] (reference default_tokenizer.py:146-148), computed by
    the fused kernel."""
    if len(text) == 0:
        # the reference fails at ``splits[0] = 1`` on an empty array (default_tokenizer.py:132)
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    _sync_rules()
    return _batch.split_offsets_one(text)


def _spans(text: str, non_zero):
    """(start, end) pairs exactly as the reference's loop walks them (default_tokenizer.py:149-158): consecutive
    boundaries, then (last boundary that was an END, or 0 when there is only one boundary, len(text))."""
    nz = non_zero.tolist()
    return zip(nz[:-1] + [nz[-1] if len(nz) > 1 else 0], nz[1:] + [len(text)])


def tokenize(text: str):
    """Yield the tokens of ``text`` (reference default_tokenizer.py:137-160)."""
    non_zero = _boundaries(text)
    if len(non_zero) > 0:
        for a, b in _spans(text, non_zero):
            token = text[a:b].strip()
            if token:
                yield token
    else:
        yield ''


def tokenize_spans(texts):
    """Additive: the tokens of a whole list of strings as spans (``latok_amd.batch.TokenSpans``: counts, (start, end) pairs,
    lazy ``tokens(i)``) -- the same result as ``[list(tokenize(t)) for t in texts]`` without one Python object per token."""
    _sync_rules()
    return _batch.token_spans_batch(texts)


def featurize(text: str):
    """Yield the tokens of ``text`` as ``LaToken`` with their summed feature vectors
    (reference default_tokenizer.py:163-191), computed on the device without building the n x 25 matrix.
    start_idx / end_idx are the unstripped span between boundaries, like the reference; the reference indexes matrix
    rows with ``np.arange(..., dtype=np.int8)`` and therefore breaks past character 127 -- any position works here."""
    if len(text) == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    _sync_rules()
    yield from _batch.featurize_one(text)
