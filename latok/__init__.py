"""latok -- the reference's import name, served by the MI355X implementation (latok_amd).

A caller written against the reference keeps its imports (reference latok/core/default_tokenizer.py:33-36,
scripts/timing/time_tokenizer.py:20-21):

    from latok.core.default_tokenizer import tokenize, featurize
    from latok.latok import _gen_parse_matrix, _gen_block_mask, _combine_matrix_rows
    from latok.core.latok_utils import gen_parse_matrix, LaToken
    import latok.core.offsets

Every one of these names IS the latok_amd object (no second copy of anything): this package only registers latok_amd's
modules under the reference's module paths.  It refuses to load when another ``latok`` distribution is importable from a
different place on sys.path (it would be shadowed silently otherwise); set LATOK_AMD_ALLOW_SHADOW=1 to load anyway.
"""
import importlib
import importlib.machinery
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _foreign_latok():
    """path of another importable ``latok`` package (not this shim), or None"""
    mine = os.path.dirname(_HERE)
    for entry in sys.path:
        root = os.path.abspath(entry or os.getcwd())
        if root == mine:
            continue
        try:
            spec = importlib.machinery.PathFinder.find_spec("latok", [root])
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.origin and os.path.dirname(os.path.abspath(spec.origin)) != _HERE:
            return spec.origin
    return None


_other = _foreign_latok()
if _other is not None and os.environ.get("LATOK_AMD_ALLOW_SHADOW", "") != "1":
    raise ImportError(f"latok (latok_amd shim at {_HERE}) would shadow another 'latok' package at {_other}; "
                      "import latok_amd directly, fix sys.path, or set LATOK_AMD_ALLOW_SHADOW=1")

_impl = importlib.import_module("latok_amd")
__version__ = _impl.__version__
for _alias, _real in (("latok.latok", "latok_amd.latok"), ("latok.core", "latok_amd.core"),
                      ("latok.core.offsets", "latok_amd.core.offsets"),
                      ("latok.core.latok_utils", "latok_amd.core.latok_utils"),
                      ("latok.core.default_tokenizer", "latok_amd.core.default_tokenizer")):
    sys.modules[_alias] = importlib.import_module(_real)
latok = sys.modules["latok.latok"]     # attribute access: ``import latok; latok.latok._gen_parse_matrix``
core = sys.modules["latok.core"]
