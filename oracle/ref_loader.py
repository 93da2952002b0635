"""TEST INFRASTRUCTURE ONLY -- loaders for the *real* reference, never imported by latok_amd/.

Two things can be loaded:

* ``load_ref_ext()``  -- the CPython extension built by ``make -C oracle ref`` from the reference's own
  ``latok/core/src/latok/latok.c`` into ``oracle/_ref/``.  It exposes the reference's three native functions
  (method table ``latok.c:373-378``): ``_gen_parse_matrix``, ``_gen_block_mask``, ``_combine_matrix_rows``.
  The built ``.so`` travels to the GPU box, so this works there too.

* ``load_ref_python()`` -- the reference's own Python glue (``latok/core/default_tokenizer.py`` etc.), imported
  from ``/root/reference`` with the extension above pre-seeded as ``latok.latok``.  Only possible in the build
  container (``/root/reference`` does not exist on the GPU box); used to generate ``tests/golden`` fixtures and to
  differential-test the restatement.
"""
import importlib.machinery
import importlib.util
import os
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("LATOK_REFERENCE_ROOT", "/root/reference")
_EXT_PATH = os.path.join(_HERE, "_ref", "latok" + sysconfig.get_config_var("EXT_SUFFIX"))

_ext = None


def ref_ext_available() -> bool:
    return os.path.exists(_EXT_PATH)


def ref_python_available() -> bool:
    return ref_ext_available() and os.path.isdir(os.path.join(REF_ROOT, "latok", "core"))


def load_ref_ext():
    """Load oracle/_ref/latok*.so under the private module name ``_latok_ref_ext``."""
    global _ext
    if _ext is None:
        if not ref_ext_available():
            raise RuntimeError("oracle/_ref is not built: run `make -C oracle ref` in the build container")
        # the init symbol is PyInit_latok, so the spec name must end in "latok"
        loader = importlib.machinery.ExtensionFileLoader("latok", _EXT_PATH)
        spec = importlib.util.spec_from_file_location("latok", _EXT_PATH, loader=loader)
        prev = sys.modules.get("latok")
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        # single-phase-init extensions register themselves in sys.modules under their short name: undo that
        if sys.modules.get("latok") is mod:
            if prev is None:
                del sys.modules["latok"]
            else:
                sys.modules["latok"] = prev
        _ext = mod
    return _ext


def load_ref_python():
    """Import the reference's own ``latok.core.default_tokenizer`` (build container only).

    Must be called in a process that has NOT imported this repo's ``latok`` alias package.
    Returns the module ``latok.core.default_tokenizer`` of the reference.
    """
    if not ref_python_available():
        raise RuntimeError("reference python sources not available (expected on the GPU box)")
    ext = load_ref_ext()
    if "latok" in sys.modules and not getattr(sys.modules["latok"], "__file__", "").startswith(REF_ROOT):
        raise RuntimeError("a different 'latok' package is already imported in this process")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import latok  # noqa: F401  (the reference package; pure python __init__)
    sys.modules["latok.latok"] = ext
    setattr(sys.modules["latok"], "latok", ext)
    import latok.core.default_tokenizer as dt
    return dt
