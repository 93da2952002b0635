"""latok_amd -- MI355X-native (gfx950, HIP) implementation of latok's character-feature-matrix + split-mask path.

Layout mirrors the reference package so it can stand in for it:

    latok_amd.latok                    <- reference C extension ``latok.latok`` (latok/core/src/latok/latok.c:373-378)
    latok_amd.core.offsets             <- reference latok/core/offsets.py
    latok_amd.core.latok_utils         <- reference latok/core/latok_utils.py
    latok_amd.core.default_tokenizer   <- reference latok/core/default_tokenizer.py
    latok_amd.batch                    <- additive: whole-batch entry points over the fused kernel

The repo's top-level ``latok/`` package registers these modules under the reference's own import names
(``from latok.core.default_tokenizer import tokenize`` works with no setup call); ``install_as_latok()`` does the same on
request for a process that must not have ``latok/`` on its path.

All compute goes through ``liblatok_hip.so`` (C ABI in include/latok_hip.h).  There is no CPU fallback: importing is
harmless, but every compute call raises ``RuntimeError`` when the library or a HIP device is missing.
"""
__version__ = "0.1.0"


def install_as_latok():
    """Register this package under the reference's import names, so unmodified reference callers keep working:

        import latok_amd; latok_amd.install_as_latok()
        from latok.core.default_tokenizer import tokenize      # reference spelling, HIP implementation
        from latok.latok import _gen_parse_matrix

    (reference import sites: latok/core/default_tokenizer.py:33-36, scripts/timing/time_tokenizer.py:20-21).
    Refuses to shadow a different, already imported ``latok`` package."""
    import importlib
    import sys
    me = sys.modules[__name__]
    other = sys.modules.get("latok")
    if other is not None and other is not me and getattr(other, "_impl", None) is not me:   # (the repo's own latok/ shim is fine)
        raise RuntimeError("a different 'latok' package is already imported")
    names = {"latok": __name__, "latok.latok": __name__ + ".latok", "latok.core": __name__ + ".core",
             "latok.core.offsets": __name__ + ".core.offsets", "latok.core.latok_utils": __name__ + ".core.latok_utils",
             "latok.core.default_tokenizer": __name__ + ".core.default_tokenizer"}
    for alias, real in names.items():
        sys.modules[alias] = importlib.import_module(real)
    return me
